#!/usr/bin/env python3
"""Stress of the classifier-file writer (CustomCLIP._write_files): many jobs into one directory, joined in different ways (explicit wait,
the next job's implicit wait, no wait at all), file digests checked against the inline writer's, thread count and host memory watched."""
import hashlib, os, sys, tempfile, threading, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ovmr_amd import modules, synth

g = np.load(os.path.join(ROOT, "tests", "golden", "tiny.npz"))
spec, SEED = synth.SPECS["tiny"], 11
S, cpb = int(g["meta_shots"]), int(g["meta_classes_per_batch"])
cm = modules.CLIPModel({k: torch.from_numpy(v) for k, v in synth.clip_state_dict(spec, SEED, jitter=True).items()}, spec)
pl = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, 2, SEED, True).items()}
labels = g["l2_eval_labels"]
img = synth.images(len(labels), spec.image_resolution, seed=1234, class_ids=labels, class_strength=0.6)
loader = [{"img": torch.from_numpy(img[s:s + cpb * S]).cuda().half(), "label": torch.from_numpy(labels[s:s + cpb * S])} for s in range(0, len(labels), cpb * S)]
q = torch.from_numpy(synth.images(4, spec.image_resolution, seed=777)).cuda().half()
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
names = ("mm_classifiers.pt", "visual_tokens.pt")
dig = lambda d: [hashlib.sha256(open(os.path.join(d, n), "rb").read()).hexdigest() for n in names]

def make(sub, a):
    cfg = modules.make_cfg(n_ctx=2, num_shots=S, eval_tau=float(g["meta_tau"]), output_dir=os.path.join(tmp, sub))
    m = modules.CustomCLIP(cfg, torch.from_numpy(g["l2_tokenized_prompts"]), cm, prompt_learner_state=pl, reserve=(64, 64, 256))
    m.ASYNC_FILE_WRITE = a
    return m

m0 = make("inline", False); m0.forward_prompt(loader); want = dig(os.path.join(tmp, "inline"))
m = make("stress", True)
m.FILE_WRITE_DELAY_S = 0.01
d = os.path.join(tmp, "stress")
t0, rss0, n = time.time(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, 300
for i in range(n):
    m.forward_prompt(loader, wait_files=False)
    for _ in range(i % 3):
        m(q)
    if i % 4 == 0:
        m.wait_files()
        assert dig(d) == want, i
    elif i % 4 == 1:
        time.sleep(0.03)                                   # the writer starts by itself
m.wait_files()
torch.cuda.synchronize()
assert dig(d) == want and sorted(os.listdir(d)) == sorted(names)
print(f"{n} jobs in {time.time() - t0:.1f} s, threads alive {threading.active_count()}, max RSS {rss0 >> 10} -> {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10} MiB, files identical to the inline writer's")
