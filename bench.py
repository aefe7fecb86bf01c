#!/usr/bin/env python3
"""Benchmark of the OVMR hot path on MI355X: images/sec for ViT-B/16 encode + fusion at
1 000 classes x 16 shots (BASELINE.json `metric`), 1/2/4/8 GPUs of one node.

One STEP = one whole classifier-generation job plus fusion-mode inference, nothing cached:
  generation  16 000 exemplar images (1000 classes x 16 shots) -> image encoder -> visual-token
              generator -> multimodal / vision prompts -> text encoder (mm, vision, zero-shot text
              classifiers) -> cross-validation argmax counts -> F1 -> fusion weights;
  inference   --queries images in batches of --query-batch (256 = BASELINE config 3) -> image encoder -> three
              classifier GEMMs -> softmax -> fused probabilities; the test loop's forwards run through
              CustomCLIP.forward_batches: every batch is its own forward of 256 images (bit-identical logits, in order),
              two of them in flight on two handles / two streams so that one batch's partial last round of tiles is
              filled by the other's work (--overlap 0: one at a time; `phases.inference_batches_in_flight`).
value = (exemplar + query images of ALL ranks) / wall time of a step, inputs resident in HBM.
With N > 1 (launched by torch.distributed.run, one process per GPU, backend nccl = RCCL) the classes
and the queries are sharded over ranks (strong scaling of the named 1k-class job); the only data-path
collectives are one all-gather of classifier rows and one all-reduce of the F1 counters.

The JSON line also carries
  roofline      the dominant kernel (fp16 MFMA GEMM at the c_fc launch shape of the job: M = batch x 197 = 152675 token
                rows at batch 775, N = 3072, K = 768, ln_2 fold + bias + QuickGELU epilogue) timed live with HIP events
                on the stream it is launched on, against the 2.5 PFLOP/s dense fp16 MFMA peak;
  cpu_baseline  the CPU oracle (a torch-CPU port of the reference path, validated against golden vectors of
                the real reference) timed on the CPUs this process may use (the cgroup quota, not the whole host: usable // 16 worker processes x 16 threads on disjoint
                class slices, 1 warm-up + 3 timed repetitions, median) on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# One-flag lines for the configurations of BASELINE.json ("metric" = configs[3]-style headline job = the default; c1 / c2 / c3 / c4 / c5 =
# configs[0] / [1] / [2] / [3] / [4] at N = 1).  A preset only overrides defaults; explicit flags win.  For c3 the VALUE is the inference phase
# alone (query images / s: generation is set-up there), for the others the whole job as for the metric.
PRESETS = {
    "metric": {"about": "ViT-B/16, 1000 classes x 16 shots + 4096 queries (the headline metric)", "set": {}},
    "c1": {"about": "CLIP ViT-B/16 zero-shot (trainers/zsclip.py), 10 classes, the test loop at batch 256; value = test images/s",
           "set": {"classes": 10, "queries": 16384}, "value": "zeroshot"},
    "c2": {"about": "ViT-B/16, 100 classes x 8 shots generation + 1024 queries", "set": {"classes": 100, "shots": 8, "queries": 1024, "classes_per_batch": 100}},
    "c3": {"about": "ViT-B/16 fusion inference at batch 256 against 1000 x 16-shot classifiers; value = inference images/s",
           "set": {"queries": 16384}, "value": "inference"},
    "c4": {"about": "ViT-B/16, 64 shots, a 10 000-class vocabulary (the reference names no count: datasets/imagenet_21k_P.py), class-sharded over 8 "
                    "ranks: ONE rank's shard (1250 classes = 80 000 exemplar images + 512 of 4096 queries) through the sharded path on this GPU, the "
                    "other seven ranks' rows and votes recorded from their own runs; value = this rank's images/s, the 8-rank figure is a projection",
           "set": {"classes": 10000, "shots": 64, "queries": 4096, "classes_per_batch": 1250, "emulate_world": 8, "emulate_rank": 0}, "value": "shard"},
    "c5": {"about": "ViT-L/14@336px, 1000 classes x 32 shots + 512 queries (the MFMA-bound stress configuration)",
           # 170 images x 577 tokens = 384 row tiles: 6.0 / 18.0 / 24.0 rounds of the 256 CUs on the N = 1024 / 3072 / 4096 GEMMs (r03y: 2 917 img/s
           # against 2 802 at 128 images and 128 classes per loader batch)
           "set": {"model": "ViT-L/14@336px", "shots": 32, "queries": 512, "batch": 170, "query_batch": 128, "classes_per_batch": 1000}},
}


DEFAULT_BATCH, DEFAULT_CLASSES_PER_BATCH = 775, 1000       # (tests/test_hip_configs.py runs the headline job with these)


def parse(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="ViT-B/16")
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--shots", type=int, default=16)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH,
                    help="exemplar images per encoder launch sequence (775 x 197 rows = 597 row tiles: 6.996 / 20.99 / 27.98 rounds of the 256 CUs "
                         "for the N = 768 / 2304 / 3072 GEMMs -- whole rounds on all three; 768 = 591 tiles pays 7 / 21 / 28 rounds for 6.93 / 20.78 / "
                         "27.70, 512 gives 4.6 rounds on N = 768, i.e. 8 % idle in the fifth)")
    ap.add_argument("--query-batch", type=int, default=256, help="query images per inference call (256 = BASELINE config 3)")
    ap.add_argument("--fuse-im2col", type=int, default=1, help="patch rows gathered inside the patch-embedding GEMM (1) or written out by an im2col pass first (0)")
    ap.add_argument("--last-q-cls", type=int, default=1, help="last vision block: Q projected for the CLS rows only (1) or for every token (0)")
    ap.add_argument("--enc-chunk", type=int, default=0, help="images per launch sequence of the image tower: 0 = the engine's choice (<= --batch, whole "
                    "rounds of tiles: ovmr_encode_chunk), n pins it")
    ap.add_argument("--overlap", type=int, default=-1, help="two query batches in flight on two streams (CustomCLIP.forward_batches): "
                    "1 / 0 force it on / off, -1 = the module's rule (batches of at most 384 images); 2 / 3: the reference's unchanged test loop "
                    "instead -- model(input) per batch + a host sync per batch -- with forward() splitting a batch over two handles (2) or not (3)")
    ap.add_argument("--fuse-qkv-attn", type=int, default=0, help="experiment build only (OVMR_HIP_LIB=.../libovmr_hip_exp.so): in_proj + attention of the vision "
                    "blocks as one launch per block (csrc/experiments/qkv_attn_fused.hip; OVMR_FQ_ABL=32 selects its 16-wave form)")
    ap.add_argument("--in-flight", type=int, default=0, help="query batches in flight in the test loop (0: the module's default, CustomCLIP.IN_FLIGHT = 2)")
    ap.add_argument("--classes-per-batch", type=int, default=DEFAULT_CLASSES_PER_BATCH,
                    help="classes per eval-set loader batch: the whole 1000-class exemplar set arrives as one batch, the engine encodes it "
                         "--batch images at a time and the classifier head runs once (r03y: 775 / 1000 against 768 / 240: +1.5-2 % end to end)")
    ap.add_argument("--gemm", type=int, default=int(os.environ.get("OVMR_GEMM", "8")))
    ap.add_argument("--attn", type=int, default=int(os.environ.get("OVMR_ATTN", "3")))
    ap.add_argument("--ln-fold", type=int, default=int(os.environ.get("OVMR_LN_FOLD", "1")),
                    help="1: ln_1/ln_2 folded into the consuming GEMM epilogue; 0: separate LayerNorm kernels")
    ap.add_argument("--preset", default="metric", choices=sorted(PRESETS),
                    help="one BASELINE.json configuration per flag: " + "; ".join(f"{k} = {v['about']}" for k, v in PRESETS.items()))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: initialise a one-rank nccl (= RCCL) process group and take the sharded path, so that the packed "
                         "all-gather and the counter all-reduce run through librccl on this GPU")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="N > 1: a PROJECTION of the N-GPU job from this one GPU -- the whole job is timed as one rank, then every rank's shard of "
                         "the N-rank job (its classes, its queries, the sharded code path with the two collectives served from the other ranks' "
                         "recorded contributions) is timed ALONE on this GPU, no process group; projected_speedup = T(1 rank) / max_R T(rank R of N)")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="with --emulate-world: time this rank's shard only (-1: every rank)")
    ap.add_argument("--stream-text", type=int, default=1,
                    help="1: the zero-shot text rows of a loader batch's classes come out of the same text-tower pass as its multimodal and "
                         "vision prompts (CustomCLIP(stream_text=True)); 0: one encode_text pass over all classes up front, as PromptLearner.__init__")
    ap.add_argument("--dist-timeout", type=float, default=300.0,
                    help="seconds a collective may take before the process group aborts this rank (it then exits non-zero; no retry)")
    ap.add_argument("--gelu-exact", type=int, default=int(os.environ.get("OVMR_GELU_EXACT", "0")),
                    help="1: QuickGELU with the reference's three fp16 rounding points; 0 (engine default): one rounding, fp32")
    ap.add_argument("--output-dir", default="tmpfs",
                    help="cfg.OUTPUT_DIR: where forward_prompt writes mm_classifiers.pt / visual_tokens.pt INSIDE the timed step (K23, "
                         "trainers/mm_classifier_one_prompt.py:276-291).  'tmpfs' = a scratch directory under /dev/shm, removed at exit; '' = no files")
    ap.add_argument("--presets", type=int, default=1,
                    help="default invocation only (preset metric, one GPU): after the headline, run the other BASELINE.json configurations (c2, c3, "
                         "c4 as one rank of eight, c5) for a few steps each as child processes and add their figures to the line as `presets`")
    ap.add_argument("--cpu-sample-classes", type=int, default=4, help="classes per CPU worker process and repetition (0: skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads per CPU worker process")
    ap.add_argument("--cpu-reps", type=int, default=5, help="timed repetitions of the faster precision (fp32 math); the fp16 leg, 2-3x slower on "
                    "these hosts, gets 2 -- enough to show it is the slower one")
    ap.add_argument("--cpu-procs", type=int, default=0, help="CPU worker processes (0: usable CPUs // --cpu-threads)")
    ap.add_argument("--cpu-timeout", type=float, default=240.0, help="give up on the CPU baseline after this many seconds")
    args = ap.parse_args(argv)
    given = {a.split("=")[0] for a in argv if a.startswith("--")}
    for key, val in PRESETS[args.preset]["set"].items():          # a preset moves defaults only: explicit flags win
        if "--" + key.replace("_", "-") not in given:
            setattr(args, key, val)
    return args


def preflight(args):
    """Before anything touches the GPU.  (1) The environment every multi-process GPU test of this repo runs under (tests/test_hip_distributed.py,
    tests/test_hip_parity.py, scripts/generate_classifier.sh): HSA_ENABLE_IPC_MODE_LEGACY=0 -- the host driver of this pool supports only dmabuf
    IPC, and with the legacy mode RCCL's (and torch's) cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.  The image
    exports it already; setdefault keeps an operator's explicit choice and makes `bench.py --gpus N` independent of the login shell.
    (2) One device per rank: `--gpus N` over RCCL needs N visible devices.  device_count() does not initialise the GPU, so a launcher that
    fails here has started nothing; the error is ONE line and the exit code 2 (no retry, no re-exec)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    if world > 1 and os.environ.get("OVMR_DIST_BACKEND", "nccl") == "nccl":
        import torch
        n = torch.cuda.device_count()
        if n < world:
            if int(os.environ.get("RANK", "0")) == 0:
                print(f"bench.py: --gpus {world} over RCCL needs {world} visible devices, this node shows {n} "
                      f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}); OVMR_DIST_BACKEND=gloo lets ranks share a device",
                      file=sys.stderr, flush=True)
            sys.exit(2)
    if "WORLD_SIZE" in os.environ and args.gpus != world and int(os.environ.get("RANK", "0")) == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); the launcher's count is used", file=sys.stderr, flush=True)


def main():
    args = parse()
    preflight(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched by hand with --gpus N: start one process per GPU as a child, before touching the GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", "29511", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist
    from ovmr_amd import modules, synth
    from ovmr_amd.data import ResidentEvalSet
    from ovmr_amd.shard import shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one GPU per rank; OVMR_DIST_BACKEND=gloo lets several ranks share one device to exercise the N > 1 path on a 1-GPU box
    backend = os.environ.get("OVMR_DIST_BACKEND", "nccl")
    dev = torch.device(f"cuda:{local if backend == 'nccl' else local % max(1, torch.cuda.device_count())}")
    torch.cuda.set_device(dev)
    sharded = world > 1 or args.force_dist          # the class-sharded path with its two collectives
    dist_info = None
    if sharded:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        # a rank that fails or times out inside a collective is aborted by the process group's watchdog and exits non-zero: the
        # launcher then ends the job -- no retry, no re-exec
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        to = datetime.timedelta(seconds=args.dist_timeout)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, timeout=to)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=to)
        try:
            dist_info = process_group_identity(dist, dev, backend, world)      # the job's first collectives: the process group's canary
        except RuntimeError as e:
            # one line per failing rank, a non-zero exit, no retry: the launcher ends the job.  The usual suspects are named -- this is the
            # point where a mis-set IPC mode or ranks doubled up on a device show
            print(f"bench.py: rank {rank} of {world}: the process group's first collective failed over {backend}: {str(e).splitlines()[0][:300]} "
                  f"(HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, device {dev}, "
                  f"HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')})", file=sys.stderr, flush=True)
            sys.exit(3)

    if PRESETS[args.preset].get("value") == "zeroshot":
        assert world == 1 and not sharded, "--preset c1 is a one-GPU line (N ranks would run N replicas of the test loop)"
        print(json.dumps(zeroshot_config(args, dev)), flush=True)
        return

    out_dir = args.output_dir
    if out_dir == "tmpfs":
        import atexit
        import shutil
        import tempfile
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        out_dir = tempfile.mkdtemp(prefix="ovmr_bench_", dir=base)
        atexit.register(shutil.rmtree, out_dir, True)
    spec, sd, pl, tok, model = make_model(args, dev, sharded, output_dir=out_dir)
    eng = model.engine
    C, S, Q, n_ctx = args.classes, args.shots, args.queries, 2
    R = spec.image_resolution

    if args.emulate_world > 1:
        assert world == 1 and not sharded, "--emulate-world runs in ONE process without a process group"
        if PRESETS[args.preset].get("value") == "shard":
            print(json.dumps(shard_of_world(args, model, spec, dev)), flush=True)
        else:
            print(json.dumps(emulate_world(args, model, spec, dev)), flush=True)
        return

    # ---- this rank's shard of the job, resident in HBM (N(0,1) images, fp16), seed 1234 + rank
    c0, c1 = shard_range(C, rank, world)
    q0, q1 = shard_range(Q, rank, world)
    ig = torch.Generator(device=dev).manual_seed(1234 + rank)
    ex_img = torch.empty(((c1 - c0) * S, 3, R, R), dtype=torch.float16, device=dev)
    for s in range(0, ex_img.shape[0], 1024):
        ex_img[s:s + 1024] = torch.randn((min(1024, ex_img.shape[0] - s), 3, R, R), generator=ig, device=dev).half()
    q_img = torch.randn((q1 - q0, 3, R, R), generator=ig, device=dev).half()
    loader = ResidentEvalSet(ex_img, torch.arange(c0, c1, device=dev), S, args.classes_per_batch, presharded=True)

    infer_only = PRESETS[args.preset].get("value") == "inference"    # c3: the classifiers are set-up, the step is the query loop
    if args.overlap in (1, 2) or (args.overlap < 0 and args.query_batch <= model.OVERLAP_MAX_BATCH):
        for slot in range(1, max(2, model.IN_FLIGHT)):
            model._twin(slot)                           # set-up, like the first handle's: the further handles forward_batches uses (not part of a step)

    def generate():
        if not sharded and not args.stream_text:
            model.zero_shot_classifier = model.prompt_learner.zero_shot_classifier = \
                model.prompt_learner.encode_zero_shot(model.tokenized_prompts)     # part of the job (:118-126)
        model.forward_prompt(loader, wait_files=False)   # rank 0's two files are written by a worker thread while the queries run; step() joins it

    def query_batches():                                 # the test loader: resident images, --query-batch at a time
        for b in range(0, q_img.shape[0], args.query_batch):
            yield q_img[b:b + args.query_batch]

    def step():
        if not infer_only:
            generate()
        outs = None
        if args.overlap >= 2:
            # the reference's UNCHANGED test loop (Dassl.pytorch/dassl/engine/trainer.py:461-482): one model(input) per batch and the
            # evaluator's host round trip per batch; 2: forward() runs the batch's two halves on two handles, 3: one handle
            model.SPLIT_FORWARD = args.overlap == 2
            for b in query_batches():
                outs = model(b)
                int(outs.max(1)[1].sum().item())
        else:
            for outs in model.forward_batches(query_batches(), stable_inputs=True, overlap=None if args.overlap < 0 else bool(args.overlap)):
                pass                                     # (the reference's loop hands each batch's logits to the evaluator)
        model.wait_files()                               # both files complete on disk INSIDE the step
        return outs

    if infer_only:
        generate()

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    t_own = time.perf_counter()
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_own = (t_own - t0) / args.steps                 # this rank's own steps, before it waited for the others at the closing barrier
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    assert out is None or bool(torch.isfinite(out).all())

    # phase split (untimed extra pass, informational)
    barrier()
    tg = time.perf_counter(); model.forward_prompt(loader, wait_files=False); torch.cuda.synchronize(); tg = time.perf_counter() - tg
    tf = time.perf_counter(); model.wait_files(); tf = time.perf_counter() - tf    # what is left of the file write once the GPU is idle (in a step it overlaps the queries)
    ti = time.perf_counter()
    if args.overlap >= 2:
        for b in query_batches():
            int(model(b).max(1)[1].sum().item())
    else:
        for _ in model.forward_batches(query_batches(), stable_inputs=True, overlap=None if args.overlap < 0 else bool(args.overlap)):
            pass
    torch.cuda.synchronize(); ti = time.perf_counter() - ti

    images_per_step = Q if infer_only else C * S + Q
    value = images_per_step * args.steps / dt
    roof = measure_roofline(eng, spec, args, dev, 0 if infer_only else (c1 - c0) * S, q1 - q0) if rank == 0 else None
    if roof is not None and (args.overlap == 1 or (args.overlap < 0 and args.query_batch <= model.OVERLAP_MAX_BATCH)) and q1 > q0:
        roof["launches_alone_note"] = ("every shape is timed with its launches running ALONE; in the job the query-batch launches (M = "
                                       f"{args.query_batch * spec.vision_tokens}) run two at a time (forward_batches), so a profiler's per-kernel-name "
                                       "average also holds durations of overlapped pairs -- per launch shape: tools/dominant_by_grid.py, "
                                       "profiles/*_dominant_kernel_by_grid.json")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(spec, sd, pl, tok, args, n_ctx) if args.cpu_sample_classes > 0 else None   # 0: profiling runs skip the CPU leg

    if sharded and dist_info is not None:
        dist_info["collective_times_us"] = time_collectives(dist, dev, C, spec.embed_dim, n_ctx)
        # one SCALE line shows the skew between ranks: every rank's own time per step (its steps done, before the closing barrier) and
        # its phase split, gathered over the group itself (host objects: any backend)
        mine = {"rank": rank, "ms_per_step_own": round(1000 * dt_own, 3), "generation_ms": round(1000 * tg, 3), "inference_ms": round(1000 * ti, 3),
                "files_join_ms": round(1000 * tf, 3),
                "classes": c1 - c0, "query_images": q1 - q0}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist_info["per_rank"] = per_rank
        own = [p["ms_per_step_own"] for p in per_rank]
        dist_info["rank_skew_ms"] = round(max(own) - min(own), 3)
    if rank == 0:
        flops_img = eng.flops_per_image()
        # executed FLOPs per image of THIS job: every loader batch and every query batch through the plan the engine runs for it
        n_ex, n_q, lb = (c1 - c0) * S, q1 - q0, args.classes_per_batch * S
        fl = sum(eng.flops_executed(min(lb, n_ex - s0)) for s0 in range(0, 0 if infer_only else n_ex, lb))
        fl += sum(eng.flops_executed(min(args.query_batch, n_q - s0)) for s0 in range(0, n_q, args.query_batch))
        flops_run = fl / max(1, (0 if infer_only else n_ex) + n_q)
        line = {
            "metric": "images/sec ViT-B/16 encode+fusion, 1k-class×16-shot, 1/2/4/8 MI355X" if args.preset == "metric" else
                      f"images/sec, BASELINE.json configuration {args.preset}: {PRESETS[args.preset]['about']}",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"OVMR classifier generation + fusion inference, {args.model}, {C} classes x {S} shots "
                                   f"({C * S} exemplar images, batch {args.batch}) + {Q} query images (batch {args.query_batch}), n_ctx 2, tau 10",
                       "parallelism": f"class/query sharding over {world} rank(s); all-gather rows + all-reduce counters"
                                      + (f" (process group {dist.get_backend()}, sharded path forced)" if args.force_dist and world == 1 else ""),
                       "preset": args.preset, "gemm_variant": args.gemm, "attn_variant": args.attn, "ln_fold": args.ln_fold, "gelu_exact": args.gelu_exact,
                       "stream_text": args.stream_text, "encoder_reserve_images": args.batch, "encoder_chunk_images": eng.encode_chunk, "images_per_step": images_per_step,
                       "files_written_in_step": bool(out_dir) and not infer_only,
                       "output_dir": ("tmpfs scratch (" + os.path.dirname(out_dir) + ")") if args.output_dir == "tmpfs" else out_dir},
            "dist": dist_info,
            "roofline": roof,
            "cpu_baseline": cpu,
            "phases": {"generation_images_per_s_rank0": round((c1 - c0) * S / tg, 1),
                       "generation_ms_rank0": round(1000 * tg, 3),
                       "files_join_ms_after_generation_alone": round(1000 * tf, 3) if out_dir else None,
                       "files_written_by": "a worker thread behind a side stream (CustomCLIP._write_files), joined at the end of every step",
                       "inference_images_per_s_rank0": round((q1 - q0) / ti, 1) if q1 > q0 else None,
                       "inference_query_batch": args.query_batch,
                       "inference_batches_in_flight": 2 if (args.overlap == 1 or (args.overlap < 0 and args.query_batch <= model.OVERLAP_MAX_BATCH)) else 1,
                       "encoder_tflops_e2e_algorithmic": round(value * flops_img / 1e12, 1),
                       "encoder_tflops_e2e_executed": round(value * flops_run / 1e12, 1),
                       "e2e_frac_of_fp16_mfma_peak": round(value * flops_run / 1e12 / (2500.0 * world), 4)},
        }
        headline_job = (args.model, args.classes, args.shots, args.queries) == ("ViT-B/16", 1000, 16, 4096)
        if args.presets and args.preset == "metric" and headline_job and world == 1 and not args.force_dist:
            torch.cuda.empty_cache()                    # (the children need at most ~40 GB of the 288: this process keeps its ~10 GB)
            line["presets"] = run_presets(args)
        if cpu and cpu.get("value"):
            # one GPU against the CPUs the cgroup grants this job (cpu["cores"] threads), NOT against the whole host
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
            line["gpu_over_cpu_note"] = f"1 GPU vs {cpu['cores']} CPU threads (of {cpu.get('host_cores')} on the host)"
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------
def run_presets(args):
    """The other BASELINE.json configurations on the same box, right behind the headline: each as a fresh child process of this script
    (its own weights, images and engine; this process has released its GPU memory), a few steps each.  Returns {preset: {value, unit,
    ms_per_step, steps, roofline_frac, workload}} -- or {"error": ...} for one that failed; the headline line does not depend on them."""
    plan = (("c1", 3, 1), ("c2", 3, 1), ("c3", 3, 1), ("c4", 2, 1), ("c5", 2, 1))
    out = {}
    for name, steps, warm in plan:
        cmd = [sys.executable, os.path.abspath(__file__), "--preset", name, "--steps", str(steps), "--warmup", str(warm),
               "--presets", "0", "--gelu-exact", str(args.gelu_exact), "--ln-fold", str(args.ln_fold)]
        # (c1's own CPU leg -- the reference's CPU-runnable configuration -- on a smaller sample than the headline's: 32 images, 3 repetitions, ~20 s)
        cmd += ["--cpu-reps", "3", "--cpu-sample-classes", "2"] if name == "c1" and not args.no_cpu_baseline else ["--no-cpu-baseline"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            rows = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
            if r.returncode != 0 or not rows:
                out[name] = {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
                continue
            d = json.loads(rows[-1])
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                         "roofline_frac": (d.get("roofline") or {}).get("frac"), "workload": d["config"]["workload"],
                         "wall_s_incl_setup": round(time.perf_counter() - t0, 1)}
            if d.get("projection"):
                out[name]["projected_images_per_s_all_ranks"] = d["projection"]["projected_images_per_s_all_ranks"]
                out[name]["projection_excludes"] = d["projection"]["not_included"]
            if d.get("phases"):
                out[name]["phases"] = {k: v for k, v in d["phases"].items() if k.endswith("images_per_s") or k.endswith("_rank0") or "xval" in k or "head" in k}
            if name == "c1" and d.get("cpu_baseline"):       # configuration 1 is the reference's CPU-runnable case: its CPU figure belongs beside it
                out[name]["cpu_baseline"] = {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample", "fp32_images_per_s", "fp16_images_per_s")}
                out[name]["gpu_over_cpu"] = d.get("gpu_over_cpu")
        except Exception as e:                               # noqa: BLE001
            out[name] = {"error": repr(e)[:300]}
    return out


# ------------------------------------------------------------------------------------------
def zeroshot_prompt_ids(num_classes, seed=4321, context_length=77):
    """Tokenised `"a photo of a {}.".format(name)` prompts (trainers/zsclip.py:22, 42-45) with random class-name ids: [SOT, a, photo, of, a,
    name tokens (1-3), ".", EOT, 0 ...] -- the ids the reference's tokenizer gives those words (tests/golden/c1_zeroshot.npz holds real ones)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ids = np.zeros((num_classes, context_length), dtype=np.int64)
    for c in range(num_classes):
        name = rng.integers(1000, 49000, size=int(rng.integers(1, 4)))
        row = [49406, 320, 1125, 539, 320, *name.tolist(), 269, 49407]
        ids[c, :len(row)] = row
    return ids


def zeroshot_config(args, dev):
    """BASELINE.json configuration 1: ZeroshotCLIP (trainers/zsclip.py:32-60) on ViT-B/16, 10 classes.  One step = the text features of the ten
    prompts (:47-50) + the whole test loop (Dassl.pytorch/dassl/engine/trainer.py:461-482) over --queries resident images in batches of
    --query-batch: model_inference (:55-60: encode, normalise, scaled product -> raw fp16 logits) two batches in flight, every batch's
    logits counted by the on-device evaluator (ovmr_eval_counts: no host round trip per batch), then evaluate() -- accuracy / macro-F1 from
    the 3 C integers.  value = test images / s.  The CPU leg is the oracle's zeroshot_logits in fp32 (what clip.load gives a CPU device,
    clip/clip.py:130-131) and in fp16, faster one reported."""
    import torch
    from ovmr_amd import modules, synth
    from ovmr_amd.evaluator import Classification
    spec = synth.SPECS[args.model]
    C, Q, R = args.classes, args.queries, spec.image_resolution
    gen = torch.Generator(device=dev).manual_seed(1234)
    sd = device_clip_state(spec, gen, dev)
    cm = modules.CLIPModel(sd, spec, str(dev))
    ids = torch.from_numpy(zeroshot_prompt_ids(C))
    zs = modules.ZeroshotCLIP(cm, ids, reserve=(args.query_batch, 256, max(C, 1024)))
    eng = zs.engine
    for k, v in (("gelu_exact", args.gelu_exact), ("fuse_im2col", args.fuse_im2col), ("enc_chunk", args.enc_chunk), ("last_q_cls", args.last_q_cls),
                 ("gemm", args.gemm), ("attn", args.attn), ("ln_fold", args.ln_fold)):
        eng.set_option(k, v)
    ig = torch.Generator(device=dev).manual_seed(1234)
    q_img = torch.empty((Q, 3, R, R), dtype=torch.float16, device=dev)
    for s0 in range(0, Q, 1024):
        q_img[s0:s0 + 1024] = torch.randn((min(1024, Q - s0), 3, R, R), generator=ig, device=dev).half()
    labels = torch.randint(0, C, (Q,), generator=ig, device=dev)
    ev = Classification(C, device=str(dev))
    two = args.overlap == 1 or (args.overlap < 0 and args.query_batch <= zs.OVERLAP_MAX_BATCH)
    if two:
        zs._twin()
    ids_dev = ids.to(dev)

    def step():
        zs.text_features = eng.encode_text_ids(ids_dev, normalize=1)              # :47-50
        ev.reset()
        b = 0
        for out in zs.inference_batches((q_img[i:i + args.query_batch] for i in range(0, Q, args.query_batch)), stable_inputs=True,
                                        overlap=None if args.overlap < 0 else bool(args.overlap)):
            ev.process(out, labels[b:b + out.shape[0]])
            b += out.shape[0]
        return ev.counts()                                                          # the pass's one host read: 3 C + 1 integers

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tp, n_pred, n_label = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(n_label.sum()) == Q and int(n_pred.sum()) == Q
    value = Q * args.steps / dt
    roof = measure_roofline(eng, spec, args, dev, 0, Q)
    cpu = None
    if not args.no_cpu_baseline and args.cpu_sample_classes > 0:
        cpu = cpu_baseline_zeroshot(spec, sd, ids, args)
    flops_img = eng.flops_per_image()
    fl = sum(eng.flops_executed(min(args.query_batch, Q - s0)) for s0 in range(0, Q, args.query_batch)) / Q
    line = {"metric": f"images/sec, BASELINE.json configuration c1: {PRESETS['c1']['about']}",
            "value": round(value, 2), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"zero-shot CLIP test pass (trainers/zsclip.py), {args.model}, {C} prompts 'a photo of a <name>.', {Q} test images "
                                   f"(batch {args.query_batch}), text features + logits + on-device evaluator inside the step",
                       "parallelism": "one GPU", "preset": "c1", "gemm_variant": args.gemm, "attn_variant": args.attn, "ln_fold": args.ln_fold,
                       "gelu_exact": args.gelu_exact, "images_per_step": Q, "inference_batches_in_flight": 2 if two else 1},
            "roofline": roof, "cpu_baseline": cpu,
            "phases": {"inference_images_per_s": round(value, 1), "encoder_tflops_e2e_algorithmic": round(value * flops_img / 1e12, 1),
                       "encoder_tflops_e2e_executed": round(value * fl / 1e12, 1), "e2e_frac_of_fp16_mfma_peak": round(value * fl / 1e12 / 2500.0, 4),
                       "accuracy_of_random_labels": round(100.0 * float(tp.sum()) / Q, 2)}}
    if cpu and cpu.get("value"):
        line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        line["gpu_over_cpu_note"] = f"1 GPU vs {cpu['cores']} CPU threads (of {cpu.get('host_cores')} on the host)"
    return line


def make_model(args, dev, sharded=False, output_dir=""):
    """The job's model as bench.py times it: synthetic CLIP-init weights drawn on the device -- same shapes / std as
    ovmr_amd.synth, no biases / affine jitter, logit_scale = ln 100 -- random class-name tokens (seed 4321), n_ctx 2, tau 10, fusion mode."""
    import torch
    from ovmr_amd import modules, synth
    spec = synth.SPECS[args.model]
    C, S, n_ctx = args.classes, args.shots, 2
    gen = torch.Generator(device=dev).manual_seed(1234)
    sd = device_clip_state(spec, gen, dev)
    pl = device_pl_state(spec, n_ctx, gen, dev)
    cm = modules.CLIPModel(sd, spec, str(dev))
    cfg = modules.make_cfg(n_ctx=n_ctx, num_shots=S, eval_mode="fusion", eval_tau=10.0, output_dir=output_dir,
                           test_batch_size=args.batch)
    tok = torch.from_numpy(synth.class_token_ids(C, seed=4321))
    model = modules.CustomCLIP(cfg, tok, cm, prompt_learner_state=pl,
                               reserve=(args.batch, max(256, min(C, 2048)), max(C, 1024)), distributed=sharded,
                               stream_text=bool(args.stream_text))
    eng = model.engine
    if getattr(args, "in_flight", 0) > 0:
        model.IN_FLIGHT = args.in_flight
    if getattr(args, "fuse_qkv_attn", 0):
        eng.set_option("fuse_qkv_attn", args.fuse_qkv_attn)
    eng.set_option("gelu_exact", args.gelu_exact)
    eng.set_option("fuse_im2col", args.fuse_im2col)
    eng.set_option("enc_chunk", args.enc_chunk)
    eng.set_option("last_q_cls", args.last_q_cls)
    eng.set_option("gemm", args.gemm)
    eng.set_option("attn", args.attn)
    eng.set_option("ln_fold", args.ln_fold)
    return spec, sd, pl, tok, model


def device_identity(dev):
    """(uuid, pci) of the PHYSICAL device behind `dev`, each None where the runtime does not report it (an all-zero UUID counts as
    not reported)."""
    import torch
    p = torch.cuda.get_device_properties(dev)
    uuid = getattr(p, "uuid", None)
    uuid = f"uuid:{uuid}" if uuid is not None and str(uuid).strip("0-") != "" else None
    pci = [getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    pci = "pci:%04x:%02x:%02x" % tuple(int(v) for v in pci) if all(v is not None for v in pci) else None
    return uuid, pci


def process_group_identity(dist, dev, backend, world):
    """What a SCALE record must prove about its process group: the collective library and its version, the rank count, and how
    many DISTINCT devices the ranks sit on (all-gathered over the group itself).  With the RCCL backend every rank must own a
    device of its own: anything else is a mis-launched job, not a scaling measurement -- every rank raises."""
    import socket
    import torch
    ids = [None] * world
    host = socket.gethostname()
    dist.all_gather_object(ids, (host,) + device_identity(dev) + (f"index:{dev.index}",))
    # distinct devices by PHYSICAL identity: UUID and PCI address are two witnesses of the same fact, and two ranks that reach one
    # physical GPU through different HIP_VISIBLE_DEVICES orderings agree on BOTH -- so the count is the larger of the two (a runtime
    # that hands out one UUID for several boards still shows distinct PCI addresses).  The per-process device index stands in only
    # where the runtime reports neither identity for some rank.
    by_uuid = len({(h, u) for h, u, _, _ in ids}) if all(u for _, u, _, _ in ids) else 0
    by_pci = len({(h, p) for h, _, p, _ in ids}) if all(p for _, _, p, _ in ids) else 0
    if by_uuid or by_pci:
        seen, source = max(by_uuid, by_pci), "uuid_or_pci"
    else:
        seen, source = len({(h, i) for h, _, _, i in ids}), "device_index"
    phys = [f"{h}/{u or p or i}" for h, u, p, i in ids]
    version = None
    if backend == "nccl":
        try:
            v = torch.cuda.nccl.version()
            version = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception:                                  # noqa: BLE001
            version = "unknown"
    info = {"backend": dist.get_backend(), "collective_library": "RCCL" if backend == "nccl" else backend, "rccl_version": version,
            "ranks": dist.get_world_size(), "devices_seen": seen, "devices_seen_by": source, "device_ids": phys,
            "hip_version": getattr(torch.version, "hip", None)}
    if backend == "nccl" and info["devices_seen"] != world:
        raise RuntimeError(f"{world} ranks over RCCL sit on {info['devices_seen']} distinct device(s): {ids}")
    return info


def time_collectives(dist, dev, C, D, n_ctx, reps=20):
    """The two data-path collectives of the sharded job at their real payloads, timed on this process group (HIP events on the current
    stream, median of `reps`): the all-gather of one rank's packed block [ceil(C / world), 3 D + n_ctx D + 2] fp16 and the all-reduce of
    the int32 [3, 2, C] counters.  With one rank (--force-dist) this is what RCCL itself costs per call without any link traffic -- the
    figure to read beside a projection (--emulate-world), not a measurement of xGMI."""
    import torch
    from ovmr_amd.shard import _staged
    world = dist.get_world_size()
    bound = -(-C // world)
    block = torch.zeros((bound, 3 * D + n_ctx * D + 2), dtype=torch.float16, device=dev)
    counts = torch.zeros((3, 2, C), dtype=torch.int32, device=dev)

    def gather():
        b = _staged(block, dist)
        out = torch.empty((world * bound, b.shape[1]), dtype=torch.float16, device=b.device)
        dist.all_gather_into_tensor(out, b)
        return out.to(dev)

    def reduce():
        c = _staged(counts, dist)
        dist.all_reduce(c)
        return c.to(dev)

    res = {}
    for name, fn in (("all_gather_rows", gather), ("all_reduce_counts", reduce)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000.0)
        res[name] = round(sorted(ts)[len(ts) // 2], 1)
    res["payload_bytes"] = {"all_gather_rows_per_rank": int(block.numel() * 2), "all_reduce_counts": int(counts.numel() * 4)}
    return res


class EmulatedPeers:
    """Stands where CustomCLIP keeps `torch.distributed` (modules.CustomCLIP._dist) for `--emulate-world`: rank `rank` of `world`
    with NO process group.  The two collectives of the sharded path (ovmr_amd/shard.py) are served on the device from what the other
    ranks would contribute -- their packed classifier blocks and their argmax counters, taken from a run of the whole job in this
    process -- so the emulated rank executes the sharded code path on its own shard and ends with the whole job's bits."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.peer_blocks = None          # [world * bound, K + 2] fp16: every rank's pack_block
        self.peer_counts = None          # int32 [3, 2, C]: the other ranks' votes (None while recording)
        self.local_counts = None

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "emulated"

    def all_gather_into_tensor(self, out, block):
        bound = block.shape[0]
        out.copy_(self.peer_blocks)
        out[self.rank * bound:(self.rank + 1) * bound] = block

    def all_reduce(self, c):
        if self.peer_counts is None:
            self.local_counts = c.clone()
        else:
            c += self.peer_counts


def emulate_world(args, model, spec, dev):
    """One JSON object: the whole job as one rank, then each rank's shard of the `--emulate-world` job alone on this GPU."""
    import torch
    from ovmr_amd.data import ResidentEvalSet
    from ovmr_amd.shard import shard_range, pack_block, local_class_bound
    N, C, S, Q, R = args.emulate_world, args.classes, args.shots, args.queries, spec.image_resolution
    D, n_ctx = spec.embed_dim, 2
    ov = None if args.overlap < 0 else bool(args.overlap)
    # the data of every rank, drawn as the N-rank job draws it (seed 1234 + rank: exemplars, then queries)
    ex, qs = [], []
    for r in range(N):
        ig = torch.Generator(device=dev).manual_seed(1234 + r)
        c0, c1 = shard_range(C, r, N)
        q0, q1 = shard_range(Q, r, N)
        e = torch.empty(((c1 - c0) * S, 3, R, R), dtype=torch.float16, device=dev)
        for s in range(0, e.shape[0], 1024):
            e[s:s + 1024] = torch.randn((min(1024, e.shape[0] - s), 3, R, R), generator=ig, device=dev).half()
        ex.append(e)
        qs.append(torch.randn((q1 - q0, 3, R, R), generator=ig, device=dev).half())
    ex_all, q_all = torch.cat(ex), torch.cat(qs)
    del ex, qs
    if args.overlap == 1 or (args.overlap < 0 and args.query_batch <= model.OVERLAP_MAX_BATCH):
        model._twin()

    def timed(step):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    def queries(q, collect=False):
        out, outs = None, []
        for out in model.forward_batches((q[b:b + args.query_batch] for b in range(0, q.shape[0], args.query_batch)),
                                         stable_inputs=True, overlap=ov):
            if collect:
                outs.append(out.clone())
        return torch.cat(outs) if collect and outs else out

    # ---- T(1 rank): the whole job, exactly bench.py's default step
    full_loader = ResidentEvalSet(ex_all, torch.arange(C, device=dev), S, args.classes_per_batch, presharded=True)

    def whole(collect=False):
        if not args.stream_text:
            model.zero_shot_classifier = model.prompt_learner.zero_shot_classifier = \
                model.prompt_learner.encode_zero_shot(model.tokenized_prompts)
        model.forward_prompt(full_loader, wait_files=False)
        out = queries(q_all, collect)
        model.wait_files()
        return out

    t1 = timed(whole)
    ref = {k: getattr(model, k).clone() for k in ("mm_classifier", "visual_classifer", "zero_shot_classifier", "fusion_weight", "visual_tokens")}
    counts_full = model.xval_counts.clone()
    ref_out_all = whole(collect=True)

    # ---- every rank's shard, alone
    bound = local_class_bound(C, N, True, max(1, args.batch // S))
    blocks = []
    for r in range(N):
        c0, c1 = shard_range(C, r, N)
        loc = torch.arange(c0, c1, device=dev)
        blocks.append(pack_block(torch.cat([ref["mm_classifier"][loc], ref["visual_classifer"][loc], ref["zero_shot_classifier"][loc],
                                            ref["visual_tokens"][loc].flatten(1)], dim=1), loc, bound))
    peer_blocks = torch.cat(blocks)
    ranks = range(N) if args.emulate_rank < 0 else [args.emulate_rank]
    per_rank = []
    for r in ranks:
        c0, c1 = shard_range(C, r, N)
        q0, q1 = shard_range(Q, r, N)
        loader = ResidentEvalSet(ex_all[c0 * S:c1 * S], torch.arange(c0, c1, device=dev), S, args.classes_per_batch, presharded=True)
        q = q_all[q0:q1]
        emu = EmulatedPeers(r, N)
        emu.peer_blocks = peer_blocks
        model._dist, model._text_streamed = emu, True

        def shard_step(collect=False):
            model.forward_prompt(loader, wait_files=False)     # (rank 0 writes the files: off its critical path, joined at the end of the step)
            out = queries(q, collect)
            model.wait_files()
            return out

        shard_step()                                        # records this rank's own votes
        emu.peer_counts = counts_full - emu.local_counts
        tr = timed(shard_step)
        out = shard_step(collect=True)
        torch.cuda.synchronize()
        tg = time.perf_counter(); model.forward_prompt(loader, wait_files=False); torch.cuda.synchronize(); tg = time.perf_counter() - tg
        model.wait_files()
        # this rank's OWN rows against the whole job's rows for the same classes (the other ranks' rows are the recorded ones): equal
        # bits where the shard and the whole job take the same kernels, 1 - cos ~1e-6 apart where the smaller head takes others
        own = torch.arange(c0, c1, device=dev)
        eq = {k: bool(torch.equal(getattr(model, k)[own], ref[k][own])) for k in ref}
        eq["fused_outputs"] = bool(out is None or torch.equal(out, ref_out_all[q0:q1]))
        worst = max(float((getattr(model, k)[own].float() - ref[k][own].float()).abs().max()) for k in ref)
        same = all(eq.values())
        per_rank.append({"rank": r, "classes": c1 - c0, "exemplar_images": (c1 - c0) * S, "query_images": q1 - q0,
                         "ms_per_step": round(1000 * tr, 3), "generation_ms": round(1000 * tg, 3),
                         "images_per_s_alone": round(((c1 - c0) * S + q1 - q0) / tr, 1),
                         "classifiers_fusion_weights_outputs_bit_equal_to_whole_job": bool(same), "bit_equal": eq,
                         "max_abs_diff_of_own_rows": worst})
    model._dist, model._text_streamed = None, bool(args.stream_text)
    # the whole job once more, BEHIND the shards: the part runs its first seconds of a process faster than its steady state (clocks settle
    # under load), so a numerator timed only in front of the shards flatters... the numerator.  Both are reported; the ratio uses their mean
    t1_after = timed(whole)
    t1_first, t1 = t1, 0.5 * (t1 + t1_after)
    worst = max(p["ms_per_step"] for p in per_rank)
    line = {"metric": f"PROJECTION of {N}-rank strong scaling from one MI355X: every rank's shard timed alone, no process group, no xGMI traffic",
            "projection": True, "emulated_world": N, "ranks_timed": [p["rank"] for p in per_rank],
            "whole_job_ms_one_rank": round(1000 * t1, 3), "whole_job_ms_timed_before_the_shards": round(1000 * t1_first, 3),
            "whole_job_ms_timed_after_the_shards": round(1000 * t1_after, 3), "whole_job_images_per_s": round((C * S + Q) / t1, 1),
            "slowest_rank_ms": worst, "projected_speedup": round(1000 * t1 / worst, 3),
            "projected_images_per_s": round((C * S + Q) / (worst / 1000), 1),
            "not_included": "the RCCL all-gather of the packed rows and the all-reduce of the counters over xGMI (here: device copies of the "
                            "recorded peer contributions), barrier skew between ranks, per-process set-up",
            "collective_payload_bytes": {"all_gather_rows_per_rank": int(bound * (3 * D + n_ctx * D + 2) * 2), "all_reduce_counts": int(3 * 2 * C * 4)},
            "steps": args.steps, "warmup": args.warmup, "per_rank": per_rank,
            "config": {"workload": f"{args.model}, {C} classes x {S} shots + {Q} queries (batch {args.query_batch}), sharded over {N} emulated ranks",
                       "preset": args.preset, "encoder_reserve_images": args.batch, "classes_per_batch": args.classes_per_batch}}
    return line


def shard_of_world(args, model, spec, dev, keep=None, peers=None):
    """BASELINE.json configuration 4 as ONE rank of `--emulate-world` ranks, for a job whose exemplar set does not fit one GPU's turn
    (10 000 classes x 64 shots = 640 000 images = 193 GB of fp16 pixels): every rank's exemplars are drawn into ONE reusable buffer
    (seed 1234 + rank, as the N-rank job draws them) and run through hot loop A once, untimed, to record that rank's packed classifier
    block; the cross-validation counters of the whole job come from one pass over all C x S exemplar features against all C rows.
    Then rank `--emulate-rank` is timed through the SHARDED code path (CustomCLIP.forward_prompt with the two collectives served from the
    recorded blocks and votes) plus its share of the queries.  The line's value is that rank's measured images/s on this GPU;
    `projection` holds the N-rank figure (total images / this rank's time), which excludes xGMI, barrier skew and start-up.
    keep: a dict that receives the tensors behind the line (tests/test_hip_configs.py checks them against the oracle).
    peers: None = every other rank's exemplars go through the encoder (the bench preset); a list = only those ranks' do, the remaining
    ranks contribute one random unit direction per class as its three classifier rows and near copies of it as its features (the GPU test: what it probes on those ranks is the
    all-gather, the votes and the counters, not the encoder a third time -- 120 960 instead of 322 560 images at 5 040 classes)."""
    import torch
    from ovmr_amd.data import ResidentEvalSet
    from ovmr_amd.shard import shard_range, pack_block, local_class_bound
    N, R0, C, S, Q, R = args.emulate_world, max(0, args.emulate_rank), args.classes, args.shots, args.queries, spec.image_resolution
    D, n_ctx = spec.embed_dim, 2
    eng = model.engine
    ov = None if args.overlap < 0 else bool(args.overlap)
    bound = local_class_bound(C, N, True, max(1, args.batch // S))
    buf = torch.empty((bound * S, 3, R, R), dtype=torch.float16, device=dev)

    def draw(r):
        """Rank r's exemplars into the shared buffer and its queries, as the N-rank job draws them."""
        ig = torch.Generator(device=dev).manual_seed(1234 + r)
        c0, c1 = shard_range(C, r, N)
        q0, q1 = shard_range(Q, r, N)
        n = (c1 - c0) * S
        for s in range(0, n, 1024):
            buf[s:s + 1024][:min(1024, n - s)] = torch.randn((min(1024, n - s), 3, R, R), generator=ig, device=dev).half()
        q = torch.randn((q1 - q0, 3, R, R), generator=ig, device=dev).half()
        return ResidentEvalSet(buf[:n], torch.arange(c0, c1, device=dev), S, args.classes_per_batch, presharded=True), q

    # ---- every rank's rows, untimed (rank R0 last: its exemplars stay in the buffer)
    t_rec = time.perf_counter()
    model._dist, model._text_streamed = None, True
    with torch.no_grad():
        model._reset_generation_state()
        blocks = [None] * N
        for r in [x for x in range(N) if x != R0] + [R0]:
            if peers is not None and r != R0 and r not in peers:
                a0, a1 = shard_range(C, r, N)
                loc = torch.arange(a0, a1, device=dev)
                sg = torch.Generator(device=dev).manual_seed(99 + r)
                # one random unit direction per class serves as its mm / vision / text row, and its S "exemplar features" are that direction
                # plus a little noise: every vote of these ranks goes to the row's own class by a wide margin (logit ~99 against |logit| < 25
                # for every other row), and no real exemplar ever votes for them -- they exercise the all-gather, the counters and the
                # all-reduce without adding near-tied argmaxes to the job
                u = torch.nn.functional.normalize(torch.randn((a1 - a0, D), generator=sg, device=dev), dim=-1)
                model.mm_classifier[loc] = model.visual_classifer[loc] = model._text_rows[loc] = u.half()
                model.visual_tokens[loc] = torch.randn((a1 - a0, n_ctx, D), generator=sg, device=dev).half()
                model.eval_feat4cls[loc] = torch.nn.functional.normalize(u[:, None, :] + 0.01 * torch.randn((a1 - a0, S, D), generator=sg, device=dev), dim=-1).half()
                blocks[r] = pack_block(torch.cat([model.mm_classifier[loc], model.visual_classifer[loc], model._text_rows[loc],
                                                  model.visual_tokens[loc].flatten(1)], dim=1), loc, bound)
                continue
            loader, q = draw(r)
            loc = model._generate_local(loader)
            blocks[r] = pack_block(torch.cat([model.mm_classifier[loc], model.visual_classifer[loc], model._text_rows[loc],
                                              model.visual_tokens[loc].flatten(1)], dim=1), loc, bound)
        peer_blocks = torch.cat(blocks)
        ref = {"mm_classifier": model.mm_classifier.clone(), "visual_classifer": model.visual_classifer.clone(),
               "zero_shot_classifier": model._text_rows.clone(), "visual_tokens": model.visual_tokens.clone()}
        # the whole job's votes: all C x S exemplar features against all C rows of the three classifiers (several workspace chunks)
        ref["fusion_weight"] = model._xval_fusion_weight(torch.arange(C, device=dev), ref["mm_classifier"], ref["visual_classifer"],
                                                         ref["zero_shot_classifier"], 10.0).clone()
        counts_full = model.xval_counts.clone()
        if keep is not None:
            keep.update(ref=ref, counts_full=counts_full, eval_feat4cls=model.eval_feat4cls.clone())
    torch.cuda.synchronize()
    t_rec = time.perf_counter() - t_rec

    if args.overlap == 1 or (args.overlap < 0 and args.query_batch <= model.OVERLAP_MAX_BATCH):
        model._twin()
    emu = EmulatedPeers(R0, N)
    emu.peer_blocks = peer_blocks
    model._dist = emu
    c0, c1 = shard_range(C, R0, N)
    q0, q1 = shard_range(Q, R0, N)

    def queries(collect=False):
        out, outs = None, []
        for out in model.forward_batches((q[b:b + args.query_batch] for b in range(0, q.shape[0], args.query_batch)),
                                         stable_inputs=True, overlap=ov):
            if collect:
                outs.append(out.clone())
        return torch.cat(outs) if collect and outs else out

    def step():
        model.forward_prompt(loader, wait_files=False)
        out = queries()
        model.wait_files()
        return out

    step()                                                   # records this rank's own votes (EmulatedPeers.all_reduce)
    emu.peer_counts = counts_full - emu.local_counts
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps

    # phase split and the cross-validation step alone (untimed extra passes)
    tg = time.perf_counter(); model.forward_prompt(loader, wait_files=False); torch.cuda.synchronize(); tg = time.perf_counter() - tg
    model.wait_files()
    ti = time.perf_counter(); out = queries(collect=True); torch.cuda.synchronize(); ti = time.perf_counter() - ti
    own = torch.arange(c0, c1, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model._xval_fusion_weight(own, model.mm_classifier, model.visual_classifer, model.zero_shot_classifier, 10.0)
    e1.record()
    torch.cuda.synchronize()
    xval_ms = e0.elapsed_time(e1)
    h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    feats = eng.encode_image(q[:args.query_batch], normalize=True)
    h0.record()
    for _ in range(10):
        eng.fused_logits(feats, model.mm_classifier, model.visual_classifer, model.zero_shot_classifier, model.fusion_weight, "fusion")
    h1.record()
    torch.cuda.synchronize()
    head_us = h0.elapsed_time(h1) * 100.0
    # the emulated rank ends with the WHOLE job's bits: every classifier row (its own recomputed, the peers' gathered), the fusion weights
    # from the summed votes, and its query outputs finite
    eq = {k: bool(torch.equal(getattr(model, k), ref[k])) for k in ref}
    assert all(eq.values()), f"the sharded rank does not reproduce the whole job: {eq}"
    assert bool(torch.isfinite(out).all()) and bool(torch.equal(model.xval_counts, counts_full))
    n_rank = (c1 - c0) * S + (q1 - q0)
    if keep is not None:
        keep.update(local_counts=emu.local_counts, exemplars=buf[:(c1 - c0) * S], queries=q, out=out, classes=(c0, c1), model=model)
    roof = measure_roofline(eng, spec, args, dev, (c1 - c0) * S, q1 - q0)
    flops_img = eng.flops_per_image()
    lb = args.classes_per_batch * S
    fl = sum(eng.flops_executed(min(lb, (c1 - c0) * S - s0)) for s0 in range(0, (c1 - c0) * S, lb))
    fl += sum(eng.flops_executed(min(args.query_batch, q1 - q0 - s0)) for s0 in range(0, q1 - q0, args.query_batch))
    flops_run = fl / n_rank
    value = n_rank / dt
    xval_flops = 3 * 2.0 * (c1 - c0) * S * C * D
    model._dist = None
    return {
        "metric": f"images/sec, BASELINE.json configuration {args.preset}: {PRESETS[args.preset]['about']}",
        "value": round(value, 2), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000 * dt, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"OVMR classifier generation + fusion inference, {args.model}, {C} classes x {S} shots, rank {R0} of {N} emulated ranks: "
                               f"{c1 - c0} classes = {(c1 - c0) * S} exemplar images (batch {args.batch}) + {q1 - q0} of {Q} query images (batch {args.query_batch}), "
                               f"cross-validation of the local {(c1 - c0) * S} rows against all {C} classifier rows, n_ctx 2, tau 10",
                   "parallelism": f"class/query sharding over {N} ranks; this rank alone on one GPU, the all-gather / all-reduce served from the peers' recorded blocks and votes",
                   "preset": args.preset, "class_count_note": "the reference names no ImageNet21k-OVR class count (datasets/imagenet_21k_P.py:16-63); 10 000 chosen",
                   "encoder_reserve_images": args.batch, "encoder_chunk_images": eng.encode_chunk, "images_per_step": n_rank},
        "projection": {"projected": True, "emulated_world": N, "rank_timed": R0,
                       "projected_images_per_s_all_ranks": round((C * S + Q) / dt, 1),
                       "not_included": "RCCL all-gather / all-reduce over xGMI (device copies of recorded contributions here), barrier skew, start-up",
                       "collective_payload_bytes": {"all_gather_rows_per_rank": int(bound * (3 * D + n_ctx * D + 2) * 2), "all_reduce_counts": int(3 * 2 * C * 4)},
                       "peer_recording_s": round(t_rec, 1),
                       "rank_reproduces_whole_job_bits": eq},
        "roofline": roof,
        "cpu_baseline": None,
        "phases": {"generation_images_per_s": round((c1 - c0) * S / tg, 1), "generation_ms": round(1000 * tg, 2),
                   "inference_images_per_s": round((q1 - q0) / ti, 1), "inference_ms": round(1000 * ti, 2),
                   "xval_counts_fusion_weights_ms": round(xval_ms, 3),
                   "xval_tflops": round(xval_flops / (xval_ms * 1e-3) / 1e12, 1),
                   "fusion_head_us_per_query_batch": round(head_us, 1),
                   "encoder_tflops_e2e_algorithmic": round(value * flops_img / 1e12, 1),
                   "encoder_tflops_e2e_executed": round(value * flops_run / 1e12, 1),
                   "e2e_frac_of_fp16_mfma_peak": round(value * flops_run / 1e12 / 2500.0, 4)},
    }


def device_clip_state(spec, gen, dev):
    import math
    import torch

    def n(shape, std):
        return torch.randn(shape, generator=gen, device=dev) * std

    sd = {}
    W, P, L, T, E = spec.vision_width, spec.vision_patch_size, spec.vision_tokens, spec.transformer_width, spec.embed_dim
    sd["visual.class_embedding"] = n((W,), W ** -0.5)
    sd["visual.positional_embedding"] = n((L, W), W ** -0.5)
    sd["visual.proj"] = n((W, E), W ** -0.5)
    sd["visual.conv1.weight"] = n((W, 3, P, P), (3 * P * P) ** -0.5)
    for nm in ("visual.ln_pre", "visual.ln_post"):
        sd[nm + ".weight"] = torch.ones(W, device=dev)
        sd[nm + ".bias"] = torch.zeros(W, device=dev)

    def blocks(prefix, width, layers):
        a, p, f = width ** -0.5, (width ** -0.5) * ((2 * layers) ** -0.5), (2 * width) ** -0.5
        for i in range(layers):
            q = f"{prefix}{i}."
            sd[q + "attn.in_proj_weight"] = n((3 * width, width), a)
            sd[q + "attn.in_proj_bias"] = torch.zeros(3 * width, device=dev)
            sd[q + "attn.out_proj.weight"] = n((width, width), p)
            sd[q + "attn.out_proj.bias"] = torch.zeros(width, device=dev)
            sd[q + "mlp.c_fc.weight"] = n((4 * width, width), f)
            sd[q + "mlp.c_fc.bias"] = torch.zeros(4 * width, device=dev)
            sd[q + "mlp.c_proj.weight"] = n((width, 4 * width), p)
            sd[q + "mlp.c_proj.bias"] = torch.zeros(width, device=dev)
            for l in ("ln_1", "ln_2"):
                sd[q + l + ".weight"] = torch.ones(width, device=dev)
                sd[q + l + ".bias"] = torch.zeros(width, device=dev)

    blocks("visual.transformer.resblocks.", W, spec.vision_layers)
    sd["token_embedding.weight"] = n((spec.vocab_size, T), 0.02)
    sd["positional_embedding"] = n((spec.context_length, T), 0.01)
    sd["ln_final.weight"] = torch.ones(T, device=dev)
    sd["ln_final.bias"] = torch.zeros(T, device=dev)
    sd["text_projection"] = n((T, E), T ** -0.5)
    sd["logit_scale"] = torch.tensor(math.log(100.0), device=dev)
    blocks("transformer.resblocks.", T, spec.transformer_layers)
    return sd


def device_pl_state(spec, n_ctx, gen, dev):
    import torch
    D = spec.embed_dim
    sd = {}
    c = torch.randn((n_ctx, D), generator=gen, device=dev)
    sd["cls_token"] = c / c.norm(dim=-1, keepdim=True)
    a, p, f = D ** -0.5, (D ** -0.5) * ((2 * spec.agg_layers) ** -0.5), (2 * D) ** -0.5
    for i in range(spec.agg_layers):
        q = f"aggregator.resblocks.{i}."
        sd[q + "attn.in_proj_weight"] = torch.randn((3 * D, D), generator=gen, device=dev) * a
        sd[q + "attn.in_proj_bias"] = torch.zeros(3 * D, device=dev)
        sd[q + "attn.out_proj.weight"] = torch.randn((D, D), generator=gen, device=dev) * p
        sd[q + "attn.out_proj.bias"] = torch.zeros(D, device=dev)
        sd[q + "mlp.c_fc.weight"] = torch.randn((4 * D, D), generator=gen, device=dev) * f
        sd[q + "mlp.c_fc.bias"] = torch.zeros(4 * D, device=dev)
        sd[q + "mlp.c_proj.weight"] = torch.randn((D, 4 * D), generator=gen, device=dev) * p
        sd[q + "mlp.c_proj.bias"] = torch.zeros(D, device=dev)
        for l in ("ln_1", "ln_2"):
            sd[q + l + ".weight"] = torch.ones(D, device=dev)
            sd[q + l + ".bias"] = torch.zeros(D, device=dev)
    return sd


def encoder_chunks(eng, n_images, loader_batch):
    """Image counts of the encoder launch sequences a stream of `n_images` produces -> {images: sequences}: the loader hands over
    `loader_batch` images at a time, the engine plans each batch (Engine.encode_plan: one sequence if it fits the reserve, else chunks
    of ovmr_encode_chunk images with a sub-round remainder folded into the last one)."""
    out = {}
    for s0 in range(0, n_images, loader_batch):
        for b in eng.encode_plan(min(loader_batch, n_images - s0)):
            out[b] = out.get(b, 0) + 1
    return out


def measure_roofline(eng, spec, args, dev, n_exemplar_images, n_query_images):
    """Dominant kernel = the fp16 MFMA GEMM of the c_fc launches (ln_2 fold + bias + QuickGELU epilogue, N = 4W, K = W): one
    kernel instantiation, launched once per block and encoder launch sequence with M = images x tokens.  A step runs it at
    several M (full exemplar batches, the last partial one, query batches); every one of those shapes is timed live with HIP
    events on the stream the kernel is launched on (torch's current stream), on random operands, and `achieved` is the
    launch-weighted figure: sum of algorithmic FLOPs / sum of launch times over the launches of one step -- what
    `rocprofv3 --kernel-trace --stats` averages for that kernel name over the same command."""
    import ctypes
    import torch
    lib = eng.lib
    N, K, L = 4 * spec.vision_width, spec.vision_width, spec.vision_tokens
    chunks = encoder_chunks(eng, n_exemplar_images, args.classes_per_batch * args.shots)
    for b, n in encoder_chunks(eng, n_query_images, args.query_batch).items():
        chunks[b] = chunks.get(b, 0) + n
    layers = spec.vision_layers - 1                                   # the last block runs the CLS row only (other kernels)
    g = torch.Generator(device=dev).manual_seed(7)
    Mmax = max(chunks) * L
    A = (torch.randn((Mmax, K), generator=g, device=dev) * 0.5).half()
    Wt = (torch.randn((N, K), generator=g, device=dev) * K ** -0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    Cm = torch.empty((Mmax, N), dtype=torch.float16, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lb, lg = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    st = torch.zeros((Mmax, K // 256, 2), device=dev)
    st[:, 0, 1] = float(K)                                            # mean 0, variance 1
    # the launch in its context: the product path runs c_fc right behind out_proj, which has just written the residual
    # stream and the row statistics c_fc reads -- so the probe launches that pair and times ONLY the c_fc launch (one
    # event pair per launch).  Back-to-back repetitions of c_fc alone find their A operand warm in the Infinity Cache and
    # read 3-4 % faster than the per-kernel average `rocprofv3 --kernel-trace --stats` reports for the real step.
    A2 = (torch.randn((Mmax, K), generator=g, device=dev) * 0.01).half()
    W2 = (torch.randn((K, K), generator=g, device=dev) * K ** -0.5).half()
    b2 = torch.zeros(K, dtype=torch.float16, device=dev)
    shapes, tot_t, tot_f, tot_n = [], 0.0, 0.0, 0
    gv = args.gemm + (0 if args.gelu_exact else 100)                 # kernel + QuickGELU form, as the engine launches c_fc
    for bsz in sorted(chunks, reverse=True):
        M = bsz * L
        if args.ln_fold:   # the c_fc launch of the product path: ln_2 folded into the epilogue (csrc/common.h EPI_LN_BIAS_QGELU)
            before = lambda: lib.ovmr_debug_gemm(0, args.gemm, p(A2), p(W2), p(b2), p(A), p(st), p(A), M, K, K, K, 3, 1.0, 0, 0, s())
            launch = lambda: lib.ovmr_debug_gemm(0, gv, p(A), p(Wt), p(lb), p(st), p(lg), p(Cm), M, N, K, N, 7, 1.0, 0, 0, s())
        else:
            before = lambda: 0
            launch = lambda: lib.ovmr_debug_gemm(0, gv, p(A), p(Wt), p(b), None, None, p(Cm), M, N, K, N, 2, 1.0, 0, 0, s())
        for _ in range(3):
            assert before() == 0 and launch() == 0
        torch.cuda.synchronize()
        # repetitions in proportion to the step's own mix, so that these extra launches do not shift the per-kernel average
        # that `rocprofv3 --stats` reports for the same command
        reps = max(3, round(30 * chunks[bsz] / max(chunks.values())))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            before()
            e0.record()
            launch()
            e1.record()
        torch.cuda.synchronize()
        us = sum(e0.elapsed_time(e1) for e0, e1 in ev) * 1000.0 / reps
        n = chunks[bsz] * layers
        shapes.append({"M": M, "launches_per_step": n, "avg_launch_us": round(us, 2), "tflops": round(2.0 * M * N * K / us / 1e6, 1)})
        tot_t += n * us
        tot_f += n * 2.0 * M * N * K
        tot_n += n
    achieved = tot_f / (tot_t * 1e-6) / 1e12
    Mfull = max(chunks) * L
    return {"bound": "mfma",
            "kernel": f"gemm_f16 variant {args.gemm}, c_fc launches N={N} K={K} ({'ln_2 fold + ' if args.ln_fold else ''}bias + QuickGELU{'' if args.gelu_exact else ', one rounding'}), "
                      f"launch-weighted over the M of one step",
            "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4),
            "avg_launch_us": round(tot_t / tot_n, 2), "flops_per_launch": tot_f / tot_n, "launches_per_step": tot_n, "shapes": shapes,
            **pmc_traffic(gv, Mfull, N, args.batch, 7 if args.ln_fold else 2)}


def pmc_traffic(variant, M, N, batch, epi=2):
    """Counters of that kernel from the committed PMC summary (tools/pmc_gemm.sh: separate rocprofv3 --pmc passes over the
    product library, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): HBM-side bytes per launch and rate,
    matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 pipes)), LDS bank-conflict fraction, held clock.
    A summary is used only if it was taken on the SAME kernel sources (ovmr_amd.build.source_sha16) -- otherwise null."""
    import glob
    from ovmr_amd.build import source_sha16
    sha = source_sha16()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_gemm_v{variant}.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("batch") != batch or d.get("kernel_source_sha16") != sha:
                continue
            for key, c in d["kernels"].items():
                for bm in (256, 128):
                    grid = ((M + bm - 1) // bm) * ((N + 255) // 256) * 512
                    if key.endswith(f"grid={grid}") and f"<{epi}," in key and "hbm_bytes_per_launch" in c:
                        return {"traffic": c["hbm_bytes_per_launch"], "traffic_unit": f"bytes/launch at M={M}",
                                "traffic_source": os.path.relpath(f, ROOT), "algorithmic_bytes": 2.0 * (M * (N // 4) + N * (N // 4) + M * N),
                                "hbm_gbps": c.get("hbm_gbps"), "hbm_frac_of_8tbps": round(c["hbm_gbps"] / 8000.0, 4) if c.get("hbm_gbps") else None,
                                "mfma_busy_frac": c.get("mfma_busy_frac"), "lds_conflict_frac": c.get("lds_conflict_frac"),
                                "clock_ghz_under_pmc": c.get("clock_ghz"), "kernel_source_sha16": sha}
        except Exception:
            pass
    return {"traffic": None, "kernel_source_sha16": sha}


def _cpu_worker(widx, nproc, threads, spec_name, sd16, pl, tok, shots, classes, reps, n_ctx, barrier, queue, kind="ovmr"):
    """One CPU worker.  kind "ovmr": `classes` classes x `shots` exemplars of its own slice through the oracle's forward_prompt + fused
    inference on 4 queries; kind "zeroshot": the oracle's zeroshot_logits (trainers/zsclip.py:55-60) on `classes` x `shots` images against
    the prompts' text features (encoded once per repetition).  1 warm-up + reps[0] timed repetitions in fp32 (what the reference computes
    after clip_model.float()) and 1 + reps[1] in fp16 (the precision its OVMR path runs in as shipped).  All workers start each repetition
    together."""
    import torch
    from oracle import ovmr_oracle as O
    from ovmr_amd import synth
    torch.set_num_threads(threads)
    spec = synth.SPECS[spec_name]
    R = spec.image_resolution
    sd32 = {k: v.float() for k, v in sd16.items()}
    g = torch.Generator().manual_seed(3 + widx)
    img = torch.randn((classes * shots, 3, R, R), generator=g)
    q = torch.randn((4, 3, R, R), generator=g)
    labels = torch.arange(classes).repeat_interleave(shots)
    mytok = tok[widx * classes:(widx + 1) * classes] if kind == "ovmr" else tok

    def job(sd, prec, x, qx):
        if kind == "zeroshot":
            tf = O.l2_normalize(O.encode_text(mytok, sd))
            for s0 in range(0, x.shape[0], 32):
                O.zeroshot_logits(x[s0:s0 + 32], tf, sd)
            return
        nc = x.shape[0] // shots                                     # (the fp16 leg runs half the classes)
        r = O.forward_prompt(x, labels[:nc * shots], mytok[:nc], sd, pl, n_ctx, 10.0, max(1, 64 // shots), prec)
        qf = O.l2_normalize(O.encode_image(qx, sd))
        O.inference_logits(qf, r["mm_classifier"].to(qf.dtype), r["vision_classifier"].to(qf.dtype), r["text_classifier"].to(qf.dtype),
                           r["fusion_weight"], sd["logit_scale"].float().exp(), "fusion")

    times32, times16 = [], []
    with torch.no_grad():
        c16 = max(1, classes // 2) * shots                           # fp16 is the slower precision on these hosts (it only has to be shown slower): half the sample
        for sdx, prec, x, qx, out, n in ((sd32, "fp32", img, q, times32, reps[0]), (sd16, "fp16", img[:c16].half(), q.half(), times16, reps[1])):
            for rep in range(n + 1):                                 # rep 0 = warm-up
                barrier.wait(timeout=600)
                t0 = time.perf_counter()
                job(sdx, prec, x, qx)
                if rep:
                    out.append(time.perf_counter() - t0)
    queue.put((widx, times32, times16))


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports the
    machine, not the container's share: a 256-thread pool on a 32-CPU quota runs ~300x slower than a 16-thread one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota)))
    return n, quota


def cpu_baseline(spec, sd, pl, tok, args, n_ctx, kind="ovmr"):
    """The oracle (torch-CPU port of the reference path) on the CPUs THIS PROCESS MAY USE -- the affinity mask capped by the cgroup
    quota (`effective_cpus`: 16 of the 256 hardware threads on the GPU boxes), not the whole host: usable // threads worker processes x
    `--cpu-threads` threads (a single 256-thread pool is far slower than 16-thread pools on this path), disjoint class slices of
    `--cpu-sample-classes` classes x shots (64 images + 4 queries at the defaults), 1 warm-up + `--cpu-reps` timed repetitions in fp32
    and 1 + 2 in fp16.  images/s of a repetition = sum over workers of images / that repetition's time, all workers running together;
    value = the MEDIAN repetition of the faster precision, min / max beside it."""
    import statistics
    import torch
    import torch.multiprocessing as mp
    from oracle import ovmr_oracle as O
    cores, quota = effective_cpus()
    threads = max(1, min(args.cpu_threads, cores))
    nproc = max(1, cores // threads)
    if args.cpu_procs > 0:
        nproc = args.cpu_procs
    S, Cs = args.shots, args.cpu_sample_classes
    reps = (max(1, args.cpu_reps), max(1, min(2, args.cpu_reps)))
    sd16 = O.convert_weights({k: v.detach().float().cpu() for k, v in sd.items()}, "fp16")
    for v in sd16.values():
        v.share_memory_()
    cpu_pl = {k: v.detach().float().cpu().share_memory_() for k, v in (pl or {}).items()}
    ctx = mp.get_context("spawn")                                  # this process has initialised the GPU: no fork
    barrier, queue = ctx.Barrier(nproc), ctx.Queue()
    t_all = time.perf_counter()
    wtok = tok[:nproc * Cs].clone() if kind == "ovmr" else tok.clone()
    procs = [ctx.Process(target=_cpu_worker, args=(i, nproc, threads, spec.name, sd16, cpu_pl, wtok, S, Cs, reps, n_ctx, barrier, queue, kind))
             for i in range(nproc)]
    saved_env = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update(OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))   # children size their pools at import
    try:
        for p in procs:
            p.start()
    finally:
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    res, deadline = [], time.time() + args.cpu_timeout
    try:
        while len(res) < nproc:
            try:
                res.append(queue.get(timeout=5))
            except Exception:
                dead = [p for p in procs if p.exitcode not in (None, 0)]
                if dead or time.time() > deadline:
                    raise RuntimeError(f"{len(dead)} CPU worker(s) failed" if dead else "CPU baseline timed out")
    except RuntimeError as e:
        for p in procs:
            if p.is_alive():
                p.terminate()
        return {"value": None, "unit": "images/s", "cores": nproc * threads, "kind": "port", "sample": f"not measured: {e}"}
    for p in procs:
        p.join(timeout=30)
    t_all = time.perf_counter() - t_all
    n_img = Cs * S + (4 if kind == "ovmr" else 0)
    # per repetition: all workers' images over that repetition's time on each worker (they start it together)
    rate32 = [sum(n_img / t32[r] for _, t32, _ in res) for r in range(reps[0])]
    n_img16 = max(1, Cs // 2) * S + (4 if kind == "ovmr" else 0)     # (the fp16 leg's half sample)
    rate16 = [sum(n_img16 / t16[r] for _, _, t16 in res) for r in range(reps[1])]
    v32, v16 = statistics.median(rate32), statistics.median(rate16)
    best, best_name = (rate32, "fp32 math on fp16-rounded weights") if v32 >= v16 else (rate16, "fp16")
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    what = (f"{Cs} class(es) x {S} shots generation + 4 fusion queries ({n_img} images) on its own class slice" if kind == "ovmr" else
            f"zero-shot logits (trainers/zsclip.py:55-60) of {n_img} images in batches of 32 against {tok.shape[0]} prompts' text features")
    return {"value": round(max(v16, v32), 2), "unit": "images/s", "cores": nproc * threads, "kind": "port",
            "min": round(min(best), 2), "median": round(statistics.median(best), 2), "max": round(max(best), 2), "repetitions": len(best),
            "precision_reported": best_name,
            "sample": f"{nproc} worker processes x {threads} threads, each {what}, 1 warm-up + {reps[0]} timed repetitions in fp32 / {reps[1]} in fp16 (on half the sample: {n_img16} images), "
                      f"median repetition; {nproc * threads} threads = the {cores} CPUs this process may use (cgroup quota) of {os.cpu_count()} host threads: "
                      f"fp32 math on fp16-rounded weights {v32:.1f} img/s (min {min(rate32):.1f}, max {max(rate32):.1f}), fp16 {v16:.1f} img/s, faster reported; "
                      f"{t_all:.0f} s wall incl. process start",
            "host_cores": os.cpu_count(), "usable_cpus": cores, "cgroup_cpu_quota": quota, "cpu_model": model, "processes": nproc, "threads_per_process": threads,
            "fp16_images_per_s": round(v16, 2), "fp32_images_per_s": round(v32, 2),
            "fp32_repetitions_images_per_s": [round(x, 2) for x in rate32], "fp16_repetitions_images_per_s": [round(x, 2) for x in rate16]}


def cpu_baseline_zeroshot(spec, sd, ids, args):
    """Configuration 1's CPU leg: the oracle's zeroshot_logits on `--cpu-sample-classes` x 16 images per worker (64 at the default)."""
    import copy
    a = copy.copy(args)
    a.shots = 16
    return cpu_baseline(spec, sd, None, ids, a, 2, kind="zeroshot")


if __name__ == "__main__":
    main()
