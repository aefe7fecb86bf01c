#!/bin/bash
# the fused in_proj + attention kernel INSIDE the encoder (experiment build, option fuse_qkv_attn): features bit-equal, then the headline job
# with and without, alternating
export OVMR_HIP_LIB=$(pwd)/ovmr_amd/lib/libovmr_hip_exp.so
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
from ovmr_amd import modules, synth
dev = torch.device("cuda:0")
spec = synth.SPECS["ViT-B/16"]
gen = torch.Generator(device=dev).manual_seed(1)
cm = modules.CLIPModel(bench.device_clip_state(spec, gen, dev), spec, str(dev))
e = cm.engine(2); e.load_state_dict({}, bench.device_pl_state(spec, 2, gen, dev)); e._pl_loaded = True; e.finalize(775, 64, 1024)
for B in (64, 300, 775):
    img = torch.randn((B, 3, 224, 224), generator=gen, device=dev).half()
    e.set_option("fuse_qkv_attn", 0); a = e.encode_image(img).clone()
    e.set_option("fuse_qkv_attn", 1); b = e.encode_image(img).clone()
    print("images", B, "features bit-equal with the fused launch:", bool(torch.equal(a, b)), flush=True)
PY
for rep in 1 2; do for f in 0 1; do for a in 0 32; do
  [ $f = 0 ] && [ $a = 32 ] && continue
  echo "fuse $f abl $a: $(OVMR_FQ_ABL=$a timeout 300 python bench.py --fuse-qkv-attn $f --steps 3 --warmup 1 --no-cpu-baseline --presets 0 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done; done
