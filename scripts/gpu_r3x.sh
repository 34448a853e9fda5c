#!/bin/bash
# round 3: all-phase up-convolution kernel: parity, D-step / cifar A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
CGAMD_HCONV_MIN=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "(forward_adjoint and (up or hup)) or fused_batch_norm" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "variants and (hconv_all or no_hup)" 2>&1 | tail -4
for h in 0 1; do
  CGAMD_HUP=$h timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep > gpurun_out/r3x_hup$h.json 2> gpurun_out/r3x_hup$h.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3x_hup$h.json").read().strip().splitlines()[-1])
l=d["resnet128_dstep"]
print("hup $h cifar ms", d["ms_per_step"], "dstep ms", l["ms"], {k:(round(v["ms_per_step"],3)) for k,v in l["kernels"].items() if "hconv" in k})
PY
done
