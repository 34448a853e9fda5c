#!/bin/bash
# round 3: RGB stem forward with the window prefetch: parity, D-step A/B over the workgroup count
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wstem or rgb or stem or pool_fused" 2>&1 | tail -4
for w in 1024; do
  CGAMD_WSTEM_WGS=$w timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep > gpurun_out/r3y_$w.json 2> gpurun_out/r3y_$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3y_$w.json").read().strip().splitlines()[-1])
l=d["resnet128_dstep"]
print("wgs $w cifar ms", d["ms_per_step"], "dstep ms", l["ms"], {k:(round(v["ms_per_step"],3)) for k,v in l["kernels"].items() if "stem" in k})
PY
done
