#!/bin/bash
# round 3: thin 1x1 kernel + split Adam chunks (tests), D-step with the BN=64 grid threshold swept
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "thin or adam or gemm_1x1 or grey" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_modular_gan_gpu.py -x -q -m gpu -k "resnet5 or cifar" 2>&1 | tail -4
for bn in 160 256 512; do
  CGAMD_HCONV_BN64_MAX=$bn timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep > gpurun_out/r3r_bn$bn.json 2> gpurun_out/r3r_bn$bn.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3r_bn$bn.json").read().strip().splitlines()[-1])
l=d["resnet128_dstep"]
print("bn64max $bn cifar ms", d["ms_per_step"], "dstep ms", l["ms"], {k:(round(v["ms_per_step"],3)) for k,v in l["kernels"].items()})
PY
done
