#!/bin/bash
# round 3: small-linear / thin kernels (tests); BigGAN leg with the 128x128-tile threshold of the
# one-tap kernel lowered (4x4x1536 layers stream 42 MB of weights per 64-pixel tile row)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "thin or lin_ or gemm_1x1" 2>&1 | tail -4
for t in 513 150; do
  rm -f gpurun_out/r3t_launches_$t.txt
  CGAMD_CONV_T128_MIN=$t CGAMD_PROF_LOG=$R/gpurun_out/r3t_launches_$t.txt timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fid --no-roofline --legs biggan128 > gpurun_out/r3t_bench_$t.json 2> gpurun_out/r3t_bench_$t.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3t_bench_$t.json").read().strip().splitlines()[-1])
l=d["biggan128"]; print("T128_MIN $t: cifar", d["ms_per_step"], "biggan ms", l["ms"], "conv eager", l["conv_kernel_ms_eager"])
PY
done
