#!/bin/bash
# round 3: conv / model tests with the small-map kernels + bench A/B (deferral on / off, sconv on / off)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modular_gan_gpu.py -x -q -m gpu > gpurun_out/r3d_tests.txt 2>&1
tail -5 gpurun_out/r3d_tests.txt
for cfg in "1 1 1" "0 0 0" "1 1 0"; do
  set -- $cfg
  CGAMD_SCONV=$1 CGAMD_SWGRAD=$2 CGAMD_DEFER_WGRAD=$3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fid > gpurun_out/r3d_bench_$1$2$3.json 2> gpurun_out/r3d_bench_$1$2$3.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3d_bench_$1$2$3.json').read().strip().splitlines()[-1])
print('sconv=$1 swgrad=$2 defer=$3: cifar ms', d['ms_per_step'], 'dstep', d.get('resnet128_dstep',{}).get('ms'), 'dstep_gp', d.get('resnet128_dstep_gp',{}).get('ms'), 'biggan', d.get('biggan128',{}).get('ms'))
PY
done
