#!/bin/bash
# round 3: hconv 64-channel tiles on small grids, A/B on the cifar shapes + steps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for m in 0 160 300; do
  echo "== CGAMD_HCONV_BN64_MAX=$m"
  CGAMD_HCONV_BN64_MAX=$m BENCH_NO_WGRAD=1 timeout 300 python scripts/bench_convs.py fixed 2>&1 | grep -v amdgpu.ids | awk '{print $1, $2, $3}'
done
for m in 0 160 0 160; do
  CGAMD_HCONV_BN64_MAX=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fid --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bn64_max=$m cifar ms', d['ms_per_step'])"
done
