#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for rep in 1 2; do for v in 0 1; do
  echo "== CGAMD_FUSED_BN=$v"
  CGAMD_FUSED_BN=$v timeout 300 python scripts/run_leg.py resnet128_dstep 20 2>/dev/null | tail -1 | python -c "import json,sys; L=json.load(sys.stdin); print('dstep', L['ms'], L['frac'], 'conv', L['conv_kernel_ms_eager'])"
  CGAMD_FUSED_BN=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-fid --no-legs 2>/dev/null | tail -1 | cut -c1-140
done; done
