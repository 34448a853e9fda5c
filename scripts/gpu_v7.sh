#!/bin/bash
TAG=${1:-v7}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "inception_preprocess_and_pool" 2>&1 | tail -3
for v in 256 0; do
  echo "== CGAMD_JACOBI_BLOCK_MIN=$v"
  CGAMD_JACOBI_BLOCK_MIN=$v timeout 300 python scripts/debug_jacobi.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/${TAG}_jacobi_debug.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('fid10k', d['fid10k']['wall_s'], d['fid10k']['split_s'], d['fid10k']['fid'])"
