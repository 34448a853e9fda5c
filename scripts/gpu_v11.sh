#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_eval_gpu.py -m gpu -q -k "syevj or fid or fp64 or kid or evaluate" 2>&1 | tail -4
timeout 300 python scripts/debug_jacobi.py 2>&1 | grep -v amdgpu.ids | tail -9
for i in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('fid10k', d['fid10k']['wall_s'], d['fid10k']['split_s'], d['fid10k']['fid'])"
done
