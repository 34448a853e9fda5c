#!/bin/bash
# round 3: register-resident-weight 64->64 kernel with pooled epilogue / up-sampled input
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_pool_fused or hconv_all" > gpurun_out/r3j_tests.txt 2>&1
tail -4 gpurun_out/r3j_tests.txt
for m in 1 0 1 0; do
  CGAMD_HCONV_RW_FUSED=$m timeout 300 python scripts/run_leg.py resnet128_dstep 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rw_fused=$m dstep ms', d['ms'], 'hconv64', d['kernels'].get('hconv_kernel<64, *>'))"
done
