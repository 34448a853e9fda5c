#!/bin/bash
# Round 2, visit B: parity of the halo-staged conv kernel + A/B timing against the one-tap kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
CGAMD_HCONV_MIN=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "test_gconv_forward_adjoint_wgrad or test_gconv_gates_residual" 2>&1 | tail -15 > gpurun_out/r2b_tests.txt
cat gpurun_out/r2b_tests.txt
for v in 0 1; do
  echo "== CGAMD_HCONV=$v" >> gpurun_out/r2b_convs.txt
  BENCH_NO_WGRAD=1 CGAMD_HCONV=$v timeout 300 python scripts/bench_convs.py hc >> gpurun_out/r2b_convs.txt 2>&1
done
cat gpurun_out/r2b_convs.txt
