#!/bin/bash
# round 3: parity at the benchmark sizes (device-resident fp64 oracle), full-size fused op cases
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
CGAMD_TEST_REPORT=1 timeout 1500 python -m pytest tests/test_modular_gan_gpu.py tests/test_kernels_gpu.py -q -m gpu -s -k "benchmark_batch or full_size" > gpurun_out/r3i_tests.txt 2>&1
grep -E "passed|failed|PASSED|FAILED|Error|worst|d_loss|g_loss|max / mean" gpurun_out/r3i_tests.txt | head -60
