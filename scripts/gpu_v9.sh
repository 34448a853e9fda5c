#!/bin/bash
TAG=${1:-v9}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_s3gan_gpu.py tests/test_modular_gan_matrix_gpu.py tests/test_architectures_gpu.py tests/test_modular_gan_gpu.py -m gpu -q -s --durations=5 -k "s3gan or matrix or single_training or disc_iters or architecture or resnet_stl or self_modulated" > gpurun_out/${TAG}_tests.txt 2>&1
grep -n "max / mean\|worst\|s3gan\|passed\|failed\|^E  \|Error\|^FAILED" gpurun_out/${TAG}_tests.txt | head -60
tail -12 gpurun_out/${TAG}_tests.txt
