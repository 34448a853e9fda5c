#!/bin/bash
# Visit v3: the parity tests added for VERDICT r01 item 6 (real sizes) and the not-unrolled step.
TAG=${1:-v3}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "full_size or statistics_groups" 2>&1 | tail -25 > gpurun_out/${TAG}_tests_k.txt
cat gpurun_out/${TAG}_tests_k.txt
timeout 1500 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -s --durations=8 -k "train_steps or benchmark_batch or full_width or biggan_deep or not_unrolled or joint_gen or batched_generator" 2>&1 | tail -60 > gpurun_out/${TAG}_tests_m.txt
cat gpurun_out/${TAG}_tests_m.txt
