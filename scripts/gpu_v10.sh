#!/bin/bash
# Re-run of the two tests that failed in the closing visit (in suite order, to catch state leaks),
# then the dispatch-threshold sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_data_parallel_gpu.py tests/test_modular_gan_gpu.py -m gpu -q -x -k "force_dp or train_steps_resnet_cifar[8] or wgangp or penalties" 2>&1 | tail -6 | tee gpurun_out/v10_tests.txt
bash scripts/gpu_sweep.sh sweep
