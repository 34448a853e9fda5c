#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modular_gan_gpu.py -q -m gpu -k "wgangp_step_resnet5" > gpurun_out/r3o_tests.log 2>&1
tail -5 gpurun_out/r3o_tests.log
grep -n "assert\|Error\|cosine\|worst" gpurun_out/r3o_tests.log | head -30
