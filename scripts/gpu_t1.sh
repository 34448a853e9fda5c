#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CGAMD_TEST_REPORT=1 timeout 900 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -s -k "wgangp_step_resnet5 and exact" 2>&1 | grep -E "cos 0.8|passed|failed|AssertionError: |wgangp d_loss" | head -30
