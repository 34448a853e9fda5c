#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CGAMD_TEST_REPORT=1 timeout 900 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -s -k "wgangp_step_resnet5" 2>&1 | grep -E "cos |passed|failed|worst|wgangp d_loss" | sort | head -60
echo "--- unfused"
CGAMD_FUSED_POOL=0 CGAMD_TEST_REPORT=1 timeout 900 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -s -k "wgangp_step_resnet5" 2>&1 | grep -E "cos |passed|failed|worst|wgangp d_loss" | sort | head -12
