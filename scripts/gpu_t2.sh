#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -s -k "penalties_against_oracle or wgangp" 2>&1 | grep -E "penalty |passed|failed|Error|assert" | head -40
