#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_eval_gpu.py -m gpu -q -x -k "accumulator or ema_weights" 2>&1 | tail -25
