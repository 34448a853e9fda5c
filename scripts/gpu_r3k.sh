#!/bin/bash
# round 3: Newton-Schulz square roots for the FID statistics: tests + FID-10k wall-clock
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q -m gpu -k "fid" > gpurun_out/r3k_tests.txt 2>&1
tail -15 gpurun_out/r3k_tests.txt
timeout 600 python - <<'PY' 2>&1 | grep -v Warning | tail -20
import sys, time, torch
sys.path.insert(0, '.')
import bench
from tests import gan_util as U
from compare_gan_amd import eval_gan_lib, eval_utils
from compare_gan_amd.metrics import fid_score as F
dev = torch.device('cuda:0')
gan, options, dataset = U.build_product('resnet_cifar10.gin', 64, dev, seed=3)
eval_utils.get_inception(dev)
from compare_gan_amd.metrics import inception_score as I
res = eval_gan_lib.evaluate_gan(gan, [F.FIDScoreTask()], num_averaging_runs=1)
print('fid', res['fid_score_mean'], F.LAST_SOLVER)
for r in F.LAST_NEWTON: print(r)
print(eval_gan_lib.LAST_TIMING)
PY
