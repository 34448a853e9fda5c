#!/bin/bash
# round 3: spectrum of the FID-10k covariances (synthetic Inception weights): is a GEMM-only square
# root certifiable there?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python - <<'PY' 2>&1 | grep -v Warning | tail -30
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from tests import gan_util as U
from compare_gan_amd import eval_gan_lib, eval_utils
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.metrics import fid_score as F
dev = torch.device('cuda:0')
gan, options, dataset = U.build_product('resnet_cifar10.gin', 64, dev, seed=3)
eval_utils.get_inception(dev)
cap = {}
orig = F.frechet_distance
def spy(real, gen, device="cuda:0"):
    cap['real'], cap['gen'] = real, gen
    return orig(real, gen, device=device)
F.frechet_distance = spy
res = eval_gan_lib.evaluate_gan(gan, [F.FIDScoreTask()], num_averaging_runs=1)
for name in ('real', 'gen'):
    x = F._activations_on_device(cap[name], dev)
    m, sigma = K.mean_cov_f64(x)
    w, _ = K.syevj_f64(sigma.clone(), max_sweeps=60, tol=1e-12)
    w = np.sort(w.cpu().numpy())[::-1]
    print(name, 'n', x.shape, 'trace', w.sum(), 'max', w[0])
    edges = [1e0, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9, 1e-10, 1e-12, 1e-14, 1e-16, 1e-18, 0, -1]
    for hi, lo in zip(edges[:-1], edges[1:]):
        print('   eigenvalues in (%g, %g]: %d' % (lo, hi, int(((w > lo) & (w <= hi)).sum())))
    colvar = x.double().var(dim=0)
    print('   zero-variance features:', int((colvar == 0).sum()), ' var<1e-12:', int((colvar < 1e-12).sum()))
PY
