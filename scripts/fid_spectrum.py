"""Diagnostic: spectrum of the covariance matrices the FID-10k leg of bench.py factorises (synthetic
reference images, seeded Inception weights, untrained generator): dead features (exactly zero
variance), extreme eigenvalues before / after deflating them (host LAPACK, fp64 -- diagnostic only).
usage: python scripts/fid_spectrum.py [num_examples]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compare_gan_amd import eval_gan_lib, eval_utils
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.metrics import fid_score
from tests import gan_util as U

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda:0")
gan, options, dataset = U.build_product("resnet_cifar10.gin", 64, dev, seed=1)
captured = {}
orig = fid_score.frechet_distance


def spy(real, gen, device="cuda:0"):
    captured["real"], captured["gen"] = real, gen
    return orig(real, gen, device=device)


fid_score.frechet_distance = spy
res = eval_gan_lib.evaluate_gan(gan, [fid_score.FIDScoreTask()], num_averaging_runs=1, num_test_examples=n)
print("fid", res["fid_score_mean"], "solver", fid_score.LAST_SOLVER, fid_score.LAST_NEWTON)
for name in ("real", "gen"):
    x = captured[name]
    m, s = K.mean_cov_f64(x.float().contiguous())
    s = s.cpu().numpy()
    dg = np.diag(s)
    dead = dg == 0.0
    print(name, "n", x.shape[0], "d", x.shape[1], "dead (diag == 0):", int(dead.sum()), "diag < 1e-12:",
          int((dg < 1e-12).sum()), "diag min nonzero %.3e max %.3e" % (dg[~dead].min(), dg.max()))
    w = np.linalg.eigvalsh(s)
    print("  full spectrum: min %.3e  #<1e-10: %d  #<1e-9: %d  max %.3e" % (w.min(), int((w < 1e-10).sum()),
                                                                            int((w < 1e-9).sum()), w.max()))
    live = ~dead
    wl = np.linalg.eigvalsh(s[np.ix_(live, live)])
    print("  deflated (%d): min %.3e  #<1e-10: %d  #<1e-9: %d  quantiles 1%% %.3e 10%% %.3e 50%% %.3e" % (
        int(live.sum()), wl.min(), int((wl < 1e-10).sum()), int((wl < 1e-9).sum()),
        np.quantile(wl, 0.01), np.quantile(wl, 0.1), np.quantile(wl, 0.5)))
