#!/usr/bin/env python
"""How well-conditioned is the generator loss of one unrolled resnet_cifar10.gin step?  It is taken
AFTER disc_iters sign-like Adam updates of D (first step: update = -lr * sign(g)), so weights whose
gradient is rounding noise move by +-lr on either side.  For several seeds, batch 8:
  exact    : the fp64 oracle
  bf16     : the same oracle with every tensor the HIP path stores in bf16 snapped to the bf16 grid
  product  : the HIP path (only with a GPU)
Printed: the D losses' and the generator loss' relative distances.  CPU-only without a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import arch_ops as oops
from tests import gan_util as U

config, bsz = "resnet_cifar10.gin", int(os.environ.get("BSZ", "8"))
seeds = [int(s) for s in (sys.argv[1:] or ["1", "2", "3", "4", "5", "6"])]
have_gpu = torch.cuda.is_available()
print("# %s batch %d, one unrolled step (5 D + 1 G); relative distances |a - b| / max(1, |b|)" % (config, bsz))
print("# seed  d_loss[4] exact   g_loss exact    g: bf16-exact   d(max): bf16-exact" +
      ("   g: product-bf16  g: product-exact  d(max): product-bf16" if have_gpu else ""))
worst = 0.0
for seed in seeds:
    rng = np.random.RandomState(500 + seed)
    images = rng.uniform(size=(6 * bsz, 32, 32, 3)).astype(np.float32)
    subs = [{"images": torch.from_numpy(images[i * bsz:(i + 1) * bsz]).double(),
             "z": U.host_uniform((bsz, 128), "z/%d" % i, -1.0, 1.0, seed, 0).double()} for i in range(6)]
    if have_gpu:
        dev = torch.device("cuda:0")
        gan, options, dataset = U.build_product(config, bsz, dev, seed=seed)
        vs_b = U.mirror_to_oracle(gan, emulate_bf16=True)
        vs_x = U.mirror_to_oracle(gan, emulate_bf16=False)
        out = gan.train_step(torch.from_numpy(images).to(dev),
                             torch.ones((6 * bsz,), dtype=torch.int32, device=dev))
        d_p, g_p = [float(x) for x in out["d_losses"]], float(out["g_loss"])
    else:
        vs_b = oops.VarStore(dtype=torch.float64, seed=seed, emulate_bf16=True)
        vs_x = oops.VarStore(dtype=torch.float64, seed=seed, emulate_bf16=False)
    d_b, g_b = U.build_oracle(config, vs_b).train_step(subs)
    d_x, g_x = U.build_oracle(config, vs_x).train_step(subs)
    rel = lambda a, b: abs(a - b) / max(1.0, abs(b))
    line = "%5d   %12.6f   %12.6f   %12.2e   %12.2e" % (
        seed, d_x[4], g_x, rel(g_b, g_x), max(rel(a, b) for a, b in zip(d_b, d_x)))
    worst = max(worst, rel(g_b, g_x))
    if have_gpu:
        srel = lambda a, b: (a - b) / max(1.0, abs(b))
        line += "   %+12.2e   %+12.2e   %12.2e" % (srel(g_p, g_b), srel(g_p, g_x),
                                                   max(rel(a, b) for a, b in zip(d_p, d_b)))
        worst = max(worst, rel(g_p, g_b))
    print(line, flush=True)
print("# worst generator-loss distance: %.2e" % worst)
