"""Per-geometry timing of the convolution launches of ONE leg of bench.py (diagnostic, eager launches
with HIP events around every gconv / gwgrad call).
usage: python scripts/prof_leg_shapes.py [resnet128_dstep|resnet128_dstep_gp|biggan128|cifar] [batch]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import gan_util as U
from compare_gan_amd.hip import kernels as K

dev = torch.device("cuda:0")
leg = sys.argv[1] if len(sys.argv) > 1 else "resnet128_dstep"
LEGS = {
    "resnet128_dstep": ("resnet_lsun-bedroom128.gin", ("penalty.fn = @no_penalty",), 64, "dstep"),
    "resnet128_dstep_gp": ("resnet_lsun-bedroom128.gin", (), 64, "dstep"),
    "biggan128": ("biggan_imagenet128.gin", (), 64, "step"),
    "cifar": ("resnet_cifar10.gin", (), 64, "step"),
}
config, binds, bsz, mode = LEGS[leg]
if len(sys.argv) > 2:
    bsz = int(sys.argv[2])
records = []


def wrap(name, fn):
    def f(geom, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(geom, *a, **kw)
        e1.record()
        taps = geom.kh * geom.kw / float(geom.U * geom.U)
        fl = 2.0 * geom.N * geom.Ho * geom.Wo * taps * geom.Ci * geom.Co
        records.append((name, geom.key(), fl, e0, e1))
        return out
    return f


K.gconv = wrap("gconv", K.gconv)
K.gwgrad = wrap("gwgrad", K.gwgrad)
gan, options, dataset = U.build_product(config, bsz, dev, seed=3, bindings=binds)
nsub = 1 if mode == "dstep" else options["disc_iters"] + 1
images, labels = next(dataset.train_batches(bsz * nsub, seed=547))
images = torch.from_numpy(images).to(dev)
labels = torch.from_numpy(labels).to(dev)
step = gan.disc_step if mode == "dstep" else gan.train_step
step(images, labels)
step(images, labels)
torch.cuda.synchronize()
del records[:]
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
step(images, labels)
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, key, fl, e0, e1 in records:
    a = agg.setdefault((name, key), [0, 0.0, fl])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
print("%s batch %d: %.2f ms (eager, with event overhead); conv launches %d" % (
    leg, bsz, t0.elapsed_time(t1), len(records)))
print("%-7s %-52s %5s %9s %9s %8s" % ("kind", "N,Hin,Win,Ci,Ho,Wo,Co,kh,kw,S,U,pt,pl", "n", "avg us",
                                      "tot ms", "TF/s"))
tot = 0.0
totfl = 0.0
for (name, key), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    totfl += fl * n
    print("%-7s %-52s %5d %9.1f %9.3f %8.1f" % (name, ",".join(map(str, key)), n, 1e3 * ms / n, ms,
                                                fl * n / (ms * 1e-3) / 1e12))
print("total conv ms %.3f, useful TFLOP %.3f" % (tot, totfl / 1e12))
