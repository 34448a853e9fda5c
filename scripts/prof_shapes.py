"""Per-geometry timing of the convolution launches of one training step (diagnostic).
usage: python scripts/prof_shapes.py [config] [batch]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gan_util as U
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
config = sys.argv[1] if len(sys.argv) > 1 else "resnet_cifar10.gin"
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
records = []


def wrap(name, fn, flops_of):
    def f(geom, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(geom, *a, **kw)
        e1.record()
        records.append((name, geom.key(), flops_of(geom), e0, e1))
        return out
    return f


def flops(g):
    taps = g.kh * g.kw / float(g.U * g.U)
    return 2.0 * g.N * g.Ho * g.Wo * taps * g.Ci * g.Co


K.gconv = wrap("gconv", K.gconv, flops)
K.gwgrad = wrap("gwgrad", K.gwgrad, flops)
gan, options, dataset = U.build_product(config, bsz, dev, seed=3)
nsub = options["disc_iters"] + 1
images, labels = next(dataset.train_batches(bsz * nsub, seed=547))
images = torch.from_numpy(images).to(dev); labels = torch.from_numpy(labels).to(dev)
gan.train_step(images, labels)
torch.cuda.synchronize()
del records[:]
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
gan.train_step(images, labels)
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, key, fl, e0, e1 in records:
    a = agg.setdefault((name, key), [0, 0.0, fl])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
print("step %.2f ms (eager, with event overhead); conv launches %d" % (t0.elapsed_time(t1), len(records)))
print("%-7s %-52s %5s %9s %9s %8s" % ("kind", "N,Hin,Win,Ci,Ho,Wo,Co,kh,kw,S,U,pt,pl", "n", "avg us", "tot ms", "TF/s"))
tot = 0.0
for (name, key), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print("%-7s %-52s %5d %9.1f %9.3f %8.1f" % (name, ",".join(map(str, key)), n, 1e3 * ms / n, ms, fl * n / (ms * 1e-3) / 1e12))
print("total conv ms", tot)
