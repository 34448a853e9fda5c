"""Per-geometry timing of the convolution launches of ONE Inception batch (diagnostic; eager launches
with HIP events around every gconv call), and the kernel families they dispatch to.
usage: python scripts/prof_inception.py [batch]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compare_gan_amd import inception
from compare_gan_amd.hip import kernels as K

dev = torch.device("cuda:0")
bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
records = []
_gconv = K.gconv


def gconv(geom, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = _gconv(geom, *a, **kw)
    e1.record()
    records.append((geom.key(), 2.0 * geom.N * geom.Ho * geom.Wo * geom.kh * geom.kw * geom.Ci * geom.Co, e0, e1))
    return out


K.gconv = gconv
_gconv_ld = K.gconv_ld


def gconv_ld(geom, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = _gconv_ld(geom, *a, **kw)
    e1.record()
    records.append((geom.key(), 2.0 * geom.N * geom.Ho * geom.Wo * geom.kh * geom.kw * geom.Ci * geom.Co, e0, e1))
    return out


K.gconv_ld = gconv_ld
net = inception.InceptionV3(dev)
x = torch.rand((bsz, 32, 32, 3), device=dev) * 255.0
for _ in range(2):
    net.features(x)
torch.cuda.synchronize()
records.clear()
K.prof_reset()
K.prof_enable(True)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
net.features(x)
t1.record()
torch.cuda.synchronize()
K.prof_enable(False)
agg = collections.OrderedDict()
for key, fl, e0, e1 in records:
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += fl
tot_ms = sum(a[1] for a in agg.values())
tot_fl = sum(a[2] for a in agg.values())
print("inception batch %d: %.2f ms wall (eager), conv launches %d, conv ms %.2f, %.1f GFLOP -> %.1f TFLOP/s" % (
    bsz, t0.elapsed_time(t1), len(records), tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9))
print("%-52s %4s %9s %9s %8s" % ("N,Hin,Win,Ci,Ho,Wo,Co,kh,kw,S,U,pt,pl", "n", "avg us", "tot ms", "TF/s"))
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %4d %9.1f %9.3f %8.1f" % (",".join(str(k) for k in key), n, 1e3 * ms / n, ms, fl / ms / 1e9))
print("kernel families:")
for k, v in sorted(K.prof_collect().items(), key=lambda kv: -kv[1]["ms"]):
    if v["launches"]:
        print("  %-34s %8.3f ms  x%d  %7.1f TF/s" % (k, v["ms"], v["launches"], v["flops"] / (v["ms"] * 1e-3) / 1e12))
