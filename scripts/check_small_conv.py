"""Correctness (on-device fp64 reference: nine shifted GEMMs in plain torch) and GPU-bound timing of
the 3x3 convolutions on small maps, forward / data gradient / weight gradient.  The dispatch switches
(CGAMD_SCONV, CGAMD_SWGRAD: 0 = off, 1 = policy, 2 = wherever the geometry fits) are read once per
process: run the script once per setting.  usage: python scripts/check_small_conv.py [quick]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from compare_gan_amd.hip import kernels as K

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
SHAPES = [
    # N, H, W, Ci, Co
    (128, 8, 8, 128, 128), (128, 16, 16, 128, 128), (64, 8, 8, 256, 256), (64, 16, 16, 256, 256),
    (64, 4, 4, 256, 256), (128, 4, 4, 512, 512), (128, 8, 8, 256, 512), (128, 8, 8, 512, 512),
    (128, 16, 16, 256, 256), (6, 4, 4, 64, 64), (3, 8, 8, 64, 192), (2, 16, 8, 128, 64),
    (128, 32, 32, 128, 128),
]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = SHAPES[:2] + SHAPES[5:6] + SHAPES[9:12]
R = 20


def ref_conv(x, w):
    n, h, w_, ci = x.shape
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    out = torch.zeros((n * h * w_, w.shape[-1]), dtype=torch.float64, device=x.device)
    for r in range(3):
        for s in range(3):
            out += xp[:, r:r + h, s:s + w_, :].reshape(-1, ci) @ w[r, s]
    return out.reshape(n, h, w_, -1)


def err(got, ref):
    g = got.detach().to(torch.float64).reshape(ref.shape)
    rms = float(ref.pow(2).mean().sqrt()) + 1e-30
    return float((g - ref).abs().max()) / rms


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(R):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / R)
    return best


print("SCONV=%s SWGRAD=%s" % (os.environ.get("CGAMD_SCONV", "-"), os.environ.get("CGAMD_SWGRAD", "-")))
print("%-24s %22s %22s %22s" % ("N,H,W,Ci,Co", "fwd us|TF  err", "dgrad us|TF  err", "wgrad us|TF  err"))
worst = 0.0
for (N, H, W, Ci, Co) in SHAPES:
    gen = torch.Generator(device=dev).manual_seed(N * 7 + H * 131 + Ci)
    xb = torch.randn((N, H, W, Ci), generator=gen, device=dev).to(BF16)
    wb = (torch.randn((3, 3, Ci, Co), generator=gen, device=dev) / math.sqrt(9 * Ci)).to(BF16)
    dyb = torch.randn((N, H, W, Co), generator=gen, device=dev).to(BF16)
    gob = torch.randn((N, H, W, Co), generator=gen, device=dev).to(BF16)
    resb = torch.randn((N, H, W, Co), generator=gen, device=dev).to(BF16)
    bias = torch.randn(Co, generator=gen, device=dev)
    geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
    bt_f, bt_b = K.weight_prep(wb.float(), want_fwd=True, want_bwd=True)
    x64, w64, dy64 = xb.double(), wb.double(), dyb.double()
    # forward: relu input gate + bias + output gate + residual, fp32 output
    conv = ref_conv(torch.relu(x64), w64) + bias.double()
    ref = torch.where(gob.double() > 0, conv, torch.zeros_like(conv)) + resb.double()
    y = K.gconv(geom, xb, bt_f, bias=bias, gate_in=xb, slope_in=0.0, gate_out=gob, slope_out=0.0,
                residual=resb, out_f32=True)
    e_f = err(y, ref)
    y2 = K.gconv(geom, xb, bt_f, bias=bias)            # plain, bf16 output
    e_f2 = err(y2, ref_conv(x64, w64) + bias.double())
    # data gradient (adjoint geometry, flipped / transposed filter)
    ref_dx = ref_conv(dy64, w64.flip(0, 1).transpose(2, 3).contiguous())
    dx = K.gconv(K.geom_adjoint(geom), dyb, bt_b, out_f32=True)
    e_d = err(dx, ref_dx)
    # weight / bias gradient with the ReLU self-gate on the input
    xp = F.pad(torch.relu(x64), (0, 0, 1, 1, 1, 1))
    dy2 = dy64.reshape(-1, Co)
    ref_dw = torch.stack([torch.stack([
        xp[:, r:r + H, s:s + W, :].reshape(-1, Ci).t() @ dy2 for s in range(3)]) for r in range(3)])
    dw, db = K.gwgrad(geom, xb, dyb, gate_in=xb, slope_in=0.0, want_dbias=True)
    e_w = max(err(dw, ref_dw), err(db, dy2.sum(dim=0)))
    fl = 2.0 * N * H * W * 9 * Ci * Co
    t_f = timed(lambda: K.gconv(geom, xb, bt_f, bias=bias, gate_in=xb, slope_in=0.0))
    t_d = timed(lambda: K.gconv(K.geom_adjoint(geom), dyb, bt_b, gate_out=xb, slope_out=0.0))
    t_w = timed(lambda: K.gwgrad(geom, xb, dyb, gate_in=xb, slope_in=0.0, want_dbias=True))
    worst = max(worst, e_f, e_d, e_w)
    print("%-24s %6.1f|%4.0f %.1e/%.1e %6.1f|%4.0f %.1e %6.1f|%4.0f %.1e" % (
        ",".join(map(str, (N, H, W, Ci, Co))), t_f, fl / t_f / 1e6, e_f, e_f2, t_d, fl / t_d / 1e6, e_d,
        t_w, fl / t_w / 1e6, e_w))
# ---- grouped weight gradients: the six 8x8 layers of a ResNet-CIFAR discriminator backward pass
# (resnet_cifar.py:119-167 blocks B3 / B4) and blocks B4 / B5 of the ResNet5 discriminator ----
for label, shapes in (("cifar D 8x8 x6", [(128, 8, 8, 128, 128)] * 6),
                      ("resnet5 D B4+B5 x6", [(128, 8, 8, 256, 512), (128, 8, 8, 256, 512),
                                              (128, 8, 8, 512, 512)] + [(128, 4, 4, 512, 512)] * 3)):
    jobs, singles = [], []
    for idx, (N, H, W, Ci, Co) in enumerate(shapes):
        gen = torch.Generator(device=dev).manual_seed(1000 + idx)
        xb = torch.randn((N, H, W, Ci), generator=gen, device=dev).to(BF16)
        dyb = torch.randn((N, H, W, Co), generator=gen, device=dev).to(BF16)
        geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
        dw = torch.zeros((3, 3, Ci, Co), device=dev)
        db = torch.zeros((Co,), device=dev)
        jobs.append((geom, xb, dyb, idx % 3 != 0, dw, db))
    K.gwgrad_multi(jobs)
    e_g = 0.0
    for (geom, xb, dyb, relu, dw, db) in jobs:
        dw1, db1 = K.gwgrad(geom, xb, dyb, gate_in=xb if relu else None, slope_in=0.0, want_dbias=True)
        e_g = max(e_g, err(dw, dw1.double()), err(db, db1.double()))
    t_g = timed(lambda: K.gwgrad_multi(jobs))
    t_s = timed(lambda: [K.gwgrad(g_, x_, d_, gate_in=x_ if r_ else None, slope_in=0.0, want_dbias=True)
                         for (g_, x_, d_, r_, _, _) in jobs])
    fl = sum(2.0 * g_.N * g_.Ho * g_.Wo * 9 * g_.Ci * g_.Co for (g_, _, _, _, _, _) in jobs)
    print("grouped wgrad %-22s one call %7.1f us (%4.0f TF)  separate calls %7.1f us  diff vs separate %.1e" % (
        label, t_g, fl / t_g / 1e6, t_s, e_g))
    worst = max(worst, e_g)
print("worst max-abs error / rms (fp32 outputs ~1e-4 expected; bf16 output ~4e-3): %.2e" % worst)
