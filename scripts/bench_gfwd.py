"""The no-gradient generator forward of the ResNet5-128 D-step, launch by launch (N = 64): plain
convolution vs the fused forms it actually runs (batch-norm + ReLU prologue in LDS, statistics
epilogue) vs the unfused alternative (bn_apply pass + plain convolution + bn_stats pass).
hipGraph of R repeats per form.  usage: python scripts/bench_gfwd.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# (H, W, Ci, Co, up) of resnet5.Generator at 128x128 (resnet5.py:60-93): per block conv1 = up-conv,
# conv2 = plain 3x3; then the RGB convolution
SHAPES = [(4, 4, 512, 512, 2), (8, 8, 512, 512, 1), (8, 8, 512, 256, 2), (16, 16, 256, 256, 1),
          (16, 16, 256, 256, 2), (32, 32, 256, 256, 1), (32, 32, 256, 128, 2), (64, 64, 128, 128, 1),
          (64, 64, 128, 64, 2), (128, 128, 64, 64, 1), (128, 128, 64, 3, 1)]
R = 20


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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R   # us


print("N = %d; us per launch (useful TFLOP/s)" % N)
print("%-22s %12s %12s %12s %12s %12s %10s %10s" % ("H,W,Ci,Co,up", "plain", "bn-prologue", "stats-epi", "both",
                                                      "residual+both", "bn_apply", "bn_stats"))
for (H, W, Ci, Co, up) in SHAPES:
    geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, up)
    x = torch.randn(N, H, W, Ci, device=dev).to(BF16)
    w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05
    bias = torch.zeros(Co, device=dev)
    res = torch.randn(N, geom.Ho, geom.Wo, Co, device=dev).to(BF16)
    bt_f, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
    mean, var = torch.zeros(Ci, device=dev), torch.ones(Ci, device=dev)
    gamma, beta = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)
    bn = (mean, var, gamma, beta, 1e-5, False)
    fl = 2.0 * N * geom.Ho * geom.Wo * 9 * Ci * Co / (up * up)
    rows = K.gconv_fused_rows(geom)
    pro = K.gconv_fused_prologue_supported(geom)
    out = []
    out.append(timed(lambda: K.gconv(geom, x, bt_f, bias=bias)))
    out.append(timed(lambda: K.gconv_fused(geom, x, bt_f, bias=bias, bn=bn)) if (rows > 0 or pro) else 0.0)
    out.append(timed(lambda: K.gconv_fused(geom, x, bt_f, bias=bias, want_stats=True)) if rows > 0 else 0.0)
    out.append(timed(lambda: K.gconv_fused(geom, x, bt_f, bias=bias, bn=bn, want_stats=True)) if rows > 0 else 0.0)
    out.append(timed(lambda: K.gconv_fused(geom, x, bt_f, bias=bias, bn=bn, want_stats=True, residual=res))
               if rows > 0 else 0.0)
    x3 = x.reshape(N, H * W, Ci)
    t_apply = timed(lambda: K.bn_apply(x3, mean, var, 1e-5, gamma, beta, False, True))
    y3 = res.reshape(N, geom.Ho * geom.Wo, Co)
    t_stats = timed(lambda: K.bn_stats(y3)) if Co >= 8 else 0.0
    print("%-22s %s %10.1f %10.1f" % (",".join(map(str, (H, W, Ci, Co, up))),
                                      " ".join("%6.1f(%4.0f)" % (t, fl / t / 1e6 if t else 0) for t in out),
                                      t_apply, t_stats))
