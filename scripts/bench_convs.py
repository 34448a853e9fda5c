"""GPU-bound timing of individual convolution launches (hipGraph of R repeats, so host launch cost
is excluded).  usage: python scripts/bench_convs.py [cifar|resnet128|...|"N,H,W,Ci,Co,k,s,up,relu;..."]
BENCH_KINDS=fwd,dgrad,wgrad selects the columns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "cifar"
# (N, H, W, Ci, Co, k, stride, up, relu_in)
SHAPES = {
    "cifar": [
        (128, 32, 32, 128, 128, 3, 1, 1, 1), (128, 16, 16, 128, 128, 3, 1, 1, 1),
        (128, 8, 8, 128, 128, 3, 1, 1, 1), (64, 32, 32, 256, 256, 3, 1, 1, 0),
        (64, 16, 16, 256, 256, 3, 1, 2, 0), (64, 16, 16, 256, 256, 3, 1, 1, 0),
        (64, 8, 8, 256, 256, 3, 1, 2, 0), (64, 8, 8, 256, 256, 3, 1, 1, 0),
        (64, 4, 4, 256, 256, 3, 1, 2, 0), (64, 1, 1, 128, 4096, 1, 1, 1, 0),
        (128, 32, 32, 3, 128, 3, 1, 1, 1), (64, 32, 32, 256, 3, 3, 1, 1, 0),
    ],
    "halo": [
        (128, 32, 32, 128, 128, 3, 1, 1, 1), (64, 32, 32, 256, 256, 3, 1, 1, 0),
        (128, 16, 16, 128, 128, 3, 1, 1, 1), (128, 64, 64, 64, 128, 3, 1, 1, 1),
    ],
    "fixed": [
        (128, 8, 8, 128, 128, 3, 1, 1, 1), (128, 16, 16, 128, 128, 3, 1, 1, 1),
        (128, 32, 32, 128, 128, 3, 1, 1, 1), (64, 32, 32, 256, 256, 3, 1, 1, 0),
        (64, 8, 8, 256, 256, 3, 1, 1, 0),
    ],
    "probe": [
        (128, 32, 32, 64, 128, 1, 1, 1, 0), (128, 32, 32, 128, 128, 1, 1, 1, 0),
        (128, 32, 32, 256, 128, 1, 1, 1, 0), (128, 32, 32, 512, 128, 1, 1, 1, 0),
        (128, 32, 32, 1024, 128, 1, 1, 1, 0), (128, 32, 32, 64, 128, 3, 1, 1, 0),
        (128, 32, 32, 128, 128, 3, 1, 1, 0), (128, 32, 32, 256, 128, 3, 1, 1, 0),
    ],
    # the unit-stride 3x3 layers of the ResNet5-128 D-step (N = 2B = 128) and G forward (B = 64)
    "hc": [
        (128, 128, 128, 64, 64, 3, 1, 1, 1), (128, 64, 64, 64, 128, 3, 1, 1, 1),
        (128, 64, 64, 128, 128, 3, 1, 1, 1), (128, 32, 32, 128, 256, 3, 1, 1, 1),
        (128, 32, 32, 256, 256, 3, 1, 1, 1), (128, 16, 16, 256, 256, 3, 1, 1, 1),
        (64, 64, 64, 128, 64, 3, 1, 2, 0), (64, 32, 32, 256, 128, 3, 1, 2, 0),
        (64, 16, 16, 256, 256, 3, 1, 2, 0), (64, 128, 128, 64, 64, 3, 1, 1, 0),
        (128, 32, 32, 128, 128, 3, 1, 1, 1), (64, 32, 32, 256, 256, 3, 1, 1, 0),
    ],
    "stem": [
        (128, 128, 128, 3, 64, 3, 1, 1, 1), (64, 128, 128, 3, 64, 3, 1, 1, 1),
        (128, 32, 32, 3, 128, 3, 1, 1, 1), (128, 128, 128, 3, 96, 3, 1, 1, 1),
    ],
    "biggan": [
        (64, 128, 128, 96, 96, 3, 1, 1, 1), (64, 64, 64, 96, 192, 3, 1, 1, 1),
        (64, 64, 64, 192, 192, 3, 1, 1, 1), (64, 8, 8, 1536, 1536, 3, 1, 1, 1),
        (64, 4, 4, 1536, 1536, 3, 1, 1, 1), (32, 64, 64, 192, 96, 3, 1, 2, 0),
    ],
    "c64": [
        (128, 128, 128, 64, 64, 3, 1, 1, 1), (64, 128, 128, 64, 64, 3, 1, 1, 0),
        (128, 64, 64, 64, 64, 3, 1, 1, 1),
    ],
    "resnet128": [
        (128, 128, 128, 64, 64, 3, 1, 1, 1), (128, 64, 64, 64, 128, 3, 1, 1, 1),
        (128, 64, 64, 128, 128, 3, 1, 1, 1), (128, 32, 32, 128, 256, 3, 1, 1, 1),
        (128, 32, 32, 256, 256, 3, 1, 1, 1), (128, 16, 16, 256, 256, 3, 1, 1, 1),
        (128, 8, 8, 512, 512, 3, 1, 1, 1), (64, 64, 64, 128, 64, 3, 1, 2, 0),
        (64, 32, 32, 256, 128, 3, 1, 2, 0), (64, 128, 128, 64, 64, 3, 1, 1, 0),
        (128, 128, 128, 3, 64, 3, 1, 1, 1), (64, 128, 128, 64, 3, 3, 1, 1, 0),
    ],
}.get(which)
if SHAPES is None:   # literal list: "N,H,W,Ci,Co,k,s,up,relu;N,H,..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in which.split(";") if t]
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


print("%-44s %10s %10s %10s   (us | TF/s useful)" % ("N,H,W,Ci,Co,k,s,up,relu", "fwd", "dgrad", "wgrad"))
for (N, H, W, Ci, Co, k, s, up, relu) in SHAPES:
    geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, s, up)
    x = torch.randn(N, H, W, Ci, device=dev).to(BF16)
    w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
    dy = torch.randn(N, geom.Ho, geom.Wo, Co, device=dev).to(BF16)
    bias = torch.zeros(Co, device=dev)
    bt_f, bt_b = K.weight_prep(w, want_fwd=True, want_bwd=True)
    fl = 2.0 * N * geom.Ho * geom.Wo * k * k * Ci * Co / (up * up)
    gi = x if relu else None
    kinds = os.environ.get("BENCH_KINDS", "fwd,dgrad,wgrad").split(",")
    t_f = timed(lambda: K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)) if "fwd" in kinds else 1e-9
    t_d = timed(lambda: K.gconv(K.geom_adjoint(geom), dy, bt_b, gate_out=gi, slope_out=0.0)) if "dgrad" in kinds else 1e-9
    t_w = 1e-9 if (os.environ.get("BENCH_NO_WGRAD") or "wgrad" not in kinds) else timed(
        lambda: K.gwgrad(geom, x, dy, gate_in=gi, slope_in=0.0, want_dbias=True))
    print("%-44s %6.1f|%4.0f %6.1f|%4.0f %6.1f|%4.0f" % (
        ",".join(map(str, (N, H, W, Ci, Co, k, s, up, relu))), t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6,
        t_w, fl / max(t_w, 1e-9) / 1e6))
