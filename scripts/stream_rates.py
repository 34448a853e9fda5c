"""Achieved HBM rates of the memory-bound kernels at the activation sizes of the legs (hipGraph of 20
launches): plain torch copy / the library's batch-norm, residual-sum and pooling passes.
usage: python scripts/stream_rates.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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


print("N = %d; us (TB/s algorithmic)" % N)
print("%-16s %8s %14s %14s %14s %14s %14s %14s %14s" % ("H,W,C", "MB", "torch copy", "bn_stats", "bn_apply", "bn_bwd",
                                                   "axpby", "pool2", "pool2_bwd"))
for (H, W, C) in [(128, 128, 64), (128, 128, 96), (64, 64, 128), (64, 64, 192), (32, 32, 256), (32, 32, 384),
                  (16, 16, 768), (8, 8, 1536)]:
    x = torch.randn(N, H, W, C, device=dev).to(BF16)
    y = torch.empty_like(x)
    dy = torch.randn(N, H, W, C, device=dev).to(BF16)
    mb = x.numel() * 2 / 1e6
    mean, var = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    x3 = x.reshape(N, H * W, C)
    dy3 = dy.reshape(N, H * W, C)
    out = []
    t = timed(lambda: y.copy_(x)); out.append((t, 2 * mb))
    t = timed(lambda: K.bn_stats(x3)); out.append((t, mb))
    t = timed(lambda: K.bn_apply(x3, mean, var, 1e-5, gamma, beta, False, True)); out.append((t, 2 * mb))
    try:
        yy = K.bn_apply(x3, mean, var, 1e-5, gamma, beta, False, True)
        t = timed(lambda: K.bn_backward(x3, yy, dy3, mean, var, 1e-5, gamma, False, True)); out.append((t, 5 * mb))
    except Exception as e:
        out.append((0.0, 0.0))
    t = timed(lambda: K.axpby(x, 1.0, dy, 1.0)); out.append((t, 3 * mb))
    try:
        t = timed(lambda: K.avgpool2(x)); out.append((t, 1.25 * mb))
        p = K.avgpool2(x)
        t = timed(lambda: K.avgpool2_bwd(p)); out.append((t, 1.25 * mb))
    except Exception as e:
        out.append((0.0, 0.0)); out.append((0.0, 0.0))
    print("%-16s %8.1f %s" % ("%d,%d,%d" % (H, W, C), mb,
                              " ".join("%7.1f(%4.2f)" % (t, b / t if t else 0) for t, b in out)))
