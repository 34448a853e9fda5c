"""Tiny-GEMM timing: the conditional-BN linears of BigGAN ([N,148] x [148,C]) through the generic
kernel vs the same problem padded to K = 160 (MFMA-tiled path).  hipGraph of 50 repeats each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
for n, k, co in ((64, 148, 1536), (64, 160, 1536), (64, 148, 192), (64, 160, 192), (64, 1000, 128),
                 (64, 1024, 128), (64, 128, 4096)):
    x = torch.randn((n, 1, 1, k), device=dev).to(torch.bfloat16)
    w = torch.randn((1, 1, k, co), device=dev) * 0.05
    bt, _ = K.weight_prep(w)
    geom = K.make_geom(n, 1, 1, k, 1, 1, co, 1, 1)
    for _ in range(3):
        K.gconv(geom, x, bt, out_f32=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        K.gconv(geom, x, bt, out_f32=True)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(50):
            y = K.gconv(geom, x, bt, out_f32=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("N %d K %d Co %d: %.1f us per launch" % (n, k, co, 1e3 * e0.elapsed_time(e1) / 200))
