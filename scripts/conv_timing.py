"""s_memtime phase stamps of workgroup 0 of fast_conv_kernel (needs the -DCG_CONV_TIMING build, `python -m compare_gan_amd.csrc.build --timing`:
CGAMD_LIB_PATH=compare_gan_amd/lib/libcgamd_timing.so).  Prints, per shape, the cycle deltas of the
four waves: entry -> descriptors -> prologue, then per K-slice [wait, barrier, stage, mfma]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.hip import _lib
lib = _lib.load()
raw = getattr(lib, "_lib", lib)
setbuf = raw.cg_debug_set_conv_timing_buffer
setbuf.restype = None
setbuf.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
SHAPES = [(128, 8, 8, 128, 128, 3, 1), (128, 16, 16, 128, 128, 3, 1), (64, 16, 16, 256, 256, 3, 0)]
for (N, H, W, Ci, Co, k, relu) in SHAPES:
    geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, 1, 1)
    x = torch.randn(N, H, W, Ci, device=dev).to(BF16)
    w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
    bias = torch.zeros(Co, device=dev)
    bt_f, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
    gi = x if relu else None
    buf = torch.zeros(4 * 256, dtype=torch.int64, device=dev)
    for _ in range(3):
        K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
    torch.cuda.synchronize()
    setbuf(buf.data_ptr())
    K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
    torch.cuda.synchronize()
    setbuf(None)
    t = buf.cpu().view(4, 256)
    nk = k * k * ((Ci + 63) // 64)
    print("== shape", (N, H, W, Ci, Co, k, relu), "nk", nk)
    for wv in range(4):
        s = t[wv]
        base = int(s[0])
        d = [int(s[i]) - base for i in range(0, 5 + 4 * nk + 8)]
        e0 = 3 + 4 * nk
        last = max(i for i in range(e0, e0 + 10) if int(s[i]) != 0)
        print(" wave %d: desc %d prologue %d | total %d | epilogue %d : %s" % (
            wv, d[1], d[2] - d[1], d[last], d[last] - d[e0],
            "/".join(str(d[i + 1] - d[i]) for i in range(e0, last))))
        rows = []
        for it in range(nk):
            b = 3 + 4 * it
            prev = d[b - 1]
            rows.append("%d/%d/%d/%d" % (d[b] - prev, d[b + 1] - d[b], d[b + 2] - d[b + 1], d[b + 3] - d[b + 2]))
        print("   per slice wait/barrier/stage/mfma:", " ".join(rows))
