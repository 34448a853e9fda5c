"""Per-workgroup cycle split of hconv_rw_kernel (timing build, see hconv_timeline.py): total / waiting
for the window + barrier / MFMA loop / epilogue.  usage: hconv_rw_timeline.py N,H,W [dgrad]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.hip import _lib
lib = _lib.load()
raw = getattr(lib, "_lib", lib)
setbuf = raw.cg_debug_set_hconv_timing_buffer
setbuf.restype = None
setbuf.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
(N, H, W) = [int(v) for v in sys.argv[1].split(",")]
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
geom = K.geom_conv_same(N, H, W, 64, 64, 3, 3, 1, 1)
x = torch.randn(N, H, W, 64, device=dev).to(BF16)
w = torch.randn(3, 3, 64, 64, device=dev) * 0.05
bias = torch.zeros(64, device=dev)
bt_f, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
go = torch.randn(N, H, W, 64, device=dev).to(BF16)
def run():
    if mode == "fwd":
        return K.gconv(geom, x, bt_f, bias=bias)
    return K.gconv(geom, x, bt_f, gate_out=go, slope_out=0.0)
buf = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    run()
torch.cuda.synchronize()
setbuf(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
setbuf(None)
t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
t = t[t[:, 0] != 0]
ntile = N * (H // 4) * (W // 32)
print("shape %s %s: %d workgroups, %d tiles, event %.1f us" % (sys.argv[1], mode, len(t), ntile, 1e3 * e0.elapsed_time(e1)))
per = ntile / max(1, len(t))
print("cycles per workgroup: total %.0f | wait %.0f | mfma %.0f | epilogue %.0f   (per tile: %.0f | %.0f | %.0f | %.0f)" % (
    t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[:, 3].mean(),
    t[:, 0].mean() / per, t[:, 1].mean() / per, t[:, 2].mean() / per, t[:, 3].mean() / per))
