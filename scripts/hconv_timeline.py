"""Workgroup timeline of hconv_kernel (needs the -DCG_CONV_TIMING build: python -m
compare_gan_amd.csrc.build --timing; CGAMD_LIB_PATH=compare_gan_amd/lib/libcgamd_timing.so).
Per workgroup: s_memtime at entry / descriptors done / first slice landed / loop done / exit,
s_memrealtime (100 MHz) at entry / exit, HW_ID.  usage: hconv_timeline.py N,H,W,Ci,Co,k,s,up,relu [dgrad]"""
import ctypes
import os
import sys
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
(N, H, W, Ci, Co, k, s, up, relu) = [int(v) for v in sys.argv[1].split(",")]
geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, s, up)
x = torch.randn(N, H, W, Ci, device=dev).to(BF16)
w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
bias = torch.zeros(Co, device=dev)
bt_f, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
gi = x if relu else None
# optional 3rd argument: fused forms of the no-gradient generator forward: any of "bn", "stats", "res"
forms = sys.argv[2].split("+") if len(sys.argv) > 2 else []
if forms:
    mean, var = torch.zeros(Ci, device=dev), torch.ones(Ci, device=dev)
    gamma, beta = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)
    bn = (mean, var, gamma, beta, 1e-5, False) if "bn" in forms else None
    res = torch.randn(N, geom.Ho, geom.Wo, Co, device=dev).to(BF16) if "res" in forms else None
    _plain = K.gconv
    K.gconv = lambda geom, x, bt_f, bias=None, gate_in=None, slope_in=0.0: K.gconv_fused(
        geom, x, bt_f, bias=bias, bn=bn, want_stats="stats" in forms, residual=res)
nwg_max = 1 << 16
buf = torch.zeros(nwg_max * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
torch.cuda.synchronize()
setbuf(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
e1.record()
torch.cuda.synchronize()
setbuf(None)
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
n = len(t)
print("shape", sys.argv[1], "workgroups", n, "event time %.1f us" % (1e3 * e0.elapsed_time(e1)))
rt0, rt1 = t[:, 5], t[:, 6]
span_us = (rt1.max() - rt0.min()) / 100.0
dur = (t[:, 4] - t[:, 0]).astype(np.float64)
dur_rt = (rt1 - rt0) / 100.0
clk = dur.sum() / max(1.0, dur_rt.sum())   # cycles per us = MHz
print("span (first entry -> last exit) %.1f us; workgroup duration mean %.1f us (min %.1f max %.1f); "
      "shader clock ~ %.0f MHz" % (span_us, dur_rt.mean(), dur_rt.min(), dur_rt.max(), clk))
print("phases (cycles, mean): descriptors %.0f | first slice landed %.0f | main loop %.0f | epilogue %.0f"
      % ((t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(),
         (t[:, 4] - t[:, 3]).mean()))
# concurrency profile: workgroups resident at 20 sample points
lo, hi = rt0.min(), rt1.max()
pts = np.linspace(lo, hi, 21)
print("resident workgroups over time:", " ".join(str(int(((rt0 <= p) & (rt1 > p)).sum())) for p in pts))
start_us = (rt0 - lo) / 100.0
print("entry time percentiles (us): p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f" % tuple(
    np.percentile(start_us, [0, 25, 50, 75, 100])))
hw = t[:, 7] & 0xffffffff
xcc = (t[:, 7] >> 32) & 0xf
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
print("workgroups per XCC:", np.bincount(xcc.astype(np.int64), minlength=8).tolist())
