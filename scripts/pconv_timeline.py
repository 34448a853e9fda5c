"""Workgroup timeline of pconv_kernel (needs the -DCG_CONV_TIMING build: python -m
compare_gan_amd.csrc.build --timing; CGAMD_LIB_PATH=compare_gan_amd/lib/libcgamd_timing.so).
Per persistent workgroup: s_memtime at entry / first window landed / exit, cycles summed over its
channel blocks (18 half-slices each) and over its epilogues, s_memrealtime (100 MHz) at entry / exit.
usage: pconv_timeline.py N,H,W,Ci,Co,relu [dgrad]"""
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
setbuf = raw.cg_debug_set_pconv_timing_buffer
setbuf.restype = None
setbuf.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
(N, H, W, Ci, Co, relu) = [int(v) for v in sys.argv[1].split(",")]
geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
x = torch.randn(N, H, W, Ci, device=dev).to(BF16)
w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05
bias = torch.zeros(Co, device=dev)
bt_f, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
gi = x if relu else None
buf = torch.zeros(512 * 8 + 512 * 8 * 4, dtype=torch.int64, device=dev)
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
raw_t = buf.cpu().numpy()
t = raw_t[:512 * 8].reshape(-1, 8)
t = t[t[:, 0] != 0]
n = len(t)
if n == 0:
    print("shape", sys.argv[1], ": the persistent kernel did not run")
    sys.exit(0)
print("shape", sys.argv[1], "workgroups", n, "event time %.1f us" % (1e3 * e0.elapsed_time(e1)))
rt0, rt1 = t[:, 5], t[:, 6]
span_us = (rt1.max() - rt0.min()) / 100.0
dur = (t[:, 4] - t[:, 0]).astype(np.float64)
dur_rt = (rt1 - rt0) / 100.0
clk = dur.sum() / max(1.0, dur_rt.sum())   # cycles per us = MHz
nb = t[:, 7].astype(np.float64)
cb = -(-Ci // 64)
items = nb / cb
print("span (first entry -> last exit) %.1f us; workgroup life mean %.1f us (min %.1f max %.1f); shader clock ~ %.0f MHz"
      % (span_us, dur_rt.mean(), dur_rt.min(), dur_rt.max(), clk))
blk = t[:, 2] / np.maximum(nb, 1)
epi = t[:, 3] / np.maximum(items, 1)
ideal = 18 * 2 * (Co if Co < 128 else 128) / 64 * 4 * 32 * 2   # MFMAs per wave per block x 32 cycles x 2 waves / SIMD
print("per workgroup: blocks %.1f (items %.1f) | prologue %.0f cycles | block %.0f cycles (MFMA-bound: %.0f -> %.0f %% busy) | "
      "epilogue + hand-over %.0f cycles per item | other %.0f cycles"
      % (nb.mean(), items.mean(), (t[:, 1] - t[:, 0]).mean(), blk.mean(), ideal, 100.0 * ideal / blk.mean(),
         epi.mean(), (dur - (t[:, 1] - t[:, 0]) - t[:, 2] - t[:, 3]).mean()))
print("block cycles percentiles: p5 %.0f p50 %.0f p95 %.0f" % tuple(np.percentile(blk, [5, 50, 95])))

# per wave: cycles inside the 18 half-slices, inside the weight waits, at the block-end barrier, in epilogues
pw = raw_t[512 * 8:512 * 8 + n * 8 * 4].reshape(n, 8, 4).astype(np.float64)
nbm = nb.mean()
print("per wave (cycles per block, mean over workgroups):  half-slices | weight waits | block-end barrier | epilogue per item")
for wv in range(8):
    print("  wave %d: %8.0f %8.0f %8.0f %8.0f" % (wv, pw[:, wv, 0].mean() / nbm, pw[:, wv, 1].mean() / nbm,
                                                 pw[:, wv, 2].mean() / nbm, pw[:, wv, 3].mean() / items.mean()))
