"""Event timing of ONE 3x3 convolution shape under the persistent kernel's ablation bits
(CGAMD_PCONV_DBG, cg_conv_pers.hip: the results are wrong, only the time means something).
usage: CGAMD_PCONV_DBG=<bits> python scripts/pconv_ablate.py N,H,W,Ci,Co,relu"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
(N, H, W, Ci, Co, relu) = [int(v) for v in sys.argv[1].split(",")]
geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
x = torch.randn(N, H, W, Ci, device=dev).to(torch.bfloat16)
w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05
bias = torch.zeros(Co, device=dev)
bt_f, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
gi = x if relu else None
for _ in range(5):
    K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
torch.cuda.synchronize()
R = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(R):
    K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / R
fl = 2.0 * N * H * W * 9 * Ci * Co
print("dbg %3s  %s  %.1f us  %.0f TF/s" % (os.environ.get("CGAMD_PCONV_DBG", "0"), sys.argv[1], us, fl / us / 1e6))
