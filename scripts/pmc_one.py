"""Eager launches of ONE convolution shape for rocprofv3 --pmc passes, and the aggregation of their
counter_collection.csv files.
usage: python scripts/pmc_one.py run N,H,W,Ci,Co,k,s,up,relu [fwd,dgrad,wgrad]
       python scripts/pmc_one.py agg DIR"""
import collections
import csv
import glob
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "agg":
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0][-60:]
            if "at::native" in row["Kernel_Name"] or "prep_" in name or "randn" in name or "distribution" in name:
                continue
            key = (name, row.get("Grid_Size", ""), row.get("LDS_Block_Size", ""))
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[key].add(row["Dispatch_Id"])
    for key in sorted(acc):
        n = max(1, len(cnt[key]))
        print(key, "dispatches", n)
        for c, v in sorted(acc[key].items()):
            print("    %-32s %16.0f" % (c, v / n))
    sys.exit(0)

import torch
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
(N, H, W, Ci, Co, k, s, up, relu) = [int(v) for v in sys.argv[2].split(",")]
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fwd", "dgrad", "wgrad"]
geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, s, up)
x = torch.randn(N, H, W, Ci, device=dev).to(BF16)
w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
dy = torch.randn(N, geom.Ho, geom.Wo, Co, device=dev).to(BF16)
bias = torch.zeros(Co, device=dev)
bt_f, bt_b = K.weight_prep(w, want_fwd=True, want_bwd=True)
gi = x if relu else None
for _ in range(3):
    if "fwd" in kinds:
        K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
    if "dgrad" in kinds:
        K.gconv(K.geom_adjoint(geom), dy, bt_b, gate_out=gi, slope_out=0.0)
    if "wgrad" in kinds:
        K.gwgrad(geom, x, dy, gate_in=gi, slope_in=0.0, want_dbias=True)
torch.cuda.synchronize()
