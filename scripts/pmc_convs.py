"""Eager launches of a few convolution shapes for a rocprofv3 --pmc pass (one process per counter set).
usage: python scripts/pmc_convs.py            -> run the launches
       python scripts/pmc_convs.py agg DIR    -> aggregate DIR/**/*counter_collection.csv per kernel"""
import os, sys, glob, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "agg":
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0][:70]
            key = (name, row.get("Grid_Size", ""), row.get("LDS_Block_Size", ""))
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[key].add(row["Dispatch_Id"])
    for key in sorted(acc):
        n = max(1, len(cnt[key]))
        print(key, "dispatches", n)
        for c, v in sorted(acc[key].items()):
            print("    %-28s %16.0f" % (c, v / n))
    sys.exit(0)

import torch
from compare_gan_amd.hip import kernels as K
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
SHAPES = [(128, 32, 32, 128, 128, 3, 1, 1, 1), (64, 32, 32, 256, 256, 3, 1, 1, 0),
          (64, 16, 16, 256, 256, 3, 1, 1, 0)]
for (N, H, W, Ci, Co, k, s, up, relu) in SHAPES:
    geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, s, up)
    x = torch.randn(N, H, W, Ci, device=dev).to(BF16)
    w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
    dy = torch.randn(N, geom.Ho, geom.Wo, Co, device=dev).to(BF16)
    bias = torch.zeros(Co, device=dev)
    bt_f, bt_b = K.weight_prep(w, want_fwd=True, want_bwd=True)
    gi = x if relu else None
    for _ in range(3):
        K.gconv(geom, x, bt_f, bias=bias, gate_in=gi, slope_in=0.0)
        K.gwgrad(geom, x, dy, gate_in=gi, slope_in=0.0, want_dbias=True)
    torch.cuda.synchronize()
