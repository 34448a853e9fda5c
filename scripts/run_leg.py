"""Runs ONE extra leg of bench.py on its own (for rocprofv3): python scripts/run_leg.py KEY [steps]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

LEGS = {
    "resnet128_dstep": ("resnet_lsun-bedroom128.gin", ("penalty.fn = @no_penalty",), 64, "dstep"),
    "resnet128_dstep_gp": ("resnet_lsun-bedroom128.gin", (), 64, "dstep"),
    "biggan128": ("biggan_imagenet128.gin", (), 64, "step"),
}
key = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg, binds, b, mode = LEGS[key]
torch.cuda.set_device(0)
from compare_gan_amd import eval_gan_lib  # noqa: F401,E402
from compare_gan_amd.gans import modular_gan  # noqa: F401,E402
print(json.dumps(bench.extra_leg(cfg, binds, b, mode, steps, 2, torch.device("cuda", 0))))
