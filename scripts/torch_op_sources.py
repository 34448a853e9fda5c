#!/usr/bin/env python
"""Which lines of the product make torch launch its own kernels inside a step?  (64
vectorized_elementwise + 29 copyBuffer launches per ResNet5 D-step although no arithmetic of the hot
path is supposed to run in torch.)  One eager unit of a bench leg under a TorchDispatchMode (it follows
the autograd worker thread too); aten ops that launch device work are grouped by the innermost
frames inside this repo.
usage: torch_op_sources.py LEG   (LEG as in scripts/run_leg_eager.py)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
from tests import gan_util as U

LEGS = {
    "resnet128_dstep": ("resnet_lsun-bedroom128.gin", ("penalty.fn = @no_penalty",), 64, "dstep"),
    "resnet128_dstep_gp": ("resnet_lsun-bedroom128.gin", (), 64, "dstep"),
    "biggan128": ("biggan_imagenet128.gin", (), 64, "step"),
    "cifar": ("resnet_cifar10.gin", (), 64, "step"),
    "sndcgan128": ("sndcgan_celebahq128.gin", (), 32, "step"),
}
key = sys.argv[1]
cfg, binds, b, mode = LEGS[key]
dev = torch.device("cuda", 0)
gan, options, dataset = U.build_product(cfg, b, dev, seed=3, bindings=binds)
nsub = 1 if mode == "dstep" else options["disc_iters"] + 1
images, labels = next(dataset.train_batches(b * nsub, seed=547))
images, labels = torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev)
step = gan.disc_step if mode == "dstep" else gan.train_step
for _ in range(2):
    step(images, labels)
torch.cuda.synchronize()
NO_LAUNCH = ("empty", "view", "reshape", "as_strided", "slice", "select", "detach", "alias", "t.",
             "transpose", "permute", "expand", "unsqueeze", "squeeze", "_unsafe_view", "narrow",
             "_local_scalar_dense", "resize_", "set_", "unbind", "split", "lift_fresh", "_reshape_alias",
             "sym_", "is_", "stride", "size", "numel", "dim", "storage_offset", "record_stream", "unfold")
by_site = collections.Counter()
by_op = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not name.startswith(NO_LAUNCH):
            frames = [f for f in traceback.extract_stack() if f.filename.startswith(ROOT) and
                      "torch_op_sources" not in f.filename]
            site = " <- ".join("%s:%d" % (f.filename.replace(ROOT + "/", ""), f.lineno)
                               for f in reversed(frames[-3:])) or "<autograd engine>"
            shape = ""
            for a in args:
                if torch.is_tensor(a):
                    shape = "%s %s" % (tuple(a.shape), str(a.dtype).replace("torch.", ""))
                    break
            by_site[(name, site, shape)] += 1
            by_op[name] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step(images, labels)
    torch.cuda.synchronize()
print("== aten ops in ONE eager %s ==" % key)
for k, n in by_op.most_common():
    print("%5d  %s" % (n, k))
print("== by call site ==")
for (op, site, shape), n in sorted(by_site.items(), key=lambda kv: (-kv[1], kv[0][1])):
    print("%4d  %-22s %-28s %s" % (n, op, shape, site))
