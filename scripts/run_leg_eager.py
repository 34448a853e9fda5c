"""ONE leg of bench.py run eagerly (no hipGraph) for the rocprofv3 --pmc passes: counters are
attributed per dispatch, and a graph replay is one dispatch.  usage: run_leg_eager.py KEY [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import gan_util as U

LEGS = {
    "resnet128_dstep": ("resnet_lsun-bedroom128.gin", ("penalty.fn = @no_penalty",), 64, "dstep"),
    "resnet128_dstep_gp": ("resnet_lsun-bedroom128.gin", (), 64, "dstep"),
    "resnet_lsun128_step": ("resnet_lsun-bedroom128.gin", (), 32, "step"),
    "biggan128": ("biggan_imagenet128.gin", (), 64, "step"),
    "biggan128_bs256": ("biggan_imagenet128.gin", (), 256, "step"),
    "cifar": ("resnet_cifar10.gin", (), 64, "step"),
    "sndcgan128": ("sndcgan_celebahq128.gin", (), 32, "step"),
}
key = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg, binds, b, mode = LEGS[key]
dev = torch.device("cuda", 0)
gan, options, dataset = U.build_product(cfg, b, dev, seed=3, bindings=binds)
nsub = 1 if mode == "dstep" else options["disc_iters"] + 1
images, labels = next(dataset.train_batches(b * nsub, seed=547))
images, labels = torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev)
step = gan.disc_step if mode == "dstep" else gan.train_step
for _ in range(steps):
    step(images, labels)
torch.cuda.synchronize()
print("ran", steps, "eager steps of", key)
