#!/usr/bin/env python
"""Writes tests/golden/tf_checkpoint/model.ckpt-7.{index,data-00000-of-00001}: a small TF-1 tensor
bundle under the reference's variable names (Adam slots as tf.train.AdamOptimizer
(name="d_opt" / "g_opt") names them, modular_gan.py:607,613), written by
compare_gan_amd.tf_checkpoint.write_bundle -- no TensorFlow is available offline, so this fixture
pins the reader against format drift, not against a TF-written file.  Values are a pure function
of the variable name (tests/test_tf_checkpoint.py recomputes them)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from compare_gan_amd import tf_checkpoint

SHAPES = {
    "discriminator/B1/down_conv2/kernel": (3, 3, 4, 6), "discriminator/B1/down_conv2/bias": (6,),
    "discriminator/B1/down_conv2/kernel/u_var": (36, 1),
    "discriminator/B1/down_conv2/kernel/d_opt": (3, 3, 4, 6), "discriminator/B1/down_conv2/kernel/d_opt_1": (3, 3, 4, 6),
    "discriminator/B1/down_conv2/bias/d_opt": (6,), "discriminator/B1/down_conv2/bias/d_opt_1": (6,),
    "generator/fc_noise/kernel": (5, 8), "generator/fc_noise/bias": (8,),
    "generator/fc_noise/kernel/g_opt": (5, 8), "generator/fc_noise/kernel/g_opt_1": (5, 8),
    "generator/fc_noise/kernel/ExponentialMovingAverage": (5, 8),
    "generator/B1/bn1/moving_mean": (8,), "generator/B1/bn1/moving_variance": (8,),
    "generator/B1/bn1/accu/accu_counter": (), "generator/final_conv/kernel": (3, 3, 8, 3),
    "beta1_power": (), "beta2_power": (), "beta1_power_1": (), "beta2_power_1": (),
}


def value(name, shape):
    seed = sum(name.encode("utf-8")) % (2 ** 31)
    return np.random.RandomState(seed).standard_normal(size=shape).astype(np.float32)


def tensors():
    t = {n: value(n, s) for n, s in SHAPES.items()}
    t["global_step"] = np.asarray(7, dtype=np.int64)
    t["global_step_disc"] = np.asarray(35, dtype=np.int64)
    t["generator/B1/bn1/accu/update_accus"] = np.asarray(0, dtype=np.int32)
    return t


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "tf_checkpoint")
    os.makedirs(out, exist_ok=True)
    tf_checkpoint.write_bundle(os.path.join(out, "model.ckpt-7"), tensors(), block_bytes=256)
    print(sorted(os.listdir(out)))
