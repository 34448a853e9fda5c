#!/usr/bin/env python
"""Where does the sampling phase of FID-10k go?  (30.2 s on the driver's box, 0.31 s on
the builder's -- the builder's run had skipped the roofline leg that precedes it in bench.py.)

Times eval_gan_lib.evaluate_gan on resnet_cifar10.gin in the states bench.py passes through:
  A  after captured (hipGraph) steps only
  B  after eager train steps with the HIP-event brackets on (the roofline leg)
  C  the same once more (state that persists?)
and, when a phase is slow, profiles 10 generator batches on the host (cProfile)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIG_DIR = os.path.join(ROOT, "tests", "golden", "example_configs")


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from compare_gan_amd import datasets, gin, runner_lib, eval_gan_lib, eval_utils
    from compare_gan_amd.gans import modular_gan  # noqa: F401
    from compare_gan_amd.hip import kernels as K
    from compare_gan_amd.metrics import fid_score as fid_lib
    from compare_gan_amd.metrics import inception_score as is_lib

    gin.parse_config_files_and_bindings([os.path.join(CONFIG_DIR, "resnet_cifar10.gin")], [])
    options = runner_lib.get_options_dict()
    dataset = datasets.get_dataset()
    gan = options["gan_class"](dataset=dataset, parameters=options, model_dir="/tmp/cg_fidb")
    gan.build(batch_size=64, device=dev, seed=3)
    nsub = options["disc_iters"] + 1
    batches = dataset.train_batches(64 * nsub, seed=547)
    images, labels = next(batches)
    images, labels = torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev)
    eval_utils.get_inception(dev)
    tasks = [is_lib.InceptionScoreTask(), fid_lib.FIDScoreTask()]

    def ev(tag):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eval_gan_lib.evaluate_gan(gan, tasks, num_averaging_runs=1)
        torch.cuda.synchronize()
        print("%-28s wall %.3f s  split %s" % (tag, time.perf_counter() - t0,
                                               {k: round(v, 3) for k, v in eval_gan_lib.LAST_TIMING.items()}),
              flush=True)
        return eval_gan_lib.LAST_TIMING["sample"]

    def host_profile(tag):
        z = torch.rand(64, 128, device=dev) * 2 - 1
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        for _ in range(10):
            gan.generate(z, None)
        torch.cuda.synchronize()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
        print("---- host profile of 10 generate() calls, %s ----" % tag)
        print(s.getvalue()[:6000], flush=True)

    if "--trace" in sys.argv:
        # first-use cost of every operation of the sampling phase, one by one
        from compare_gan_amd.hip import kernels as K2

        def lap(tag, fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            print("  %-34s %.4f s" % (tag, time.perf_counter() - t0), flush=True)
            return out
        z = torch.rand(64, 128, device=dev) * 2 - 1
        imgs = []
        for i in range(4):
            imgs.append(lap("generate #%d" % i, lambda: gan.generate(z, None)))
        ev_imgs = [lap("to_eval_images #%d" % i, lambda x=x: eval_utils.to_eval_images(x)) for i, x in enumerate(imgs[:2])]
        local = lap("torch.cat", lambda: torch.cat(ev_imgs * 40, dim=0))
        lap("isnan.any", lambda: bool(torch.isnan(local).any()))
        lap("isnan.any again", lambda: bool(torch.isnan(local).any()))
        lap("inception transform (128 img)", lambda: eval_utils.inception_transform_np(local[:128], 64))
        lap("inception transform again", lambda: eval_utils.inception_transform_np(local[:128], 64))
    if "--eager-first" in sys.argv:
        gan.train_step(images, labels)
        ev("0 after one eager step")
    run = gan.capture_train_step()
    for _ in range(5):
        run(images, labels)
    a = ev("A after captured steps")
    a2 = ev("A2 again")
    K.prof_reset()
    K.prof_enable(True)
    for _ in range(3):
        gan.train_step(images, labels)
    torch.cuda.synchronize()
    K.prof_enable(False)
    K.prof_collect()
    b = ev("B after eager+prof steps")
    c = ev("C again")
    for _ in range(2):
        run(images, labels)
    d = ev("D after replays again")
    if max(a, a2, b, c, d) > 2.0:
        host_profile("slow state")


if __name__ == "__main__":
    main()
