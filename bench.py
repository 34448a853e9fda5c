#!/usr/bin/env python
"""Benchmark of the hot path: one unrolled GAN training step (disc_iters D updates + 1 G update)
of resnet_cifar10.gin (SN-ResNet, non-saturating loss) at batch 64 per GPU, bf16, synthetic data.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Rank 0 prints ONE JSON line.  `value` = real images consumed per second by the whole job
(B_global * (disc_iters + 1) / t_step), inputs resident in HBM before the
timed region.  On N = 1 two extra legs run AFTER the timed region:
  roofline     : the convolution kernel families re-run with HIP-event brackets on the launch
                 stream (cg_prof_*); achieved = useful FLOPs / kernel time of the dominant family
                 against the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline : the CPU oracle (restatement of the reference, PyTorch-CPU fp32, all host cores)
                 on the same workload for a bounded sample.  The reference's own TF1 path cannot be
                 installed here, hence kind = "port".
Self-calibration (boxes of one pool differ by +-15 % in sustained clocks): before the warm-up the
captured step is replayed for `--preheat-s` seconds (reported as `preheat_s`; `--warmup` keeps its
meaning: W untimed steps right before the timed region), the shader / memory clocks are read before
and after the timed region, and two microbenchmarks of the SAME process -- a pure bf16 MFMA loop and
a float4 copy -- give `roofline.peak_measured` / `hbm_measured`, so that `frac_of_measured` is
comparable between boxes while `frac` stays against the datasheet peak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIG_DIR = os.path.join(ROOT, "tests", "golden", "example_configs")
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--config", default="resnet_cifar10.gin")
    p.add_argument("--batch-per-gpu", type=int, default=64)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-fid", action="store_true", help="skip the FID-10k wall-clock leg")
    p.add_argument("--no-calibration", action="store_true",
                   help="skip the in-process MFMA / HBM / GEMM calibration launches (profiling passes: "
                        "the calibration GEMM would be averaged into the convolution families)")
    p.add_argument("--cpu-budget-s", type=float, default=20.0)
    p.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--no-legs", action="store_true",
                   help="skip the resnet128 D-step / BigGAN-128 legs")
    p.add_argument("--legs", default="", help="comma list of legs to run (default: all)")
    p.add_argument("--biggan-batch", type=int, default=64)
    p.add_argument("--biggan-big-batch", type=int, default=256,
                   help="per-GPU batch of the biggan128_bs256 leg (C5: global 2048 on 8 GPUs)")
    p.add_argument("--preheat-s", type=float, default=1.0,
                   help="seconds of untimed step replays before the warm-up (clock ramp)")
    p.add_argument("--cpu-mode", default="step", help=argparse.SUPPRESS)
    p.add_argument("--cpu-bindings", default="", help=argparse.SUPPRESS)
    return p.parse_args()


def cpu_baseline(config, batch, budget_s=20.0, mode="step", no_penalty=False):
    """The oracle on the host cores, bounded: whole unrolled steps (mode "step") or single
    discriminator sub-steps (mode "dstep": G forward without gradient, D forward + backward, TF-Adam
    on D -- the unit of the resnet128_dstep leg) are timed until `budget_s` seconds of CPU work have
    been spent (at least one)."""
    from oracle import arch_ops as oops
    from tests import gan_util as U
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    vs = oops.VarStore(dtype=torch.float32, seed=1)
    over = {"penalty": "no_penalty"} if no_penalty else {}
    ora = U.build_oracle(config, vs, **over)
    nsub = ora.disc_iters + 1 if mode == "step" else 1
    h, w, c = ora.image_shape
    gen = torch.Generator().manual_seed(547)
    subs = [{"images": torch.rand(batch, h, w, c, generator=gen),
             "z": torch.rand(batch, 128, generator=gen) * 2 - 1,
             "alpha": torch.rand(batch, 1, 1, 1, generator=gen)} for _ in range(nsub)]

    def dstep(sub):
        ora._ensure_opts()   # pylint: disable=protected-access
        with torch.no_grad():
            generated = ora.G(sub["z"], None)
        d_loss, _, _ = ora.create_loss(sub["images"], generated, None, None, sub["alpha"])
        ora.d_opt.step(torch.autograd.grad(d_loss, ora.d_vars()))

    def one():
        if mode == "step":
            ora.train_step(subs)
        else:
            dstep(subs[0])

    # the first unit creates the variables and the optimiser slots and pages the oneDNN kernels in
    # (5x run-to-run spread when it was part of the sample): run it untimed
    t_first = time.time()
    one()
    t_first = time.time() - t_first
    steps, t0 = 0, time.time()
    while True:
        one()
        steps += 1
        dt = time.time() - t0
        if dt >= budget_s or steps >= 8:
            break
    what = ("%d full unrolled step(s) (%d D + 1 G sub-steps each)" % (steps, ora.disc_iters)
            if mode == "step" else "%d discriminator sub-step(s)%s" % (
                steps, " without the penalty" if no_penalty else ""))
    return {"value": round(batch * nsub * steps / dt, 2), "unit": "img/s",
            "cores": cores, "kind": "port", "batch": batch,
            "sample": "%s of %s at batch %d in %.1f s, fp32 PyTorch-CPU restatement of the "
                      "reference (oracle/); one untimed unit before them (variable creation, %.1f s)" % (
                          what, config, batch, dt, t_first)}


_PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
PMC_TRAFFIC_FILE = next((p for p in (os.path.join(_PROFILES, "r06_pmc_traffic.json"),
                                     os.path.join(_PROFILES, "r05_pmc_traffic.json"),
                                     os.path.join(_PROFILES, "r04_pmc_traffic.json"))
                         if os.path.exists(p)), os.path.join(_PROFILES, "r06_pmc_traffic.json"))
PMC_TRAFFIC_NOTE = ("HBM bytes per launch of this kernel family = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                    "from two rocprofv3 --pmc passes of this workload (scripts/pmc_traffic.py -> "
                    "profiles/%s, one table per workload; counters cannot be read from inside the "
                    "timed process, so this is the committed summary of those passes, not a property "
                    "of this run); null when that summary is absent" % os.path.basename(PMC_TRAFFIC_FILE))


def pmc_traffic(family, workload="cifar"):
    """PMC counters cannot be read from inside the timed process: the committed summary of the
    separate `rocprofv3 --pmc` passes over this same command is reported (bytes per launch)."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            d = json.load(f)
        d = d.get("workloads", {}).get(workload, d if workload == "cifar" else {})
        return d["families"][family]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def read_clocks():
    """Current shader / memory clock in MHz from the amdgpu sysfs tables (the active level carries a
    `*`), or None where the box does not expose them."""
    import glob
    out = {}
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
        val = None
        for path in sorted(glob.glob("/sys/class/drm/card*/device/" + name)):
            try:
                with open(path) as f:
                    for line in f:
                        if "*" in line:
                            val = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                if val is not None:
                    break
            except (OSError, ValueError, IndexError):
                continue
        out[key] = val
    return out


def cpu_baseline_guarded(config, batch, budget_s, mode="step", no_penalty=False):
    """Runs cpu_baseline in a child process under a hard timeout so that a slow or oversubscribed
    host can never stall the benchmark line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", config,
           "--batch-per-gpu", str(batch), "--cpu-budget-s", str(budget_s), "--cpu-mode", mode,
           "--cpu-bindings", "no_penalty" if no_penalty else ""]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=max(120.0, 8 * budget_s),
                           env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        note = "cpu baseline child failed: rc=%d %s" % (r.returncode, r.stderr[-300:])
    except subprocess.TimeoutExpired:
        note = "cpu baseline exceeded its hard timeout"
    return {"value": None, "unit": "img/s", "cores": 0, "kind": "port", "sample": note}


def family_table(fam, n_prof):
    """Per-kernel-family rows of cg_prof_collect (HIP-event brackets on the launch stream)."""
    return {k: {"ms_per_step": round(v["ms"] / n_prof, 4),
                "avg_launch_us": round(1e3 * v["ms"] / v["launches"], 2),
                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                "algorithmic_GBs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                "launches_per_step": v["launches"] / n_prof}
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}


def extra_leg(config, bindings, batch, mode, steps, warmup, dev, survey_tflop=None, calib=None,
              traffic_key=None):
    """A second workload measured in the same process AFTER the headline line's timed region:
    mode "dstep" = ONE discriminator sub-step (fresh z, G forward, D forward/backward, D Adam), the
    unit BASELINE.json's north star names for the 128x128 ResNet; mode "step" = one full unrolled
    train step.  Replayed from a hipGraph, timed with a synchronize on both sides; then the same
    work runs eagerly with HIP-event brackets around every convolution launch for the per-kernel
    table and the count of useful FLOPs (useful MACs x 2 of the convolution / linear launches;
    attention, normalisation and element-wise work is NOT counted, so the fraction is conservative)."""
    from compare_gan_amd import datasets, gin, runner_lib
    from compare_gan_amd.hip import kernels as K
    gin.clear_config()
    gin.parse_config_files_and_bindings([os.path.join(CONFIG_DIR, config)], list(bindings))
    options = runner_lib.get_options_dict()
    dataset = datasets.get_dataset()
    gan = options["gan_class"](dataset=dataset, parameters=options, model_dir="/tmp/cg_bench_leg")
    gan.build(batch_size=batch, device=dev, seed=3)
    nsub = 1 if mode == "dstep" else options["disc_iters"] + 1
    batches = dataset.train_batches(batch * nsub, seed=547)
    pool = []
    for _ in range(2):
        images, labels = next(batches)
        if dataset.num_classes:
            labels = np.random.RandomState(7).randint(0, dataset.num_classes, size=labels.shape).astype(np.int32)
        pool.append((torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev)))
    eager = gan.disc_step if mode == "dstep" else gan.train_step
    run = gan.capture_disc_step() if mode == "dstep" else gan.capture_train_step()
    for i in range(warmup):
        run(*pool[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = run(*pool[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss = float(out["d_loss"] if mode == "dstep" else out["g_loss"])
    if not np.isfinite(loss):
        raise RuntimeError("non-finite loss in leg %s: %r" % (config, loss))
    K.prof_reset()
    K.prof_enable(True)
    n_prof = 2
    for i in range(n_prof):
        eager(*pool[i % 2])
    torch.cuda.synchronize()
    K.prof_enable(False)
    fam = {k: v for k, v in K.prof_collect().items() if v["launches"] > 0}
    counted = sum(v["flops"] for v in fam.values()) / n_prof / 1e12
    conv_ms = sum(v["ms"] for v in fam.values()) / n_prof
    leg = {
        "workload": "%s%s: %s, batch %d per GPU, %dx%dx%d, bf16, synthetic, hipGraph replay" % (
            config, (" [" + "; ".join(bindings) + "]") if bindings else "",
            "ONE discriminator sub-step (G forward + D forward/backward + D Adam update)"
            if mode == "dstep" else "full unrolled train step (%d D + 1 G sub-steps)" % options["disc_iters"],
            batch, *dataset.image_shape),
        "batch": batch, "steps": steps, "warmup": warmup,
        "ms": round(1e3 * dt, 4),
        "img_per_s": round(batch * nsub / dt, 1),
        "useful_tflop_counted": round(counted, 4),
        "useful_tflop_survey": survey_tflop,
        "tflops": round(counted / dt, 2),
        "frac": round(counted / dt / PEAK_BF16_TFLOPS, 4),
        "peak": PEAK_BF16_TFLOPS,
        "conv_kernel_ms_eager": round(conv_ms, 4),
        "kernels": family_table(fam, n_prof),
    }
    if calib:
        leg["peak_measured"] = round(calib["mfma_bf16_tflops"], 1)
        leg["frac_of_measured"] = round(counted / dt / calib["mfma_bf16_tflops"], 4)
    if traffic_key:
        leg["traffic"] = {k: pmc_traffic(k, traffic_key) for k in leg["kernels"]
                          if pmc_traffic(k, traffic_key) is not None}
        leg["traffic_source"] = PMC_TRAFFIC_NOTE
    del run, eager, out
    gan._graph = None   # pylint: disable=protected-access
    del gan
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return leg


def main():
    teardown_hung = False
    args = parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.config, args.batch_per_gpu, args.cpu_budget_s,
                                      args.cpu_mode, args.cpu_bindings == "no_penalty")))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # CGAMD_BENCH_DEVICE: every rank on ONE device (the dry run of the N > 1 entry point on a 1-GPU
    # box, tests/test_data_parallel_gpu.py; two RCCL ranks cannot share a device, so that run also
    # sets CGAMD_DIST_BACKEND=gloo and CGAMD_DP_GRAPH=0)
    dev_index = int(os.environ.get("CGAMD_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_dp = os.environ.get("CGAMD_FORCE_DP", "") == "1"   # exercise the RCCL path on 1 GPU
    # one process per GPU: joins the RCCL group and switches batch norm to cross-replica
    # statistics (the reference's data-parallel semantics, arch_ops.py:258-263)
    from compare_gan_amd.tpu import tpu_ops
    tpu_ops.init_replicas(dev)

    from compare_gan_amd import datasets, gin, runner_lib
    from compare_gan_amd import eval_gan_lib  # noqa: F401  (registers eval_z)
    from compare_gan_amd.gans import modular_gan  # noqa: F401
    from compare_gan_amd.hip import kernels as K

    gin.parse_config_files_and_bindings([os.path.join(CONFIG_DIR, args.config)], [])
    options = runner_lib.get_options_dict()
    dataset = datasets.get_dataset()
    bsz = args.batch_per_gpu
    gan = options["gan_class"](dataset=dataset, parameters=options, model_dir="/tmp/cg_bench")
    gan.build(batch_size=bsz, device=dev, seed=3)
    nsub = options["disc_iters"] + 1

    # synthetic inputs, resident in HBM before the timed region (datasets.py:136-145 semantics)
    batches = dataset.train_batches(bsz * nsub, seed=547 + rank)
    pool = []
    for _ in range(4):
        images, labels = next(batches)
        pool.append((torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev)))

    use_graph = not args.no_graph
    if (world > 1 or force_dp) and os.environ.get("CGAMD_DP_GRAPH", "1") == "0":
        use_graph = False     # operator override: eager launches around the RCCL all-reduces
    step_fn = gan.train_step
    if use_graph:
        # the whole step -- including the RCCL gradient all-reduces under data parallelism -- is
        # replayed from one hipGraph.  A capture the runtime refuses is an ERROR (a silently eager
        # run would be reported as if it were the captured one): CGAMD_DP_GRAPH=0 asks for eager
        # launches explicitly
        step_fn = gan.capture_train_step()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # disclosed pre-heat: the clocks of an idle box ramp for several hundred ms under load, which a
    # 0.2 s timed region right after 5 warm-up steps would sit inside of
    preheat_steps, t_pre = 0, time.perf_counter()
    while time.perf_counter() - t_pre < args.preheat_s:
        for i in range(4):
            step_fn(*pool[i % len(pool)])
        torch.cuda.synchronize()
        preheat_steps += 4
    preheat_s = time.perf_counter() - t_pre
    for i in range(args.warmup):
        step_fn(*pool[i % len(pool)])
    sync_all()
    clocks_before = read_clocks()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step_fn(*pool[i % len(pool)])
    sync_all()
    dt = time.perf_counter() - t0
    clocks_after = read_clocks()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    g_loss = float(out["g_loss"])
    if not np.isfinite(g_loss):
        raise SystemExit("non-finite generator loss after the timed region: %r" % g_loss)

    result = {
        "metric": "train img/s (G+D step)",
        "value": round(bsz * world * nsub * args.steps / dt, 2),
        "unit": "img/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "%s full unrolled train step: %d D sub-steps + 1 G sub-step, "
                               "batch %d per GPU, %dx%dx%d" % (
                                   args.config, options["disc_iters"], bsz, *dataset.image_shape),
                   "global_batch": bsz * world, "images_per_step": bsz * world * nsub,
                   "parallelism": "dp%d" % world, "hip_graph": bool(use_graph)},
        "preheat_s": round(preheat_s, 3), "preheat_steps": preheat_steps,
        "clocks": {"before": clocks_before, "after": clocks_after},
    }
    calib = None
    if rank == 0 and world == 1 and not args.no_calibration:
        # measured roofline denominators of THIS box, right after the timed region
        calib = K.calibrate(dev)
        result["calibration"] = {
            "mfma_bf16_tflops": round(calib["mfma_bf16_tflops"], 1),
            "hbm_copy_gbs": round(calib["hbm_copy_gbs"], 1),
            # the same MFMA loop on zero operands (clock not power-capped), and an
            # 8192^3 GEMM through this library's own convolution main loop (1x1 convolution)
            "mfma_zero_tflops": round(calib["mfma_zero_tflops"], 1),
            "gemm_tflops": round(calib["gemm_tflops"], 1),
            # DERIVED, not read: 1024 SIMDs x 1024 FLOP per cycle (one 32x32x16 MFMA per 32 cycles
            # and SIMD) -> the shader clock at which back-to-back MFMA issue gives this rate.  The
            # 2.5 PFLOP/s datasheet peak is that arithmetic at 2.4 GHz; the sysfs clock table of the
            # pool's boxes does not follow the load (it reads 105-157 MHz throughout), so the
            # sustained clock cannot be read directly
            "mfma_equivalent_clock_mhz": round(calib["mfma_bf16_tflops"] * 1e12 / (1024 * 1024) / 1e6, 0),
            "how": "pure v_mfma_f32_32x32x16_bf16 loop on random operands for %.0f ms (2048 "
                   "workgroups of 4 waves, 8 independent accumulators) and a float4 copy of %d MiB "
                   "(read + write bytes), HIP events, same process" % (calib["mfma_ms"], calib["copy_mb"]),
            "clocks_after": read_clocks()}

    if rank == 0 and world == 1 and not args.no_roofline:
        # HIP-event brackets on the launch stream around every convolution launch, eager replays of
        # the same step (event records cannot be captured into the hipGraph)
        K.prof_reset()
        K.prof_enable(True)
        n_prof = max(1, min(args.steps, 3))
        for i in range(n_prof):
            gan.train_step(*pool[i % len(pool)])
        torch.cuda.synchronize()
        K.prof_enable(False)
        fam = {k: v for k, v in K.prof_collect().items() if v["launches"] > 0}
        name, st = max(fam.items(), key=lambda kv: kv[1]["ms"])
        achieved = st["flops"] / (st["ms"] * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "mfma", "kernel": name, "achieved": round(achieved, 2),
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "peak_measured": round(calib["mfma_bf16_tflops"], 1) if calib else None,
            "frac_of_measured": round(achieved / calib["mfma_bf16_tflops"], 4) if calib else None,
            "hbm_peak": PEAK_HBM_GBS, "hbm_measured": round(calib["hbm_copy_gbs"], 1) if calib else None,
            "traffic": pmc_traffic(name),
            "traffic_source": PMC_TRAFFIC_NOTE,
            "launches_per_step": st["launches"] / n_prof,
            "avg_launch_us": round(1e3 * st["ms"] / st["launches"], 3),
            "avg_launch_gflop": round(st["flops"] / st["launches"] / 1e9, 3),
            "avg_launch_algorithmic_MB": round(st["bytes"] / st["launches"] / 1e6, 3),
            "flops_definition": "useful MACs x 2 of the launches of this kernel (zero-inserted taps "
                                "not counted), summed / summed kernel time",
            "kernels": family_table(fam, n_prof),
        }
    if rank == 0 and world == 1 and not args.no_fid:
        # second half of BASELINE.json's metric: FID-10k wall-clock (10,000 generated vs 10,000
        # synthetic reference images, 157 batches of 64, Inception features, fp64 statistics)
        from compare_gan_amd.metrics import fid_score as fid_lib
        from compare_gan_amd.metrics import inception_score as is_lib
        n_eval = dataset.eval_test_samples
        # the feature extractor is built (weights drawn / loaded, MFMA operand images prepared)
        # before the clock starts, as a trained graph would be loaded once per process
        from compare_gan_amd import eval_utils
        t_setup = time.perf_counter()
        eval_utils.get_inception(dev)
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t_setup
        # one small untimed evaluation first (two generator batches through every phase), as the
        # train steps have their warm-up: on a freshly leased box the first use of each kernel
        # pages its code object in from a cold image (30 s in the sampling phase on the
        # driver's fresh box, 0.3 s on a warm one).  Its wall-clock is REPORTED (cold_start_s)
        t_cold = time.perf_counter()
        eval_gan_lib.evaluate_gan(gan, [is_lib.InceptionScoreTask(), fid_lib.FIDScoreTask()],
                                  num_averaging_runs=1, num_test_examples=128)
        torch.cuda.synchronize()
        t_cold = time.perf_counter() - t_cold
        cold_split = {k: round(v, 3) for k, v in eval_gan_lib.LAST_TIMING.items()}
        t0 = time.perf_counter()
        res = eval_gan_lib.evaluate_gan(gan, [is_lib.InceptionScoreTask(), fid_lib.FIDScoreTask()],
                                        num_averaging_runs=1)
        torch.cuda.synchronize()
        t_eval = time.perf_counter() - t0
        result["fid10k"] = {
            "wall_s": round(t_eval + t_setup, 3), "eval_s": round(t_eval, 3),
            "num_examples": n_eval,
            "extractor_setup_s": round(t_setup, 3),
            "cold_start_s": round(t_cold, 3), "cold_start_split_s": cold_split,
            "wall_definition": "building the Inception extractor (weights, MFMA operand images) + "
                               "sampling + features of 2 x 10,000 images + fp64 statistics; NOT "
                               "included: cold_start_s, one untimed 128-image evaluation before it "
                               "(first use of every kernel of the evaluation path)",
            "split_s": {k: round(v, 3) for k, v in eval_gan_lib.LAST_TIMING.items()},
            # which symmetric-square-root path the statistics took (metrics/fid_score.py: the GEMM-only
            # Newton-Schulz iteration when it certifies itself, the Jacobi eigen-solver otherwise)
            "sqrt_solver": dict(fid_lib.LAST_SOLVER),
            "tridiagonal_certificate": {k: (float("%.4g" % v) if isinstance(v, float) else v)
                                        for k, v in fid_lib.LAST_TRIDIAG.items()},
            "newton_schulz": [{k: (float("%.4g" % v) if isinstance(v, float) else v)
                               for k, v in rec.items()} for rec in fid_lib.LAST_NEWTON],
            "fid": round(float(res["fid_score_mean"]), 4),
            "inception_score": round(float(res["inception_score_mean"]), 4),
            "note": "synthetic reference images; seeded (untrained) Inception weights, the trained "
                    "graph is not available offline -- wall-clock is meaningful, the values are not "
                    "comparable with published FIDs"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_guarded(args.config, bsz, args.cpu_budget_s)
    if rank == 0 and world == 1 and not args.no_legs:
        # the north star's target configurations as driver-visible numbers.
        # The headline model is released first: its captured graph pins its memory pool.
        step_fn = None
        gan._graph = None   # pylint: disable=protected-access
        gan = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        legs = [
            # D-step = 2*[B*G_fwd + 3*(2B)*D_fwd] = 3.53 useful TFLOP at B = 64
            ("resnet128_dstep", "resnet_lsun-bedroom128.gin", ("penalty.fn = @no_penalty",), 64,
             "dstep", 20, 3, 3.53),
            ("resnet128_dstep_gp", "resnet_lsun-bedroom128.gin", (), 64, "dstep", 10, 2, None),
            # C4 of BASELINE.json as written: resnet_lsun-bedroom128.gin, the whole unrolled step (5 D
            # sub-steps with the WGAN-GP double backward + 1 G sub-step) at 32 per GPU
            ("resnet_lsun128_step", "resnet_lsun-bedroom128.gin", (), 32, "step", 6, 2, None),
            # one BigGAN-128 iteration (2 D + 1 G) ~ 0.6 TFLOP per image of batch
            ("biggan128", "biggan_imagenet128.gin", (), args.biggan_batch, "step", 4, 2, None),
            # C5 of BASELINE.json: global batch 2048 on 8 GPUs = 256 per GPU
            ("biggan128_bs256", "biggan_imagenet128.gin", (), args.biggan_big_batch, "step", 2, 1,
             None),
            # C3 of BASELINE.json: sndcgan_celebahq128.gin at 32 per GPU (sndcgan.py:36-127: the 4x4 /
            # stride-2 convolutions and deconvolutions no other leg runs)
            ("sndcgan128", "sndcgan_celebahq128.gin", (), 32, "step", 10, 2, None),
        ]
        for key, cfg, binds, b, mode, k, w, survey in legs:
            if args.legs and key not in args.legs.split(","):
                continue
            try:
                result[key] = extra_leg(cfg, binds, b, mode, k, w, dev, survey, calib,
                                        "resnet128_dstep" if key == "resnet128_dstep" else None)
            except Exception as e:  # pylint: disable=broad-except
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                result[key] = {"error": repr(e)[:400]}
        if "resnet128_dstep" in result and "error" not in result["resnet128_dstep"] and \
                not args.no_cpu_baseline:
            # the oracle's discriminator sub-step on the host cores, at a REDUCED batch (8 instead of
            # 64: the CPU path is linear in the batch) -- img/s is per image
            result["resnet128_dstep"]["cpu_baseline"] = cpu_baseline_guarded(
                "resnet_lsun-bedroom128.gin", 8, args.cpu_budget_s, "dstep", True)

    if world > 1 or force_dp:
        import gc
        import torch.distributed as dist
        dist.barrier()
        # orderly teardown: the captured graph holds RCCL kernel nodes and the model holds the
        # communication streams -- release graph, model and cached blocks, THEN the communicator
        torch.cuda.synchronize()
        step_fn = None
        if getattr(gan, "_graph", None) is not None:
            gan._graph = None   # pylint: disable=protected-access
        gan = None
        out = None
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        # the communicator goes down on a helper thread with a deadline: a teardown that hangs (never
        # seen on one rank; N > 1 has not met hardware yet) must not cost the measured line
        import threading
        torn_down = threading.Event()

        def _destroy():
            try:
                dist.destroy_process_group()
            finally:
                torn_down.set()
        threading.Thread(target=_destroy, daemon=True).start()
        teardown_hung = not torn_down.wait(30.0)
    # RCCL prints its version banner through C stdio; flush it so that the JSON line is the LAST
    # line on stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # pylint: disable=broad-except
        pass
    if rank == 0:
        if "fid10k" in result:
            # the FID-10k figures once more as flat keys at the END of the line: a reader that keeps
            # only the tail of a long line still sees them
            f = result["fid10k"]
            result["fid10k_wall_s"] = f["wall_s"]
            result["fid10k_cold_start_s"] = f["cold_start_s"]
            result["fid10k_split_s"] = f["split_s"]
        # ... and so do the north-star legs (they sat mid-line and were cut off)
        for key in ("resnet128_dstep", "resnet128_dstep_gp", "resnet_lsun128_step", "biggan128",
                    "biggan128_bs256", "sndcgan128"):
            leg = result.get(key)
            if isinstance(leg, dict) and "ms" in leg:
                result[key + "_ms"] = leg["ms"]
                result[key + "_frac"] = leg["frac"]
                result[key + "_img_per_s"] = leg["img_per_s"]
        sys.stdout.write(json.dumps(result) + "\n")
    sys.stdout.flush()
    sys.stderr.flush()
    if (world > 1 or force_dp) and (os.environ.get("CGAMD_BENCH_HARD_EXIT", "0") == "1" or
                                    teardown_hung):
        # operator override, or a communicator teardown past its 30 s deadline: skip the
        # interpreter's teardown of HIP / RCCL objects
        os._exit(0)


if __name__ == "__main__":
    main()
