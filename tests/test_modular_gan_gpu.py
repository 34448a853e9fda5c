"""Model-level parity on the MI355X: generator / discriminator forward, losses and the gradients of
one D sub-step and one G sub-step of the example configs against the CPU oracle on identical
weights, images, z and labels; then whole unrolled train steps.

Two oracles are used (both fp64 arithmetic, see oracle/arch_ops.py VarStore.emulate_bf16):
  * "exact": the plain restatement of the reference.  The HIP path stores activations, their
    gradients and the MFMA weight operands in bf16; ReLU networks have piecewise-constant
    input-gradients, so a 2^-9 relative perturbation of D's input flips a few ReLU masks and moves
    deep gradients by O(sqrt(eps)) -- an exact-vs-bf16 effect that the fp64 restatement with
    emulated bf16 storage reproduces on the CPU alone (cosine 0.976 at the deepest G layer of
    resnet_cifar10.gin, tests/test_oracle_pins.py::test_bf16_storage_sensitivity).
  * "bf16-storage": the same restatement with every stored tensor snapped to the bf16 grid, which
    removes that effect and leaves only accumulation-order / rounding-placement noise.

Stated tolerances:
  images      max |diff| <= 0.03 (values in [0,1]), mean |diff| <= 5e-3                 (exact)
  losses      |diff| <= 2e-2 * max(1, |ref|)                                             (exact)
              generator loss taken AFTER several Adam updates of D within a step: 4e-2 (its
              spread is 0.3-2.9 % between builds that differ in fp32 summation order only)
  gradients   per tensor: cosine >= COS and rel-L2 <= REL, or (tensors that are ~0 by
              construction, e.g. a bias in front of batch norm) |diff| <= 2e-3 * largest grad norm
              exact oracle:        COS 0.97, REL 0.35 (sndcgan_celebahq128.gin, whose generator
                                   gradient crosses 4 deconvolutions + 7 D convolutions at
                                   128x128: COS 0.90, REL 0.45)
              bf16-storage oracle: COS 0.99, REL 0.15
  WGAN-GP     the penalty's own gradient (double backward) is checked in isolation:
              cosine >= 0.999, rel-L2 <= 0.05 against the bf16-storage oracle; the full d_loss
              gradient uses generator outputs as fakes (real-vs-real noise makes the Wasserstein
              term a difference of two nearly equal sums, ill-conditioned in ANY 8-bit-mantissa
              storage: measured cosine 0.86-0.95 there with the penalty term at 0.9999).
  The batch-2 / batch-8 cases above are WIRING guards (their bands are as wide as the bf16-storage
  oracle itself is from the exact one at those sizes); the tight checks run at the sizes that are
  benchmarked, against an fp64 oracle resident on the device: every gradient of the ResNet5-128
  D sub-step at batch 64 at cosine >= 0.999 / rel-L2 <= 0.06 (measured worst 0.99958), BigGAN
  ch = 96 at batch 64 (D and G sub-steps, >= 0.999), BigGAN at 256 px -- the three tests named
  *_at_the_benchmark_batch / test_biggan_256px below.
"""
import numpy as np
import pytest
import torch

from tests import gan_util as U

pytestmark = pytest.mark.gpu

SEED = 3
TOL = {False: (0.97, 0.35), True: (0.99, 0.15)}


def _check_grads(named_product, oracle_grads, what, cos_min, rel_max, bias_tol=None):
    """Per-variable cosine / relative L2 of the product's gradients against the oracle's; variables
    whose absolute error is below 0.2 % of the largest gradient norm are not judged by ratio.
    bias_tol: (cos_min, rel_max) for variables named */bias (see the call site that passes it).
    CGAMD_TEST_REPORT=1 prints every variable's figures (sorted) before asserting."""
    import os
    norms = [float(g.norm()) for g in oracle_grads]
    big = max(norms)
    rows = []
    for (name, p), go in zip(named_product, oracle_grads):
        assert p.grad is not None, "%s: %s has no gradient" % (what, name)
        go = go.detach().cpu()
        diff = float((p.grad.detach().double().cpu().reshape(-1) - go.reshape(-1)).norm())
        if diff <= 2e-3 * big:
            continue
        rows.append((U.cosine(p.grad, go), U.rel_l2(p.grad, go), name))
    if os.environ.get("CGAMD_TEST_REPORT"):
        for c, r, name in sorted(rows):
            print("  %-70s cos %.5f rel %.4f" % (name, c, r))
    for c, r, name in rows:
        lo, hi = bias_tol if (bias_tol is not None and name.endswith("/bias")) else (cos_min, rel_max)
        assert c >= lo and r <= hi, "%s: grad of %s cosine %.5f rel-L2 %.4f" % (what, name, c, r)
    return min([(c, name) for c, _, name in rows], default=(1.0, None))


def _substep_inputs(gan, dataset, bsz, sub, step, conditional):
    rng = np.random.RandomState(100 + sub)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    labels = torch.from_numpy(rng.randint(0, dataset.num_classes or 1, size=bsz).astype(np.int32))
    return images, labels


@pytest.mark.parametrize("emulate", [False, True], ids=["exact", "bf16-storage"])
@pytest.mark.parametrize("config,bsz", [("resnet_cifar10.gin", 8), ("dcgan_celeba64.gin", 4),
                                        ("sndcgan_celebahq128.gin", 2)])
def test_forward_and_gradients(dev, config, bsz, emulate):
    _forward_and_gradients(dev, config, bsz, emulate)


def test_forward_and_gradients_at_the_benchmark_batch(dev):
    """resnet_cifar10.gin at batch 64 per sub-step -- the shape bench.py measures, where the
    dispatcher selects the 128-pixel-row tiles, the deep rings and the halo splits it never uses at
    batch 8 -- against the bf16-storage oracle."""
    _forward_and_gradients(dev, "resnet_cifar10.gin", 64, True)


def test_self_modulated_batch_norm_generator(dev):
    """G.batch_norm_fn = @self_modulated_batch_norm (arch_ops.py:370-420: gamma / beta from a
    two-layer MLP on z, per sample) in resnet_cifar10.gin: forward and gradients, including the
    MLP's own weights, against the bf16-storage oracle."""
    from oracle import arch_ops as oops
    from oracle import architectures as OA
    _forward_and_gradients(
        dev, "resnet_cifar10.gin", 8, True,
        bindings=("G.batch_norm_fn = @self_modulated_batch_norm",),
        oracle_overrides=dict(g_cfg=lambda: OA.ArchConfig(
            batch_norm_fn="self_modulated_batch_norm", bn_cfg=oops.BNConfig(0.9, 1e-5))))


@pytest.mark.parametrize("config,bsz,tol,tol_g", [
    ("sndcgan_celebahq128.gin", 32, (0.998, 0.06), (0.99, 0.15)),
    ("dcgan_celeba64.gin", 16, (0.997, 0.08), (0.997, 0.08))])
def test_forward_and_gradients_at_the_baseline_batch(dev, config, bsz, tol, tol_g):
    """BASELINE.json configs[2] / configs[0] at THEIR batch sizes: sndcgan_celebahq128.gin at 32 per
    GPU (sndcgan.py:36-127: 4x4 / stride-2 and 3x3 convolutions with spectral norm, 4x4 / stride-2
    deconvolutions with batch norm, 128x128) and dcgan_celeba64.gin at 16 (dcgan.py:39-129: 5x5 /
    stride-2 both ways, batch norm in G and D), generator forward, D sub-step and G sub-step losses
    and every gradient against the bf16-storage oracle resident on the device (per-tap fp64 GEMMs,
    oracle/arch_ops.py conv2d_same_gemm / conv2d_transpose_same_gemm).  The batch-2 / batch-4 cases
    above stay as wiring guards with their wide exact-oracle band; this is the tight check.
    Tolerance per variable (cosine, rel-L2), measured in round 4: sndcgan at 32 worst cosine 0.99877
    / rel-L2 0.0495 (discriminator/d_conv7/bias, everything else >= 0.999) -> 0.998 / 0.06; dcgan at
    16 worst 0.99789 / 0.0650 (generator/g_fc1/kernel: 16 samples through four batch norms) ->
    0.997 / 0.08.  The ResNet5 test at batch 64 holds 0.999 / 0.06; these two run at a half and a
    quarter of that batch because BASELINE.json names them so.  The generator gradients of sndcgan
    cross four deconvolutions with batch norm and seven discriminator convolutions at 128x128: its
    first layer measured cosine 0.99413 / rel-L2 0.108 (generator/g_fc1/kernel), so the G sub-step
    carries the bf16-storage band of the small-batch tests (0.99 / 0.15); its D sub-step the tight
    one."""
    _forward_and_gradients(dev, config, bsz, True, oracle_device=dev, tol=tol, tol_g=tol_g)


def _forward_and_gradients(dev, config, bsz, emulate, bindings=(), oracle_overrides=None,
                           oracle_device="cpu", tol=None, tol_g=None):
    from compare_gan_amd.architectures import arch_ops as ops
    od = oracle_device
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED, bindings=bindings)
    vs = U.mirror_to_oracle(gan, emulate_bf16=emulate, device=od)
    ora = U.build_oracle(config, vs, **(oracle_overrides or {}))
    if bindings:
        assert any("sbn/" in n for n, _ in gan.store.trainable_variables("generator"))
    cos_min, rel_max = tol if tol is not None else TOL[emulate]
    if config.startswith("sndcgan") and not emulate:
        # batch 2, exact oracle: a wiring guard only (the generator gradient crosses 4 deconvolutions
        # + 7 D convolutions at 128x128 and the exact-vs-bf16 mask flips dominate at 2 samples); the
        # tight check of this config is test_forward_and_gradients_at_the_baseline_batch (batch 32)
        cos_min, rel_max = 0.90, 0.45
    images, labels = _substep_inputs(gan, dataset, bsz, 0, 0, False)
    z = U.host_uniform((bsz, options["z_dim"]), "z/0", -1.0, 1.0, SEED, 0)

    # ---- generator forward (training mode) ----
    with ops.use_store(gan.store):
        zd = gan.z_generator([bsz, options["z_dim"]], name="z/0")
        assert torch.equal(zd.cpu(), z), "device z differs from the host Philox stream"
        with torch.no_grad():
            gen = gan.generator(zd, y=None, is_training=True)
    with torch.no_grad():
        gen_o = ora.G(z.double().to(od), None).cpu()
    diff = (gen.cpu().double() - gen_o).abs()
    assert float(diff.max()) <= 0.03 and float(diff.mean()) <= 5e-3, (float(diff.max()), float(diff.mean()))

    # ---- D sub-step: loss + gradients w.r.t. D's variables ----
    gen_in = gen_o.float()   # both sides see the SAME fake images from here on
    feats = {"images": images.to(dev), "generated": gen_in.to(dev)}
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    with ops.use_store(gan.store):
        gan.create_loss(feats, labels.to(dev))
    gan.d_loss.backward()
    d_loss_o, _, logits_o = ora.create_loss(images.double().to(od), gen_in.double().to(od), None, None)
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 2e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    w = _check_grads(gan.store.trainable_variables("discriminator"), grads_o, config + " D-step",
                     cos_min, rel_max)
    print(config, "D-step worst grad cosine", w)

    # ---- G sub-step: loss + gradients w.r.t. G's variables (fresh D forward, D frozen) ----
    gan._set_requires_grad(gan.d_opt, False)
    gan._set_requires_grad(gan.g_opt, True)
    gan._zero_grads(gan.g_opt)
    with ops.use_store(gan.store):
        feats = {"images": images.to(dev), "_generator_step": True,
                 "generated": gan.generator(zd, y=None, is_training=True)}
        gan.create_loss(feats, labels.to(dev))
    gan.g_loss.backward()
    gen_o2 = ora.G(z.double().to(od), None)
    _, g_loss_o, _ = ora.create_loss(images.double().to(od), gen_o2, None, None, with_penalty=False)
    ggrads_o = torch.autograd.grad(g_loss_o, ora.g_vars())
    assert abs(float(gan.g_loss.detach()) - float(g_loss_o.detach())) <= 2e-2 * max(
        1.0, abs(float(g_loss_o.detach())))
    gc, gr = tol_g if tol_g is not None else (cos_min, rel_max)
    w = _check_grads(gan.store.trainable_variables("generator"), ggrads_o, config + " G-step", gc, gr)
    print(config, "G-step worst grad cosine", w)
    # spectral-norm vectors and BN moving averages moved in lock-step with the oracle
    for name, v in gan.store.vars.items():
        if name.endswith("u_var") or "moving_" in name:
            assert U.rel_l2(v, vs.vars[name]) <= 2e-2, name


@pytest.mark.parametrize("bsz", [8, 64])
def test_train_steps_resnet_cifar(dev, bsz):
    """Full unrolled steps (5 D sub-steps + 1 G sub-step each) of resnet_cifar10.gin through
    train_step() vs the bf16-storage oracle, at a toy batch (two steps) and at the batch bench.py
    measures (64: the dispatcher picks the kernel variants of the benchmark there).

    Every sub-step is compared FROM IDENTICAL STATES (U.stepwise_parity, installed as the step's
    sub_step_hook): losses within 2e-3, each network's update at cosine >= 0.98.  Round 4 compared
    the whole step with a free-running oracle and had to widen the generator-loss band to 4e-2;
    the round-5 sweeps show why that comparison cannot be tight -- it is chaotic (over eight seeds
    product, bf16-storage oracle and exact oracle end up 1e-4 ... 2e-1 apart in the generator loss,
    profiles/r05_gloss_spread.txt) -- while from identical states the sub-steps agree to 3e-4
    (profiles/r05_gloss_stepwise.txt).  The free-running D losses of THIS seed are still reported
    and held to the 2e-2 of round 1.  Step counters follow modular_gan_test.py:175-177."""
    config = "resnet_cifar10.gin"
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED)
    ora = U.build_oracle(config, U.mirror_to_oracle(gan, emulate_bf16=True))
    free = U.build_oracle(config, U.mirror_to_oracle(gan, emulate_bf16=True)) if bsz <= 8 else None
    nsub = options["disc_iters"] + 1
    nsteps = 2 if bsz <= 8 else 1     # (the fp64 oracle needs ~20 s per step at batch 64)
    for step in range(nsteps):
        rng = np.random.RandomState(500 + step)
        images = rng.uniform(size=(nsub * bsz,) + dataset.image_shape).astype(np.float32)
        labels = np.ones((nsub * bsz,), dtype=np.int32)
        subs = [{"images": torch.from_numpy(images[i * bsz:(i + 1) * bsz]).double(),
                 "z": U.host_uniform((bsz, 128), "z/%d" % i, -1.0, 1.0, SEED, step).double()}
                for i in range(nsub)]
        check = U.stepwise_parity(gan, ora, subs, lr_d=2e-4)
        gan.sub_step_hook = check
        try:
            out = gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
        finally:
            gan.sub_step_hook = None
        d_o, g_o = check.finish(out)
        d_p = [float(x) for x in out["d_losses"]]
        print("step", step, "d", d_p, d_o, "g", float(out["g_loss"]), g_o, "update cosines", check.rows)
        assert len(check.rows) == nsub
        for name, v in gan.store.trainable_variables():
            assert float(v.detach().abs().max()) > 0.0
        if free is not None and step == 0:
            # the same step free-running (the comparison of rounds 1-4): D losses at their old band
            d_f, g_f = free.train_step(subs)
            print("free-running oracle d", d_f, "g", g_f)
            for a, b in zip(d_p, d_f):
                assert abs(a - b) <= 2e-2 * max(1.0, abs(b)), (a, b)
    assert int(gan.global_step.item()) == nsteps
    assert int(gan.global_step_disc.item()) == nsteps * options["disc_iters"]


def _wgangp_setup(dev, emulate, bsz=2, oracle_device="cpu", penalty=True):
    """Product + oracle of resnet_lsun-bedroom128.gin on identical variables; real images, fakes
    (the ORACLE's generator output) and the penalty's alpha as CPU tensors."""
    config = "resnet_lsun-bedroom128.gin"
    binds = () if penalty else ("penalty.fn = @no_penalty",)
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED, bindings=binds)
    vs = U.mirror_to_oracle(gan, emulate_bf16=emulate, device=oracle_device)
    ora = U.build_oracle(config, vs, **({} if penalty else {"penalty": "no_penalty"}))
    rng = np.random.RandomState(7)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    z = U.host_uniform((bsz, options["z_dim"]), "z/0", -1.0, 1.0, SEED, 0)
    with torch.no_grad():
        fake = ora.G(z.double().to(oracle_device), None).float().cpu()
    alpha = U.host_uniform((bsz,), "wgangp_penalty/alpha", 0.0, 1.0, SEED, 0)
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    return gan, ora, images, fake, alpha


@pytest.mark.parametrize("penalty", [False, True], ids=["no_penalty", "wgangp"])
def test_resnet128_d_substep_at_the_benchmark_batch(dev, penalty):
    """The units bench.py measures as `resnet128_dstep` / `resnet128_dstep_gp`: the D sub-step of
    resnet_lsun-bedroom128.gin at batch 64 (128 images through D, 128x128), loss and every D
    gradient against the bf16-storage oracle.  The oracle runs on the device here -- the same fp64
    restatement on plain torch ops (convolutions as per-tap fp64 GEMMs, oracle/arch_ops.py
    conv2d_same_gemm, pinned to the direct-loop restatement by tests/test_oracle_direct.py); on the
    CPU one case takes minutes.  The dispatcher picks the kernels of the benchmark at this size
    (pooled epilogues, hwgrad with pooled dy, grouped small-map weight gradients, sconv on the 4x4
    block) instead of the small-grid variants the batch-2 tests exercise.
    Tolerance: cosine >= 0.999, rel-L2 <= 0.06 per variable -- ten times tighter than the
    bf16-storage band of the batch-2 tests (0.99 / 0.15), because 128 images average the
    rounding-boundary flips out; measured in round 3: worst cosine 0.99958 (B5/same_conv1/bias) with
    and without the penalty, losses equal to 5 significant digits."""
    from compare_gan_amd.architectures import arch_ops as ops
    bsz = 64
    gan, ora, images, fake, alpha = _wgangp_setup(dev, True, bsz=bsz, oracle_device=dev,
                                                  penalty=penalty)
    with ops.use_store(gan.store):
        gan.create_loss({"images": images.to(dev), "generated": fake.to(dev)}, None)
    assert (gan.penalty_loss is not None) == penalty
    gan.d_loss.backward()
    d_loss_o, _, _ = ora.create_loss(images.double().to(dev), fake.double().to(dev), None, None,
                                     alpha.double().to(dev))
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    print("resnet128 D sub-step bs64 d_loss", float(gan.d_loss.detach()), float(d_loss_o.detach()))
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 2e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    w = _check_grads(gan.store.trainable_variables("discriminator"), grads_o,
                     "resnet128 D sub-step bs64", 0.999, 0.06)
    print("resnet128 D sub-step bs64 worst grad cosine", w)


def test_train_step_resnet_lsun128(dev):
    """resnet_lsun-bedroom128.gin AS WRITTEN (BASELINE.json configs[3]; bench.py leg
    `resnet_lsun128_step`): one whole unrolled train_step() -- five discriminator sub-steps with the
    WGAN-GP double backward (penalty_lib.py:59-82, modular_gan.py:512-604) and one generator sub-step --
    against the bf16-storage oracle SUB-STEP BY SUB-STEP from identical states (U.stepwise_parity: losses,
    the gradient recovered from the Adam slots, the update, the step counters).  Batch 8, the oracle
    resident on the device (fp64 per-tap GEMMs in plain torch; on the CPU the six double-backward
    sub-steps at 128 x 128 take many minutes).  The penalty's interpolation coefficients are the
    product's own Philox stream (penalty_lib.alpha_name per sub-step)."""
    from compare_gan_amd.gans import penalty_lib
    config = "resnet_lsun-bedroom128.gin"
    bsz = 8
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED)
    assert options["disc_iters"] == 5
    ora = U.build_oracle(config, U.mirror_to_oracle(gan, emulate_bf16=True, device=dev))
    nsub = options["disc_iters"] + 1
    rng = np.random.RandomState(321)
    images = rng.uniform(size=(nsub * bsz,) + tuple(dataset.image_shape)).astype(np.float32)
    labels = np.ones((nsub * bsz,), dtype=np.int32)
    subs = [{"images": torch.from_numpy(images[i * bsz:(i + 1) * bsz]).double().to(dev),
             "z": U.host_uniform((bsz, options["z_dim"]), "z/%d" % i, -1.0, 1.0, SEED, 0).double().to(dev),
             "alpha": U.host_uniform((bsz,), penalty_lib.alpha_name(i), 0.0, 1.0, SEED, 0).double().to(dev)}
            for i in range(nsub)]
    # measured (round 6, batch 8): losses within 3e-4, gradient cosines >= 0.9990 (D, through the
    # double backward) / 0.99996 (G), update cosines >= 0.985 / 0.991
    check = U.stepwise_parity(gan, ora, subs, lr_d=1e-4, lr_g=1e-4, cos_grad=0.998, cos_min=0.97,
                              cos_min_g=0.88)
    gan.sub_step_hook = check
    try:
        out = gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    finally:
        gan.sub_step_hook = None
    d_o, g_o = check.finish(out)
    print("lsun128 step d", [float(x) for x in out["d_losses"]], d_o, "g", float(out["g_loss"]), g_o,
          "cosines (update, gradient)", [(r[0][0], round(r[1], 5), round(r[2], 5)) for r in check.rows])
    assert len(check.rows) == nsub
    assert int(gan.global_step.item()) == 1 and int(gan.global_step_disc.item()) == 5


def test_wgangp_step_with_layer_norm(dev):
    """resnet_lsun-bedroom128.gin with `D.layer_norm = True` -- the pairing layer norm exists for
    (resnet_ops.py:162-173 under penalty_lib.py:59-82): Wasserstein loss + gradient penalty at
    128x128, batch 2; the double backward crosses every layer norm of the discriminator
    (LayerNormBwdFn / cg_layer_norm_bwd_bwd).  Loss and every D gradient against the bf16-storage
    oracle; the band is the one of test_wgangp_step_resnet5 (batch 2: summation order decides a few
    rounding flips that the double backward amplifies)."""
    from compare_gan_amd.architectures import arch_ops as ops
    from oracle import architectures as OA
    config = "resnet_lsun-bedroom128.gin"
    bsz = 2
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED,
                                            bindings=("D.layer_norm = True",))
    assert any("/ln1/" in n for n, _ in gan.store.trainable_variables("discriminator"))
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    ora = U.build_oracle(config, vs, d_cfg=lambda: OA.ArchConfig(spectral_norm=False, layer_norm=True))
    rng = np.random.RandomState(7)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    z = U.host_uniform((bsz, options["z_dim"]), "z/0", -1.0, 1.0, SEED, 0)
    with torch.no_grad():
        fake = ora.G(z.double(), None).float()
    alpha = U.host_uniform((bsz,), "wgangp_penalty/alpha", 0.0, 1.0, SEED, 0)
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    with ops.use_store(gan.store):
        gan.create_loss({"images": images.to(dev), "generated": fake.to(dev)}, None)
    assert gan.penalty_loss is not None
    gan.d_loss.backward()
    d_loss_o, _, _ = ora.create_loss(images.double(), fake.double(), None, None, alpha.double())
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    print("layer-norm wgangp d_loss", float(gan.d_loss.detach()), float(d_loss_o.detach()))
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 3e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    w = _check_grads(gan.store.trainable_variables("discriminator"), grads_o,
                     "wgangp + layer norm D-step", 0.98, 0.20, bias_tol=(0.95, 0.35))
    print("layer-norm wgangp worst grad cosine", w)


def test_wgangp_penalty_gradient(dev):
    """The gradient penalty alone (penalty_lib.py:59-82): value and its gradient w.r.t. every D
    kernel -- the double backward through D, all on HIP kernels -- at 128x128, batch 2."""
    from compare_gan_amd.architectures import arch_ops as ops
    from compare_gan_amd.gans import penalty_lib
    from oracle import gan as ogan
    gan, ora, images, fake, alpha = _wgangp_setup(dev, True)
    with ops.use_store(gan.store):
        pen = penalty_lib.get_penalty_loss(x=images.to(dev), x_fake=fake.to(dev), y=None,
                                           is_training=True, discriminator=gan.discriminator)
    pen.backward()
    pen_o = ogan.wgangp_penalty(lambda x, yy, t: ora.D(x, yy, t), images.double(), fake.double(),
                                None, True, alpha.double().reshape(-1, 1, 1, 1))
    grads_o = torch.autograd.grad(pen_o, ora.d_vars(), allow_unused=True)
    assert abs(float(pen.detach()) - float(pen_o.detach())) <= 1e-3 * abs(float(pen_o.detach()))
    checked = 0
    for (name, p), go in zip(gan.store.trainable_variables("discriminator"), grads_o):
        if name.endswith("/bias"):
            # d penalty / d bias == 0: the input-gradient of a ReLU network does not depend
            # (differentiably) on its biases
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        c, r = U.cosine(p.grad, go), U.rel_l2(p.grad, go)
        assert c >= 0.999 and r <= 0.05, "penalty grad of %s cosine %.5f rel-L2 %.4f" % (name, c, r)
        checked += 1
    assert checked == 19


@pytest.mark.parametrize("emulate", [False, True], ids=["exact", "bf16-storage"])
def test_wgangp_step_resnet5(dev, emulate):
    """resnet_lsun-bedroom128.gin: Wasserstein loss + lambda * gradient penalty at the full
    128x128 resolution, batch 2; fakes are generator outputs."""
    from compare_gan_amd.architectures import arch_ops as ops
    gan, ora, images, fake, alpha = _wgangp_setup(dev, emulate)
    with ops.use_store(gan.store):
        gan.create_loss({"images": images.to(dev), "generated": fake.to(dev)}, None)
    pen_p = float(gan.penalty_loss)
    gan.d_loss.backward()
    d_loss_o, _, _ = ora.create_loss(images.double(), fake.double(), None, None, alpha.double())
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    print("wgangp d_loss", float(gan.d_loss.detach()), float(d_loss_o.detach()), "penalty", pen_p)
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 3e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    # bf16-storage oracle: same rounding points, but the fp32 summation ORDER inside a convolution
    # differs (MFMA tiles, split-K groups), so individual bf16 roundings flip and the double backward
    # of the penalty amplifies that: the first-block shortcut kernel sits at cosine 0.988-0.993
    # depending on the kernel variant in use; everything else is > 0.995
    # exact oracle: the Wasserstein part of these gradients is a difference of two nearly equal sums
    # over ReLU masks (real minus fake, batch 2), so which masks flip under 8-bit-mantissa storage
    # decides the figure: the bias gradients of the LAST block (4x4, plain kernels, nothing fused)
    # measure 0.86-0.96 depending on where the earlier layers round (0.964 with the poolings as
    # separate kernels, 0.864 with them fused into the convolutions, round 2).  The tight checks are
    # the bf16-storage oracle here (>= 0.98) and the penalty's own gradient (>= 0.999,
    # test_wgangp_penalty_gradient); the exact comparison only guards against wiring errors.
    # bias gradients at batch 2 are sums of a few hundred dy values whose sign pattern follows the
    # flipped masks: the mid-block shortcut biases measure 0.979-0.993 against the bf16-storage oracle
    # depending on which kernel variant (summation order) serves the layers above them, so they carry
    # 0.95 here; the SAME variables are held to cosine >= 0.999 at the benchmark batch
    # (test_resnet128_d_substep_at_the_benchmark_batch, worst 0.99958 over all variables).
    w = _check_grads(gan.store.trainable_variables("discriminator"), grads_o, "wgangp D-step",
                     0.98 if emulate else 0.80, 0.20 if emulate else 0.60,
                     bias_tol=(0.95, 0.35) if emulate else None)
    print("wgangp worst grad cosine", w)


def _biggan_family_forward_and_gradients(dev, label, bind, g_over, d_over, arch=None,
                                         min_g_grads=40, fwd_tol=(0.05, 5e-3),
                                         d_tol=(0.98, 0.2), g_tol=(0.97, 0.3), check_u=True,
                                         bsz=2, g_step=True, oracle_device="cpu",
                                         image_shape=None):
    """Generator forward, D sub-step and G sub-step losses and gradients of a BigGAN-family
    architecture under biggan_imagenet128.gin against the bf16-storage oracle (128x128)."""
    from compare_gan_amd.architectures import arch_ops as ops
    from oracle import architectures as OA
    from oracle import arch_ops as oops
    config = "biggan_imagenet128.gin"
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED, bindings=bind)
    vs = U.mirror_to_oracle(gan, emulate_bf16=True, device=oracle_device)
    od = oracle_device
    overrides = dict(
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="conditional_batch_norm", spectral_norm=True,
                                    bn_cfg=oops.BNConfig(0.9, 1e-5, use_moving_averages=False),
                                    sn_cfg=oops.SNConfig(singular_value="auto"), **g_over),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=True, sn_cfg=oops.SNConfig(singular_value="auto"),
                                    **d_over))
    if arch is not None:
        overrides["architecture"] = arch
    if image_shape is not None:
        overrides["image_shape"] = image_shape
    ora = U.build_oracle(config, vs, **overrides)
    rng = np.random.RandomState(11)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    labels = ((torch.arange(bsz) * 487 + 3) % 1000).to(torch.int32)      # 3, 490, 977, ...
    sampled = ((torch.arange(bsz) * 395 + 5) % 1000).to(torch.int32)     # 5, 400, 795, ...
    z = U.host_normal((bsz, options["z_dim"]), "z/0", 0.0, 1.0, SEED, 0)
    with ops.use_store(gan.store):
        zd = gan.z_generator([bsz, options["z_dim"]], name="z/0")
        assert float((zd.cpu() - z).abs().max()) <= 1e-5, "device z differs from the host stream"
        sy = gan._get_one_hot_labels(sampled.to(dev))
        with torch.no_grad():
            gen = gan.generator(zd, y=sy, is_training=True)
    with torch.no_grad():
        gen_o = ora.G(z.double().to(od), ora.one_hot(sampled.to(od))).cpu()
    diff = (gen.cpu().double() - gen_o).abs()
    print(label, "generator output max / mean abs diff", float(diff.max()), float(diff.mean()))
    assert float(diff.max()) <= fwd_tol[0] and float(diff.mean()) <= fwd_tol[1], (
        float(diff.max()), float(diff.mean()))

    # the generator forward above ran one power iteration on G's u vectors on both sides; D next
    gen_in = gen_o.float()
    feats = {"images": images.to(dev), "generated": gen_in.to(dev), "sampled_labels": sampled.to(dev)}
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    with ops.use_store(gan.store):
        gan.create_loss(feats, labels.to(dev))
    gan.d_loss.backward()
    d_loss_o, _, _ = ora.create_loss(images.double().to(od), gen_in.double().to(od), labels.to(od),
                                     sampled.to(od))
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    print(label + " d_loss", float(gan.d_loss.detach()), float(d_loss_o.detach()))
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 3e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    w = _check_grads(gan.store.trainable_variables("discriminator"), grads_o, label + " D-step",
                     d_tol[0], d_tol[1])
    print(label + " D-step worst grad cosine", w)

    if not g_step:
        return
    gan._set_requires_grad(gan.d_opt, False)
    gan._set_requires_grad(gan.g_opt, True)
    gan._zero_grads(gan.g_opt)
    with ops.use_store(gan.store):
        feats = {"images": images.to(dev), "_generator_step": True, "sampled_labels": sampled.to(dev),
                 "generated": gan.generator(zd, y=sy, is_training=True)}
        gan.create_loss(feats, labels.to(dev))
    gan.g_loss.backward()
    gen_o2 = ora.G(z.double().to(od), ora.one_hot(sampled.to(od)))
    _, g_loss_o, _ = ora.create_loss(images.double().to(od), gen_o2, labels.to(od), sampled.to(od),
                                     with_penalty=False)
    ggrads_o = torch.autograd.grad(g_loss_o, ora.g_vars(), allow_unused=True)
    print(label + " g_loss", float(gan.g_loss.detach()), float(g_loss_o.detach()))
    assert abs(float(gan.g_loss.detach()) - float(g_loss_o.detach())) <= 3e-2 * max(
        1.0, abs(float(g_loss_o.detach())))
    named = [(n, p) for (n, p), go in zip(gan.store.trainable_variables("generator"), ggrads_o)
             if go is not None and p.grad is not None]
    ggo = [go for go in ggrads_o if go is not None]
    assert len(named) == len(ggo) >= min_g_grads
    w = _check_grads(named, ggo, label + " G-step", g_tol[0], g_tol[1])
    print(label + " G-step worst grad cosine", w)
    worst_u = max((U.rel_l2(v, vs.vars[name]), name) for name, v in gan.store.vars.items()
                  if name.endswith("u_var"))
    print(label, "worst power-iteration vector rel-L2", worst_u)
    if check_u:
        assert worst_u[0] <= 3e-2, worst_u


def test_biggan_forward_and_gradients(dev):
    """biggan_imagenet128.gin (class-conditional hinge, conditional BN on hierarchical z + embedded
    labels, spectral norm "auto" in G and D, self-attention at 64x64, projection discriminator) at
    128x128, batch 2, with the reference's own width binding ch = 32 to keep the fp64 oracle fast."""
    _biggan_family_forward_and_gradients(
        dev, "biggan", ["resnet_biggan.Generator.ch = 32", "resnet_biggan.Discriminator.ch = 32"],
        dict(hierarchical_z=True, embed_y=True, ch=32), dict(project_y=True, ch=32))


def test_biggan_full_width_d_step(dev):
    """biggan_imagenet128.gin at its own width ch = 96 (the benchmark's channel counts: 96 ... 1536,
    multiples of 32 that are not multiples of 64 -- the half-empty last channel block of the halo
    kernels), batch 2: generator forward and the D sub-step against the bf16-storage oracle
    (VERDICT r01, item 6)."""
    _biggan_family_forward_and_gradients(
        dev, "biggan-ch96", [], dict(hierarchical_z=True, embed_y=True, ch=96),
        dict(project_y=True, ch=96), g_step=False)


def test_biggan_at_the_benchmark_batch(dev):
    """bench.py's `biggan128` leg: biggan_imagenet128.gin at its own width (ch = 96) and batch 64
    -- generator forward, D sub-step and G sub-step (losses, every gradient, the power-iteration
    vectors) against the bf16-storage oracle resident on the device (see
    test_resnet128_d_substep_at_the_benchmark_batch).  The BigGAN family's standard tolerances."""
    _biggan_family_forward_and_gradients(
        dev, "biggan-ch96-bs64", [], dict(hierarchical_z=True, embed_y=True, ch=96),
        dict(project_y=True, ch=96), bsz=64, oracle_device=dev,
        # measured in round 3: generator output max / mean |diff| 0.016 / 1.1e-3, worst D-step gradient
        # cosine 0.99951, worst G-step 0.99961, power-iteration vectors within 3.3e-7
        fwd_tol=(0.03, 2e-3), d_tol=(0.999, 0.06), g_tol=(0.999, 0.06))


def test_biggan_d_substep_at_the_c5_batch(dev):
    """BASELINE.json configs[4] / bench.py's `biggan128_bs256` leg: biggan_imagenet128.gin at 256 per
    GPU (global 2048 on 8 GPUs, example_configs/biggan_imagenet128.gin:15) -- the dispatcher's
    choices change with the batch (tile sizes, pixel splits of the weight gradients, grouped
    launches), and the tests stopped at 64 (VERDICT r04 missing 4).  Generator forward (256
    samples) and the D sub-step on 512 images -- loss, every gradient -- against the bf16-storage
    oracle resident on the device; the G sub-step's shapes are the generator's at the same batch
    and the discriminator's at half of it, both covered here.  Tolerances of the batch-64 test."""
    torch.cuda.empty_cache()
    _biggan_family_forward_and_gradients(
        dev, "biggan-ch96-bs256", [], dict(hierarchical_z=True, embed_y=True, ch=96),
        dict(project_y=True, ch=96), bsz=256, oracle_device=dev, g_step=False,
        fwd_tol=(0.03, 2e-3), d_tol=(0.999, 0.06))
    torch.cuda.empty_cache()


def test_biggan_g_substep_at_half_the_c5_batch(dev):
    """The generator sub-step of the same configuration at 128 per GPU (the parity
    of the G sub-step stopped at batch 64).  256 was tried first: the fp64 autograd graph of the
    device-resident oracle through G and D at 128 x 128 needs more than the 288 GB of the card (269 GB
    allocated when it gave up), so this is the largest power of two whose oracle fits; the kernels'
    batch-dependent choices at 256 (tile sizes, pixel splits, grouped launches) are covered by the D
    sub-step test above on 512 images and the generator forward at 256.  Generator forward with
    gradients, the discriminator on the generated images, hinge generator loss, every generator
    gradient and the power-iteration vectors against the bf16-storage oracle; the D sub-step runs
    first, as in a training step.  gans/modular_gan.py:486-508, resnet_biggan.py:99-151,344-425."""
    torch.cuda.empty_cache()
    _biggan_family_forward_and_gradients(
        dev, "biggan-ch96-bs128-g", [], dict(hierarchical_z=True, embed_y=True, ch=96),
        dict(project_y=True, ch=96), bsz=128, oracle_device=dev, g_step=True,
        fwd_tol=(0.03, 2e-3), d_tol=(0.999, 0.06), g_tol=(0.999, 0.06))
    torch.cuda.empty_cache()


def test_biggan_256px(dev):
    """resnet_biggan at 256x256 (resnet_biggan.py:205-221,344-361: seven blocks, attention at 64x64
    in G after B4 and at 128x128 in D after B1 -- 16,384 queries x 4,096 keys), width ch = 32 (the
    attention kernel needs key width ch / 8 >= 4), batch
    2, z_dim 140 (seven 20-dim chunks of the hierarchical z): generator forward, D and G sub-steps
    against the bf16-storage oracle resident on the device.  Nothing above 128 px had run before
    round 3."""
    _biggan_family_forward_and_gradients(
        dev, "biggan-256px",
        ['dataset.name = "imagenet_256"', "options.z_dim = 140",
         "resnet_biggan.Generator.ch = 32", "resnet_biggan.Discriminator.ch = 32"],
        dict(hierarchical_z=True, embed_y=True, ch=32), dict(project_y=True, ch=32), bsz=2,
        oracle_device=dev, image_shape=(256, 256, 3), min_g_grads=40)


def test_biggan_512px(dev):
    """resnet_biggan at 512x512 (resnet_biggan.py:205-221,344-361: eight blocks -- channel
    multipliers 16,16,8,8,4,2,1,1 / 1,1,2,4,8,8,16,16 -- attention at 64x64 in G after B4 and at
    128x128 in D after B2), width ch = 32, batch 2, z_dim 160 (eight 20-dim chunks of the
    hierarchical z): generator forward, D and G sub-steps against the bf16-storage oracle resident
    on the device.  The 512 px branch of the architecture tables had never executed before round 4."""
    _biggan_family_forward_and_gradients(
        dev, "biggan-512px",
        ['dataset.name = "imagenet_512"', "options.z_dim = 160",
         "resnet_biggan.Generator.ch = 32", "resnet_biggan.Discriminator.ch = 32"],
        dict(hierarchical_z=True, embed_y=True, ch=32), dict(project_y=True, ch=32), bsz=2,
        oracle_device=dev, image_shape=(512, 512, 3), min_g_grads=40)


def test_biggan_deep_forward_and_gradients(dev):
    """The same settings on resnet_biggan_deep_arch: bottleneck blocks,
    channel-dropping zero-insertion shortcuts in G (cg_unpool2 with the main branch as residual),
    pooled + channel-appending shortcuts in D, z concatenated with the label embedding for every
    conditional batch norm, attention at 64x64 in both networks; ch = 32.

    Batch 8: at batch 2 the 40 conditional batch norms of this generator amplify every flipped
    ReLU mask so much that the bf16-storage oracle itself only reaches cosine 0.84-0.86 against the
    exact one on the G-step gradients (profiles/r01_oracle_sensitivity.txt), which left nothing to
    assert in round 1 (floor 0.75).  At batch 8 the network is well conditioned and the test
    carries the BigGAN family's standard tolerances (measured in round 2: generator output mean
    |diff| 1.6e-3, every D-step and G-step gradient at cosine >= 0.9998, power-iteration vectors
    within 2e-7)."""
    _biggan_family_forward_and_gradients(
        dev, "biggan-deep",
        ['options.architecture = "resnet_biggan_deep_arch"',
         "resnet_biggan_deep.Generator.ch = 32", "resnet_biggan_deep.Discriminator.ch = 32"],
        dict(embed_y=True, ch=32), dict(project_y=True, ch=32), arch="resnet_biggan_deep_arch",
        min_g_grads=60, bsz=8)


@pytest.mark.parametrize("config,bsz", [("resnet_lsun-bedroom128.gin", 4), ("resnet_cifar10.gin", 16),
                                        ("biggan_imagenet128.gin", 4)])
def test_generator_fused_batch_norm_matches_unfused(dev, config, bsz):
    """The no-gradient generator forward of the discriminator sub-steps fuses batch norm around the
    convolutions (statistics from the producer's epilogue, normalisation + ReLU in the consumer's
    LDS tile: arch_ops.PendingBN).  It must produce the image of the unfused path (the one the
    gradient step and the oracle tests use) up to statistics summation order, and update the
    moving averages identically."""
    from compare_gan_amd.architectures import arch_ops as ops
    bind = ["resnet_biggan.Generator.ch = 64", "resnet_biggan.Discriminator.ch = 64"] \
        if "biggan" in config else []
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED, bindings=bind)
    z = U.host_uniform((bsz, options["z_dim"]), "z/test", -1.0, 1.0, SEED, 0).float().to(dev)
    y = None
    if gan.conditional:
        labels = torch.arange(bsz, dtype=torch.int32, device=dev) % dataset.num_classes
        y = gan._get_one_hot_labels(labels)   # pylint: disable=protected-access
    mov = [n for n in gan.store.vars if n.endswith("moving_mean") or n.endswith("moving_variance")]
    saved = {n: gan.store.vars[n].detach().clone() for n in mov}
    with ops.use_store(gan.store):
        with torch.no_grad():
            fused = gan.generator(z, y=y, is_training=True)
        mov_fused = {n: gan.store.vars[n].detach().clone() for n in mov}
        with torch.no_grad():
            for n in mov:
                gan.store.vars[n].copy_(saved[n])
        for p in gan.g_opt.params:
            p.requires_grad_(True)
        plain = gan.generator(z, y=y, is_training=True)   # autograd graph -> unfused kernels
    torch.cuda.synchronize()
    d = (fused.double() - plain.detach().double()).abs()
    # (conditional BN over 4 samples amplifies single bf16 roundings: BigGAN's budget is 3x wider)
    mean_tol = 3e-3 if "biggan" in config else 1e-3
    assert float(d.max()) <= 0.03 and float(d.mean()) <= mean_tol, (float(d.max()), float(d.mean()))
    for n in mov:
        a, b = mov_fused[n].double(), gan.store.vars[n].detach().double()
        assert float((a - b).abs().max()) <= 1e-4 * (1.0 + float(b.abs().max())), n


@pytest.mark.parametrize("config,bsz,penalty", [
    ("resnet_cifar10.gin", 8, "wgangp_penalty"),       # spectrally normalised D under a gradient penalty
    ("resnet_cifar10.gin", 8, "dragan_penalty"),
    ("resnet_cifar10.gin", 8, "l2_penalty"),
    ("sndcgan_celebahq128.gin", 2, "wgangp_penalty"),  # D rescales its input (x * 2 - 1) INSIDE D
    ("dcgan_celeba64.gin", 4, "dragan_penalty"),
])
def test_penalties_against_oracle(dev, config, bsz, penalty):
    """penalty_lib.{wgangp,dragan,l2}_penalty (penalty_lib.py:33-102) on discriminators the example
    configs do not pair them with: value and gradient w.r.t. every D kernel against the
    bf16-storage oracle.  The gradient is taken w.r.t. the [0,1] image (sndcgan.py:108 keeps its
    rescaling inside D), the spectral norm's power iteration runs again in the penalty's D call."""
    from compare_gan_amd.architectures import arch_ops as ops
    from compare_gan_amd.gans import penalty_lib
    from oracle import gan as ogan
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED,
                                            bindings=["penalty.fn = @%s" % penalty])
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    ora = U.build_oracle(config, vs)
    rng = np.random.RandomState(17)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    fake = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    with ops.use_store(gan.store):
        pen = penalty_lib.get_penalty_loss(x=images.to(dev), x_fake=fake.to(dev), y=None,
                                           is_training=True, discriminator=gan.discriminator)
    pen.backward()
    d_fn = lambda x, yy, t: ora.D(x, yy, t)
    if penalty == "wgangp_penalty":
        alpha = U.host_uniform((bsz,), "wgangp_penalty/alpha", 0.0, 1.0, SEED, 0)
        pen_o = ogan.wgangp_penalty(d_fn, images.double(), fake.double(), None, True,
                                    alpha.double().reshape(-1, 1, 1, 1))
    elif penalty == "dragan_penalty":
        noise = U.host_uniform(tuple(images.shape), "dragan_penalty/random_uniform/0", 0.0, 1.0,
                               SEED, 0)
        pen_o = ogan.dragan_penalty(d_fn, images.double(), None, True, noise.double())
    else:
        kernels = [v for n, v in zip(ora.d_var_names(), ora.d_vars()) if n.endswith("/kernel")]
        pen_o = ogan.l2_penalty(kernels)
    grads_o = torch.autograd.grad(pen_o, ora.d_vars(), allow_unused=True)
    print(penalty, config, float(pen.detach()), float(pen_o.detach()))
    assert abs(float(pen.detach()) - float(pen_o.detach())) <= 5e-3 * max(
        1e-3, abs(float(pen_o.detach())))
    checked = 0
    for (name, p), go in zip(gan.store.trainable_variables("discriminator"), grads_o):
        if not name.endswith("/kernel") or go is None:
            continue
        assert p.grad is not None, name
        if float(go.norm()) < 1e-12:
            continue
        c, r = U.cosine(p.grad, go), U.rel_l2(p.grad, go)
        # (double backward through 7 leaky-ReLU convolutions at 128x128, batch 2: the 131072 x 1 head
        # of sndcgan measures 0.990; everything else is > 0.995)
        tol = (0.9999, 0.01) if penalty == "l2_penalty" else (0.98, 0.20)
        assert c >= tol[0] and r <= tol[1], "%s grad of %s cosine %.5f rel-L2 %.4f" % (
            penalty, name, c, r)
        checked += 1
    assert checked >= 4


@pytest.mark.parametrize("config,bsz", [("resnet_cifar10.gin", 8), ("biggan_imagenet128.gin", 2)])
def test_captured_step_matches_eager(dev, config, bsz):
    """capture_train_step(): (1) capturing -- warm-up steps included -- leaves every variable,
    optimizer slot and step counter as it found them; (2) replaying the hipGraph is the eager
    step bit for bit (same kernels in the same order, nothing atomic), also for the EMA shadows
    and the spectral-norm vectors."""
    bindings = ("options.batch_size = %d" % bsz,)
    if config.startswith("biggan"):
        bindings += ("resnet_biggan.Generator.ch = 32", "resnet_biggan.Discriminator.ch = 32")
    eager, options, dataset = U.build_product(config, bsz, dev, seed=5, bindings=bindings)
    graph, _, _ = U.build_product(config, bsz, dev, seed=5, bindings=bindings)
    nsub = options["disc_iters"] + 1
    it = dataset.train_batches(bsz * nsub, seed=21)
    batches = [next(it) for _ in range(2)]
    before = graph.state_dict()
    run = graph.capture_train_step()
    after = graph.state_dict()
    for k, v in before.items():
        assert torch.equal(v, after[k]), "capture changed %s" % k
    for images, labels in batches:
        x, y = torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev)
        oe = eager.train_step(x, y)
        og = run(x, y)
        assert torch.equal(oe["g_loss"], og["g_loss"])
    torch.cuda.synchronize()
    se, sg = eager.state_dict(), graph.state_dict()
    assert int(sg["global_step"]) == 2
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    assert not bad, bad[:8]


def test_batched_generator_forward_matches_separate_calls(dev):
    """train_step() runs the generator forwards of the discriminator sub-steps as ONE batched
    no-gradient call with one set of batch-norm statistics per sub-step
    (ModularGAN._generate_for_disc, ops.statistics_groups) -- the arithmetic of the reference's
    separate calls (modular_gan.py:464-467).  The batched call may dispatch other kernel variants
    (larger grids), so the images agree to bf16 accumulation noise, and the moving averages take
    the sub-steps' updates in order."""
    from compare_gan_amd.architectures import arch_ops as ops
    config, bsz = "resnet_cifar10.gin", 16
    bind = ("standardize_batch.use_moving_averages = True",)
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED, bindings=bind)
    n = options["disc_iters"]
    images = torch.zeros((bsz,) + dataset.image_shape, dtype=torch.float32, device=dev)
    labels = torch.zeros((bsz,), dtype=torch.int32, device=dev)
    mov = [k for k in gan.store.vars if k.endswith("moving_mean") or k.endswith("moving_variance")]
    assert mov
    saved = {k: gan.store.vars[k].detach().clone() for k in mov}
    with ops.use_store(gan.store):
        fs = [gan._preprocess(images, labels, i)[0] for i in range(n + 1)]   # pylint: disable=protected-access
        gan._generate_for_disc(fs)                                          # pylint: disable=protected-access
        joint = [fs[i]["generated"].clone() for i in range(n)]
        mov_joint = {k: gan.store.vars[k].detach().clone() for k in mov}
        with torch.no_grad():
            for k in mov:
                gan.store.vars[k].copy_(saved[k])
            separate = [gan.generator(fs[i]["z"], y=None, is_training=True) for i in range(n)]
    torch.cuda.synchronize()
    assert "generated" not in fs[n]     # the generator sub-step keeps its own (gradient) forward
    for i in range(n):
        d = (joint[i].double() - separate[i].double()).abs()
        assert float(d.max()) <= 0.03 and float(d.mean()) <= 1e-3, (i, float(d.max()), float(d.mean()))
    for k in mov:
        a, b = mov_joint[k].double(), gan.store.vars[k].detach().double()
        assert float((a - b).abs().max()) <= 1e-4 * (1.0 + float(b.abs().max())), k


def test_joint_gen_for_disc_step_against_oracle(dev):
    """ModularGAN.experimental_joint_gen_for_disc = True (modular_gan.py:444-463): ONE generator
    call on the z of all discriminator sub-steps, batch-norm statistics over the joint batch, the
    images split afterwards; one full unrolled step against the bf16-storage oracle."""
    config, bsz = "resnet_cifar10.gin", 8
    bind = ("ModularGAN.experimental_joint_gen_for_disc = True",)
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED, bindings=bind)
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    ora = U.build_oracle(config, vs, joint_gen_for_disc=True)
    ora_sep = U.build_oracle(config, U.mirror_to_oracle(gan, emulate_bf16=True))
    nsub = options["disc_iters"] + 1
    rng = np.random.RandomState(500)
    images = rng.uniform(size=(nsub * bsz,) + dataset.image_shape).astype(np.float32)
    labels = np.ones((nsub * bsz,), dtype=np.int32)
    subs = [{"images": torch.from_numpy(images[i * bsz:(i + 1) * bsz]).double(),
             "z": U.host_uniform((bsz, 128), "z/%d" % i, -1.0, 1.0, SEED, 0).double()}
            for i in range(nsub)]
    # sub-step by sub-step from identical states (see test_train_steps_resnet_cifar)
    check = U.stepwise_parity(gan, ora, subs, lr_d=2e-4)
    gan.sub_step_hook = check
    try:
        out = gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    finally:
        gan.sub_step_hook = None
    d_o, g_o = check.finish(out)
    d_s, _ = ora_sep.train_step(subs)
    d_p = [float(x) for x in out["d_losses"]]
    print("joint: product", d_p, "oracle", d_o, "oracle with separate calls", d_s)
    # the option is not a no-op: joint statistics move the first loss away from the separate calls'
    assert abs(d_o[0] - d_s[0]) > 1e-6


def test_not_unrolled_step_against_oracle(dev):
    """ModularGAN.train_step_not_unrolled(): the reference's GPU graph (modular_gan.py:533-584,
    SURVEY App. A.7) -- one sub-batch per call, the G update only on every disc_iters-th call, on
    the same z and images, through the generator forward built BEFORE the D update and a fresh
    forward of the updated D.  Three calls with disc_iters = 2 against the bf16-storage oracle."""
    config, bsz = "resnet_cifar10.gin", 8
    gan, options, dataset = U.build_product(config, bsz, dev, seed=SEED,
                                            bindings=("options.disc_iters = 2",))
    assert not gan.unroll_graph(use_tpu=False) and gan.unroll_graph(use_tpu=True)
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    ora = U.build_oracle(config, vs, disc_iters=2)
    g_before = {n: v.detach().clone() for n, v in gan.store.trainable_variables("generator")}
    for call in range(3):
        rng = np.random.RandomState(900 + call)
        images = rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32)
        out = gan.train_step_not_unrolled(torch.from_numpy(images).to(dev),
                                          torch.ones((bsz,), dtype=torch.int32, device=dev))
        # z is keyed by (seed, "z/<call within the cycle>", global_step); global_step moves only
        # with the G update
        step = 0 if call < 2 else 1
        sub = {"images": torch.from_numpy(images).double(),
               "z": U.host_uniform((bsz, 128), "z/%d" % (call % 2), -1.0, 1.0, SEED, step).double()}
        d_o, g_o = ora.train_step_not_unrolled(sub)
        d_p, g_p = float(out["d_losses"][0]), float(out["g_loss"])
        print("call", call, "d", d_p, d_o, "g", g_p, g_o)
        assert abs(d_p - d_o) <= 2e-2 * max(1.0, abs(d_o))
        # (ONE discriminator update from identical states lies between the resync and this loss)
        assert abs(g_p - g_o) <= 2e-2 * max(1.0, abs(g_o))
        moved = any(not torch.equal(v, g_before[n])
                    for n, v in gan.store.trainable_variables("generator"))
        assert moved == (call >= 1), "generator update on the wrong call (%d)" % call
        assert (g_p != 0.0) == (call == 1)
        U.resync_oracle(gan, ora)
    assert int(gan.global_step.item()) == 1 and int(gan.global_step_disc.item()) == 3
