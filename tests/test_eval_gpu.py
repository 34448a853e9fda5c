"""Eval path on the MI355X: Inception features, FID, inception
score and evaluate_gan against the CPU oracle.

Tolerances: FID / IS statistics run in fp64 on both sides -> relative 1e-6 on FID given identical
activations (the reference's own pin is 89.091 +- 1e-4, metrics/fid_score_test.py:31-40);
Inception activations cross 94 bf16-stored convolutions -> cosine >= 0.999 and rel-L2 <= 0.05
against the fp64 oracle with emulated bf16 storage."""
import numpy as np
import pytest
import torch

from oracle import fid as ofid
from tests import gan_util as U

pytestmark = pytest.mark.gpu


def test_fid_matches_reference_golden(dev):
    from compare_gan_amd.metrics import fid_score
    real = np.ones((100, 2), dtype=np.float32)
    real[:50, 0] = 2
    gen = np.ones((100, 2), dtype=np.float32) * 9
    gen[50:, 0] = 2
    got = fid_score.compute_fid_from_activations(real, gen)   # the reference test's argument order
    assert abs(got - 89.091) <= 1e-4, got


@pytest.mark.parametrize("n,d", [(300, 64), (64, 96), (2000, 256), (300, 512)])
def test_fid_matches_oracle(dev, n, d):
    """Includes the rank-deficient case n < d (covariance with zero eigenvalues)."""
    from compare_gan_amd.metrics import fid_score
    rng = np.random.RandomState(n + d)
    a = (rng.randn(n, d) * rng.rand(d) * 2 + rng.randn(d)).astype(np.float32)
    b = (rng.randn(n, d) * rng.rand(d) * 3 + 0.5).astype(np.float32)
    ref = ofid.frechet_distance(a, b)
    got = fid_score.frechet_distance(a, b, device=dev)
    assert abs(got - ref) <= 1e-6 * abs(ref) + 1e-8, (got, ref)
    same = fid_score.frechet_distance(a, a, device=dev)
    assert abs(same) <= 1e-6 * float(np.trace(np.cov(a.T))), same


@pytest.mark.parametrize("n_real,n_gen,d,mbs", [(64, 64, 32, 1024), (300, 257, 64, 100),
                                                (2048, 2048, 2048, 1024)])
def test_kid_matches_oracle(dev, n_real, n_gen, d, mbs):
    """metrics/kid_score.py:44-149 -- device fp64 block estimator vs the NumPy restatement,
    ragged block sizes included.  Tolerance: 1e-9 relative (fp64 GEMM summation order)."""
    from compare_gan_amd.metrics import kid_score
    rng = np.random.RandomState(n_real + d)
    real = rng.rand(n_real, d).astype(np.float32)
    gen = (rng.rand(n_gen, d) * 1.1 + 0.05).astype(np.float32)
    want = ofid.kid(gen, real, max_batch_size=mbs)
    got = kid_score.kid(torch.from_numpy(gen).to(dev), torch.from_numpy(real).to(dev),
                        max_batch_size=mbs, device=dev)
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (got, want)
    got2, err = kid_score.kid(gen, real, max_batch_size=mbs, return_stderr=True, device=dev)
    assert got2 == got and (np.isnan(err) if n_real // mbs < 5 else err >= 0)


@pytest.mark.parametrize("n,d,cond", [(1500, 256, 1e3), (4000, 512, 1e6), (3000, 2048, 1e4)])
def test_fid_newton_schulz_path(dev, n, d, cond):
    """The GEMM-only square-root path (metrics/fid_score.py: _sqrt_newton_schulz) on full-rank
    covariances with a prescribed condition number: it must certify itself, agree with the Jacobi
    path to 1e-9 and with the oracle (tfgan's SVD rule) to 1e-6 relative; a covariance with
    eigenvalues under tfgan's 1e-10 threshold must be REFUSED (the rule is active there) and still
    come out right through the Jacobi fallback."""
    from compare_gan_amd.metrics import fid_score
    rng = np.random.RandomState(n + d)
    # features with singular values spread geometrically over `cond`^(1/2)
    basis = np.linalg.qr(rng.standard_normal((d, d)))[0]
    scales = np.geomspace(1.0, cond ** -0.5, d)
    a = (rng.standard_normal((n, d)) * scales) @ basis.T + rng.standard_normal((1, d))
    b = (rng.standard_normal((n, d)) * scales[::-1]) @ basis.T * 0.9
    a, b = a.astype(np.float32), b.astype(np.float32)
    old = fid_score._SOLVER
    try:
        fid_score._SOLVER = "auto"
        got = fid_score.frechet_distance(a, b, device=dev)
        assert fid_score.LAST_SOLVER == {"sqrt_sigma": "newton-schulz", "trace_sqrt": "newton-schulz"}
        fid_score._SOLVER = "jacobi"
        jac = fid_score.frechet_distance(a, b, device=dev)
        assert fid_score.LAST_SOLVER["sqrt_sigma"].startswith("jacobi")
    finally:
        fid_score._SOLVER = old
    assert abs(got - jac) <= 1e-9 * abs(jac), (got, jac)
    if d <= 512:
        ref = ofid.frechet_distance(a, b)
        assert abs(got - ref) <= 1e-6 * abs(ref), (got, ref)
    # eigenvalues of 1e-12 (below the 1e-10 rule): not certifiable -> Jacobi, same answer as forced
    tiny = a.copy()
    tiny[:, : d // 4] *= 1e-6
    try:
        fid_score._SOLVER = "auto"
        g2 = fid_score.frechet_distance(tiny, b, device=dev)
        used = dict(fid_score.LAST_SOLVER)
        fid_score._SOLVER = "jacobi"
        j2 = fid_score.frechet_distance(tiny, b, device=dev)
    finally:
        fid_score._SOLVER = old
    assert used["sqrt_sigma"].startswith("jacobi"), used
    assert abs(g2 - j2) <= 1e-9 * abs(j2)


@pytest.mark.parametrize("n,d", [(100, 256), (3000, 512), (600, 1024)])
def test_fid_tridiagonal_trace_path(dev, n, d):
    """The trace of the second matrix square root from eigenvalues by tridiagonalisation + bisection
    (metrics/fid_score.py: _trace_sqrt_values_only; fid_score.py:58-75 via tfgan trace_sqrt_product): on
    small-norm features with dead and nearly dead channels (what the seeded extractor of bench.py
    produces; rank-deficient for n < d) the certificate holds and the distance agrees with the
    values-only Jacobi solve to 1e-9 and with the oracle to 1e-6; at other feature scales (x 1e2, 1e4,
    1e6: the rounding level of an absolute-accuracy eigenvalue moves across tfgan's 1e-10 cut-off)
    whichever path the certificate picks gives the oracle's number; a certificate that cannot hold
    takes the Jacobi fallback."""
    from compare_gan_amd.metrics import fid_score
    rng = np.random.RandomState(n + d)
    scales = np.geomspace(1e-1, 1e-6, d)
    scales[::7] = 0.0                                         # dead channels
    a = (np.maximum(rng.standard_normal((n, d)), 0.0) * scales).astype(np.float32)
    b = (np.maximum(rng.standard_normal((n, d)) + 0.3, 0.0) * scales[::-1] * 0.5).astype(np.float32)
    old = (fid_score._SOLVER, fid_score._TRIDIAG)
    try:
        fid_score._SOLVER = "jacobi"       # (no Newton-Schulz: dead channels rule it out anyway)
        fid_score._TRIDIAG = 1
        got = fid_score.frechet_distance(a, b, device=dev)
        used, cert = dict(fid_score.LAST_SOLVER), dict(fid_score.LAST_TRIDIAG)
        fid_score._TRIDIAG = 0
        jac = fid_score.frechet_distance(a, b, device=dev)
        assert fid_score.LAST_SOLVER["trace_sqrt"] == "jacobi"
        fid_score._TRIDIAG = 1
        scaled = []
        for sc in (1e2, 1e4, 1e6):
            val = fid_score.frechet_distance(a * sc, b * sc, device=dev)
            scaled.append((sc, val, fid_score.LAST_SOLVER["trace_sqrt"], dict(fid_score.LAST_TRIDIAG)))
        # a certificate that cannot hold: the fallback is taken and gives the Jacobi number
        fid_score._TRIDIAG_ACCEPT, keep = 1e-30, fid_score._TRIDIAG_ACCEPT
        try:
            refused = fid_score.frechet_distance(a, b, device=dev)
            assert fid_score.LAST_SOLVER["trace_sqrt"] == "jacobi" and not fid_score.LAST_TRIDIAG["accepted"]
        finally:
            fid_score._TRIDIAG_ACCEPT = keep
    finally:
        fid_score._SOLVER, fid_score._TRIDIAG = old
    assert used["trace_sqrt"] == "tridiagonal+bisection" and cert["accepted"], (used, cert)
    assert abs(got - jac) <= 1e-9 * abs(jac), (got, jac)
    assert refused == jac
    ref = ofid.frechet_distance(a, b)
    assert abs(got - ref) <= 1e-6 * abs(ref), (got, ref)
    # whatever the certificate decides at other feature scales, the number is the oracle's
    for sc, val, path, c in scaled:
        ref_sc = ofid.frechet_distance(a * sc, b * sc)
        assert abs(val - ref_sc) <= 1e-6 * abs(ref_sc), (sc, path, val, ref_sc, c)
    print("tridiagonal trace path n %d d %d: certificate %s | scaled: %s" % (
        n, d, cert, [(sc, path, "%.2e" % (2 * c["bound"] / c["scale"])) for sc, _, path, c in scaled]))


@pytest.mark.parametrize("n,d", [(200, 256), (3000, 512), (5000, 2048)])
def test_fid_root_from_jacobi_rows(dev, n, d):
    """The first matrix square root rebuilt from the rows the one-sided Jacobi solve leaves behind (g_i =
    lambda_i v_i: sqrt = G^T diag(f(|lambda|) / lambda^2) G, metrics/fid_score.py; tfgan
    _symmetric_matrix_square_root behind fid_score.py:58-75) against the form with an accumulated
    eigenvector matrix: the distance agrees to 1e-9 -- on graded covariances with dead channels,
    rank-deficient (n < d) included -- and with the oracle to 1e-6."""
    from compare_gan_amd.metrics import fid_score
    rng = np.random.RandomState(n + d)
    scales = np.geomspace(3.0, 1e-5, d)
    scales[::11] = 0.0
    a = (np.maximum(rng.standard_normal((n, d)), 0.0) * scales).astype(np.float32)
    b = (np.maximum(rng.standard_normal((n, d)) + 0.2, 0.0) * scales[::-1]).astype(np.float32)
    old = (fid_score._SOLVER, fid_score._ROOT_FROM_G)
    try:
        fid_score._SOLVER = "jacobi"
        fid_score._ROOT_FROM_G = 1
        rows = fid_score.frechet_distance(a, b, device=dev)
        assert fid_score.LAST_SOLVER["sqrt_sigma"] == "jacobi (rows)"
        fid_score._ROOT_FROM_G = 0
        vecs = fid_score.frechet_distance(a, b, device=dev)
        assert fid_score.LAST_SOLVER["sqrt_sigma"] == "jacobi"
    finally:
        fid_score._SOLVER, fid_score._ROOT_FROM_G = old
    assert abs(rows - vecs) <= 1e-9 * abs(vecs), (rows, vecs)
    if d <= 512:
        ref = ofid.frechet_distance(a, b)
        assert abs(rows - ref) <= 1e-6 * abs(ref), (rows, ref)


def test_inception_score_matches_oracle(dev):
    from compare_gan_amd.metrics import inception_score
    rng = np.random.RandomState(3)
    logits = (rng.randn(512, 1008) * 2).astype(np.float32)
    got = inception_score.classifier_score_from_logits(logits, dev)
    ref = ofid.classifier_score_from_logits(logits)
    assert abs(got - ref) <= 1e-9 * ref


def test_inception_features_match_oracle(dev):
    from compare_gan_amd import inception
    from oracle import inception as oinc
    weights = inception.make_weights(seed=7)
    net = inception.InceptionV3(dev, weights=weights)
    rng = np.random.RandomState(5)
    images = (rng.rand(2, 32, 32, 3) * 255).astype(np.float32)
    pool3, logits = net.features(torch.from_numpy(images).to(dev))
    assert tuple(pool3.shape) == (2, 2048) and tuple(logits.shape) == (2, 1008)
    p_ref, l_ref = oinc.features(inception.SPEC, weights, images, emulate_bf16=True)
    assert U.cosine(pool3, p_ref) >= 0.999 and U.rel_l2(pool3, p_ref) <= 0.05, (
        U.cosine(pool3, p_ref), U.rel_l2(pool3, p_ref))
    assert U.cosine(logits, l_ref) >= 0.999 and U.rel_l2(logits, l_ref) <= 0.05
    # batched transform == per-batch features, ragged last batch
    images3 = (rng.rand(5, 32, 32, 3) * 255).astype(np.float32)
    f_all, _ = net.transform(images3, batch_size=2)
    f_one, _ = net.features(torch.from_numpy(images3[4:5]).to(dev))
    assert tuple(f_all.shape) == (5, 2048)
    # (a batch of 5 and a batch of 1 may dispatch different kernel variants: bf16 rounding noise)
    assert U.rel_l2(f_all[4:5], f_one) <= 1e-2
    import compare_gan_amd.inception as inc
    old_chunk, inc._CHUNK = inc._CHUNK, 2
    try:
        f_cut, _ = net.transform(images3, batch_size=2)      # ragged last batch of 1
    finally:
        inc._CHUNK = old_chunk
    assert tuple(f_cut.shape) == (5, 2048) and U.rel_l2(f_cut, f_all) <= 1e-2


def test_evaluate_gan_small(dev):
    """evaluate_gan end to end on a freshly initialised ResNet-CIFAR GAN with 128 test examples:
    keys and aggregation of eval_gan_lib.py:196-212, determinism of the evaluation noise."""
    from compare_gan_amd import eval_gan_lib
    from compare_gan_amd.metrics import fid_score, inception_score, kid_score
    gan, options, dataset = U.build_product("resnet_cifar10.gin", 8, dev, seed=3)
    tasks = [inception_score.InceptionScoreTask(), fid_score.FIDScoreTask(),
             kid_score.KIDScoreTask()]
    with pytest.warns(RuntimeWarning, match="SYNTHETIC") if not eval_gan_lib.eval_utils._INCEPTION \
            else __import__("contextlib").nullcontext():
        r1 = eval_gan_lib.evaluate_gan(gan, tasks, num_averaging_runs=2, num_test_examples=128)
    assert r1["inception_weights_synthetic"] == 1.0     # no trained weight file offline: tagged
    assert np.isfinite(r1["kid_score_mean"])
    for key in ("inception_score", "fid_score", "kid_score"):
        for suffix in ("_mean", "_std", "_list"):
            assert key + suffix in r1
    assert r1["inception_score_mean"] >= 1.0 - 1e-6 and np.isfinite(r1["fid_score_mean"])
    assert len(r1["fid_score_list"].split("_")) == 2
    r2 = eval_gan_lib.evaluate_gan(gan, tasks, num_averaging_runs=2, num_test_examples=128)
    assert r1["fid_score_list"] == r2["fid_score_list"]      # same latents for each evaluation
    assert eval_gan_lib.evaluate_gan(gan, [], 1, num_test_examples=64) is None


def test_nan_detection(dev):
    from compare_gan_amd import eval_utils
    def bad():
        return torch.full((4, 8, 8, 3), float("nan"), device=dev)
    with pytest.raises(eval_utils.NanFoundError):
        eval_utils.sample_fake_dataset(bad, 2)


@pytest.mark.parametrize("shape,off", [((64, 32, 32, 3), 0), ((3, 7, 5, 1), 1), ((2, 1, 1, 3), 0)])
def test_fake_image_sink(dev, shape, off):
    """cg_scale_count_nan_f32 (eval_utils.FakeImageSink): 255 * x written into the batch's slot of
    the set's buffer, bit-identical to the separate torch passes it replaces (one fp32 multiply),
    NaNs counted (np.isnan semantics: infinities are not NaNs, eval_utils.py:156-159); odd sizes
    and unaligned slices take the scalar path."""
    from compare_gan_amd import eval_utils
    from compare_gan_amd.hip import kernels as K
    g = torch.Generator().manual_seed(5)
    batches = [torch.rand(shape, generator=g).to(dev) for _ in range(3)]
    sink = eval_utils.FakeImageSink(3)
    for b in batches:
        sink.add(b)
    images, nan_found = sink.finish()
    want = torch.cat(batches, dim=0) * 255.0
    if shape[-1] == 1:
        want = want.repeat(1, 1, 1, 3)
    assert not nan_found and torch.equal(images, want)
    # the raw kernel on an unaligned slice, with NaNs and infinities
    n = int(np.prod(shape))
    x = torch.rand(n + off, generator=g).to(dev)[off:]    # off = 1: 4 bytes past a 16-byte boundary
    x[0] = float("nan")
    x[n // 2] = float("nan")
    x[n - 1] = float("inf")
    out = torch.empty(n + off, device=dev)[off:]
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    K.scale_count_nan(x, 255.0, out, count)
    K.scale_count_nan(x, 255.0, out, count)
    assert int(count.item()) == 4
    ok = ~torch.isnan(x)
    assert torch.equal(out[ok], (x * 255.0)[ok]) and bool(torch.isnan(out[~ok]).all())


def test_evaluate_gan_raises_on_nan(dev):
    """eval_gan_lib.py:165-169: a NaN anywhere in the generated set aborts the evaluation."""
    from compare_gan_amd import eval_gan_lib, eval_utils
    from compare_gan_amd.metrics import fid_score
    gan, _, _ = U.build_product("resnet_cifar10.gin", 8, dev, seed=3)
    with torch.no_grad():
        for p in gan.g_opt.params:        # (a diverged generator: every weight, the last layer included)
            p.fill_(float("nan"))
    with pytest.raises(eval_utils.NanFoundError):
        eval_gan_lib.evaluate_gan(gan, [fid_score.FIDScoreTask()], 1, num_test_examples=128)


def _small_biggan(dev, bsz, extra=()):
    bind = ["resnet_biggan.Generator.ch = 32", "resnet_biggan.Discriminator.ch = 32"] + list(extra)
    gan, options, dataset = U.build_product("biggan_imagenet128.gin", bsz, dev, seed=3, bindings=bind)
    from oracle import architectures as OA
    from oracle import arch_ops as oops
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    ora = U.build_oracle(
        "biggan_imagenet128.gin", vs,
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="conditional_batch_norm", spectral_norm=True,
                                    bn_cfg=oops.BNConfig(0.9, 1e-5, use_moving_averages=False),
                                    sn_cfg=oops.SNConfig(singular_value="auto"),
                                    hierarchical_z=True, embed_y=True, ch=32),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=True, sn_cfg=oops.SNConfig(singular_value="auto"),
                                    project_y=True, ch=32))
    return gan, options, dataset, vs, ora


def test_accumulator_batch_norm_eval(dev):
    """batch norms configured with use_moving_averages=False (BigGAN) normalise
    at eval time with accu_mean / accu_counter, filled by eval_gan_lib._update_bn_accumulators
    (arch_ops.py:122-191, eval_gan_lib.py:65-92): three fill batches, then an eval forward, against
    the oracle's accumulated_moments_for_inference on the same z / labels.  The fill and the read
    stay on the device (no .item() per batch norm call)."""
    from compare_gan_amd import eval_gan_lib
    bsz = 4
    gan, options, dataset, vs, ora = _small_biggan(dev, bsz)
    zs = [U.host_normal((bsz, options["z_dim"]), "fill_z/%d" % i, 0.0, 1.0, 3, 0).float()
          for i in range(4)]
    labels = [torch.tensor([(7 * i + j) % 1000 for j in range(bsz)], dtype=torch.int32)
              for i in range(4)]
    calls = [0]

    def fill_pass(index):
        calls[0] += 1
        return gan.generate(zs[index].to(dev), labels[0].to(dev), use_ema=False)

    consumed = eval_gan_lib._update_bn_accumulators(gan, fill_pass, bsz, 3 * bsz, first_index=0)   # pylint: disable=protected-access
    assert calls[0] == 3 and consumed == 3
    # oracle: the same three fill passes
    for n in vs.vars:
        if n.endswith("accu/update_accus"):
            vs.vars[n].fill_(1)
    with torch.no_grad():
        for i in range(3):
            ora.G(zs[i].double(), ora.one_hot(labels[0]), is_training=False)
    for n in vs.vars:
        if n.endswith("accu/update_accus"):
            vs.vars[n].fill_(0)
    checked = 0
    for n, v in gan.store.vars.items():
        if n.endswith("accu/accu_counter"):
            assert abs(float(v) - 3.0) < 1e-6, (n, float(v))
        if n.endswith("accu/accu_mean") or n.endswith("accu/accu_variance"):
            ref = vs.vars[n].double()
            got = v.detach().cpu().double()
            scale = float(ref.abs().max()) + 1e-6
            assert float((got - ref).abs().max()) <= 2e-2 * scale, (n, float((got - ref).abs().max()), scale)
            checked += 1
    assert checked >= 20   # 11 batch norms x (mean, variance)
    with torch.no_grad():
        img_o = ora.G(zs[3].double(), ora.one_hot(labels[1]), is_training=False)
    img_p = gan.generate(zs[3].to(dev), labels[1].to(dev), use_ema=False)
    d = (img_p.detach().cpu().double() - img_o).abs()
    assert float(d.max()) <= 0.05 and float(d.mean()) <= 5e-3, (float(d.max()), float(d.mean()))


def test_generate_with_ema_weights(dev):
    """modular_gan.py:266-285,498-508: after a training step with the EMA active (ema_start_step = 0,
    decay 0.5) the shadow variables follow s <- s - (1 - d)(s - theta) and generate(use_ema=True)
    evaluates the generator on them (and leaves the live weights untouched)."""
    bsz = 2
    # (G.spectral_norm off: the power iteration advances on every generator call, which would make
    # two evaluations of the same weights differ in the last bits)
    bind = ["resnet_biggan.Generator.ch = 32", "resnet_biggan.Discriminator.ch = 32",
            "ModularGAN.ema_decay = 0.5", "ModularGAN.ema_start_step = 0", "G.spectral_norm = False"]
    gan, options, dataset = U.build_product("biggan_imagenet128.gin", bsz, dev, seed=3, bindings=bind)
    nsub = options["disc_iters"] + 1
    rng = np.random.RandomState(5)
    images = rng.uniform(size=(nsub * bsz,) + tuple(dataset.image_shape)).astype(np.float32)
    labels = rng.randint(0, 1000, size=nsub * bsz).astype(np.int32)
    before = [p.detach().clone() for p in gan.g_opt.params]
    gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    torch.cuda.synchronize()
    # shadow = 0.5 * old + 0.5 * new for every generator variable
    for p0, p1, e in zip(before, gan.g_opt.params, gan.g_opt.ema):
        want = 0.5 * p0.double() + 0.5 * p1.detach().double()
        assert float((e.double() - want).abs().max()) <= 1e-6 * (1.0 + float(want.abs().max()))
    live = [p.detach().clone() for p in gan.g_opt.params]
    z = U.host_normal((bsz, options["z_dim"]), "ema_z", 0.0, 1.0, 3, 0).float().to(dev)
    lab = torch.tensor([1, 2], dtype=torch.int32, device=dev)
    from compare_gan_amd import eval_gan_lib
    eval_gan_lib._update_bn_accumulators(   # pylint: disable=protected-access
        gan, lambda index: gan.generate(z, lab, use_ema=False), bsz, 2 * bsz)
    img_ema = gan.generate(z, lab, use_ema=True)
    img_live = gan.generate(z, lab, use_ema=False)
    for p, l in zip(gan.g_opt.params, live):
        assert torch.equal(p.detach(), l)          # the swap is undone
    assert float((img_ema - img_live).abs().max()) > 0.0
    # reference: evaluate with the shadows copied in by hand
    with torch.no_grad():
        for p, e in zip(gan.g_opt.params, gan.g_opt.ema):
            p.copy_(e)
    img_ref = gan.generate(z, lab, use_ema=False)
    assert torch.equal(img_ema, img_ref)


@pytest.mark.parametrize("config,bind", [
    ("resnet_cifar10.gin", []),
    ("biggan_imagenet128.gin", ["resnet_biggan.Generator.ch = 32", "resnet_biggan.Discriminator.ch = 32",
                                "ModularGAN.ema_decay = 0.5", "ModularGAN.ema_start_step = 0",
                                "G.spectral_norm = False"])])
def test_captured_sampler_equals_generate(dev, config, bind):
    """ModularGAN.make_sampler (the evaluation's sampling loop, eval_gan_lib.py:95-140 /
    modular_gan.py:266-285): one hipGraph replay per batch returns bit for bit what generate() returns --
    unconditional and class-conditional with EMA shadows -- and follows the variables when a training
    step updates them in place after the capture."""
    bsz = 4
    gan, options, dataset = U.build_product(config, bsz, dev, seed=11, bindings=bind)
    sample = gan.make_sampler(bsz)
    assert hasattr(sample, "graph")
    nsub = options["disc_iters"] + 1
    rng = np.random.RandomState(7)
    for round_ in range(2):
        for k in range(3):
            z = U.host_uniform((bsz, options["z_dim"]), "z/%d/%d" % (round_, k), -1.0, 1.0, 11, 0).float().to(dev)
            lab = None
            if gan.conditional:
                lab = torch.from_numpy(rng.randint(0, dataset.num_classes, size=bsz).astype(np.int32)).to(dev)
            got = sample(z, lab).clone()
            want = gan.generate(z, lab)
            assert torch.equal(got, want), (config, round_, k, float((got - want).abs().max()))
        if round_ == 0:
            images = rng.uniform(size=(nsub * bsz,) + tuple(dataset.image_shape)).astype(np.float32)
            labels = rng.randint(0, max(1, dataset.num_classes), size=nsub * bsz).astype(np.int32)
            before = want.clone()
            gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
            torch.cuda.synchronize()
            assert not torch.equal(gan.generate(z, lab), before)   # the weights did move
