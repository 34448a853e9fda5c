"""oracle/ (torch restatement) against oracle/direct.py (independent direct-loop NumPy
restatement): TF 'SAME' convolution, conv2d_transpose and TF-Adam have no reference-held vector
(SURVEY.md section 8c), so single-implementation risk is removed by making two restatements that
share no code agree -- on DCGAN's 5x5 / stride-2 geometry (dcgan.py:109-122, asymmetric padding: 1
before, 2 after) among others."""
import numpy as np
import pytest
import torch

from oracle import arch_ops as oops
from oracle import direct
from oracle import gan as ogan

CASES = [
    # name, N, H, W, Ci, Co, k, stride
    ("dcgan_d_5x5_s2", 2, 8, 8, 3, 5, 5, 2),        # dcgan.py:109-122 (64x64 in the config)
    ("dcgan_odd_7x7_map", 1, 7, 7, 2, 3, 5, 2),     # odd size: pad_total odd -> extra pixel after
    ("sndcgan_4x4_s2", 2, 8, 8, 3, 4, 4, 2),
    ("resnet_3x3_s1", 2, 5, 6, 3, 4, 3, 1),
    ("one_by_one", 1, 4, 4, 3, 2, 1, 1),
    ("k3_s2_odd", 1, 5, 5, 2, 2, 3, 2),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_same_conv_restatements_agree(case):
    _, n, h, w, ci, co, k, stride = case
    rng = np.random.RandomState(sum(case[1:]))
    x = rng.standard_normal((n, h, w, ci))
    wt = rng.standard_normal((k, k, ci, co))
    ref = direct.conv2d_same(x, wt, stride)
    got = oops.conv2d_same(torch.from_numpy(x), torch.from_numpy(wt), stride).numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    # the per-tap GEMM form the device-resident oracle uses
    gemm = oops.conv2d_same_gemm(torch.from_numpy(x), torch.from_numpy(wt), stride).numpy()
    np.testing.assert_allclose(gemm, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_transpose_restatements_agree(case):
    """conv2d_transpose(SAME): x lives in the output space of the forward conv of an [h,w] map."""
    _, n, h, w, ci, co, k, stride = case
    rng = np.random.RandomState(7 + sum(case[1:]))
    ho, wo = -(-h // stride), -(-w // stride)
    x = rng.standard_normal((n, ho, wo, ci))
    wt = rng.standard_normal((k, k, co, ci))        # [kh,kw,Cout,Cin] (arch_ops.py:583-585)
    ref = direct.conv2d_transpose_same(x, wt, (h, w), stride)
    got = oops.conv2d_transpose_same(torch.from_numpy(x), torch.from_numpy(wt), (h, w), stride).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    # the per-tap GEMM form the oracle takes when it runs on a device (parity tests at the BASELINE
    # batch sizes of dcgan_celeba64.gin / sndcgan_celebahq128.gin), and its gradient
    xt = torch.from_numpy(x).requires_grad_(True)
    gemm = oops.conv2d_transpose_same_gemm(xt, torch.from_numpy(wt), (h, w), stride)
    np.testing.assert_allclose(gemm.detach().numpy(), ref, rtol=1e-12, atol=1e-12)
    dy = rng.standard_normal(ref.shape)
    (gx,) = torch.autograd.grad((gemm * torch.from_numpy(dy)).sum(), xt)
    np.testing.assert_allclose(gx.numpy(), direct.conv2d_same(dy, wt, stride), rtol=1e-12, atol=1e-11)


def test_conv_transpose_is_the_adjoint_of_conv():
    """<conv(x), y> == <x, conv_transpose(y)> for the direct-loop pair itself (5x5 / s2)."""
    rng = np.random.RandomState(3)
    x = rng.standard_normal((2, 9, 9, 3))
    wt = rng.standard_normal((5, 5, 3, 4))
    y = rng.standard_normal((2, 5, 5, 4))
    lhs = float((direct.conv2d_same(x, wt, 2) * y).sum())
    # the transpose's filter layout [kh,kw,Cout,Cin] is the forward filter's [kh,kw,Ci,Co] as it is
    rhs = float((x * direct.conv2d_transpose_same(y, wt, (9, 9), 2)).sum())
    assert abs(lhs - rhs) <= 1e-10 * max(1.0, abs(lhs))


def test_tf_adam_restatements_agree():
    rng = np.random.RandomState(11)
    theta0 = rng.standard_normal((3, 4))
    grads = [rng.standard_normal((3, 4)) * 10.0 ** rng.randint(-6, 2) for _ in range(7)]
    for (lr, b1, b2, eps) in ((2e-4, 0.5, 0.999, 1e-8), (1e-4, 0.0, 0.9, 1e-8)):
        ref, m_ref, v_ref = direct.tf_adam(theta0, grads, lr, b1, b2, eps, len(grads))
        p = torch.from_numpy(theta0.copy()).requires_grad_(True)
        opt = ogan.TFAdam([p], lr, b1, b2, eps)
        for g in grads:
            opt.step([torch.from_numpy(g)])
        np.testing.assert_allclose(p.detach().numpy(), ref, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(opt.m[0].numpy(), m_ref, rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(opt.v[0].numpy(), v_ref, rtol=1e-12, atol=1e-300)


# ---- round 3: pooling / un-pooling, spectral norm, batch norm, attention, losses, EMA ------------
def test_unpool_and_poolings_agree():
    rng = np.random.RandomState(11)
    for shape in ((2, 3, 5, 4), (1, 4, 4, 3), (3, 7, 2, 2)):
        x = rng.randn(*shape)
        xt = torch.from_numpy(x)
        np.testing.assert_array_equal(direct.unpool(x), oops.unpool(xt).numpy())
        np.testing.assert_allclose(direct.avg_pool2_same(x), oops.avg_pool2(xt).numpy(), rtol=0, atol=1e-15)
        if shape[1] >= 2 and shape[2] >= 2:
            np.testing.assert_array_equal(direct.max_pool2_valid(x), oops.max_pool2(xt).numpy())


@pytest.mark.parametrize("shape,mode", [((3, 3, 8, 16), "left"), ((1, 1, 20, 6), "right"),
                                         ((12, 40), "auto"), ((64, 5), "auto")])
def test_spectral_norm_agrees(shape, mode):
    rng = np.random.RandomState(sum(shape))
    w = rng.randn(*shape)
    vs = oops.VarStore(dtype=torch.float64)
    wt = torch.from_numpy(w.copy()).requires_grad_(True)
    # the oracle creates u from its own stream; read it back and hand the SAME vector to the loops
    wbar_o = oops.spectral_norm(vs, wt, "layer/kernel", singular_value=mode, update=False)
    u0 = vs.vars["layer/kernel/u_var"].detach().numpy().copy()
    wbar_d, u_new, sigma = direct.spectral_norm(w, u0, mode)
    np.testing.assert_allclose(wbar_o.detach().numpy(), wbar_d, rtol=1e-12, atol=1e-14)
    # with update=True the persisted vector is the new one
    oops.spectral_norm(vs, wt, "layer/kernel", singular_value=mode, update=True)
    np.testing.assert_allclose(vs.vars["layer/kernel/u_var"].numpy(), u_new, rtol=1e-12, atol=1e-14)
    # sigma is the top singular value's estimate from below, never above it
    assert 0.0 < sigma <= np.linalg.svd(w.reshape(-1, shape[-1]), compute_uv=False)[0] * (1 + 1e-12)


def test_batch_norm_training_mode_agrees():
    rng = np.random.RandomState(5)
    x = rng.randn(4, 3, 5, 6) * 2.0 + 0.7
    vs = oops.VarStore(dtype=torch.float64)
    cfg = oops.BNConfig()
    out_o = oops.standardize_batch(vs, torch.from_numpy(x), True, "bn/", cfg)
    out_d, mean, var = direct.batch_norm_train(x, eps=cfg.epsilon)
    np.testing.assert_allclose(out_o.numpy(), out_d, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(mean, x.reshape(-1, 6).mean(0), rtol=1e-13)
    np.testing.assert_allclose(var, x.reshape(-1, 6).var(0), rtol=1e-10)      # biased
    gamma, beta = rng.rand(4, 6) + 0.5, rng.randn(4, 6)                       # conditional: per sample
    got, _, _ = direct.batch_norm_train(x, gamma, beta, eps=cfg.epsilon)
    want = out_o.numpy() * gamma[:, None, None, :] + beta[:, None, None, :]
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12)


def test_attention_agrees():
    rng = np.random.RandomState(9)
    theta, phi, g = rng.randn(2, 7, 3), rng.randn(2, 5, 3), rng.randn(2, 5, 4)
    want = torch.softmax(torch.from_numpy(theta) @ torch.from_numpy(phi).transpose(1, 2), dim=-1) @ torch.from_numpy(g)
    np.testing.assert_allclose(direct.attention(theta, phi, g), want.numpy(), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("kind", ["non_saturating", "wasserstein", "least_squares", "hinge"])
def test_gan_losses_agree(kind):
    from oracle import gan as ogan
    rng = np.random.RandomState(2)
    r, f = rng.randn(9, 1) * 3.0, rng.randn(9, 1) * 3.0       # |logit| up to ~9: the stable form matters
    rt, ft = torch.from_numpy(r), torch.from_numpy(f)
    want = getattr(ogan, kind)(d_real=torch.sigmoid(rt), d_fake=torch.sigmoid(ft), d_real_logits=rt,
                               d_fake_logits=ft)
    got = direct.gan_losses(kind, r, f)
    for a, b in zip(got, want):
        assert abs(float(a) - float(b)) <= 1e-12 * max(1.0, abs(float(b))), (kind, a, b)


def test_ema_agrees():
    from oracle import gan as ogan
    rng = np.random.RandomState(4)
    s, p = rng.randn(5, 3), rng.randn(5, 3)
    st = [torch.from_numpy(s.copy())]
    ogan.ema_update(st, [torch.from_numpy(p)], 0.999)
    np.testing.assert_allclose(st[0].numpy(), direct.ema(s, p, 0.999), rtol=1e-15)


class _QuadraticD(object):
    """D(x) = (a . x)^2 + b . x per sample, through torch autograd (the shape of a discriminator
    call: (probabilities, logits, features))."""

    def __init__(self, a, b):
        self.a = torch.from_numpy(a.copy()).requires_grad_(True)
        self.b = torch.from_numpy(b.copy()).requires_grad_(True)

    def __call__(self, x, y, is_training):
        xf = x.reshape(x.shape[0], -1)
        logits = (xf @ self.a) ** 2 + xf @ self.b
        return torch.sigmoid(logits), logits.reshape(-1, 1), None


def test_gradient_penalties_against_the_closed_form():
    """The oracle's WGAN-GP and DRAGAN penalties -- value AND the gradients that reach the
    discriminator's parameters through the double backward -- against the chain rule written out by
    hand for a discriminator whose input gradient has a closed form."""
    rng = np.random.RandomState(8)
    x, xf = rng.rand(5, 2, 3, 2), rng.rand(5, 2, 3, 2)
    a, b = rng.randn(12) * 0.6, rng.randn(12) * 0.4
    alpha = rng.rand(5, 1, 1, 1)
    d = _QuadraticD(a, b)
    pen = ogan.wgangp_penalty(d, torch.from_numpy(x), torch.from_numpy(xf), None, True,
                              torch.from_numpy(alpha))
    ga, gb = torch.autograd.grad(pen, [d.a, d.b])
    want, da, db = direct.gradient_penalty_quadratic(direct.interpolate(x, xf, alpha), a, b)
    assert abs(float(pen.detach()) - want) <= 1e-12 * max(1.0, want)
    np.testing.assert_allclose(ga.numpy(), da, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gb.numpy(), db, rtol=1e-10, atol=1e-12)
    noise = rng.rand(*x.shape)
    d2 = _QuadraticD(a, b)
    pen2 = ogan.dragan_penalty(d2, torch.from_numpy(x), None, True, torch.from_numpy(noise))
    ga2, gb2 = torch.autograd.grad(pen2, [d2.a, d2.b])
    want2, da2, db2 = direct.gradient_penalty_quadratic(direct.dragan_perturb(x, noise), a, b)
    assert abs(float(pen2.detach()) - want2) <= 1e-12 * max(1.0, want2)
    np.testing.assert_allclose(ga2.numpy(), da2, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gb2.numpy(), db2, rtol=1e-10, atol=1e-12)
    # and the closed form itself against central differences in a
    eps = 1e-6
    for k in (0, 5, 11):
        ap, am = a.copy(), a.copy()
        ap[k] += eps
        am[k] -= eps
        xh = direct.interpolate(x, xf, alpha)
        fd = (direct.gradient_penalty_quadratic(xh, ap, b)[0] - direct.gradient_penalty_quadratic(xh, am, b)[0]) / (2 * eps)
        assert abs(fd - da[k]) <= 1e-6 * max(1.0, abs(da[k]))


def test_l2_penalty_agrees():
    rng = np.random.RandomState(1)
    ws = [rng.randn(3, 3, 4, 5), rng.randn(7, 2), rng.randn(1, 1, 6, 6)]
    want = float(np.mean([0.5 * (w ** 2).sum() for w in ws]))     # penalty_lib.py:85-102: mean of tf.nn.l2_loss
    got = float(ogan.l2_penalty([torch.from_numpy(w) for w in ws]))
    assert abs(got - want) <= 1e-13 * want


def test_inception_score_agrees():
    from oracle import fid as ofid
    rng = np.random.RandomState(6)
    logits = rng.randn(40, 17) * 4.0
    assert abs(direct.inception_score(logits) - ofid.classifier_score_from_logits(logits)) <= 1e-12 * 17
    # closed forms: identical rows -> 1; one-hot rows spread evenly over K classes -> K
    assert abs(direct.inception_score(np.tile(logits[:1], (8, 1))) - 1.0) <= 1e-12
    onehot = np.full((12, 4), -1e3)
    onehot[np.arange(12), np.arange(12) % 4] = 1e3
    assert abs(ofid.classifier_score_from_logits(onehot) - 4.0) <= 1e-9
    assert abs(direct.inception_score(onehot) - 4.0) <= 1e-9
