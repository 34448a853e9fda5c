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
