"""Op-level parity: every C-ABI kernel family (through compare_gan_amd.hip.kernels -> libcgamd.so)
against the CPU oracle on the same seeded inputs.

Inputs are drawn in bf16 so that both sides see identical values; the oracle computes in fp64.
Tolerances (stated per check): bf16 outputs within 2 bf16 ulps (2^-7 rel) + 2^-8 * rms(ref)
absolute; fp32 outputs (fp32 accumulation of exact bf16 products) within 1e-4 rel + 1e-4 * rms.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import arch_ops as oops
from oracle import fid as ofid
from oracle import gan as ogan
from oracle import rng as orng
from tests.util import assert_close_bf16, assert_close_f32, rand_bf16

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def K():
    from compare_gan_amd.hip import kernels
    kernels.lib()
    return kernels


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _ref_conv(x, w, stride, up, gate_slope=None):
    """oracle forward of the gather conv: [lrelu] -> [unpool] -> conv2d SAME."""
    if gate_slope is not None:
        x = torch.where(x > 0, x, gate_slope * x)
    if up == 2:
        x = oops.unpool(x)
    return oops.conv2d_same(x, w, stride)


CONV_CASES = [
    # name, N, H, W, Ci, Co, k, stride, up
    ("gemm_1x1", 2, 8, 8, 64, 128, 1, 1, 1),
    ("d_3x3_128", 4, 16, 16, 128, 128, 3, 1, 1),
    ("rgb_in_3x3", 4, 16, 16, 3, 128, 3, 1, 1),
    ("rgb_out_3x3", 2, 16, 16, 64, 3, 3, 1, 1),
    # more than 128 channels beside an RGB tensor (resnet_cifar.py:108-111: the generator's last
    # convolution 256 -> 3 and its data gradient, a 3 -> 256 convolution): two channel tiles of the stem kernel
    ("rgb_in_3x3_co256", 3, 32, 32, 3, 256, 3, 1, 1),
    ("rgb_in_3x3_co200", 2, 16, 16, 3, 200, 3, 1, 1),
    ("rgb_out_3x3_ci256", 3, 32, 32, 256, 3, 3, 1, 1),
    ("up_3x3", 2, 4, 4, 256, 256, 3, 1, 2),
    ("s2_4x4", 2, 16, 16, 64, 128, 4, 2, 1),
    ("s2_5x5_asym", 2, 16, 16, 64, 128, 5, 2, 1),
    ("odd_shapes", 3, 5, 7, 24, 40, 3, 1, 1),
    ("wide_co", 1, 8, 8, 96, 192, 3, 1, 1),
    ("up_1x1", 2, 8, 8, 64, 32, 1, 1, 2),
    ("s2_5x5_rgb", 2, 32, 32, 3, 64, 5, 2, 1),
    ("odd_stride2", 2, 9, 9, 16, 24, 3, 2, 1),
    # single-channel images (mnist / fashion-mnist: infogan, dcgan on 28x28x1)
    ("grey_4x4_s2", 4, 28, 28, 1, 64, 4, 2, 1),
    ("grey_3x3", 2, 16, 16, 1, 32, 3, 1, 1),
    ("grey_5x5_s2", 2, 28, 28, 1, 64, 5, 2, 1),
    # fast (LDS-DMA) path: multi-tile, ragged M / Co tiles, phases, strides
    ("fast_ragged", 3, 10, 10, 128, 192, 3, 1, 1),
    ("fast_big", 8, 32, 32, 128, 128, 3, 1, 1),
    ("fast_up_big", 4, 16, 16, 256, 256, 3, 1, 2),
    ("fast_up_1x1", 2, 8, 8, 128, 64, 1, 1, 2),
    ("fast_s2_4x4", 2, 32, 32, 128, 256, 4, 2, 1),
    ("fast_s2_5x5", 2, 16, 16, 128, 128, 5, 2, 1),
    ("fast_s2_3x3", 4, 16, 16, 256, 256, 3, 2, 1),
    ("fast_1x1_wide", 2, 16, 16, 192, 384, 1, 1, 1),
    # channel counts that are multiples of 32 but not 64 (BigGAN ch = 96): half-empty last K slice
    ("c32", 2, 8, 8, 32, 64, 3, 1, 1),
    ("c96_big", 4, 16, 16, 96, 96, 3, 1, 1),
    ("c160_up", 2, 8, 8, 160, 96, 3, 1, 2),
    ("c96_s2", 2, 16, 16, 96, 192, 4, 2, 1),
    # all-taps (halo) weight gradient: 4-wide tiles with 4 images per slice (batch not a multiple
    # of 4), 8x8 tiles, 16x4 tiles over several tile rows / columns, many splits
    ("halo_4x4", 6, 4, 4, 128, 64, 3, 1, 1),
    ("halo_4x4_c96", 5, 4, 4, 96, 160, 3, 1, 1),
    ("halo_8x8", 3, 8, 8, 64, 128, 3, 1, 1),
    ("halo_64x64", 2, 64, 64, 64, 64, 3, 1, 1),
    ("halo_4x16", 3, 4, 16, 64, 72, 3, 1, 1),
    ("halo_16x4", 3, 16, 4, 64, 64, 3, 1, 1),
    # halo-staged forward / data-gradient kernel (cg_conv_halo.hip; selected for these small grids
    # by the "hconv_all" variant below): 16x16 and 8x32 tiles, ragged channel tile, 64-channel
    # tile, non-square maps, zero-insertion phases, 1x1 filters
    ("hc_16x16", 3, 16, 16, 128, 192, 3, 1, 1),
    ("hc_32x32_c64", 2, 32, 32, 64, 64, 3, 1, 1),
    ("hc_64x32", 1, 64, 32, 64, 128, 3, 1, 1),
    ("hc_up16", 2, 16, 16, 128, 128, 3, 1, 2),
    ("hc_up32_c64", 1, 32, 32, 64, 64, 3, 1, 2),
    ("hc_1x1", 2, 32, 32, 128, 136, 1, 1, 1),
    ("hc_c96", 2, 32, 32, 96, 160, 3, 1, 1),
    # RGB outputs through the halo-staged kernel (scalar epilogue of its 64-channel tile): 8x32 tiles
    # with two channel blocks, 16x16 tiles
    ("hc_rgb_out_32", 26, 32, 32, 128, 3, 3, 1, 1),
    ("hc_rgb_out_16", 100, 16, 16, 64, 3, 3, 1, 1),
    # several 16x32 tiles per image, three channel blocks, three out-channel tiles with a ragged
    # last one
    ("pc_64x64_c192", 1, 64, 64, 192, 128, 3, 1, 1),
    ("pc_32x64_co320", 3, 32, 64, 64, 320, 3, 1, 1),
    # 64 -> 64 channels with register-resident weights (hconv_rw_kernel; "hconv_all" variant)
    ("hc_rw", 3, 32, 64, 64, 64, 3, 1, 1),
    ("hc_c160_up", 1, 16, 16, 160, 96, 3, 1, 2),
    # all-phase zero-insertion kernel (hup_kernel, "hconv_all" variant; "no_hup" keeps the per-phase
    # form covered): non-square map, 3 channel blocks with a ragged last one, partial channel tile
    ("hup_rect", 2, 8, 32, 96, 72, 3, 1, 2),
    ("hup_3blocks", 1, 16, 16, 160, 128, 3, 1, 2),
    # window-staged RGB-input kernels (cg_conv_halo.hip: wstem_*): 8x32 and 16x16 tiles, 64 / 96 /
    # 128 output channels
    ("wstem_32", 3, 32, 32, 3, 64, 3, 1, 1),
    ("wstem_64x32_c96", 1, 64, 32, 3, 96, 3, 1, 1),
    ("wstem_16_c128", 5, 16, 16, 3, 128, 3, 1, 1),
    # small-map kernels (cg_conv_small.hip: sconv / swgrad): four 4x4 images per tile with a ragged
    # last tile, 8x8 tiles, several 8x8 tiles per image (non-square), 3 channel blocks, 16-pixel
    # k-steps on 16-wide rows
    ("small_4x4", 6, 4, 4, 64, 64, 3, 1, 1),
    ("small_4x4_c192", 5, 4, 4, 192, 128, 3, 1, 1),
    ("small_8x8", 3, 8, 8, 128, 192, 3, 1, 1),
    ("small_16x8", 2, 16, 8, 128, 64, 3, 1, 1),
    ("small_16x16", 2, 16, 16, 64, 128, 3, 1, 1),
    # thin 1x1 convolutions (K <= 8: thin_conv_kernel): the final linear layer of a discriminator
    # (its data gradient is 1 -> 512), a 4-channel 1x1 with a vector store, and a ragged channel tail
    ("thin_linear", 128, 1, 1, 512, 1, 1, 1, 1),
    ("thin_1x1_c4", 3, 5, 5, 4, 40, 1, 1, 1),
    ("thin_co12", 2, 3, 3, 8, 12, 1, 1, 1),
    ("thin_rgb_1x1", 3, 16, 16, 3, 96, 1, 1, 1),
    # small linear layers with K % 8 != 0 (small_linear_kernel): conditional-BN projection, label
    # embedding, ragged rows / channels
    ("lin_148", 64, 1, 1, 148, 192, 1, 1, 1),
    ("lin_148_wide", 200, 1, 1, 148, 1536, 1, 1, 1),   # conditional-BN projection at a ragged batch
    ("lin_1000", 70, 1, 1, 1000, 128, 1, 1, 1),
    ("lin_ragged", 5, 1, 1, 21, 20, 1, 1, 1),
    # few outputs, long K (rowdot_linear_kernel): the final linear of the DCGAN / SNDCGAN discriminators
    ("rowdot_k8192", 5, 1, 1, 8192, 1, 1, 1, 1),
    ("rowdot_co3", 3, 1, 1, 4096, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_gconv_forward_adjoint_wgrad(K, dev, case):
    name, N, H, W, Ci, Co, k, stride, up = case
    g = _gen(sum(ord(c) for c in name))
    x64, xb = rand_bf16((N, H, W, Ci), g)
    w64, wb = rand_bf16((k, k, Ci, Co), g, 1.0 / math.sqrt(k * k * Ci))
    bias = torch.randn(Co, generator=g, dtype=torch.float32)
    geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, stride, up)
    wd = wb.to(torch.float32).to(dev)
    bt_f, bt_b = K.weight_prep(wd, want_fwd=True, want_bwd=True)
    xd = xb.to(dev)

    # forward + bias, bf16 and fp32 outputs
    xr = x64.clone().requires_grad_(True)
    wr = w64.clone().requires_grad_(True)
    ref = _ref_conv(xr, wr, stride, up) + bias.to(torch.float64)
    assert (geom.Ho, geom.Wo) == (ref.shape[1], ref.shape[2])
    y = K.gconv(geom, xd, bt_f, bias=bias.to(dev))
    assert_close_bf16(y, ref.detach(), name + " fwd bf16")
    y32 = K.gconv(geom, xd, bt_f, bias=bias.to(dev), out_f32=True)
    assert_close_f32(y32, ref.detach(), name + " fwd f32")

    # gradients of sum(ref * dy) from torch autograd on the oracle
    dy64, dyb = rand_bf16(tuple(ref.shape), g)
    (ref * dy64).sum().backward()
    dyd = dyb.to(dev)
    dx = K.gconv(K.geom_adjoint(geom), dyd, bt_b, out_f32=True)
    assert_close_f32(dx, xr.grad, name + " dgrad f32", rtol=2e-4, abs_rms=2e-4)
    dw, db = K.gwgrad(geom, xd, dyd, want_dbias=True)
    assert_close_f32(dw, wr.grad, name + " wgrad", rtol=2e-4, abs_rms=2e-4)
    assert_close_f32(db, dy64.sum(dim=(0, 1, 2)), name + " dbias", rtol=2e-4, abs_rms=2e-4)


def _close_on_device(got, ref, what, rel, abs_rms):
    """assert_close_* for tensors too large to ship to the host: same bound, evaluated on the GPU."""
    g = got.detach().to(torch.float64).reshape(ref.shape)
    rms = float(ref.pow(2).mean().sqrt()) + 1e-30
    excess = (g - ref).abs() - (rel * ref.abs() + abs_rms * rms)
    bad = int((excess > 0).sum())
    assert bad == 0, "%s: %d/%d elements out of tolerance (worst excess %.3g, rms %.3g)" % (
        what, bad, ref.numel(), float(excess.max()), rms)


def _ref_conv3x3_dev64(x, w):
    """3x3 SAME convolution as nine shifted fp64 GEMMs on the device (plain torch, no kernel of
    libcgamd.so): x [N,H,W,Ci], w [3,3,Ci,Co] -> [N,H,W,Co]."""
    n, h, w_, ci = x.shape
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    out = torch.zeros((n * h * w_, w.shape[-1]), dtype=torch.float64, device=x.device)
    for r in range(3):
        for s_ in range(3):
            out += xp[:, r:r + h, s_:s_ + w_, :].reshape(-1, ci) @ w[r, s_]
    return out.reshape(n, h, w_, -1)


FULL_SIZE_CASES = [
    # the geometries bench.py and the microbenchmarks actually run
    ("cifar_D_32x32", 128, 32, 32, 128, 128),
    ("cifar_G_32x32", 64, 32, 32, 256, 256),
    ("cifar_D_8x8", 128, 8, 8, 128, 128),
    ("resnet128_D_64x64", 128, 64, 64, 128, 128),
    ("resnet128_D_128x128", 128, 128, 128, 64, 64),
    ("resnet128_D_4x4", 128, 4, 4, 512, 512),
    # four whole 8x8 images per halo tile (hconv_kernel<64, *, 3, 0>): blocks B4 / B5 of the ResNet5
    # discriminator at the benchmark batch, and a batch that leaves the last image group ragged
    ("resnet128_D_8x8_c512", 128, 8, 8, 512, 512),
    ("mi_ragged_n98", 98, 8, 8, 256, 512),
    ("cifar_G_rgb_out", 64, 32, 32, 256, 3),
    ("resnet128_G_rgb_out", 32, 128, 128, 64, 3),
    # BigGAN at ch = 96 (resnet_biggan.py:99-151, 344-425): channel counts of 64 k + 32 -- the second
    # channel half of the last 64-channel group is empty and its waves skip their matrix work
    # (hconv_kernel's jn / half_k, hwgrad_kernel's idle roles)
    ("biggan_128x128_c96", 24, 128, 128, 96, 96),
    ("biggan_64x64_c96_c192", 24, 64, 64, 96, 192),
    ("biggan_64x64_c192", 24, 64, 64, 192, 192),
    ("biggan_32x32_c192_c96", 40, 32, 32, 192, 96),
]


@pytest.mark.parametrize("case", FULL_SIZE_CASES, ids=[c[0] for c in FULL_SIZE_CASES])
def test_gconv_full_size_shapes(K, dev, case):
    """Forward, data gradient, weight and bias gradient of the 3x3 convolutions at the batch sizes
    of the benchmark (the dispatcher picks other kernel variants -- tile sizes, ring depths, halo
    splits -- there than at the toy sizes above).  The CPU oracle would need minutes per case at
    these sizes; the reference here is nine shifted fp64 GEMMs in plain torch on the device."""
    name, N, H, W, Ci, Co = case
    g = torch.Generator(device=dev).manual_seed(sum(ord(c) for c in name))
    xb = torch.randn((N, H, W, Ci), generator=g, device=dev, dtype=torch.float32).to(BF16)
    wb = (torch.randn((3, 3, Ci, Co), generator=g, device=dev, dtype=torch.float32) /
          math.sqrt(9 * Ci)).to(BF16)
    dyb = torch.randn((N, H, W, Co), generator=g, device=dev, dtype=torch.float32).to(BF16)
    bias = torch.randn(Co, generator=g, device=dev, dtype=torch.float32)
    geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
    bt_f, bt_b = K.weight_prep(wb.to(torch.float32), want_fwd=True, want_bwd=True)
    x64, w64, dy64 = xb.double(), wb.double(), dyb.double()

    ref = _ref_conv3x3_dev64(x64, w64) + bias.double()
    y = K.gconv(geom, xb, bt_f, bias=bias)
    _close_on_device(y, ref, name + " fwd bf16", 2.0 * 2.0 ** -8, 2.0 ** -8)
    del ref, y
    if Co >= 8:
        # the form a residual block's second convolution / a data gradient takes: ReLU on the input,
        # gate tensor and residual in the epilogue (resnet_ops.py:165-181)
        go = torch.randn((N, H, W, Co), generator=g, device=dev, dtype=torch.float32).to(BF16)
        res = torch.randn((N, H, W, Co), generator=g, device=dev, dtype=torch.float32).to(BF16)
        ref = _ref_conv3x3_dev64(torch.relu(x64), w64) * (go > 0).double() + res.double()
        y = K.gconv(geom, xb, bt_f, gate_in=xb, slope_in=0.0, gate_out=go, slope_out=0.0, residual=res)
        _close_on_device(y, ref, name + " gated fwd bf16", 2.0 * 2.0 ** -8, 2.0 ** -8)
        del ref, y, go, res
    # data gradient = convolution of dy with the flipped, transposed filter
    ref_dx = _ref_conv3x3_dev64(dy64, w64.flip(0, 1).transpose(2, 3).contiguous())
    dx = K.gconv(K.geom_adjoint(geom), dyb, bt_b, out_f32=True)
    _close_on_device(dx, ref_dx, name + " dgrad f32", 2e-4, 2e-4)
    del ref_dx, dx
    # weight / bias gradient, ReLU self-gate on the input (the form every block uses)
    xr = torch.relu(x64)
    xp = F.pad(xr, (0, 0, 1, 1, 1, 1))
    dy2 = dy64.reshape(-1, Co)
    ref_dw = torch.stack([torch.stack([
        xp[:, r:r + H, s_:s_ + W, :].reshape(-1, Ci).t() @ dy2 for s_ in range(3)]) for r in range(3)])
    dw, db = K.gwgrad(geom, xb, dyb, gate_in=xb, slope_in=0.0, want_dbias=True)
    _close_on_device(dw, ref_dw, name + " wgrad", 2e-4, 2e-4)
    _close_on_device(db, dy2.sum(dim=0), name + " dbias", 2e-4, 2e-4)


@pytest.mark.parametrize("shape", [(3, 8, 8, 64), (2, 16, 16, 128), (4, 4, 4, 512), (2, 32, 32, 72),
                                   (2, 16, 16, 3)])   # (RGB: the first block of a discriminator)
def test_layer_norm(K, dev, shape):
    """cg_layer_norm_fwd / _bwd (arch_ops.py:448-450: tf.contrib.layers.layer_norm defaults --
    per-sample moments over (H, W, C), per-channel gamma / beta, variance_epsilon 1e-12) and the
    autograd Function, against the oracle's restatement with fp64 autograd."""
    from compare_gan_amd.hip import functional as Fn
    from oracle import arch_ops as oops2
    g = _gen(sum(shape))
    x64, xb = rand_bf16(shape, g, 1.5)
    x64 = x64 + 0.25
    xb = x64.float().to(BF16)
    x64 = xb.double()
    c = shape[-1]
    gamma = (1.0 + 0.3 * torch.randn(c, generator=g)).float()
    beta = (0.3 * torch.randn(c, generator=g)).float()
    dy64, dyb = rand_bf16(shape, g)
    vs = oops2.VarStore()
    vs.vars["ln/beta"] = beta.double().requires_grad_(True)
    vs.vars["ln/gamma"] = gamma.double().requires_grad_(True)
    xr = x64.clone().requires_grad_(True)
    ref = oops2.layer_norm(vs, xr, True, "ln")
    gx, gg, gb = torch.autograd.grad((ref * dy64).sum(), [xr, vs.vars["ln/gamma"], vs.vars["ln/beta"]])
    xd = xb.to(dev).requires_grad_(True)
    gd = gamma.to(dev).requires_grad_(True)
    bd = beta.to(dev).requires_grad_(True)
    y = Fn.layer_norm(xd, gd, bd)
    assert_close_bf16(y, ref.detach(), "layer_norm fwd")
    dx, dg, db = torch.autograd.grad(y, [xd, gd, bd], grad_outputs=dyb.to(dev))
    assert_close_bf16(dx, gx, "layer_norm dx", ulps=3.0, abs_rms=2.0 ** -7)
    assert_close_f32(dg, gg, "layer_norm dgamma", rtol=2e-3, abs_rms=2e-3)
    assert_close_f32(db, gb, "layer_norm dbeta", rtol=2e-3, abs_rms=2e-3)


@pytest.mark.parametrize("shape", [(3, 8, 8, 64), (2, 16, 16, 128), (2, 16, 16, 3)])
def test_layer_norm_double_backward(K, dev, shape):
    """Second order through layer_norm (cg_layer_norm_bwd_bwd): the gradient of a penalty on the
    input-gradient, (||d sum(y * w) / dx||_2 - 1)^2 summed over the samples (penalty_lib.py:59-82
    through resnet_ops.py:162-173 with D.layer_norm = True), with respect to x AND gamma, against
    fp64 autograd of the oracle's restatement (create_graph=True on both sides)."""
    from compare_gan_amd.hip import functional as Fn
    from oracle import arch_ops as oops2
    g = _gen(7 + sum(shape))
    x64, xb = rand_bf16(shape, g, 1.5)
    xb = (x64 + 0.25).float().to(BF16)
    x64 = xb.double()
    c = shape[-1]
    gamma = (1.0 + 0.3 * torch.randn(c, generator=g)).float()
    beta = (0.3 * torch.randn(c, generator=g)).float()
    w64, wb = rand_bf16(shape, g)

    def penalty(y, x, w):
        (gx,) = torch.autograd.grad((y * w).sum(), x, create_graph=True)
        nrm = gx.reshape(shape[0], -1).double().pow(2).sum(dim=1).add(1e-8).sqrt()
        return ((nrm - 1.0) ** 2).sum()

    vs = oops2.VarStore()
    vs.vars["ln/beta"] = beta.double().requires_grad_(True)
    vs.vars["ln/gamma"] = gamma.double().requires_grad_(True)
    xr = x64.clone().requires_grad_(True)
    pen_r = penalty(oops2.layer_norm(vs, xr, True, "ln"), xr, w64)
    gx_r, gg_r = torch.autograd.grad(pen_r, [xr, vs.vars["ln/gamma"]])
    xd = xb.to(dev).requires_grad_(True)
    gd = gamma.to(dev).requires_grad_(True)
    bd = beta.to(dev).requires_grad_(True)
    pen = penalty(Fn.layer_norm(xd, gd, bd), xd, wb.to(dev))
    assert abs(float(pen) - float(pen_r)) <= 2e-2 * max(1.0, abs(float(pen_r))), (float(pen), float(pen_r))
    gx, gg = torch.autograd.grad(pen, [xd, gd])
    from tests.gan_util import cosine, rel_l2
    assert cosine(gx, gx_r) >= 0.999 and rel_l2(gx, gx_r) <= 0.05, (cosine(gx, gx_r), rel_l2(gx, gx_r))
    assert cosine(gg, gg_r) >= 0.999 and rel_l2(gg, gg_r) <= 0.05, (cosine(gg, gg_r), rel_l2(gg, gg_r))


def test_gwgrad_multi_grouped(K, dev):
    """cg_gwgrad_multi: the weight (+ bias) gradients of several layers in one call -- the small-map
    ones share a launch (swgrad_kernel), the others run through cg_gwgrad -- against the fp64 oracle
    convolution; ReLU input gates on some jobs, the sizes of the ResNet-CIFAR discriminator's 8x8
    blocks (resnet_cifar.py:119-167) at batch 8 among them."""
    g = _gen(91)
    shapes = [(8, 8, 8, 128, 128, True), (8, 8, 8, 128, 128, False), (5, 4, 4, 64, 192, True),
              (2, 16, 16, 64, 64, False), (2, 32, 32, 64, 64, True), (3, 8, 8, 96, 64, False)]
    jobs, refs = [], []
    for (N, H, W, Ci, Co, relu) in shapes:
        x64, xb = rand_bf16((N, H, W, Ci), g)
        dy64, dyb = rand_bf16((N, H, W, Co), g)
        geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
        w0 = torch.zeros((3, 3, Ci, Co), dtype=torch.float64, requires_grad=True)
        xin = torch.relu(x64) if relu else x64
        (_ref_conv(xin, w0, 1, 1) * dy64).sum().backward()
        dw = torch.full((3, 3, Ci, Co), float("nan"), device=dev)
        db = torch.full((Co,), float("nan"), device=dev)
        jobs.append((geom, xb.to(dev), dyb.to(dev), relu, dw, db))
        refs.append((w0.grad, dy64.sum(dim=(0, 1, 2))))
    K.gwgrad_multi(jobs)
    for (N, H, W, Ci, Co, relu), (_, _, _, _, dw, db), (rw, rb) in zip(shapes, jobs, refs):
        name = "multi %dx%dx%dx%d->%d" % (N, H, W, Ci, Co)
        assert_close_f32(dw, rw, name + " wgrad", rtol=2e-4, abs_rms=2e-4)
        assert_close_f32(db, rb, name + " dbias", rtol=2e-4, abs_rms=2e-4)


def test_deferred_wgrads_match_immediate(K, dev):
    """Fn.deferred_wgrads(): the recorded weight gradients, run as one cg_gwgrad_multi call at the
    end of the context (or when the spectral-norm backward asks for them), equal the ones computed
    inside the backward pass -- bit for bit where the same kernel runs, to rounding otherwise."""
    from compare_gan_amd.hip import functional as Fn
    g = _gen(17)
    N, H, W, C = 8, 8, 8, 128
    _, xb = rand_bf16((N, H, W, C), g)
    ws = [rand_bf16((3, 3, C, C), g, 0.03)[1].float().to(dev).requires_grad_(True) for _ in range(3)]
    bs = [torch.zeros(C, device=dev, requires_grad=True) for _ in range(3)]
    geom = K.geom_conv_same(N, H, W, C, C, 3, 3, 1, 1)

    def run(defer):
        h = xb.to(dev)
        for w, b in zip(ws, bs):
            h = Fn.gconv(h, w, b, gate_in=h, spec=Fn.ConvSpec(geom, slope_in=0.0))
        loss = Fn.relu_mean(h).sum()
        with Fn.deferred_wgrads(defer):
            grads = torch.autograd.grad(loss, ws + bs)
        return [t.clone() for t in grads]

    immediate = run(False)
    deferred = run(True)
    for a, b in zip(immediate, deferred):
        assert torch.isfinite(b).all()
        assert_close_f32(b, a.double().cpu(), "deferred vs immediate", rtol=1e-4, abs_rms=1e-4)


def test_deferred_wgrads_with_a_torch_op_on_the_weight_path(K, dev):
    """The self-attention projections run with zero-padded kernels (arch_ops.conv2d pad_out_to): the
    padding's backward is a torch slice of dw INSIDE the backward pass, so inside
    Fn.deferred_wgrads() such a weight gradient must be complete when the convolution's backward
    returns (its split reduction is not left recorded).  Bit-identical to the undeferred pass."""
    from compare_gan_amd.hip import functional as Fn
    g = _gen(29)
    N, H, W, C = 2, 64, 64, 32
    _, xb = rand_bf16((N, H, W, C), g)
    w_small = rand_bf16((1, 1, C, 4), g, 0.1)[1].float().to(dev).requires_grad_(True)
    w_plain = rand_bf16((3, 3, C, 64), g, 0.05)[1].float().to(dev).requires_grad_(True)
    g1 = K.geom_conv_same(N, H, W, C, 32, 1, 1, 1, 1)
    g3 = K.geom_conv_same(N, H, W, 32, 64, 3, 3, 1, 1)

    def run(defer):
        wp = torch.nn.functional.pad(w_small, (0, 28))
        h = Fn.gconv(xb.to(dev), wp, None, spec=Fn.ConvSpec(g1))
        h = Fn.gconv(h, w_plain, None, gate_in=h, spec=Fn.ConvSpec(g3, slope_in=0.0))
        loss = Fn.relu_mean(h).sum()
        with Fn.deferred_wgrads(defer):
            grads = torch.autograd.grad(loss, [w_small, w_plain])
        return [t.clone() for t in grads]

    immediate = run(False)
    deferred = run(True)
    for a, b in zip(immediate, deferred):
        assert float(a.abs().max()) > 0.0
        assert torch.equal(a, b)


def _deferred_reduction_ops(K, dev, seed=23):
    g = _gen(seed)
    cases = [  # N, H, W, Ci, Co, relu, pooled
        (6, 32, 32, 128, 128, True, False), (4, 32, 32, 64, 64, False, True), (9, 32, 32, 3, 128, False, False),
        (5, 32, 32, 3, 64, False, True), (8, 16, 16, 128, 256, True, False), (3, 64, 64, 64, 64, True, False),
        (40, 8, 8, 512, 512, False, False), (8, 16, 16, 256, 128, False, False),
        (16, 32, 32, 128, 256, False, "s2"), (8, 64, 64, 64, 128, False, "s2")]   # 4x4 stride 2 (sndcgan.py:109-121)
    ops = []
    for (N, H, W, Ci, Co, relu, pooled) in cases:
        _, xb = rand_bf16((N, H, W, Ci), g)
        if pooled == "s2":
            geom = K.geom_conv_same(N, H, W, Ci, Co, 4, 4, 2, 1)
            pooled = False
        else:
            geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
        _, dyb = rand_bf16((N, geom.Ho // 2, geom.Wo // 2, Co) if pooled else (N, geom.Ho, geom.Wo, Co), g)
        ops.append((geom, xb.to(dev), dyb.to(dev), relu, pooled))

    def run_all(defer=None, ops_=None):
        outs = []
        for geom, x, dy, relu, pooled in (ops if ops_ is None else ops_):
            gi = x if relu else None
            if pooled:
                outs.extend(K.gwgrad_pooled(geom, x, dy, gate_in=gi, want_dbias=True, defer=defer))
            else:
                outs.extend(K.gwgrad(geom, x, dy, gate_in=gi, slope_in=0.0, want_dbias=True, defer=defer))
        return outs
    return cases, ops, run_all


def test_deferred_reductions_are_bit_identical(K, dev):
    """cgDeferCtx (cg_gwgrad_deferred / cg_gwgrad_pooled_deferred / cg_defer_flush): the split
    reductions behind the weight-gradient kernels (halo: float4 partials, 8 split lanes; RGB stem:
    strided partials; pooled-gradient forms; a layer with the bias partials in the reused workspace,
    whose dw reduction must NOT be deferred), recorded in a caller-owned context and run in one launch
    per form, give bit for bit the tensors of the separate launches."""
    cases, ops, run_all = _deferred_reduction_ops(K, dev)
    ref = [t.clone() for t in run_all()]
    ctx = K.DeferCtx()
    outs = run_all(ctx)
    pending = ctx.pending()
    ctx.flush()
    assert pending >= 8, pending          # (most of the layers above split their pixels)
    assert ctx.pending() == 0 and not ctx.keep
    for i, (a, b) in enumerate(zip(ref, outs)):
        assert torch.equal(a, b), "output %d of case %s" % (i % 2, cases[i // 2])
    # an aborted recording leaves nothing behind: calls without a context reduce at once, and the
    # context is reusable
    run_all(ctx)
    ctx.abort()
    assert ctx.pending() == 0
    again = run_all()
    for a, b in zip(ref, again):
        assert torch.equal(a, b)
    outs = run_all(ctx)
    ctx.flush()
    for a, b in zip(ref, outs):
        assert torch.equal(a, b)


def test_deferred_reductions_of_two_replicas_do_not_mix(K, dev):
    """the C-ABI is re-entrant, no state besides what the caller owns.  Two replicas in
    one process (two streams, two contexts, two sets of tensors) record their reductions
    INTERLEAVED -- replica A's call, then B's, ... -- with a third party calling without a context in
    between; each context flushes on its own stream and holds exactly its own reductions; all three
    results are bit-identical to the immediate form.  (the process-wide switch
    this replaces sent whatever was recorded to whichever stream flushed first.)"""
    cases, ops_a, run_all = _deferred_reduction_ops(K, dev, seed=23)
    _, ops_b, _ = _deferred_reduction_ops(K, dev, seed=29)
    ref_a = [t.clone() for t in run_all(None, ops_a)]
    ref_b = [t.clone() for t in run_all(None, ops_b)]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ca, cb = K.DeferCtx(), K.DeferCtx()
    outs_a, outs_b, outs_c = [], [], []
    for oa, ob in zip(ops_a, ops_b):
        with torch.cuda.stream(sa):
            outs_a.extend(run_all(ca, [oa]))
        with torch.cuda.stream(sb):
            outs_b.extend(run_all(cb, [ob]))
        outs_c.extend(run_all(None, [oa]))        # no context: reduced at once, recorded nowhere
    na, nb = ca.pending(), cb.pending()
    assert na == nb and na >= 8
    with torch.cuda.stream(sb):
        cb.flush()
    assert ca.pending() == na and cb.pending() == 0     # B's flush did not touch A's reductions
    with torch.cuda.stream(sa):
        ca.flush()
    torch.cuda.synchronize()
    for name, ref, outs in (("A", ref_a, outs_a), ("B", ref_b, outs_b), ("none", ref_a, outs_c)):
        for i, (a, b) in enumerate(zip(ref, outs)):
            assert torch.equal(a, b), "replica %s output %d of case %s" % (name, i % 2, cases[i // 2])


FUSED_FULL_SIZE = [
    # name, N, H, W, Ci, Co, relu_in, residual -- the pooled convolutions of the ResNet5-128 D-step at
    # the benchmark's batch (2 x 64 images): B0 conv2 and its RGB shortcut at 128x128, B1 shortcut
    ("B0_conv2_128x128", 128, 128, 128, 64, 64, True, True),
    ("B0_rgb_shortcut_128x128", 128, 128, 128, 3, 64, False, False),
    ("B1_shortcut_64x64", 128, 64, 64, 64, 128, False, False),
]


@pytest.mark.parametrize("case", FUSED_FULL_SIZE, ids=[c[0] for c in FUSED_FULL_SIZE])
def test_conv_pool_fused_full_size(K, dev, case):
    """ConvPoolFn at the sizes bench.py's resnet128_dstep leg runs: the pooled
    epilogue (FUSE = 2 / wstem pooled), the data gradient through the up-sampled read of the pooled
    dy (in_up) and the weight / bias gradients from the pooled dy (pooled hwgrad / wstem_wgrad),
    against fp64 autograd over per-tap GEMMs in plain torch on the device."""
    from compare_gan_amd.hip import functional as Fn
    name, N, H, W, Ci, Co, relu_in, with_res = case
    g = torch.Generator(device=dev).manual_seed(sum(ord(c) for c in name))
    xb = torch.randn((N, H, W, Ci), generator=g, device=dev).to(BF16)
    wb = (torch.randn((3, 3, Ci, Co), generator=g, device=dev) / math.sqrt(9 * Ci)).to(BF16)
    bias = torch.randn(Co, generator=g, device=dev)
    rb = torch.randn((N, H // 2, W // 2, Co), generator=g, device=dev).to(BF16)
    dyb = torch.randn((N, H // 2, W // 2, Co), generator=g, device=dev).to(BF16)
    geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
    import os
    if not K.gconv_pool_supported(geom) and "0" in (os.environ.get("CGAMD_HCONV"),
                                                    os.environ.get("CGAMD_WSTEM")):
        pytest.skip("the kernels that carry the fused pooling are switched off in this variant")
    assert K.gconv_pool_supported(geom)
    # reference
    xr = xb.double().requires_grad_(True)
    wr = wb.double().requires_grad_(True)
    br = bias.double().requires_grad_(True)
    conv = oops.conv2d_same_gemm(torch.relu(xr) if relu_in else xr, wr, 1) + br
    ref = F.avg_pool2d(conv.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    if with_res:
        ref = ref + rb.double()
    gx_r, gw_r, gb_r = torch.autograd.grad((ref * dyb.double()).sum(), [xr, wr, br])
    ref = ref.detach()
    del conv
    # product
    xd = xb.clone().requires_grad_(True)
    wd = wb.float().requires_grad_(True)
    bd = bias.clone().requires_grad_(True)
    y = Fn.conv_pool(xd, wd, bd, residual_p=rb if with_res else None,
                     gate_in=xd if relu_in else None,
                     spec=Fn.ConvSpec(geom, slope_in=0.0 if relu_in else None))
    _close_on_device(y, ref, name + " pooled fwd", 2.0 * 2.0 ** -8, 2.0 ** -8)
    gx, gw, gb = torch.autograd.grad(y, [xd, wd, bd], grad_outputs=dyb)
    _close_on_device(gw, gw_r, name + " wgrad from pooled dy", 2e-4, 2e-4)
    _close_on_device(gb, gb_r, name + " dbias from pooled dy", 2e-4, 2e-4)
    if Ci > 3:   # the image itself needs no gradient in a D sub-step without a penalty
        _close_on_device(gx, gx_r, name + " dgrad (in_up)", 2.0 * 2.0 ** -8, 2.0 ** -8)


@pytest.mark.parametrize("case", [
    # name, N, H, W, Ci, Co, up, per_sample: generator layers of the D-step leg (batch 64)
    ("G_64x64_c128", 64, 64, 64, 128, 128, 1, False),
    ("G_up_32to64_c256_128", 64, 32, 32, 256, 128, 2, False),
    ("G_128x128_c64", 64, 128, 128, 64, 64, 1, False),
], ids=lambda c: c[0])
def test_gconv_fused_batch_norm_full_size(K, dev, case):
    """cg_gconv_fused (FUSE = 1: batch-norm + ReLU prologue in LDS, statistics epilogue) at the
    generator's sizes in bench.py's legs, against the fp64 restatement on the device."""
    from tests.util import bf16_round
    name, N, H, W, Ci, Co, up, per_sample = case
    g = torch.Generator(device=dev).manual_seed(sum(ord(c) for c in name))
    xb = torch.randn((N, H, W, Ci), generator=g, device=dev).to(BF16)
    wb = (torch.randn((3, 3, Ci, Co), generator=g, device=dev) / math.sqrt(9 * Ci)).to(BF16)
    bias = torch.randn(Co, generator=g, device=dev)
    gamma = 1.0 + 0.3 * torch.randn(Ci, generator=g, device=dev)
    beta = 0.3 * torch.randn(Ci, generator=g, device=dev)
    x64 = xb.double()
    mean = x64.mean(dim=(0, 1, 2)).float()
    var = (x64.pow(2).mean(dim=(0, 1, 2)) - x64.mean(dim=(0, 1, 2)).pow(2)).float()
    geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, up)
    assert K.gconv_fused_rows(geom) > 0
    bt_f, _ = K.weight_prep(wb.float())
    act = bf16_round(torch.relu((x64 - mean.double()) * torch.rsqrt(var.double() + 1e-5) *
                                gamma.double() + beta.double()))
    if up == 2:
        z = act.new_zeros((N, 2 * H, 2 * W, Ci))
        z[:, ::2, ::2, :] = act
        act = z
    ref = oops.conv2d_same_gemm(act, wb.double(), 1) + bias.double()
    del act
    out, part = K.gconv_fused(geom, xb, bt_f, bias=bias,
                              bn=(mean, var, gamma, beta, 1e-5, per_sample), want_stats=True)
    _close_on_device(out, ref, name + " fused fwd", 3.0 * 2.0 ** -8, 2.0 ** -6)
    stored = out.double()
    m2, v2 = K.bn_finalize(part, N * geom.Ho * geom.Wo)
    m_ref = stored.mean(dim=(0, 1, 2))
    _close_on_device(m2, m_ref, name + " fused mean", 1e-4, 1e-4)
    _close_on_device(v2, stored.pow(2).mean(dim=(0, 1, 2)) - m_ref.pow(2), name + " fused var", 1e-3,
                     1e-4)


@pytest.mark.parametrize("size", [8, 32])
@pytest.mark.parametrize("slope", [0.0, 0.2])
def test_gconv_gates_residual(K, dev, slope, size):
    """out = d(gate_out) * (conv(lrelu(x)) + b) + residual and its wgrad with gated operands
    (size 32 is eligible for the halo-staged kernel, cg_conv_halo.hip)."""
    g = _gen(7)
    N, H, W, Ci, Co, k = 2, size, size, 128, 64, 3
    x64, xb = rand_bf16((N, H, W, Ci), g)
    w64, wb = rand_bf16((k, k, Ci, Co), g, 0.05)
    go64, gob = rand_bf16((N, H, W, Co), g)
    r64, rb = rand_bf16((N, H, W, Co), g)
    geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, 1, 1)
    bt_f, _ = K.weight_prep(wb.to(torch.float32).to(dev))
    xd = xb.to(dev)
    conv = _ref_conv(x64, w64, 1, 1, gate_slope=slope)
    dgate = torch.where(go64 > 0, torch.ones_like(go64), torch.full_like(go64, slope))
    ref = dgate * conv + r64
    y = K.gconv(geom, xd, bt_f, gate_in=xd, slope_in=slope, gate_out=gob.to(dev), slope_out=slope,
                residual=rb.to(dev), out_f32=True)
    # lrelu(x) with slope != 0 is rounded to bf16 before the MFMA -> looser tolerance
    assert_close_f32(y, ref, "gates fwd", rtol=1e-4 if slope == 0 else 5e-3,
                     abs_rms=1e-4 if slope == 0 else 5e-3)
    # wgrad with gate on x and gate on dy
    dy64, dyb = rand_bf16((N, H, W, Co), g)
    xr = torch.where(x64 > 0, x64, slope * x64)
    wr = w64.clone().requires_grad_(True)
    (oops.conv2d_same(xr, wr, 1) * (dgate * dy64)).sum().backward()
    dw, _ = K.gwgrad(geom, xd, dyb.to(dev), gate_in=xd, slope_in=slope, gate_dy=gob.to(dev),
                     slope_dy=slope)
    assert_close_f32(dw, wr.grad, "gated wgrad", rtol=1e-4 if slope == 0 else 1e-2,
                     abs_rms=1e-4 if slope == 0 else 1e-2)
    if slope == 0:
        # ReLU self-gate only (the transposed-read LDS-DMA weight-gradient path), with bias grads
        wr2 = w64.clone().requires_grad_(True)
        (oops.conv2d_same(xr, wr2, 1) * dy64).sum().backward()
        dw2, db2 = K.gwgrad(geom, xd, dyb.to(dev), gate_in=xd, slope_in=0.0, want_dbias=True)
        assert_close_f32(dw2, wr2.grad, "relu-gated wgrad", rtol=1e-4, abs_rms=1e-4)
        assert_close_f32(db2, dy64.sum(dim=(0, 1, 2)), "relu-gated dbias", rtol=2e-4, abs_rms=2e-4)


@pytest.mark.parametrize("case", [
    # name, N, H, W, Ci, Co, k, up, per_sample, residual
    ("bn_c64", 2, 32, 32, 64, 64, 3, 1, False, False),
    # 64 -> 64 channels on whole 128 x 128 maps (one channel block, 8 x 32 tiles), plain and per-sample
    # coefficients
    ("bn_c64_128_res", 3, 128, 128, 64, 64, 3, 1, False, True),
    ("cbn_c64_128", 2, 128, 128, 64, 64, 3, 1, True, False),
    ("bn_c128_res", 3, 16, 16, 128, 192, 3, 1, False, True),
    ("cbn_up", 2, 16, 16, 128, 64, 3, 2, True, False),
    ("cbn_1x1", 2, 32, 32, 64, 128, 1, 1, True, True),
    ("cbn_c96", 2, 16, 16, 96, 64, 3, 1, True, False),
    ("bn_up_c160_res", 1, 16, 32, 160, 96, 3, 2, False, True),
    # RGB outputs (hconv_kernel<64, *, *, 3>): the prologue only, 8x32 and 16x16 tiles
    ("bn_rgb_out", 2, 32, 32, 128, 3, 3, 1, False, False),
    ("cbn_rgb_out_c96", 3, 16, 16, 96, 3, 3, 1, True, False),
], ids=lambda c: c[0])
def test_gconv_fused_batch_norm(K, dev, case):
    """cg_gconv_fused: relu(batch_norm(x)) applied in LDS in front of the convolution
    (arch_ops.py:289-313,423-445 + resnet_ops.py:165) and the per-channel partial sums of the
    stored output for the next batch norm (cg_bn_finalize), against the oracle: the normalised
    activation is rounded to bf16 exactly where the unfused path stores it."""
    from tests.util import bf16_round
    name, N, H, W, Ci, Co, k, up, per_sample, with_res = case
    g = _gen(sum(ord(c) for c in name))
    x64, xb = rand_bf16((N, H, W, Ci), g)
    w64, wb = rand_bf16((k, k, Ci, Co), g, 1.0 / math.sqrt(k * k * Ci))
    bias = torch.randn(Co, generator=g, dtype=torch.float32)
    gshape = (N, Ci) if per_sample else (Ci,)
    gamma = (1.0 + 0.3 * torch.randn(gshape, generator=g)).float()
    beta = (0.3 * torch.randn(gshape, generator=g)).float()
    eps = 1e-5
    mean = x64.mean(dim=(0, 1, 2)).float()
    var = (x64.pow(2).mean(dim=(0, 1, 2)) - x64.mean(dim=(0, 1, 2)).pow(2)).float()
    geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, 1, up)
    assert K.gconv_fused_prologue_supported(geom) and (K.gconv_fused_rows(geom) > 0) == (Co >= 8)
    bt_f, _ = K.weight_prep(wb.to(torch.float32).to(dev))
    res64, resb = rand_bf16((N, geom.Ho, geom.Wo, Co), g)
    # oracle: fp32 statistics as given, normalisation in fp64, bf16 storage of the activation
    gm = gamma.double().reshape((N, 1, 1, Ci) if per_sample else (1, 1, 1, Ci))
    bt_ = beta.double().reshape((N, 1, 1, Ci) if per_sample else (1, 1, 1, Ci))
    xhat = (x64 - mean.double()) * torch.rsqrt(var.double() + eps)
    act = bf16_round(torch.relu(xhat * gm + bt_))
    ref = _ref_conv(act, w64, 1, up) + bias.double()
    if with_res:
        ref = ref + res64
    out, part = K.gconv_fused(geom, xb.to(dev), bt_f, bias=bias.to(dev),
                              residual=resb.to(dev) if with_res else None,
                              bn=(mean.to(dev), var.to(dev), gamma.to(dev), beta.to(dev), eps,
                                  per_sample), want_stats=Co >= 8)
    # the activation is rounded to bf16 before the MFMA: a value on a rounding boundary may land on
    # the other side than in the fp64 oracle -> 2^-8 relative noise on single products
    assert_close_bf16(out, ref, name + " fused fwd", ulps=3.0, abs_rms=2.0 ** -6)
    if Co < 8:   # no statistics epilogue on narrow outputs: refused, not silently dropped
        assert part is None
        with pytest.raises(Exception, match="not covered"):
            K.gconv_fused(geom, xb.to(dev), bt_f, bias=bias.to(dev),
                          bn=(mean.to(dev), var.to(dev), gamma.to(dev), beta.to(dev), eps, per_sample),
                          want_stats=True)
        return
    # statistics of the STORED values
    stored = out.detach().float().cpu().double()
    cnt = N * geom.Ho * geom.Wo
    m2, v2 = K.bn_finalize(part, cnt)
    m_ref = stored.mean(dim=(0, 1, 2))
    v_ref = stored.pow(2).mean(dim=(0, 1, 2)) - m_ref.pow(2)
    assert_close_f32(m2, m_ref, name + " fused mean", rtol=1e-4, abs_rms=1e-4)
    assert_close_f32(v2, v_ref, name + " fused var", rtol=1e-3, abs_rms=1e-4)
    # statistics only (no prologue) on the same convolution
    out2, part2 = K.gconv_fused(geom, xb.to(dev), bt_f, bias=bias.to(dev), want_stats=True)
    ref2 = _ref_conv(x64, w64, 1, up) + bias.double()
    assert_close_bf16(out2, ref2, name + " stats-only fwd")
    m3, _ = K.bn_finalize(part2, cnt)
    assert_close_f32(m3, out2.detach().float().cpu().double().mean(dim=(0, 1, 2)),
                     name + " stats-only mean", rtol=1e-4, abs_rms=1e-4)


@pytest.mark.parametrize("C,HW", [(64, 64), (256, 16), (40, 9), (3, 1024)])
def test_batch_norm_statistics_groups(K, dev, C, HW):
    """cg_bn_stats_groups / cg_bn_apply_groups: one launch over a batch that stands for `groups`
    separate calls is bit for bit the separate calls (arch_ops.py:289-313 per call; the moving
    averages take one update per call, in order: arch_ops.py:105-114)."""
    groups, per = 3, 4
    g = _gen(C * 7 + HW)
    _, xb = rand_bf16((groups * per, HW, C), g, 2.0)
    xb = xb.to(dev)
    mm = torch.randn(C, generator=g).float().to(dev)
    mv = (1.0 + torch.rand(C, generator=g)).float().to(dev)
    mm_g, mv_g = mm.clone(), mv.clone()
    mean_g, var_g = K.bn_stats(xb, mm_g, mv_g, 0.9, groups=groups)
    assert mean_g.shape == (groups, C)
    gamma = (1.0 + 0.3 * torch.randn((groups * per, C), generator=g)).float().to(dev)
    beta = (0.3 * torch.randn((groups * per, C), generator=g)).float().to(dev)
    y_g = K.bn_apply(xb, mean_g, var_g, 1e-5, gamma, beta, True, True)
    for i in range(groups):
        xs = xb[i * per:(i + 1) * per].contiguous()
        m, v = K.bn_stats(xs, mm, mv, 0.9)
        assert torch.equal(m, mean_g[i]) and torch.equal(v, var_g[i]), "group %d statistics" % i
        y = K.bn_apply(xs, m, v, 1e-5, gamma[i * per:(i + 1) * per].contiguous(),
                       beta[i * per:(i + 1) * per].contiguous(), True, True)
        assert torch.equal(y, y_g[i * per:(i + 1) * per]), "group %d apply" % i
    assert torch.equal(mm, mm_g) and torch.equal(mv, mv_g), "moving averages"


@pytest.mark.parametrize("case", [
    # name, per-group N, H, W, Ci, Co, up, per_sample
    ("grp_c64", 2, 32, 32, 64, 128, 1, False),
    ("grp_cbn_up", 2, 16, 16, 128, 64, 2, True),
    ("grp_c256_16", 4, 16, 16, 256, 256, 1, False),
], ids=lambda c: c[0])
def test_gconv_fused_statistics_groups(K, dev, case):
    """cg_gconv_fused with bn_stat_group and cg_bn_finalize_groups: the batched call with one set
    of batch-norm statistics per group of samples equals the per-group calls bit for bit (same
    kernel, same tiles), for the normalised input AND for the statistics it emits."""
    name, per, H, W, Ci, Co, up, per_sample = case
    groups = 3
    N = groups * per
    g = _gen(sum(ord(c) for c in name))
    _, xb = rand_bf16((N, H, W, Ci), g)
    _, wb = rand_bf16((3, 3, Ci, Co), g, 1.0 / math.sqrt(9 * Ci))
    xb = xb.to(dev)
    bias = torch.randn(Co, generator=g, dtype=torch.float32).to(dev)
    gshape = (N, Ci) if per_sample else (Ci,)
    gamma = (1.0 + 0.3 * torch.randn(gshape, generator=g)).float().to(dev)
    beta = (0.3 * torch.randn(gshape, generator=g)).float().to(dev)
    bt_f, _ = K.weight_prep(wb.to(torch.float32).to(dev))
    mean, var = K.bn_stats(xb.reshape(N, H * W, Ci), groups=groups)
    geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, up)
    geom1 = K.geom_conv_same(per, H, W, Ci, Co, 3, 3, 1, up)
    assert K.gconv_fused_rows(geom) > 0
    out, part = K.gconv_fused(geom, xb, bt_f, bias=bias,
                              bn=(mean, var, gamma, beta, 1e-5, per_sample), want_stats=True)
    cnt = N * geom.Ho * geom.Wo
    m_g, v_g = K.bn_finalize(part, cnt, groups=groups, phases=K.gconv_fused_phases(geom))
    for i in range(groups):
        sl = slice(i * per, (i + 1) * per)
        gm = gamma[sl].contiguous() if per_sample else gamma
        bt_ = beta[sl].contiguous() if per_sample else beta
        o1, p1 = K.gconv_fused(geom1, xb[sl].contiguous(), bt_f, bias=bias,
                               bn=(mean[i].contiguous(), var[i].contiguous(), gm, bt_, 1e-5,
                                   per_sample), want_stats=True)
        assert torch.equal(o1, out[sl]), "%s group %d output" % (name, i)
        m1, v1 = K.bn_finalize(p1, cnt // groups)
        assert torch.equal(m1, m_g[i]) and torch.equal(v1, v_g[i]), "%s group %d stats" % (name, i)


@pytest.mark.parametrize("case", [
    # name, N, H, W, Ci, Co, relu_in, residual
    ("pool_c64_128", 2, 32, 32, 64, 128, True, True),
    ("pool_c128_64_16", 3, 16, 16, 128, 64, True, False),
    ("pool_c64_64_16", 2, 16, 16, 64, 64, False, True),
    # 64 -> 64 on 32-wide maps: the register-resident-weight kernel's pooled / up-sampled-input forms
    # when CGAMD_HCONV_RW_MIN lets these small grids take it (variant "hconv_all" below)
    ("pool_c64_64_32x64", 2, 32, 64, 64, 64, True, True),
    ("pool_c64_64_64x32_plain", 1, 64, 32, 64, 64, False, False),
    ("pool_c96", 2, 32, 32, 96, 192, True, True),
    ("pool_rgb_64", 2, 32, 32, 3, 64, True, False),
    ("pool_rgb_128_16", 3, 16, 16, 3, 128, False, False),
], ids=lambda c: c[0])
def test_conv_pool_fused(K, dev, case):
    """ConvPoolFn: avgpool2x2(conv3x3(relu?(x)) + b) + r in one kernel (resnet_ops.py:131-133,
    165-181) and its gradients from the pooled-resolution dy (data gradient through the
    nearest-neighbour-upsampled read, weight / bias gradients through cg_gwgrad_pooled), against
    torch autograd on the fp64 oracle."""
    from compare_gan_amd.hip import functional as Fn
    name, N, H, W, Ci, Co, relu_in, with_res = case
    g = _gen(sum(ord(c) for c in name))
    x64, xb = rand_bf16((N, H, W, Ci), g)
    w64, wb = rand_bf16((3, 3, Ci, Co), g, 1.0 / math.sqrt(9 * Ci))
    bias = torch.randn(Co, generator=g, dtype=torch.float32)
    r64, rb = rand_bf16((N, H // 2, W // 2, Co), g)
    geom = K.geom_conv_same(N, H, W, Ci, Co, 3, 3, 1, 1)
    import os
    if not K.gconv_pool_supported(geom) and "0" in (os.environ.get("CGAMD_HCONV"),
                                                    os.environ.get("CGAMD_WSTEM")):
        pytest.skip("the kernels that carry the fused pooling are switched off in this variant")
    assert K.gconv_pool_supported(geom)
    xr = x64.clone().requires_grad_(True)
    wr = w64.clone().requires_grad_(True)
    br = bias.double().clone().requires_grad_(True)
    conv = _ref_conv(xr, wr, 1, 1, gate_slope=0.0 if relu_in else None) + br
    ref = F.avg_pool2d(conv.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    if with_res:
        ref = ref + r64
    dy64, dyb = rand_bf16(tuple(ref.shape), g)
    (ref * dy64).sum().backward()

    xd = xb.to(dev).requires_grad_(True)
    wd = wb.to(torch.float32).to(dev).requires_grad_(True)
    bd = bias.to(dev).requires_grad_(True)
    rd = rb.to(dev).requires_grad_(True) if with_res else None
    spec = Fn.ConvSpec(geom, slope_in=0.0 if relu_in else None)
    y = Fn.conv_pool(xd, wd, bd, rd, xd.detach() if relu_in else None, spec, Ci <= 4)
    assert_close_bf16(y, ref.detach(), name + " fwd")
    grads = torch.autograd.grad(y, [xd, wd, bd] + ([rd] if with_res else []),
                                grad_outputs=dyb.to(dev))
    if Ci > 4:
        assert_close_bf16(grads[0], xr.grad, name + " dx", ulps=2.0, abs_rms=2.0 ** -7)
    assert_close_f32(grads[1], wr.grad, name + " dw", rtol=2e-4, abs_rms=2e-4)
    assert_close_f32(grads[2], br.grad, name + " db", rtol=2e-4, abs_rms=2e-4)
    if with_res:
        assert torch.equal(grads[3].cpu().double(), dy64)


@pytest.mark.parametrize("size", [12, 32])
def test_stem_relu_gate(K, dev, size):
    """Image-like input (Ci = 3) with the ReLU input gate of a D block's first convolution
    (resnet_ops.py:165): forward, weight and bias gradients (size 32: window-staged kernels)."""
    g = _gen(21)
    N, H, W, Ci, Co, k = 3, size, size, 3, 64, 3
    x64, xb = rand_bf16((N, H, W, Ci), g)
    w64, wb = rand_bf16((k, k, Ci, Co), g, 0.2)
    bias = torch.randn(Co, generator=g, dtype=torch.float32)
    geom = K.geom_conv_same(N, H, W, Ci, Co, k, k, 1, 1)
    bt_f, _ = K.weight_prep(wb.to(torch.float32).to(dev))
    xd = xb.to(dev)
    wr = w64.clone().requires_grad_(True)
    ref = oops.conv2d_same(torch.relu(x64), wr, 1) + bias.to(torch.float64)
    y = K.gconv(geom, xd, bt_f, bias=bias.to(dev), gate_in=xd, slope_in=0.0)
    assert_close_bf16(y, ref.detach(), "stem relu fwd")
    dy64, dyb = rand_bf16(tuple(ref.shape), g)
    (ref * dy64).sum().backward()
    dw, db = K.gwgrad(geom, xd, dyb.to(dev), gate_in=xd, slope_in=0.0, want_dbias=True)
    assert_close_f32(dw, wr.grad, "stem relu wgrad", rtol=2e-4, abs_rms=2e-4)
    assert_close_f32(db, dy64.sum(dim=(0, 1, 2)), "stem dbias", rtol=2e-4, abs_rms=2e-4)


DECONV_CASES = [("dcgan_5x5", 2, 4, 4, 64, 32, 5, 2), ("sndcgan_4x4", 2, 8, 8, 32, 16, 4, 2),
                ("sndcgan_3x3_s1", 2, 8, 8, 64, 3, 3, 1)]


@pytest.mark.parametrize("case", DECONV_CASES, ids=[c[0] for c in DECONV_CASES])
def test_deconv(K, dev, case):
    """conv2d_transpose(SAME) = adjoint geometry of the forward conv with kernel [kh,kw,Cout,Cin]."""
    name, N, H, W, Cin, Cout, k, s = case
    g = _gen(11)
    x64, xb = rand_bf16((N, H, W, Cin), g)
    w64, wb = rand_bf16((k, k, Cout, Cin), g, 0.1)
    Hy, Wy = H * s, W * s
    xr = x64.clone().requires_grad_(True)
    wr = w64.clone().requires_grad_(True)
    ref = oops.conv2d_transpose_same(xr, wr, (Hy, Wy), s)
    fgeom = K.geom_conv_same(N, Hy, Wy, Cout, Cin, k, k, s, 1)  # F: y-space -> x-space
    assert (fgeom.Ho, fgeom.Wo) == (H, W)
    _, bt_b = K.weight_prep(wb.to(torch.float32).to(dev), want_fwd=False, want_bwd=True)
    y = K.gconv(K.geom_adjoint(fgeom), xb.to(dev), bt_b, out_f32=True)
    assert_close_f32(y, ref.detach(), name + " deconv fwd")
    dy64, dyb = rand_bf16(tuple(ref.shape), g)
    (ref * dy64).sum().backward()
    # dx = F(dy) ; dw = wgrad_F(in = dy, "dy" = x)
    bt_f, _ = K.weight_prep(wb.to(torch.float32).to(dev), want_fwd=True)
    dx = K.gconv(fgeom, dyb.to(dev), bt_f, out_f32=True)
    assert_close_f32(dx, xr.grad, name + " deconv dx", rtol=2e-4, abs_rms=2e-4)
    dw, _ = K.gwgrad(fgeom, dyb.to(dev), xb.to(dev))
    assert_close_f32(dw, wr.grad, name + " deconv dw", rtol=2e-4, abs_rms=2e-4)


def test_linear_as_1x1(K, dev):
    g = _gen(3)
    B, Kin, Nout = 64, 128, 4096
    x64, xb = rand_bf16((B, Kin), g)
    w64, wb = rand_bf16((Kin, Nout), g, 0.05)
    geom = K.make_geom(B, 1, 1, Kin, 1, 1, Nout, 1, 1)
    bt_f, _ = K.weight_prep(wb.to(torch.float32).to(dev).reshape(1, 1, Kin, Nout))
    y = K.gconv(geom, xb.to(dev), bt_f, out_f32=True)
    assert_close_f32(y, x64 @ w64, "linear fwd")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(27, 128), (1152, 128), (128, 1), (148, 1536), (2304, 96)])
def test_spectral_norm(K, dev, mode, shape):
    g = _gen(5)
    Kd, Co = shape
    w = torch.randn(shape, generator=g, dtype=torch.float32) * 0.05
    u0 = torch.randn((Kd, 1) if mode == 0 else (1, Co), generator=g, dtype=torch.float32)
    vs = oops.VarStore(dtype=torch.float64)
    vs.vars["w/u_var"] = u0.to(torch.float64).clone()
    wr = w.to(torch.float64).clone().requires_grad_(True)
    wbar = oops.spectral_norm(vs, wr, "w", 1e-12, "left" if mode == 0 else "right")
    u_d = u0.reshape(-1).to(dev).clone()
    v, sigma, inv_sigma = K.spectral_norm(w.to(dev), u_d, mode)
    assert_close_f32(u_d, vs.vars["w/u_var"].reshape(-1), "u'", rtol=1e-4, abs_rms=1e-4)
    sig_ref = (w.to(torch.float64) / wbar.detach()).mean()
    assert abs(float(sigma) - float(sig_ref)) <= 2e-5 * abs(float(sig_ref)), (float(sigma), float(sig_ref))
    wbar_d = K.scale_f32(w.to(dev), inv_sigma)
    assert_close_f32(wbar_d, wbar.detach(), "w_bar", rtol=1e-4, abs_rms=1e-5)
    # backward through w/sigma with u, v constant
    dwbar = torch.randn(shape, generator=g, dtype=torch.float32)
    (wbar * dwbar.to(torch.float64)).sum().backward()
    a_k, b_co = (u_d, v) if mode == 0 else (v, u_d)
    dw = K.sn_backward(dwbar.to(dev), w.to(dev), a_k, b_co, sigma)
    assert_close_f32(dw, wr.grad, "sn backward", rtol=2e-4, abs_rms=2e-4)


def test_batch_norm_golden(K, dev):
    """The reference's own golden vector (architectures/arch_ops_test.py:32-61), eps = 1e-3."""
    x = torch.tensor([[[[5, 7, 2]], [[5, 8, 8]]], [[[1, 2, 0]], [[4, 0, 4]]],
                      [[[6, 2, 6]], [[5, 0, 5]]], [[[2, 4, 2]], [[6, 4, 1]]]], dtype=torch.float32)
    expected = np.asarray(
        [[[[0.4375205, 1.30336881, -0.58830315]], [[0.4375205, 1.66291881, 1.76490951]]],
         [[[-1.89592218, -0.49438119, -1.37270737]], [[-0.14584017, -1.21348119, 0.19610107]]],
         [[[1.02088118, -0.49438119, 0.98050523]], [[0.4375205, -1.21348119, 0.58830321]]],
         [[[-1.31256151, 0.22471881, -0.58830315]], [[1.02088118, 0.22471881, -0.98050523]]]],
        dtype=np.float32)
    x3 = x.reshape(4, 2, 3).to(BF16).to(dev)  # small integers are exact in bf16
    mean, var = K.bn_stats(x3)
    y = K.bn_apply(x3, mean, var, 1e-3)
    assert_close_bf16(y, torch.from_numpy(expected).to(torch.float64).reshape(4, 2, 3),
                      "BN golden", ulps=2.0, abs_rms=2.0 ** -9)


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("shape", [(8, 16, 256), (32, 1, 1024), (4, 64, 24)])
def test_batch_norm_fwd_bwd(K, dev, per_sample, shape):
    g = _gen(9)
    N, HW, C = shape
    eps = 1e-5
    x64, xb = rand_bf16(shape, g)
    x64 = x64 * 1.5 + 0.3
    xb = x64.to(torch.float32).to(BF16)
    x64 = xb.to(torch.float64)
    pshape = (N, C) if per_sample else (C,)
    gamma = torch.randn(pshape, generator=g, dtype=torch.float32) * 0.5 + 1.0
    beta = torch.randn(pshape, generator=g, dtype=torch.float32) * 0.2
    xr = x64.clone().requires_grad_(True)
    gr = gamma.to(torch.float64).requires_grad_(True)
    br = beta.to(torch.float64).requires_grad_(True)
    mean_r = xr.mean(dim=(0, 1))
    var_r = (xr * xr).mean(dim=(0, 1)) - mean_r ** 2
    xhat = (xr - mean_r) * torch.rsqrt(var_r + eps)
    gg = gr.reshape(N, 1, C) if per_sample else gr
    bb = br.reshape(N, 1, C) if per_sample else br
    ref = torch.relu(xhat * gg + bb)
    xd = xb.to(dev)
    mean, var = K.bn_stats(xd)
    assert_close_f32(mean, mean_r.detach(), "bn mean", rtol=1e-5, abs_rms=1e-5)
    assert_close_f32(var, var_r.detach(), "bn var", rtol=1e-4, abs_rms=1e-4)
    y = K.bn_apply(xd, mean, var, eps, gamma.to(dev), beta.to(dev), per_sample, relu=True)
    assert_close_bf16(y, ref.detach(), "bn apply")
    dy64, dyb = rand_bf16(shape, g)
    (ref * dy64).sum().backward()
    dx, dgam, dbet = K.bn_backward(xd, y, dyb.to(dev), mean, var, eps, gamma.to(dev), per_sample,
                                   relu=True, batch_stats=True)
    # y's relu mask is taken from the bf16-rounded y: identical sign pattern except exact zeros
    assert_close_bf16(dx, xr.grad, "bn dx", ulps=4.0, abs_rms=2.0 ** -6)
    assert_close_f32(dgam, gr.grad, "bn dgamma", rtol=2e-3, abs_rms=2e-3)
    assert_close_f32(dbet, br.grad, "bn dbeta", rtol=2e-3, abs_rms=2e-3)


def test_bn_moving_average(K, dev):
    mm = torch.zeros(8, device=dev)
    mv = torch.ones(8, device=dev)
    mean = torch.arange(8, dtype=torch.float32, device=dev)
    var = torch.full((8,), 3.0, device=dev)
    K.bn_update_moving(mm, mv, mean, var, 0.9)
    torch.testing.assert_close(mm.cpu(), 0.1 * torch.arange(8, dtype=torch.float32), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(mv.cpu(), torch.full((8,), 1.2), rtol=1e-6, atol=1e-7)


def test_elementwise_and_pooling(K, dev):
    g = _gen(13)
    x64, xb = rand_bf16((2, 8, 8, 24), g)
    d64, db = rand_bf16((2, 8, 8, 24), g)
    xd = xb.to(dev)
    assert_close_bf16(K.lrelu(xd, 0.2), oops.lrelu(x64, 0.2), "lrelu")
    assert_close_bf16(K.lrelu_bwd(xd, db.to(dev), 0.1),
                      torch.where(x64 > 0, d64, 0.1 * d64), "lrelu bwd")
    assert_close_bf16(K.axpby(xd, 0.5, db.to(dev), 2.0), 0.5 * x64 + 2.0 * d64, "axpby")
    assert_close_bf16(K.avgpool2(xd), oops.avg_pool2(x64), "avgpool")
    assert_close_bf16(K.maxpool2(xd), oops.max_pool2(x64), "maxpool")
    p64, pb = rand_bf16((2, 4, 4, 24), g)
    xr = x64.clone().requires_grad_(True)
    (oops.avg_pool2(xr) * p64).sum().backward()
    assert_close_bf16(K.avgpool2_bwd(pb.to(dev)), xr.grad, "avgpool bwd")
    xr = x64.clone().requires_grad_(True)
    (oops.max_pool2(xr) * p64).sum().backward()
    assert_close_bf16(K.maxpool2_bwd(xd, pb.to(dev)), xr.grad, "maxpool bwd")
    # odd channel count takes the scalar path
    y64, yb = rand_bf16((1, 4, 4, 3), g)
    assert_close_bf16(K.avgpool2(yb.to(dev)), oops.avg_pool2(y64), "avgpool c=3")


def test_spatial_reduce_heads_rowdot(K, dev):
    g = _gen(17)
    x64, xb = rand_bf16((4, 8, 8, 128), g)
    xd = xb.to(dev)
    ref = torch.relu(x64).mean(dim=(1, 2))
    assert_close_bf16(K.spatial_reduce(xd, xd, 1.0 / 64), ref, "relu+mean")
    assert_close_bf16(K.spatial_reduce(xd, None, 1.0), x64.sum(dim=(1, 2)), "sum", abs_rms=2.0 ** -7)
    d64, db = rand_bf16((4, 128), g)
    xr = x64.clone().requires_grad_(True)
    (torch.relu(xr).mean(dim=(1, 2)) * d64).sum().backward()
    assert_close_bf16(K.spatial_reduce_bwd(xd, db.to(dev), (4, 8, 8, 128), 1.0 / 64), xr.grad,
                      "relu+mean bwd")
    # heads
    h = torch.randn((2, 4, 4, 3), generator=g, dtype=torch.float32)
    for kind, fn in ((0, torch.sigmoid), (1, lambda t: (torch.tanh(t) + 1.0) / 2.0)):
        hr = h.to(torch.float64).requires_grad_(True)
        out = fn(hr)
        yd = K.head(h.to(dev), kind)
        assert_close_f32(yd, out.detach(), "head %d" % kind, rtol=1e-5, abs_rms=1e-6)
        dyh = torch.randn(h.shape, generator=g, dtype=torch.float32)
        (out * dyh.to(torch.float64)).sum().backward()
        assert_close_bf16(K.head_bwd(yd, kind, dyh.to(dev)), hr.grad, "head bwd %d" % kind)
    # rowdot / one-hot / casts / colsum
    a64, ab = rand_bf16((16, 96), g)
    b64, bb = rand_bf16((16, 96), g)
    assert_close_f32(K.rowdot(ab.to(dev), bb.to(dev)), (a64 * b64).sum(1, keepdim=True), "rowdot")
    do = torch.randn((16, 1), generator=g, dtype=torch.float32)
    da, dbb = K.rowdot_bwd(ab.to(dev), bb.to(dev), do.to(dev))
    assert_close_bf16(da, do.to(torch.float64) * b64, "rowdot da")
    assert_close_bf16(dbb, do.to(torch.float64) * a64, "rowdot db")
    labels = torch.tensor([1, 0, 9, 3], dtype=torch.int32)
    oh = K.one_hot(labels.to(dev), 10).cpu().to(torch.float32)
    assert torch.equal(oh, F.one_hot(labels.long(), 10).float())
    f = torch.randn(1000, generator=g, dtype=torch.float32)
    assert torch.equal(K.cast_f32_to_bf16(f.to(dev)).cpu(), f.to(BF16))
    assert torch.equal(K.cast_bf16_to_f32(f.to(BF16).to(dev)).cpu(), f.to(BF16).to(torch.float32))
    assert torch.equal(K.cast_f32_to_bf16(f.to(dev), 2.0, -1.0).cpu(), (f * 2.0 - 1.0).to(BF16))
    c64, cb = rand_bf16((1000, 40), g)
    assert_close_f32(K.colsum(cb.to(dev)), c64.sum(0), "colsum", rtol=1e-5, abs_rms=1e-5)


@pytest.mark.parametrize("kind", ["non_saturating", "wasserstein", "least_squares", "hinge"])
def test_gan_losses(K, dev, kind):
    g = _gen(19)
    B = 48
    logits = torch.randn((2 * B, 1), generator=g, dtype=torch.float32) * 2
    lr = logits.to(torch.float64).requires_grad_(True)
    real, fake = lr[:B], lr[B:]
    d_loss, d_real, d_fake, g_loss = ogan.get_losses(kind, torch.sigmoid(real), torch.sigmoid(fake),
                                                     real, fake)
    losses, dd, dg = K.gan_loss(K.LOSS_KINDS[kind], logits.to(dev))
    ref = torch.stack([d_loss, d_real, d_fake, g_loss]).detach()
    assert_close_f32(losses, ref, kind + " losses", rtol=1e-5, abs_rms=1e-6)
    gd, = torch.autograd.grad(d_loss, lr, retain_graph=True)
    gg, = torch.autograd.grad(g_loss, lr)
    assert_close_f32(dd, gd.reshape(-1), kind + " d grads", rtol=1e-5, abs_rms=1e-6)
    assert_close_f32(dg, gg.reshape(-1), kind + " g grads", rtol=1e-5, abs_rms=1e-6)


def test_gradient_penalty(K, dev):
    g = _gen(23)
    B, per = 8, 3 * 16 * 16
    gr = (torch.randn((B, per), generator=g, dtype=torch.float32) * 0.05)
    g64 = gr.to(torch.float64).requires_grad_(True)
    slopes = torch.sqrt(0.0001 + (g64 ** 2).sum(1))
    pen = ((slopes - 1.0) ** 2).mean()
    sl, p = K.gradient_penalty(gr.to(dev))
    assert_close_f32(sl, slopes.detach(), "slopes", rtol=1e-5, abs_rms=1e-6)
    assert abs(float(p) - float(pen)) < 1e-5 * abs(float(pen))
    (10.0 * pen).backward()
    up = torch.tensor([10.0], device=dev)
    assert_close_bf16(K.gradient_penalty_bwd(gr.to(dev), sl, up), g64.grad, "gp bwd")
    x = torch.rand((B, 4, 4, 3), generator=g)
    xf = torch.rand((B, 4, 4, 3), generator=g)
    al = torch.rand((B,), generator=g)
    ref = x.double() + al.double().reshape(B, 1, 1, 1) * (xf.double() - x.double())
    assert_close_bf16(K.interpolate(x.to(dev), xf.to(dev), al.to(dev)), ref, "interpolate")


def test_adam_ema_multi(K, dev):
    g = _gen(29)
    shapes = [(3, 3, 16, 32), (32,), (100000,), (7,)]
    params = [torch.randn(s, generator=g, dtype=torch.float32) for s in shapes]
    p64 = [p.to(torch.float64).clone() for p in params]
    opt = ogan.TFAdam(p64, lr=2e-4, beta1=0.5, beta2=0.999)
    shadow64 = [p.clone() for p in p64]
    pd = [p.to(dev).clone() for p in params]
    gd = [torch.zeros_like(p) for p in pd]
    md = [torch.zeros_like(p) for p in pd]
    vd = [torch.zeros_like(p) for p in pd]
    ed = [p.clone() for p in pd]
    table = K.AdamTable(pd, gd, md, vd, ed)
    step = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(3):
        grads = [torch.randn(s, generator=g, dtype=torch.float32) for s in shapes]
        for gdst, gsrc in zip(gd, grads):
            gdst.copy_(gsrc)
        decay = 0.9 if it >= 1 else 0.0
        table.adam(2e-4, 0.5, 0.999, 1e-8, 1.0, step, ema_decay=0.9, ema_start=1)
        K.counter_add(step, 1)
        opt.step([gg.to(torch.float64) for gg in grads])
        ogan.ema_update(shadow64, p64, decay)
    assert int(step.item()) == 3
    for i in range(len(shapes)):
        assert_close_f32(pd[i], p64[i], "adam param %d" % i, rtol=1e-5, abs_rms=1e-6)
        assert_close_f32(ed[i], shadow64[i], "ema %d" % i, rtol=1e-5, abs_rms=1e-6)
    # gather / scatter of the gradient bucket
    flat = torch.empty(table.total_elems, device=dev)
    table.gather(flat)
    ref = torch.cat([t.reshape(-1) for t in gd])
    assert torch.equal(flat, ref)
    flat.mul_(0.5)
    table.scatter(flat)
    assert torch.equal(torch.cat([t.reshape(-1) for t in gd]), ref * 0.5)


def test_rng_matches_oracle_bitstream(K, dev):
    step = torch.tensor(5, dtype=torch.int64, device=dev)
    u = K.random(0, -1.0, 1.0, 547, 3, 1, step, (1001,), dev).cpu().numpy()
    ref = orng.uniform(1001, -1.0, 1.0, 547, 3, 1, 5)
    assert np.array_equal(u, ref)
    n = K.random(1, 0.0, 1.0, 42, 7, 0, None, (64, 120), dev).cpu().numpy().reshape(-1)
    refn = orng.normal(64 * 120, 0.0, 1.0, 42, 7, 0, 0)
    np.testing.assert_allclose(n, refn, rtol=2e-5, atol=2e-6)
    assert abs(n.mean()) < 0.05 and abs(n.std() - 1.0) < 0.05
    lab = K.random_labels(1000, 9, 2, 3, step, 4097, dev).cpu().numpy()
    assert np.array_equal(lab, orng.labels(4097, 1000, 9, 2, 3, 5))
    # semantics of tpu_random_test.py: same (op, step) -> same; other step / replica -> different
    u2 = K.random(0, -1.0, 1.0, 547, 3, 1, step, (1001,), dev).cpu().numpy()
    assert np.array_equal(u, u2)
    step2 = torch.tensor(6, dtype=torch.int64, device=dev)
    assert not np.array_equal(u, K.random(0, -1.0, 1.0, 547, 3, 1, step2, (1001,), dev).cpu().numpy())
    assert not np.array_equal(u, K.random(0, -1.0, 1.0, 547, 3, 2, step, (1001,), dev).cpu().numpy())


@pytest.mark.parametrize("dims", [(2, 256, 128, 12, 48), (1, 1024, 256, 24, 96),
                                  (2, 128, 128, 48, 128),      # MFMA path, Dk padded to 64
                                  (2, 96, 40, 12, 48)])         # ragged lengths: streaming fallback
def test_attention(K, dev, dims):
    _attention_case(K, dev, dims)


@pytest.mark.parametrize("nw8", ["0", "1"])
@pytest.mark.parametrize("dims", [(2, 512, 256, 12, 48), (1, 1024, 512, 24, 96), (1, 256, 256, 40, 128)])
def test_attention_workgroup_forms(K, dev, dims, nw8, monkeypatch):
    """4-wave (128 rows) and 8-wave (256 rows) workgroups of the MFMA kernels, forced either way
    (the policy picks 8 waves only on grids of >= 512 workgroups, i.e. at the benchmark sizes)."""
    monkeypatch.setenv("CGAMD_ATTN_NW8", nw8)
    _attention_case(K, dev, dims)


def _attention_case(K, dev, dims):
    B, Lq, Lk, Dk, Dv = dims
    g = _gen(31)
    t64, tb = rand_bf16((B, Lq, Dk), g, 0.5)
    p64, pb = rand_bf16((B, Lk, Dk), g, 0.5)
    g64, gb = rand_bf16((B, Lk, Dv), g)
    tr, pr, gr = [t.clone().requires_grad_(True) for t in (t64, p64, g64)]
    ref = torch.softmax(tr @ pr.transpose(1, 2), dim=-1) @ gr
    out, lse = K.attention_fwd(tb.to(dev), pb.to(dev), gb.to(dev))
    # the probabilities are rounded to bf16 before the second MFMA (as in every flash-style kernel):
    # 2^-7 * rms absolute slack on top of the bf16 output rounding
    assert_close_bf16(out, ref.detach(), "attention fwd", ulps=3.0, abs_rms=2.0 ** -7)
    do64, dob = rand_bf16((B, Lq, Dv), g)
    (ref * do64).sum().backward()
    dt, dp, dg = K.attention_bwd(tb.to(dev), pb.to(dev), gb.to(dev), out, lse, dob.to(dev))
    # ds = p * (dp - delta) cancels to ~1e-2 of its operands and delta is formed from the
    # bf16-rounded `out` (as in every flash-style backward) -> 2^-5 * rms absolute slack
    assert_close_bf16(dt, tr.grad, "dtheta", ulps=4.0, abs_rms=2.0 ** -5)
    assert_close_bf16(dp, pr.grad, "dphi", ulps=4.0, abs_rms=2.0 ** -5)
    assert_close_bf16(dg, gr.grad, "dg", ulps=4.0, abs_rms=2.0 ** -6)


def test_fid_statistics(K, dev):
    rng = np.random.RandomState(0)
    x = rng.randn(500, 96).astype(np.float32) * rng.rand(96).astype(np.float32) + 1.0
    mean, cov = K.mean_cov_f64(torch.from_numpy(x).to(dev))
    m_ref, c_ref = ofid.mean_cov(x)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cov.cpu().numpy(), c_ref, rtol=1e-10, atol=1e-12)
    a = rng.randn(70, 50)
    b = rng.randn(50, 33)
    c = K.gemm_f64(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    np.testing.assert_allclose(c, a @ b, rtol=1e-12, atol=1e-12)
    c2 = K.gemm_f64(torch.from_numpy(a.T.copy()).to(dev), torch.from_numpy(b.T.copy()).to(dev),
                    ta=True, tb=True).cpu().numpy()
    np.testing.assert_allclose(c2, a @ b, rtol=1e-12, atol=1e-12)
    # eigen-decomposition of a PSD matrix (rank-deficient like a covariance with n < d)
    s = c_ref.copy()
    w, v = K.syevj_f64(torch.from_numpy(s.copy()).to(dev))
    w, v = w.cpu().numpy(), v.cpu().numpy()
    np.testing.assert_allclose(np.sort(w), np.linalg.eigvalsh(s), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(v.T @ np.diag(w) @ v, s, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(v @ v.T, np.eye(96), atol=1e-10)
    logits = rng.randn(300, 1008).astype(np.float32) * 3
    sc = K.inception_score_f64(torch.from_numpy(logits).to(dev))
    assert abs(float(sc) - ofid.classifier_score_from_logits(logits)) < 1e-9 * float(sc)


@pytest.mark.parametrize("d,rank", [(256, 256), (512, 200), (96, 96), (288, 288)])
def test_syevj_block_form(K, dev, d, rank):
    """cg_syevj_f64 on matrices large enough for the block form of the Jacobi sweeps (d >= 256,
    d % 64 == 0; 96 and 288 take the scalar rounds): eigenvalues against numpy's eigvalsh, the
    reconstruction V^T diag(w) V and the orthonormality of V -- full-rank and rank-deficient PSD
    matrices (a covariance of n < d samples has d - n zero eigenvalues), and an indefinite one."""
    rng = np.random.RandomState(d + rank)
    x = rng.randn(rank, d) * (0.2 + rng.rand(d) * 3.0)
    a = x.T @ x / rank
    for name, mat in (("psd", a), ("indefinite", a - 0.5 * np.diag(rng.rand(d)) * np.trace(a) / d)):
        w, v = K.syevj_f64(torch.from_numpy(mat.copy()).to(dev), max_sweeps=60, tol=1e-12)
        w, v = w.cpu().numpy(), v.cpu().numpy()
        scale = np.abs(np.linalg.eigvalsh(mat)).max()
        np.testing.assert_allclose(np.sort(w), np.linalg.eigvalsh(mat), rtol=1e-9,
                                   atol=1e-11 * scale, err_msg=name)
        np.testing.assert_allclose(v.T @ np.diag(w) @ v, mat, rtol=1e-9, atol=1e-10 * scale,
                                   err_msg=name)
        np.testing.assert_allclose(v @ v.T, np.eye(d), atol=1e-10, err_msg=name)


@pytest.mark.parametrize("n,kind", [(2, "random"), (3, "random"), (37, "random"), (128, "psd"),
                                    (300, "graded"), (512, "rank"), (1000, "psd"), (2048, "graded"),
                                    (257, "diagonal"), (200, "tridiagonal")])
def test_tridiagonal_eigenvalues(K, dev, n, kind):
    """cg_sytrd_eigvals_f64 (Householder tridiagonalisation + Sturm bisection: the eigenvalues behind the
    trace of the second matrix square root, metrics/fid_score.py:58-75 via tfgan) against LAPACK
    (numpy.linalg.eigvalsh, fp64): the l2 norm of the eigenvalue errors within a QUARTER of 4 sqrt(n) u
    |A|_F and the largest one within HALF of a quarter of that (n >= 64) -- the |E|_F and |E|_2
    metrics/fid_score.py certifies its result with (measured: 1 ... 17 u |A|_F); ascending order; |A|_F
    returned; cg_spectral_sqrt_bound_f64 equals its definition and really bounds the error of the sum
    against the sum over LAPACK's eigenvalues."""
    u = 2.220446049250313e-16
    rng = np.random.RandomState(n)
    if kind == "random":
        a = rng.randn(n, n)
        a = a + a.T
    elif kind == "psd":
        x = rng.randn(3 * n, n) * (rng.rand(n) * 2)
        a = x.T @ x / (3 * n - 1)
    elif kind == "graded":       # spectrum over 14 decades, as the covariances of dead-ish features
        q, _ = np.linalg.qr(rng.randn(n, n))
        a = (q * np.logspace(-15, -1, n)) @ q.T
        a = 0.5 * (a + a.T)
    elif kind == "rank":         # fewer samples than features
        x = rng.randn(n // 4, n)
        x -= x.mean(axis=0)
        a = x.T @ x / (n // 4 - 1)
    elif kind == "diagonal":
        a = np.diag(rng.randn(n))
    else:
        a = np.diag(rng.randn(n)) + np.diag(rng.randn(n - 1), 1)
        a = a + np.triu(a, 1).T
    want = np.linalg.eigvalsh(a)
    fro_want = np.linalg.norm(a)
    w, fro = K.sytrd_eigvals_f64(torch.from_numpy(a).to(dev))
    got = w.cpu().numpy()
    assert abs(float(fro.item()) - fro_want) <= 1e-12 * fro_want
    assert np.all(np.diff(got) >= 0)
    delta_rel = 4.0 * np.sqrt(n) * u
    delta2_rel = delta_rel / (4.0 if n >= 64 else 1.0)
    delta, delta2 = delta_rel * fro_want, delta2_rel * fro_want
    err = np.abs(got - want).max()                     # what Weyl bounds by |E|_2
    err_f = np.sqrt(((got - want) ** 2).sum())         # what Hoffman-Wielandt bounds by |E|_F
    assert err <= 0.5 * delta2 and err_f <= 0.25 * delta, (kind, n, err / (u * fro_want), err_f / (u * fro_want))
    eps = 1e-10
    f = lambda s: np.where(s < eps, s, np.sqrt(s))
    out = K.spectral_sqrt_bound_f64(w, eps, delta_rel, delta2_rel, fro).cpu().numpy()
    ag = np.abs(got)
    assert abs(out[0] - f(ag).sum()) <= 1e-12 * max(1.0, f(ag).sum())
    above, below = ag - delta2 >= eps, ag + delta2 < eps
    g2 = (1.0 / (4.0 * (ag[above] - delta2))).sum() + below.sum()
    bound = delta * np.sqrt(g2) + (~above & ~below).sum() * np.sqrt(eps + 2 * delta2)
    assert abs(out[1] - bound) <= 1e-9 * bound
    assert abs(f(ag).sum() - f(np.abs(want)).sum()) <= out[1]
    print("tridiagonal eigenvalues n %d %s: max error %.1f u|A|_F, l2 error %.1f u|A|_F (assumed |E|_F %.0f u|A|_F), "
          "sum error %.2e <= bound %.2e" % (n, kind, err / (u * fro_want), err_f / (u * fro_want), delta_rel / u,
                                          abs(f(ag).sum() - f(np.abs(want)).sum()), out[1]))


def test_inception_preprocess_and_pool(K, dev):
    rng = np.random.RandomState(1)
    img = (rng.rand(2, 32, 32, 3) * 255).astype(np.float32)
    ref = ofid.inception_preprocess(img, 299)
    y = K.inception_preprocess(torch.from_numpy(img).to(dev))
    assert_close_bf16(y, torch.from_numpy(ref), "inception preprocess", ulps=2.0, abs_rms=2.0 ** -9)
    g = _gen(37)
    x64, xb = rand_bf16((2, 9, 9, 16), g)
    ref_max = F.max_pool2d(x64.permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    assert_close_bf16(K.pool2d(xb.to(dev), 3, 2, 0, 0, 4, 4), ref_max, "maxpool 3x3/2 valid")
    ref_avg = F.avg_pool2d(x64.permute(0, 3, 1, 2), 3, 1, 1, count_include_pad=False).permute(0, 2, 3, 1)
    assert_close_bf16(K.pool2d(xb.to(dev), 3, 1, 1, 1, 9, 9), ref_avg, "avgpool 3x3/1 same")
    # the same poolings on channel slices, finished by bias + ReLU (cg_pool2d_ld)
    wide64, wide = rand_bf16((3, 17, 17, 8 + 24 + 16), g)
    bias = torch.randn((24,), generator=g, dtype=torch.float32)
    xs64 = wide64[..., 8:32]
    out = torch.full((3, 17, 17, 16 + 24 + 8), 5.0, dtype=BF16, device=dev)
    K.pool2d_ld(wide.to(dev)[..., 8:32], 3, 1, 1, 1, 17, 17, out[..., 16:40], bias=bias.to(dev), relu=True)
    ref = F.avg_pool2d(xs64.permute(0, 3, 1, 2), 3, 1, 1, count_include_pad=False).permute(0, 2, 3, 1)
    assert_close_bf16(out[..., 16:40], (ref + bias.double()).clamp(min=0.0), "avgpool slices + bias + relu")
    assert bool((out[..., :16] == 5.0).all()) and bool((out[..., 40:] == 5.0).all())
    out2 = torch.zeros((3, 8, 8, 32), dtype=BF16, device=dev)
    K.pool2d_ld(wide.to(dev)[..., 8:32], 3, 2, 0, 0, 8, 8, out2[..., 8:])
    ref_max = F.max_pool2d(xs64.permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    assert_close_bf16(out2[..., 8:], ref_max, "maxpool slices")
    assert torch.equal(out2[..., 8:].contiguous(),
                       K.pool2d(wide.to(dev)[..., 8:32].contiguous(), 3, 2, 0, 0, 8, 8))


@pytest.mark.parametrize("case", [
    # N, H, W, Ci, Co, kh, kw, stride, padding, in_extra (before, after), out_extra (before, after)
    (3, 17, 17, 64, 96, 1, 1, 1, "SAME", (32, 8), (64, 16)),       # sibling 1x1 head read from a slice
    (2, 17, 17, 160, 192, 1, 7, 1, "SAME", (0, 32), (192, 384)),   # 1x7 into the middle of a block output
    (2, 17, 17, 128, 128, 7, 1, 1, "SAME", (64, 0), (0, 8)),
    (2, 35, 35, 96, 96, 3, 3, 2, "VALID", (0, 0), (384, 288)),     # reduction block: stride 2, VALID
    (4, 8, 8, 384, 384, 3, 1, 1, "SAME", (448, 0), (320, 384)),    # mixed_9 split branch
    (130, 8, 8, 64, 72, 1, 1, 1, "SAME", (0, 0), (8, 0)),          # > one workgroup per CU; Co % 8 only
])
def test_gconv_on_channel_slices(K, dev, case):
    """cg_gconv_ld (the concatenations of the Inception graph behind eval_utils.py:165-175 written in
    place): conv + bias + ReLU reading a channel slice of a wider tensor and writing into a channel
    slice of another equals -- bit for bit, same kernel arithmetic -- cg_gconv on dense copies, and
    leaves the neighbouring channels of the output untouched; against the fp64 oracle within the
    bf16 tolerance of the file header."""
    N, H, W, Ci, Co, kh, kw, stride, padding, (ib, ia), (ob, oa) = case
    g = _gen(sum(case[:8]))
    wide64, wide = rand_bf16((N, H, W, ib + Ci + ia), g)
    w64, wb = rand_bf16((kh, kw, Ci, Co), g, scale=1.0 / math.sqrt(kh * kw * Ci))
    bias = torch.randn((Co,), generator=g, dtype=torch.float32)
    if padding == "SAME":
        geom = K.geom_conv_same(N, H, W, Ci, Co, kh, kw, stride, 1)
    else:
        geom = K.make_geom(N, H, W, Ci, (H - kh) // stride + 1, (W - kw) // stride + 1, Co, kh, kw, stride, 1, 0, 0)
    assert K.gconv_ld_supported(geom, ib + Ci + ia, ob + Co + oa)
    bt = K.weight_prep(wb.float().to(dev), want_fwd=True)[0]
    wide_d = wide.to(dev)
    x_view = wide_d[..., ib:ib + Ci]
    out_wide = torch.full((N, geom.Ho, geom.Wo, ob + Co + oa), 7.0, dtype=BF16, device=dev)
    got = K.gconv_ld(geom, x_view, bt, out_wide[..., ob:ob + Co], bias=bias.to(dev), relu=True)
    dense = K.gconv(geom, x_view.contiguous(), bt, bias=bias.to(dev), act_out=0.0)
    assert torch.equal(got, dense)
    assert bool((out_wide[..., :ob] == 7.0).all()) and bool((out_wide[..., ob + Co:] == 7.0).all())
    x64 = wide64[..., ib:ib + Ci]
    xp = x64.permute(0, 3, 1, 2)
    if padding == "SAME":
        xp = F.pad(xp, (geom.pl, kw - 1 - geom.pl, geom.pt, kh - 1 - geom.pt))
    ref = F.conv2d(xp, w64.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1) + bias.double()
    assert_close_bf16(got, ref.clamp(min=0.0), "gconv_ld %s" % (case,))
    # fp32 output, no ReLU (the logits layer's form)
    out32 = torch.zeros((N, geom.Ho, geom.Wo, Co + 8), dtype=torch.float32, device=dev)
    K.gconv_ld(geom, x_view, bt, out32[..., 8:], bias=None, relu=False)
    assert_close_f32(out32[..., 8:], ref - bias.double(), "gconv_ld fp32 %s" % (case,))
    # ReLU on the leading output channels only (a merged convolution whose trailing columns another
    # kernel finishes: cg_pool2d_ld adds bias and ReLU behind the pooling)
    if Co >= 16:
        rc = (Co // 2) // 8 * 8
        part = torch.zeros((N, geom.Ho, geom.Wo, Co), dtype=BF16, device=dev)
        K.gconv_ld(geom, x_view, bt, part, bias=bias.to(dev), relu=rc)
        want = torch.cat([ref[..., :rc].clamp(min=0.0), ref[..., rc:]], dim=-1)
        assert_close_bf16(part, want, "gconv_ld relu_cols %s" % (case,))
        assert torch.equal(part[..., :rc], got[..., :rc])
    # what is not a slice is refused
    with pytest.raises(ValueError):
        K.gconv_ld(geom, x_view.transpose(1, 2), bt, out_wide[..., ob:ob + Co])   # H == W: shape fits, pitch not


@pytest.mark.parametrize("case", [
    # N, H, W, Ci, Co, kh, kw, padding          what it exercises in cg_conv_fast.hip
    (64, 35, 35, 96, 96, 3, 3, "SAME"),        # two-deep ring, half-empty last K block
    (100, 35, 35, 64, 96, 3, 3, "SAME"),       # single buffer, 96 of 128 columns
    (128, 17, 17, 160, 160, 1, 7, "SAME"),     # 128-wide tiles (grid too small for the wide ones), K 160
    (256, 17, 17, 160, 192, 7, 1, "SAME"),     # 192-wide tile, two-deep ring, K 160
    (400, 17, 17, 192, 192, 1, 1, "SAME"),     # 192-wide tile, single buffer
    (400, 17, 17, 192, 160, 1, 1, "SAME"),     # 192-wide tile with 32 empty columns
    (8, 73, 73, 96, 192, 3, 3, "VALID"),       # K 96 on the 128-wide tiles
    (16, 147, 147, 32, 64, 3, 3, "SAME"),      # 64-wide tile, every K block half empty
    (16, 149, 149, 32, 32, 3, 3, "VALID"),     # 32-wide tile, the same
    (4, 8, 8, 160, 128, 3, 3, "SAME"),         # 8-wave split-K form, K 160 (slices alternate between groups)
    (4, 8, 8, 96, 128, 3, 3, "SAME"),          # ... and K 96
])
def test_fast_conv_ragged_channel_counts(K, dev, case):
    """The one-tap MFMA kernel on channel counts that are odd multiples of 32 (the Inception graph of
    eval_utils.py:165-175: 32 / 96 / 160 / 288 inputs, 96 / 160 / 192 outputs; BigGAN ch = 96): the
    upper half of a half-empty last 64-channel K block is skipped, and 192-wide output tiles
    replace quarter-empty 128-wide ones.  conv + bias + ReLU against the fp64 oracle, tolerance of the
    file header."""
    N, H, W, Ci, Co, kh, kw, padding = case
    g = _gen(sum(case[:7]))
    x64, xb = rand_bf16((N, H, W, Ci), g)
    w64, wb = rand_bf16((kh, kw, Ci, Co), g, scale=1.0 / math.sqrt(kh * kw * Ci))
    bias = torch.randn((Co,), generator=g, dtype=torch.float32)
    if padding == "SAME":
        geom = K.geom_conv_same(N, H, W, Ci, Co, kh, kw, 1, 1)
    else:
        geom = K.make_geom(N, H, W, Ci, H - kh + 1, W - kw + 1, Co, kh, kw, 1, 1, 0, 0)
    bt = K.weight_prep(wb.float().to(dev), want_fwd=True)[0]
    got = K.gconv(geom, xb.to(dev), bt, bias=bias.to(dev), act_out=0.0)
    xp = x64.permute(0, 3, 1, 2)
    if padding == "SAME":
        xp = F.pad(xp, (geom.pl, kw - 1 - geom.pl, geom.pt, kh - 1 - geom.pt))
    ref = F.conv2d(xp.float(), w64.permute(3, 2, 0, 1).float()).double().permute(0, 2, 3, 1) + bias.double()
    # (fp32 reference convolution of bf16-exact operands: its own error, ~1e-6 relative, is far inside
    # the bf16 output tolerance; the fp64 one takes minutes at these sizes on the host)
    assert_close_bf16(got, ref.clamp(min=0.0), "fast conv %s" % (case,))


def test_error_codes(K, dev):
    from compare_gan_amd.hip._lib import CgamdError
    x = torch.zeros((1, 4, 4, 8), dtype=BF16, device=dev)
    with pytest.raises(ValueError):
        K.gconv(K.make_geom(1, 4, 4, 8, 4, 4, 8, 3, 3, 1, 1, 1, 1), x.cpu(), x)
    geom = K.make_geom(1, 4, 4, 8, 4, 4, 8, 3, 3, 1, 3, 1, 1)  # U=3 is not a power of two
    bt = torch.zeros((8, 72), dtype=BF16, device=dev)
    with pytest.raises(CgamdError):
        K.gconv(geom, x, bt)
    with pytest.raises(CgamdError):
        K.avgpool2(torch.zeros((1, 3, 4, 8), dtype=BF16, device=dev))


# The convolution dispatcher picks a kernel variant from the grid size and a few environment
# overrides that are read once per process.  Every variant must stay parity-green, including the ones
# the default policy does not select for the small test shapes: run the convolution cases again in a
# child process per override.
CONV_VARIANT_ENVS = [
    ("no_splitk", {"CGAMD_CONV_SK": "0"}),              # 4-wave kernels on small grids
    ("ring2", {"CGAMD_CONV_SK": "0", "CGAMD_CONV_NS": "2"}),
    ("single_buffer", {"CGAMD_CONV_SK": "0", "CGAMD_CONV_NS": "1"}),
    ("tiles128", {"CGAMD_CONV_SK": "0", "CGAMD_CONV_T128_MIN": "1"}),   # 128x128 tiles everywhere
    ("splitk_tiles128", {"CGAMD_CONV_T128_MIN": "1"}),
    # groups of 3 M tiles walk the N tiles together (the order of weight-heavy layers; ragged last group)
    ("tile_groups", {"CGAMD_CONV_SK": "0", "CGAMD_CONV_T128_MIN": "1", "CGAMD_CONV_GM": "3"}),
    ("halo_forward", {"CGAMD_HALO": "1"}),             # experimental halo-staged forward kernel
    ("one_tap_wgrad", {"CGAMD_NO_HALO_WGRAD": "1"}),   # one-tap-per-workgroup weight gradient
    # halo-staged forward / weight-gradient kernels wherever they apply
    ("hconv_all", {"CGAMD_HCONV_MIN": "1", "CGAMD_HWGRAD_MIN": "1", "CGAMD_HCONV_RW_MIN": "1",
                   "CGAMD_HUP": "2"}),
    ("no_hup", {"CGAMD_HCONV_MIN": "1", "CGAMD_HWGRAD_MIN": "1", "CGAMD_HCONV_RW_MIN": "1",
                "CGAMD_HUP": "0"}),
    ("no_hconv", {"CGAMD_HCONV": "0", "CGAMD_HWGRAD": "0", "CGAMD_WSTEM": "0", "CGAMD_HCONV_RW": "0"}),
    # small-map kernels (cg_conv_small.hip) wherever their geometry fits / nowhere
    ("small_all", {"CGAMD_SCONV": "2", "CGAMD_SWGRAD": "2"}),
    ("no_small", {"CGAMD_SCONV": "0", "CGAMD_SWGRAD": "0"}),
    # 512 weight-gradient workgroups (the policy of the large-batch layers: hwgrad_plan)
    ("hwgrad_512_blocks", {"CGAMD_HWGRAD_BLOCKS": "512", "CGAMD_HWGRAD_MIN": "1"}),
    # 128-wide tiles where the default takes 192-wide ones (Inception's 160 / 192-channel layers)
    ("no_wide_tiles", {"CGAMD_CONV_BN_WIDE": "0"},
     "test_fast_conv_ragged_channel_counts or test_gconv_on_channel_slices"),
]
_VARIANT_DEFAULT_CASES = (
    "test_gconv_forward_adjoint_wgrad or test_gconv_gates_residual or "
    "test_stem_relu_gate or (test_conv_pool_fused and not full_size) or "
    "(test_gconv_fused_batch_norm and not full_size) or "
    "test_gconv_fused_statistics_groups")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", CONV_VARIANT_ENVS, ids=[v[0] for v in CONV_VARIANT_ENVS])
def test_conv_kernel_variants(dev, variant):
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(variant[1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = variant[2] if len(variant) > 2 else _VARIANT_DEFAULT_CASES
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q",
                        "-x", "-k", cases],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "variant %s:\n%s\n%s" % (variant[0], r.stdout[-3000:], r.stderr[-1000:])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 4, 4, 64), (3, 5, 7, 24), (1, 8, 8, 3)],
                         ids=["vec", "ragged_vec", "scalar"])
def test_unpool_standalone(K, dev, shape):
    """cg_unpool2 / cg_unpool2_bwd (resnet_ops.py:35-56 outside a convolution): bit-exact copy into
    the even pixels, residual added with one bf16 rounding; autograd of the Function pair."""
    from compare_gan_amd.hip import functional as Fn
    g = _gen(sum(shape))
    x_ref, x = rand_bf16(shape, g)
    n, h, w, c = shape
    r_ref, r = rand_bf16((n, 2 * h, 2 * w, c), g)
    up = K.unpool2(x.to(dev))
    assert torch.equal(up.cpu().double(), oops.unpool(x_ref))
    up_r = K.unpool2(x.to(dev), r.to(dev))
    want = (oops.unpool(x_ref) + r_ref).float().to(torch.bfloat16).double()
    assert torch.equal(up_r.cpu().double(), want)
    dy_ref, dy = rand_bf16((n, 2 * h, 2 * w, c), g)
    assert torch.equal(K.unpool2_bwd(dy.to(dev)).cpu().double(), dy_ref[:, ::2, ::2, :])
    xd = x.to(dev).requires_grad_(True)
    rd = r.to(dev).requires_grad_(True)
    out = Fn.unpool2(xd, rd)
    gx, gr = torch.autograd.grad(out, [xd, rd], grad_outputs=dy.to(dev))
    assert torch.equal(gx.cpu().double(), dy_ref[:, ::2, ::2, :])
    assert torch.equal(gr.cpu().double(), dy_ref)


def _bf(t):
    return t.float().to(torch.bfloat16).double()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 4])
def test_fork_n_sums_gradients_in_one_launch(K, dev, n):
    """Fn.fork_n / cg_sum4: the gradient of a tensor with three or four consumers (the input of the
    self-attention block, arch_ops.py:709-758) = the fp32 sum of the contributions, rounded once."""
    from compare_gan_amd.hip import functional as Fn
    g = _gen(40 + n)
    x = torch.randn((2, 5, 7, 24), generator=g).to(BF16).to(dev).requires_grad_(True)
    ws = [torch.randn((2, 5, 7, 24), generator=g).to(BF16).to(dev) for _ in range(n)]
    parts = Fn.fork_n(x, n)
    assert len(parts) == n
    loss = sum((p.float() * w.float()).sum() for p, w in zip(parts, ws))
    (gx,) = torch.autograd.grad(loss, [x])
    ref = sum(w.double() for w in ws)
    assert gx.dtype == BF16
    assert_close_bf16(gx, ref.cpu(), "fork_n gradient")
    # direct entry point, ragged tail (not a multiple of 8 elements)
    a, b, c = (torch.randn(1003, generator=g).to(BF16).to(dev) for _ in range(3))
    out = K.sum4(a, b, c)
    assert_close_bf16(out, (a.double() + b.double() + c.double()).cpu(), "sum4 of three")


@pytest.mark.parametrize("N,HW,C,mean", [(128, 64, 128, True), (16, 16, 512, True), (6, 16, 1536, False),
                                         (3, 5, 24, False), (2, 4, 2056, True)],
                         ids=["cifar", "resnet5", "biggan", "small_ragged", "wide"])
def test_pooled_head(K, dev, N, HW, C, mean):
    """cg_pooled_head_fwd / _bwd (relu -> reduce_mean / reduce_sum over [1, 2] -> linear(C -> 1),
    resnet_cifar.py:154-157 + arch_ops.py:538-556, in one launch per direction) against the fp64
    restatement with the bf16 rounding points of the separate launches: pooled, d pooled and dx are
    bf16 tensors; an extra gradient through `pooled` (projection discriminators) is added; then the
    autograd Function against the separate Functions it replaces (same roundings: bf16 noise only)."""
    from compare_gan_amd.hip import functional as Fn
    g = _gen(N + C)
    x64, x = rand_bf16((N, HW, C), g)
    w = (torch.randn(C, generator=g) * 0.1).float()
    b = torch.randn(1, generator=g).float()
    dl = torch.randn(N, generator=g).float()
    _, ext = rand_bf16((N, C), g, scale=0.05)
    scale = 1.0 / HW if mean else 1.0
    pooled_ref = _bf(scale * torch.relu(x64).sum(1))
    w16 = _bf(w)
    logit_ref = pooled_ref @ w16 + b.double()
    logit, pooled = K.pooled_head_fwd(x.to(dev), w.to(dev), b.to(dev), scale)
    # pooled: one bf16 rounding of an fp32 sum -> at most one ulp from the fp64 reference's rounding
    assert_close_bf16(pooled, pooled_ref, "pooled", ulps=1.01, abs_rms=1e-6)
    p_dev = pooled.cpu().double()
    assert_close_f32(logit, (p_dev @ w16 + b.double()).reshape(N, 1), "logit", 1e-5, 1e-5)
    for use_ext in (False, True):
        dl16 = _bf(dl)
        dp = _bf(dl16[:, None] * w16[None, :])
        if use_ext:
            dp = _bf(dp + ext.double())
        dx_ref = torch.where(x64 > 0, _bf(scale * dp)[:, None, :].expand(N, HW, C), torch.zeros((), dtype=torch.float64))
        dw_ref = (p_dev * dl16[:, None]).sum(0)
        db_ref = dl16.sum().reshape(1)
        dx, dw, db = K.pooled_head_bwd(x.to(dev), w.to(dev), scale, pooled, dlogit=dl.to(dev),
                                       dpooled=ext.to(dev) if use_ext else None)
        assert torch.equal(dx.cpu().double(), dx_ref), "dx (ext %s)" % use_ext
        assert_close_f32(dw, dw_ref, "dw", 1e-5, 1e-6)
        assert_close_f32(db, db_ref, "dbias", 1e-5, 1e-6)
    # only through pooled (no logit gradient), no parameter gradients
    dx, dw, db = K.pooled_head_bwd(x.to(dev), w.to(dev), scale, pooled, dlogit=None, dpooled=ext.to(dev))
    assert dw is None and db is None
    dx_ref = torch.where(x64 > 0, _bf(scale * ext.double())[:, None, :].expand(N, HW, C), torch.zeros((), dtype=torch.float64))
    assert torch.equal(dx.cpu().double(), dx_ref)
    # the autograd Function against the separate Functions
    side = int(round(math.sqrt(HW)))
    shape = (N, side, side, C) if side * side == HW else (N, HW, 1, C)
    xs = [x.to(dev).reshape(shape).clone().requires_grad_(True) for _ in range(2)]
    ws = [w.to(dev).reshape(C, 1).clone().requires_grad_(True) for _ in range(2)]
    bs = [b.to(dev).clone().requires_grad_(True) for _ in range(2)]
    lo_f, po_f = Fn.PooledHeadFn.apply(xs[0], ws[0], bs[0], scale)
    po_u = Fn.SpatialReduceFn.apply(xs[1], xs[1].detach(), scale)
    geom = K.make_geom(N, 1, 1, C, 1, 1, 1, 1, 1)
    lo_u = Fn.gconv(po_u.reshape(N, 1, 1, C), ws[1].reshape(1, 1, C, 1), bs[1], None, None, None,
                    Fn.ConvSpec(geom, transpose=False, slope_in=None, out_f32=True), False, None).reshape(N, 1)
    assert torch.equal(po_f, po_u) or float((po_f.float() - po_u.float()).abs().max()) <= 2.0 ** -7 * float(po_u.float().abs().max())
    assert_close_f32(lo_f, lo_u.double().cpu(), "logit vs separate", 2e-3, 2e-3)
    gl = dl.to(dev).reshape(N, 1)
    gp = ext.to(dev)
    torch.autograd.backward([lo_f, po_f], [gl, gp])
    torch.autograd.backward([lo_u, po_u], [gl, gp])
    assert U_cos(xs[0].grad, xs[1].grad) >= 0.9999
    assert_close_f32(ws[0].grad, ws[1].grad.double().cpu(), "dw vs separate", 2e-2, 2e-3)
    assert_close_f32(bs[0].grad, bs[1].grad.double().cpu(), "db vs separate", 1e-3, 1e-4)


def U_cos(a, b):
    a, b = a.double().reshape(-1).cpu(), b.double().reshape(-1).cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.gpu
def test_sigma_folded_into_the_attention_projection(K, dev):
    """arch_ops.py:755-758 `x + sigma * conv1x1(attn_g, w)` as ONE convolution (sigma folded into the
    kernel by Fn.ScaleWeightFn, x as the epilogue's residual) against the separate form
    (convolution, then Fn.ScaledResidualFn): outputs and the gradients w.r.t. x, attn_g, w and sigma,
    at sigma = 0 (the initial value: the output must be x exactly) and at sigma != 0.  The two forms
    round at different points (sigma * w to bf16 vs the convolution's output to bf16): bf16 noise."""
    from compare_gan_amd.hip import functional as Fn
    g = _gen(77)
    N, H, W, Cg, C = 3, 16, 16, 32, 64
    _, x = rand_bf16((N, H, W, C), g)
    _, a = rand_bf16((N, H, W, Cg), g)
    w0 = (torch.randn(1, 1, Cg, C, generator=g) * 0.2).float()
    _, dy = rand_bf16((N, H, W, C), g)
    geom = K.geom_conv_same(N, H, W, Cg, C, 1, 1, 1, 1)
    spec = Fn.ConvSpec(geom, transpose=False, slope_in=None, out_f32=False)
    # ScaleWeightFn on its own: w_eff = sigma * w, d w = sigma * g, d sigma = <g, w>
    wv = w0.to(dev).clone().requires_grad_(True)
    sv = torch.tensor(0.37, device=dev, requires_grad=True)
    weff = Fn.ScaleWeightFn.apply(wv, sv)
    gw = torch.randn(w0.shape, generator=g).float().to(dev)
    weff.backward(gw)
    assert_close_f32(weff, (w0.double() * 0.37), "w_eff", 1e-6, 1e-7)
    assert_close_f32(wv.grad, gw.double().cpu() * 0.37, "d w", 1e-6, 1e-7)
    assert_close_f32(sv.grad.reshape(1), (gw.double().cpu() * w0.double()).sum().reshape(1), "d sigma", 1e-5, 1e-6)
    for sigma in (0.0, 0.6):
        outs = []
        for folded in (True, False):
            xs = x.to(dev).clone().requires_grad_(True)
            as_ = a.to(dev).clone().requires_grad_(True)
            ws = w0.to(dev).clone().requires_grad_(True)
            ss = torch.tensor(sigma, device=dev, requires_grad=True)
            if folded:
                y = Fn.gconv(as_, Fn.ScaleWeightFn.apply(ws, ss), None, xs, None, None, spec, False, None)
            else:
                o = Fn.gconv(as_, ws, None, None, None, None, spec, False, None)
                y = Fn.ScaledResidualFn.apply(xs, o, ss.reshape(1))
            y.backward(dy.to(dev))
            outs.append((y.detach(), xs.grad, as_.grad, ws.grad, ss.grad))
        (yf, dxf, daf, dwf, dsf), (yu, dxu, dau, dwu, dsu) = outs
        if sigma == 0.0:
            assert torch.equal(yf, x.to(dev)) and torch.equal(yu, x.to(dev))
            assert float(daf.float().abs().max()) == 0.0 and float(dwf.abs().max()) == 0.0
        assert U_cos(yf, yu) >= 0.99999 and float((yf.float() - yu.float()).abs().max()) <= 0.05
        assert torch.equal(dxf, dxu)                       # dy passes through both forms unchanged
        if sigma != 0.0:
            assert U_cos(daf, dau) >= 0.9999 and U_cos(dwf, dwu) >= 0.9999
        assert abs(float(dsf) - float(dsu)) <= 5e-3 * max(1.0, abs(float(dsu))), (float(dsf), float(dsu))
