"""Shared helpers for the parity tests (oracle = checker, HIP path = thing under test)."""
import torch

BF16 = torch.bfloat16


def bf16_round(t):
    """fp64/fp32 tensor rounded to bf16 and back (what the HIP path stores)."""
    return t.to(torch.float32).to(BF16).to(torch.float64)


def rand_bf16(shape, gen, scale=1.0):
    """(fp64 CPU tensor holding bf16-representable values, same values as a bf16 tensor)."""
    x = (torch.randn(shape, generator=gen, dtype=torch.float32) * scale).to(BF16)
    return x.to(torch.float64), x


def assert_close_bf16(got, ref, what, ulps=2.0, abs_rms=2.0 ** -8):
    """got: device tensor (bf16 or fp32); ref: fp64 CPU tensor.  Tolerance: `ulps` bf16 ulps
    relative (2^-8 each) + abs_rms * rms(ref) absolute (accumulation-order noise near zero)."""
    g = got.detach().to("cpu").to(torch.float64).reshape(ref.shape)
    rms = float(ref.pow(2).mean().sqrt()) + 1e-30
    err = (g - ref).abs()
    bound = ulps * 2.0 ** -8 * ref.abs() + abs_rms * rms
    bad = err > bound
    if bad.any():
        idx = int(torch.argmax((err - bound).reshape(-1)))
        raise AssertionError(
            "%s: %d/%d elements out of tolerance; worst at flat %d: got %.6g ref %.6g (rms %.3g)"
            % (what, int(bad.sum()), bad.numel(), idx, float(g.reshape(-1)[idx]),
               float(ref.reshape(-1)[idx]), rms))


def assert_close_f32(got, ref, what, rtol=1e-4, abs_rms=1e-4):
    g = got.detach().to("cpu").to(torch.float64).reshape(ref.shape)
    rms = float(ref.pow(2).mean().sqrt()) + 1e-30
    err = (g - ref).abs()
    bound = rtol * ref.abs() + abs_rms * rms
    bad = err > bound
    if bad.any():
        idx = int(torch.argmax((err - bound).reshape(-1)))
        raise AssertionError(
            "%s: %d/%d elements out of tolerance; worst at flat %d: got %.8g ref %.8g (rms %.3g)"
            % (what, int(bad.sum()), bad.numel(), idx, float(g.reshape(-1)[idx]),
               float(ref.reshape(-1)[idx]), rms))
