"""A SECOND, independent restatement of the arithmetic whose parity cannot be pinned to a
reference-held vector (the reference's tests assert no values for them and
TF1 cannot be installed): TF 'SAME' convolution, its transpose and the TF form of Adam, written as
direct NumPy loops straight from the TF1 contract -- no torch, no F.conv2d, nothing shared with
oracle/arch_ops.py or oracle/gan.py.  tests/test_oracle_direct.py asserts that the two
restatements agree (DCGAN's 5x5 / stride-2 geometry among the cases,
compare_gan/architectures/dcgan.py:109-122): a slip in either one shows up as a disagreement.
Test infrastructure only -- nothing under compare_gan_amd/ imports this package.

References: tf.nn.conv2d / conv2d_transpose with padding='SAME' (arch_ops.py:559-592; SURVEY App.
A.1: out = ceil(in / stride), pad_total = max((out - 1) * stride + k - in, 0), pad_before =
pad_total // 2 -- the extra pixel goes AFTER); tf.train.AdamOptimizer (modular_gan.py:480-483;
SURVEY App. A.5: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), theta -= lr_t * m / (sqrt(v) + eps)).
"""
import math

import numpy as np


def same_geometry(size, k, stride):
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2


def conv2d_same(x, w, stride):
    """x [N,H,W,Ci], w [kh,kw,Ci,Co] -> [N,ceil(H/s),ceil(W/s),Co]; plain loops over the output."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    ho, pt = same_geometry(h, kh, stride)
    wo, pl = same_geometry(wd, kw, stride)
    out = np.zeros((n, ho, wo, co))
    for b in range(n):
        for oy in range(ho):
            for ox in range(wo):
                acc = np.zeros(co)
                for r in range(kh):
                    iy = oy * stride - pt + r
                    if iy < 0 or iy >= h:
                        continue
                    for s in range(kw):
                        ix = ox * stride - pl + s
                        if ix < 0 or ix >= wd:
                            continue
                        acc += x[b, iy, ix, :] @ w[r, s]
                out[b, oy, ox, :] = acc
    return out


def conv2d_transpose_same(x, w, out_hw, stride):
    """tf.nn.conv2d_transpose(x, w, output_shape, strides, 'SAME'): the gradient of conv2d_same
    w.r.t. its input, scattered output pixel by output pixel.  x [N,h,w,Cin] lives in the OUTPUT
    space of the forward convolution whose input has spatial size out_hw; w [kh,kw,Cout,Cin]."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    n, h, wd, _ = x.shape
    kh, kw, co, _ = w.shape
    hy, wy = out_hw
    ho, pt = same_geometry(hy, kh, stride)
    wo, pl = same_geometry(wy, kw, stride)
    assert (ho, wo) == (h, wd), "x is not the output of a SAME convolution of a %dx%d map" % out_hw
    out = np.zeros((n, hy, wy, co))
    for b in range(n):
        for oy in range(h):
            for ox in range(wd):
                for r in range(kh):
                    iy = oy * stride - pt + r
                    if iy < 0 or iy >= hy:
                        continue
                    for s in range(kw):
                        ix = ox * stride - pl + s
                        if ix < 0 or ix >= wy:
                            continue
                        out[b, iy, ix, :] += w[r, s] @ x[b, oy, ox, :]
    return out


def tf_adam(theta, grads, lr, beta1, beta2, eps, steps):
    """`steps` updates of tf.train.AdamOptimizer on one tensor with the given per-step gradients;
    returns (theta, m, v).  Scalar loops: t counts from 1, epsilon sits OUTSIDE the square root
    and is not bias-corrected."""
    theta = np.array(theta, dtype=np.float64).copy()
    m = np.zeros_like(theta)
    v = np.zeros_like(theta)
    flat_t, flat_m, flat_v = theta.reshape(-1), m.reshape(-1), v.reshape(-1)
    for t in range(1, steps + 1):
        g = np.asarray(grads[t - 1], dtype=np.float64).reshape(-1)
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        for i in range(flat_t.size):
            flat_m[i] = beta1 * flat_m[i] + (1.0 - beta1) * g[i]
            flat_v[i] = beta2 * flat_v[i] + (1.0 - beta2) * g[i] * g[i]
            flat_t[i] -= lr_t * flat_m[i] / (math.sqrt(flat_v[i]) + eps)
    return theta, m, v
