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


# ------------------------------------------------------------------------------------------------
# Round 3: the remaining "parity unpinned" arithmetic of SURVEY section 8c, again as plain NumPy
# written from the reference lines / the TF contract only (nothing shared with oracle/arch_ops.py,
# oracle/gan.py): zero-insertion un-pooling and the two poolings, one spectral-norm power iteration,
# training-mode batch norm, the self-attention core, the four GAN losses, the EMA update.
# ------------------------------------------------------------------------------------------------
def unpool(x):
    """resnet_ops.py:35-56: [N,H,W,C] -> [N,2H,2W,C], value at the even pixels, zeros elsewhere."""
    x = np.asarray(x, dtype=np.float64)
    n, h, w, c = x.shape
    out = np.zeros((n, 2 * h, 2 * w, c))
    for i in range(h):
        for j in range(w):
            out[:, 2 * i, 2 * j, :] = x[:, i, j, :]
    return out


def avg_pool2_same(x):
    """tf.nn.pool(x, [2,2], "AVG", "SAME", strides=[2,2]) (resnet_ops.py:131-133): out = ceil(in/2);
    a window hanging over the border averages only the elements that exist."""
    x = np.asarray(x, dtype=np.float64)
    n, h, w, c = x.shape
    ho, wo = -(-h // 2), -(-w // 2)
    out = np.zeros((n, ho, wo, c))
    for i in range(ho):
        for j in range(wo):
            win = x[:, 2 * i:min(2 * i + 2, h), 2 * j:min(2 * j + 2, w), :]
            out[:, i, j, :] = win.sum(axis=(1, 2)) / (win.shape[1] * win.shape[2])
    return out


def max_pool2_valid(x):
    """tf.layers.max_pooling2d(pool_size=[2,2], strides=2), VALID (arch_ops.py:741,750)."""
    x = np.asarray(x, dtype=np.float64)
    n, h, w, c = x.shape
    out = np.zeros((n, h // 2, w // 2, c))
    for i in range(h // 2):
        for j in range(w // 2):
            out[:, i, j, :] = x[:, 2 * i:2 * i + 2, 2 * j:2 * j + 2, :].max(axis=(1, 2))
    return out


def _l2n(v, eps):
    """tf.math.l2_normalize over all elements: v * rsqrt(max(sum(v^2), eps))."""
    return v / math.sqrt(max(float(np.sum(v * v)), eps))


def spectral_norm(w, u, singular_value="left", eps=1e-12):
    """arch_ops.py:479-535, one power-iteration round.  w [..., Co] is viewed as W [K, Co]; u is
    [K, 1] for "left", [1, Co] for "right" ("auto": left iff K <= Co, :489-490).
    Returns (w / sigma, u_new, sigma): left: v = l2n(W^T u), u' = l2n(W v), sigma = u'^T W v (:507-509,
    525); right: v = l2n(u W^T), u' = l2n(v W), sigma = v W u'^T (:511-513, 527)."""
    w = np.asarray(w, dtype=np.float64)
    w2 = w.reshape(-1, w.shape[-1])
    if singular_value == "auto":
        singular_value = "left" if w2.shape[0] <= w2.shape[1] else "right"
    u = np.asarray(u, dtype=np.float64)
    if singular_value == "left":
        v = _l2n(w2.T.dot(u), eps)                  # [Co, 1]
        u_new = _l2n(w2.dot(v), eps)                # [K, 1]
        sigma = float(u_new.T.dot(w2).dot(v).item())
    else:
        v = _l2n(u.dot(w2.T), eps)                  # [1, K]
        u_new = _l2n(v.dot(w2), eps)                # [1, Co]
        sigma = float(v.dot(w2).dot(u_new.T).item())
    return (w2 / sigma).reshape(w.shape), u_new, sigma


def batch_norm_train(x, gamma=None, beta=None, eps=1e-5):
    """standardize_batch + scale / offset in training mode (arch_ops.py:289-319): per-channel
    sufficient statistics over (N, H, W) normalised without a shift (tf.nn.sufficient_statistics +
    tf.nn.normalize_moments(shift=None), :294-297: mean = sum(x) / n, variance = sum(x^2) / n -
    mean^2, i.e. the biased variance), then (x - mean) * rsqrt(var + eps) * gamma + beta
    (tf.nn.batch_normalization, :306-312).  gamma / beta: [C], or [N, C] (conditional)."""
    x = np.asarray(x, dtype=np.float64)
    c = x.shape[-1]
    flat = x.reshape(-1, c)
    mean = flat.sum(axis=0) / flat.shape[0]
    var = (flat * flat).sum(axis=0) / flat.shape[0] - mean * mean
    out = (x - mean) / np.sqrt(var + eps)
    for p, apply in ((gamma, lambda o, q: o * q), (beta, lambda o, q: o + q)):
        if p is not None:
            p = np.asarray(p, dtype=np.float64)
            out = apply(out, p.reshape((p.shape[0],) + (1,) * (x.ndim - 2) + (c,)) if p.ndim == 2 else p)
    return out, mean, var


def attention(theta, phi, g):
    """arch_ops.py:744-753: softmax(theta phi^T) g per sample, row by row.
    theta [B, Lq, Dk], phi [B, Lk, Dk], g [B, Lk, Dv] -> [B, Lq, Dv]."""
    theta, phi, g = (np.asarray(t, dtype=np.float64) for t in (theta, phi, g))
    out = np.zeros(theta.shape[:2] + (g.shape[2],))
    for b in range(theta.shape[0]):
        for q in range(theta.shape[1]):
            s = phi[b].dot(theta[b, q])
            e = np.exp(s - s.max())
            out[b, q] = (e / e.sum()).dot(g[b])
    return out


def _sigmoid_xent(logits, label):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x, 0) - x * z + log(1 + exp(-|x|))."""
    x = np.asarray(logits, dtype=np.float64)
    return np.maximum(x, 0.0) - x * label + np.log1p(np.exp(-np.abs(x)))


def gan_losses(kind, d_real_logits, d_fake_logits):
    """(d_loss, d_loss_real, d_loss_fake, g_loss) of gans/loss_lib.py:53-148; least_squares takes
    the sigmoid of the logits as its predictions (modular_gan.py: d_real = sigmoid(d_real_logits))."""
    r = np.asarray(d_real_logits, dtype=np.float64)
    f = np.asarray(d_fake_logits, dtype=np.float64)
    if kind == "non_saturating":                    # loss_lib.py:53-80
        dr, df = _sigmoid_xent(r, 1.0).mean(), _sigmoid_xent(f, 0.0).mean()
        return dr + df, dr, df, _sigmoid_xent(f, 1.0).mean()
    if kind == "wasserstein":                       # :83-104
        dr, df = -r.mean(), f.mean()
        return dr + df, dr, df, -df
    if kind == "least_squares":                     # :107-128
        pr, pf = 1.0 / (1.0 + np.exp(-r)), 1.0 / (1.0 + np.exp(-f))
        dr, df = ((pr - 1.0) ** 2).mean(), (pf ** 2).mean()
        return 0.5 * (dr + df), dr, df, 0.5 * ((pf - 1.0) ** 2).mean()
    if kind == "hinge":                             # :131-148
        dr, df = np.maximum(1.0 - r, 0.0).mean(), np.maximum(1.0 + f, 0.0).mean()
        return dr + df, dr, df, -f.mean()
    raise ValueError(kind)


def ema(shadow, value, decay):
    """tf.train.ExponentialMovingAverage.apply: shadow -= (1 - decay) * (shadow - value)."""
    return shadow - (1.0 - decay) * (shadow - value)


def gradient_penalty_quadratic(x_hat, a, b):
    """The WGAN-GP / DRAGAN penalty (penalty_lib.py:59-82, 33-56) for the closed-form discriminator
    D(x) = (a . x)^2 + b . x per sample (its input gradient is 2 (a . x) a + b, no autograd needed):
    mean over the batch of (sqrt(1e-4 + |grad|^2) - 1)^2, and d penalty / d a, d penalty / d b --
    what the double backward of the product has to deliver -- by the chain rule written out."""
    x_hat = np.asarray(x_hat, dtype=np.float64)
    n = x_hat.shape[0]
    xf = x_hat.reshape(n, -1)
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    pen, da, db = 0.0, np.zeros_like(a), np.zeros_like(b)
    for i in range(n):
        s = float(a.dot(xf[i]))
        g = 2.0 * s * a + b
        slope = math.sqrt(1e-4 + float(g.dot(g)))
        pen += (slope - 1.0) ** 2 / n
        dg = (2.0 * (slope - 1.0) / slope) * g / n          # d pen / d g
        da += 2.0 * s * dg + 2.0 * float(dg.dot(a)) * xf[i]  # g = 2 (a.x) a + b
        db += dg
    return pen, da, db


def interpolate(x, x_fake, alpha):
    """penalty_lib.py:74: x + alpha (x_fake - x), alpha [B,1,1,1]."""
    return np.asarray(x, np.float64) + np.asarray(alpha, np.float64) * (
        np.asarray(x_fake, np.float64) - np.asarray(x, np.float64))


def dragan_perturb(x, noise):
    """penalty_lib.py:46-49: x + std(x) * (noise - 0.5) with the population standard deviation over
    ALL elements of the batch (tf.nn.moments over every axis), clipped to [0, 1]; noise ~ U[0, 1)."""
    x = np.asarray(x, dtype=np.float64)
    std = math.sqrt(float(((x - x.mean()) ** 2).mean()))
    return np.clip(x + std * (np.asarray(noise, np.float64) - 0.5), 0.0, 1.0)


def inception_score(logits):
    """tfgan.eval.classifier_score_from_logits (metrics/inception_score.py:39-48): with p(y|x_i) =
    softmax(logits_i) and p(y) = mean_i p(y|x_i), exp(mean_i sum_y p(y|x_i) (log p(y|x_i) - log p(y)))
    -- sample by sample with scalar-free Python loops over the rows."""
    logits = np.asarray(logits, dtype=np.float64)
    probs = []
    for row in logits:
        e = np.exp(row - row.max())
        probs.append(e / e.sum())
    marginal = sum(probs) / len(probs)
    total = 0.0
    for p in probs:
        nz = p > 0.0                      # 0 * log 0 = 0
        total += float((p[nz] * (np.log(p[nz]) - np.log(marginal[nz]))).sum())
    return math.exp(total / len(probs))
