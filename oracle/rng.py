"""ORACLE -- test infrastructure only.

NumPy restatement of the framework's stateless counter RNG (Philox4x32-10 keyed by (seed, op_id),
counter = (element quad index, stream id, step)).  The reference's own generator
(tpu/tpu_random.py:54-154, tf.contrib.stateless) is a TF op whose bit stream is not reproducible
outside TensorFlow; only its SEMANTICS are part of the contract (deterministic per (op, step),
distinct across steps and replicas: tpu/tpu_random_test.py:87-168) and those are what
tests/test_rng.py checks.  This file pins the bit stream of the HIP implementation.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
  c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3)]
  k0 = np.uint32(k0)
  k1 = np.uint32(k1)
  for _ in range(10):
    p0 = M0 * c0.astype(np.uint64)
    p1 = M1 * c2.astype(np.uint64)
    hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
    hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
    c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
    with np.errstate(over="ignore"):
      k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
      k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
  return c0, c1, c2, c3


def _key(seed, op_id):
  seed = int(seed) & (2 ** 64 - 1)
  k0 = (seed & 0xFFFFFFFF) ^ ((int(op_id) * 0x9E3779B1) & 0xFFFFFFFF)
  k1 = ((seed >> 32) & 0xFFFFFFFF) ^ 0x85EBCA6B
  return k0, k1


def _raw(n, seed, op_id, stream_id, step):
  nq = (n + 3) // 4
  q = np.arange(nq, dtype=np.uint64)
  c0 = (q & MASK).astype(np.uint32)
  c1 = ((q >> np.uint64(32)).astype(np.uint32)) ^ np.uint32((int(stream_id) << 8) & 0xFFFFFFFF)
  c2 = np.full(nq, int(step) & 0xFFFFFFFF, dtype=np.uint32)
  c3 = np.full(nq, (int(step) >> 32) & 0xFFFFFFFF, dtype=np.uint32)
  k0, k1 = _key(seed, op_id)
  return philox4x32_10(c0, c1, c2, c3, k0, k1)


def uniform(n, lo, hi, seed, op_id, stream_id=0, step=0):
  r = _raw(n, seed, op_id, stream_id, step)
  sc = np.float32(hi) - np.float32(lo)
  u = [(x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) for x in r]
  out = np.stack([np.float32(lo) + sc * v for v in u], axis=1).reshape(-1)
  return out[:n]


def normal(n, mean, stddev, seed, op_id, stream_id=0, step=0):
  r0, r1, r2, r3 = _raw(n, seed, op_id, stream_id, step)
  f = np.float32(2.0 ** -24)
  u1 = ((r0 >> np.uint32(8)).astype(np.float32) + np.float32(1)) * f
  u2 = (r1 >> np.uint32(8)).astype(np.float32) * f
  u3 = ((r2 >> np.uint32(8)).astype(np.float32) + np.float32(1)) * f
  u4 = (r3 >> np.uint32(8)).astype(np.float32) * f
  ra = np.sqrt(np.float32(-2) * np.log(u1))
  rb = np.sqrt(np.float32(-2) * np.log(u3))
  ta = np.float32(6.283185307179586) * u2
  tb = np.float32(6.283185307179586) * u4
  m, s = np.float32(mean), np.float32(stddev)
  out = np.stack([m + s * ra * np.cos(ta), m + s * ra * np.sin(ta), m + s * rb * np.cos(tb),
                  m + s * rb * np.sin(tb)], axis=1).reshape(-1)
  return out[:n].astype(np.float32)


def labels(n, K, seed, op_id, stream_id=0, step=0):
  r = _raw(n, seed, op_id, stream_id, step)
  out = np.stack([((x.astype(np.uint64) * np.uint64(K)) >> np.uint64(32)).astype(np.int32)
                  for x in r], axis=1).reshape(-1)
  return out[:n]
