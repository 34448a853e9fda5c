"""ORACLE -- test infrastructure only.

CPU restatement (PyTorch-CPU fp64) of the Inception feature path the reference reaches through
tfgan.eval.run_inception on the frozen 2015 graph (eval_utils.py:165-175): preprocess
(oracle.fid.inception_preprocess), conv + bias + ReLU stacks, 3x3 max / average pools (TF 'SAME'
average pooling divides by the number of valid taps), channel concatenation, global average pool
-> pool_3, dense -> logits.  The graph's trained weights are not available offline, so this oracle
and the product share a table of seeded weights ("parity unpinned" against the real graph; what is
checked is that the HIP path computes THIS network correctly).  `spec` is the op table of
compare_gan_amd.inception (passed in by the test; nothing here imports the product)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import arch_ops as oops
from oracle import fid as ofid


def _bf16(t):
  return t.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def _conv(x, w, b, stride, padding, emulate):
  if padding == "SAME":
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = oops.same_pads(x.shape[1], kh, stride)
    _, pl, pr = oops.same_pads(x.shape[2], kw, stride)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
  else:
    xp = x.permute(0, 3, 1, 2)
  wq = _bf16(w) if emulate else w
  y = F.conv2d(xp, wq.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1) + b
  y = torch.relu(y)
  return _bf16(y) if emulate else y


def _pool(x, kind, k, s, same):
  xn = x.permute(0, 3, 1, 2)
  p = (k - 1) // 2 if same else 0
  if kind == "max":
    y = F.max_pool2d(xn, k, s, p)
  else:
    y = F.avg_pool2d(xn, k, s, p, count_include_pad=False)
  return y.permute(0, 2, 3, 1)


def run(spec, weights, x, emulate_bf16=False):
  for op in spec:
    if op[0] == "conv":
      _, name, _, _, _, stride, padding = op
      x = _conv(x, weights[name + "/kernel"].double(), weights[name + "/bias"].double(), stride,
                padding, emulate_bf16)
    elif op[0] == "max":
      x = _pool(x, "max", op[1], op[2], False)
    elif op[0] == "avg3":
      x = _pool(x, "avg", 3, 1, True)
      if emulate_bf16:
        x = _bf16(x)
    elif op[0] == "max3s1":
      x = _pool(x, "max", 3, 1, True)
    elif op[0] == "mixed":
      x = torch.cat([run(b, weights, x, emulate_bf16) for b in op[2]], dim=3)
    elif op[0] == "split":
      x = torch.cat([run(b, weights, x, emulate_bf16) for b in op[1]], dim=3)
    else:
      raise ValueError(op)
  return x


def features(spec, weights, images_0_255, emulate_bf16=False):
  """images [B,H,W,3] in [0,255] (numpy) -> (pool_3 [B,2048], logits [B,1008]) fp64 tensors."""
  x = torch.from_numpy(np.asarray(ofid.inception_preprocess(images_0_255, 299)))
  if emulate_bf16:
    x = _bf16(x)
  x = run(spec, weights, x, emulate_bf16)
  pool3 = x.mean(dim=(1, 2))
  if emulate_bf16:
    pool3 = _bf16(pool3)
  wl = weights["logits/kernel"].double()
  logits = pool3 @ (_bf16(wl) if emulate_bf16 else wl) + weights["logits/bias"].double()
  return pool3, logits
