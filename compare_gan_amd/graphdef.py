"""Frozen TensorFlow GraphDef (.pb) -> the weight dictionary of compare_gan_amd.inception.

Reference: compare_gan/eval_utils.py:41-49 evaluates with the frozen 2015 Inception graph
(`inceptionv1_for_inception_score.pb`, fetched by tfgan.eval.run_inception at run time).  Offline
that file has to be supplied by the user; this module turns it into the {name: array} dictionary
InceptionV3.load_weights() takes, without TensorFlow: a GraphDef is a protocol-buffer message whose
wire format is decoded here directly (only what is needed: node names / ops / attrs and the Const
tensors).

  GraphDef   { repeated NodeDef node = 1; }
  NodeDef    { string name = 1; string op = 2; repeated string input = 3; map<string, AttrValue> attr = 5; }
  AttrValue  { float f = 4; bool b = 5; DataType type = 6; TensorProto tensor = 8; ... }
  TensorProto{ DataType dtype = 1; TensorShapeProto tensor_shape = 2; bytes tensor_content = 4;
               repeated float float_val = 5; repeated int32 int_val = 7; }
  TensorShapeProto { repeated Dim dim = 2 { int64 size = 1; } }

The 2015 graph keeps batch norm as its own op behind every convolution
(Conv2D -> BatchNormWithGlobalNormalization(mean, variance, beta, gamma; variance_epsilon = 1e-3,
scale_after_normalization = False) -> Relu): the converter folds it into the kernel and a bias, the
form inception.py runs (conv + bias + ReLU in one kernel):
  w'[..., co] = w[..., co] * rsqrt(var[co] + eps) (* gamma[co] if scale_after_normalization)
  b'[co]      = beta[co] - mean[co] * rsqrt(var[co] + eps) (* gamma[co] ...)

inception_weights_from_graphdef(path) is what eval_utils calls for a `.pb` path;
write_graphdef() emits the same wire format (used by the round-trip test and to ship fixtures).
"""
import struct

import numpy as np

DT_FLOAT, DT_INT32 = 1, 3
BN_EPSILON = 1e-3   # variance_epsilon of every BatchNormWithGlobalNormalization node of the graph


# ---- protocol-buffer wire format ------------------------------------------------------------------
def _varint(buf, pos):
  out, shift = 0, 0
  while True:
    b = buf[pos]
    pos += 1
    out |= (b & 0x7F) << shift
    if not b & 0x80:
      return out, pos
    shift += 7


def _fields(buf):
  """Yields (field number, wire type, value) of one message; length-delimited values as memoryview."""
  pos, n = 0, len(buf)
  while pos < n:
    tag, pos = _varint(buf, pos)
    field, wt = tag >> 3, tag & 7
    if wt == 0:
      val, pos = _varint(buf, pos)
    elif wt == 1:
      val, pos = bytes(buf[pos:pos + 8]), pos + 8
    elif wt == 2:
      ln, pos = _varint(buf, pos)
      val, pos = buf[pos:pos + ln], pos + ln
    elif wt == 5:
      val, pos = bytes(buf[pos:pos + 4]), pos + 4
    else:
      raise ValueError("unsupported protobuf wire type %d" % wt)
    yield field, wt, val


def _parse_tensor(buf):
  dtype, shape, content, floats, ints = 0, [], None, [], []
  for f, wt, v in _fields(buf):
    if f == 1:
      dtype = v
    elif f == 2:
      for f2, _, v2 in _fields(v):
        if f2 == 2:
          size = 0
          for f3, _, v3 in _fields(v2):
            if f3 == 1:
              size = v3
          shape.append(size)
    elif f == 4:
      content = bytes(v)
    elif f == 5:
      floats += (list(np.frombuffer(bytes(v), dtype="<f4")) if wt == 2
                 else [struct.unpack("<f", v)[0]])
    elif f == 7:
      if wt == 2:
        pos = 0
        while pos < len(v):
          x, pos = _varint(v, pos)
          ints.append(x)
      else:
        ints.append(v)
  np_dtype = {DT_FLOAT: "<f4", DT_INT32: "<i4"}.get(dtype)
  if np_dtype is None:
    return None
  count = int(np.prod(shape)) if shape else 1
  if content is not None:
    arr = np.frombuffer(content, dtype=np_dtype)
  else:
    vals = floats if dtype == DT_FLOAT else ints
    arr = np.asarray(vals, dtype=np_dtype)
    if arr.size == 1 and count > 1:
      arr = np.full(count, arr[0], dtype=np_dtype)   # a splat constant
  return arr.reshape(shape).copy()


def read_graphdef(path):
  """{node name: {"op": str, "inputs": [str], "attrs": {name: python value / ndarray}}}."""
  with open(path, "rb") as f:
    buf = memoryview(f.read())
  nodes = {}
  for f, _, v in _fields(buf):
    if f != 1:
      continue
    name, op, inputs, attrs = "", "", [], {}
    for f2, _, v2 in _fields(v):
      if f2 == 1:
        name = bytes(v2).decode()
      elif f2 == 2:
        op = bytes(v2).decode()
      elif f2 == 3:
        inputs.append(bytes(v2).decode())
      elif f2 == 5:
        key, val = None, None
        for f3, _, v3 in _fields(v2):
          if f3 == 1:
            key = bytes(v3).decode()
          elif f3 == 2:
            for f4, wt4, v4 in _fields(v3):
              if f4 == 8:
                val = _parse_tensor(v4)
              elif f4 == 4:
                val = struct.unpack("<f", v4)[0]
              elif f4 in (5, 6, 3):
                val = v4
        attrs[key] = val
    nodes[name] = {"op": op, "inputs": inputs, "attrs": attrs}
  return nodes


def _enc_varint(x):
  out = bytearray()
  while True:
    b = x & 0x7F
    x >>= 7
    if x:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _enc_field(field, payload):
  return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_tensor(arr):
  arr = np.ascontiguousarray(arr)
  dtype = DT_FLOAT if arr.dtype.kind == "f" else DT_INT32
  arr = arr.astype("<f4" if dtype == DT_FLOAT else "<i4")
  shape = b"".join(_enc_field(2, _enc_varint((1 << 3) | 0) + _enc_varint(int(d))) for d in arr.shape)
  return (_enc_varint((1 << 3) | 0) + _enc_varint(dtype) + _enc_field(2, shape) +
          _enc_field(4, arr.tobytes()))


def write_graphdef(path, consts, other_nodes=()):
  """Writes {name: ndarray} as Const nodes (plus (name, op, inputs, {attr: float}) placeholder
  nodes) in GraphDef wire format."""
  out = bytearray()
  for name, arr in consts.items():
    attr_val = _enc_field(1, b"value") + _enc_field(2, _enc_field(8, _enc_tensor(arr)))
    attr_dt = _enc_field(1, b"dtype") + _enc_field(
        2, _enc_varint((6 << 3) | 0) + _enc_varint(DT_FLOAT if np.asarray(arr).dtype.kind == "f" else DT_INT32))
    node = (_enc_field(1, name.encode()) + _enc_field(2, b"Const") + _enc_field(5, attr_dt) +
            _enc_field(5, attr_val))
    out += _enc_field(1, node)
  for name, op, inputs, attrs in other_nodes:
    node = _enc_field(1, name.encode()) + _enc_field(2, op.encode())
    for i in inputs:
      node += _enc_field(3, i.encode())
    for k, v in attrs.items():
      node += _enc_field(5, _enc_field(1, k.encode()) + _enc_field(
          2, _enc_varint((4 << 3) | 5) + struct.pack("<f", float(v))))
    out += _enc_field(1, node)
  with open(path, "wb") as f:
    f.write(bytes(out))


# ---- the 2015 Inception graph -> inception.py names ----------------------------------------------
def _block_map():
  """inception.py convolution name -> scope of the convolution in the frozen graph."""
  m = {n: n for n in ("conv", "conv_1", "conv_2", "conv_3", "conv_4")}
  for blk in ("mixed", "mixed_1", "mixed_2"):
    m.update({blk + "/b0_1x1": blk + "/conv", blk + "/b1_1x1": blk + "/tower/conv",
              blk + "/b1_5x5": blk + "/tower/conv_1", blk + "/b2_1x1": blk + "/tower_1/conv",
              blk + "/b2_3x3a": blk + "/tower_1/conv_1", blk + "/b2_3x3b": blk + "/tower_1/conv_2",
              blk + "/b3_pool_1x1": blk + "/tower_2/conv"})
  m.update({"mixed_3/b0_3x3": "mixed_3/conv", "mixed_3/b1_1x1": "mixed_3/tower/conv",
            "mixed_3/b1_3x3a": "mixed_3/tower/conv_1", "mixed_3/b1_3x3b": "mixed_3/tower/conv_2"})
  for blk in ("mixed_4", "mixed_5", "mixed_6", "mixed_7"):
    m.update({blk + "/b0_1x1": blk + "/conv", blk + "/b1_1x1": blk + "/tower/conv",
              blk + "/b1_1x7": blk + "/tower/conv_1", blk + "/b1_7x1": blk + "/tower/conv_2",
              blk + "/b2_1x1": blk + "/tower_1/conv", blk + "/b2_7x1a": blk + "/tower_1/conv_1",
              blk + "/b2_1x7a": blk + "/tower_1/conv_2", blk + "/b2_7x1b": blk + "/tower_1/conv_3",
              blk + "/b2_1x7b": blk + "/tower_1/conv_4", blk + "/b3_pool_1x1": blk + "/tower_2/conv"})
  m.update({"mixed_8/b0_1x1": "mixed_8/tower/conv", "mixed_8/b0_3x3": "mixed_8/tower/conv_1",
            "mixed_8/b1_1x1": "mixed_8/tower_1/conv", "mixed_8/b1_1x7": "mixed_8/tower_1/conv_1",
            "mixed_8/b1_7x1": "mixed_8/tower_1/conv_2", "mixed_8/b1_3x3": "mixed_8/tower_1/conv_3"})
  for blk in ("mixed_9", "mixed_10"):
    m.update({blk + "/b0_1x1": blk + "/conv", blk + "/b1_1x1": blk + "/tower/conv",
              blk + "/b1_1x3": blk + "/tower/mixed/conv", blk + "/b1_3x1": blk + "/tower/mixed/conv_1",
              blk + "/b2_1x1": blk + "/tower_1/conv", blk + "/b2_3x3": blk + "/tower_1/conv_1",
              blk + "/b2_1x3": blk + "/tower_1/mixed/conv",
              blk + "/b2_3x1": blk + "/tower_1/mixed/conv_1",
              blk + "/b3_pool_1x1": blk + "/tower_2/conv"})
  return m


GRAPH_SCOPE = _block_map()
LOGITS_WEIGHTS, LOGITS_BIASES = "softmax/weights", "softmax/biases"


def inception_weights_from_nodes(nodes):
  """The folded {name: float32 ndarray} dictionary of inception.InceptionV3.load_weights()."""
  def const(name):
    if name not in nodes or nodes[name]["attrs"].get("value") is None:
      raise KeyError("the graph has no Const tensor %r (is this the 2015 Inception graph?)" % name)
    return np.asarray(nodes[name]["attrs"]["value"], dtype=np.float64)

  out = {}
  for ours, scope in GRAPH_SCOPE.items():
    w = const(scope + "/conv2d_params")                      # HWIO
    beta = const(scope + "/batchnorm/beta")
    mean = const(scope + "/batchnorm/moving_mean")
    var = const(scope + "/batchnorm/moving_variance")
    bn = nodes.get(scope + "/batchnorm", {"attrs": {}})["attrs"]
    eps = float(bn.get("variance_epsilon") or BN_EPSILON)
    inv = 1.0 / np.sqrt(var + eps)
    if bn.get("scale_after_normalization"):
      inv = inv * const(scope + "/batchnorm/gamma")
    out[ours + "/kernel"] = (w * inv).astype(np.float32)
    out[ours + "/bias"] = (beta - mean * inv).astype(np.float32)
  out["logits/kernel"] = const(LOGITS_WEIGHTS).astype(np.float32)
  out["logits/bias"] = const(LOGITS_BIASES).astype(np.float32)
  return out


def inception_weights_from_graphdef(path):
  return inception_weights_from_nodes(read_graphdef(path))
