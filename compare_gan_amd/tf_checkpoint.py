"""TensorFlow-1 checkpoints (tensor bundles: `<prefix>.index` + `<prefix>.data-NNNNN-of-MMMMM`) ->
the state of a ModularGAN, without TensorFlow.

Reference: the Estimator saves / restores every global variable of the model under its TF name
(compare_gan/gans/modular_gan.py:266-285 loads them for the TF-Hub export; runner_lib.py:193-206
walks `model.ckpt-<step>` prefixes).  The variables of this package carry the SAME names and
layouts (HWIO kernels, `<kernel>/u_var`, `moving_mean`, `accu/...`,
`<var>/ExponentialMovingAverage`, `global_step`, `global_step_disc`), so importing is a rename-free
copy; only the optimiser slots are spelled differently:

  TF (tf.train.AdamOptimizer(lr, name="d_opt"), modular_gan.py:607,613; slot_creator names a slot
  `<primary>/<optimizer name>`, the second one of the same primary gets the `_1` suffix):
      <var>/d_opt   = m        <var>/d_opt_1 = v        (g_opt / g_opt_1 for the generator)
      beta1_power, beta2_power (+ `_1` for the second optimiser): not needed, the step counters
      carry the same information
  here (ModularGAN.state_dict): <var>/d_opt/Adam = m, <var>/d_opt/Adam_1 = v

On-disk format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*):
  .index   an SSTable (the LevelDB table format): data blocks of prefix-compressed (key, value)
           entries with restart points, each block followed by a 1-byte compression type and a
           masked CRC32C; an index block (separator key -> BlockHandle{offset, size} as varints); a
           metaindex block; a 48-byte footer (two BlockHandles, padding, magic 0xdb4775248b80fb57).
           Key "" holds a BundleHeaderProto {num_shards = 1, endianness = 2, version = 3}; every
           other key is a variable name with a BundleEntryProto {dtype = 1, shape = 2
           (TensorShapeProto: repeated dim = 2 {size = 1}), shard_id = 3, offset = 4, size = 5,
           crc32c = 6 (fixed32, masked CRC32C of the tensor bytes)}.  The bundle writer does not
           compress its blocks (table::kNoCompression); a snappy block is reported as unsupported.
  .data-*  the tensors' bytes, little-endian, row-major, at [offset, offset + size) of shard shard_id.

write_bundle() emits the same format (several data blocks, restart interval 16, real prefix
compression) -- used by the round-trip test and scripts/make_tf_checkpoint_fixture.py; no real
TF-written file is available offline, so the reader is pinned by this independent writer plus the
LevelDB / bundle format constants (magic number, masked CRC32C) -- "parity unpinned" against a
TF-written bundle until one is supplied.
"""
import os
import struct

import numpy as np

from compare_gan_amd.graphdef import _enc_field, _enc_varint, _fields, _varint
from compare_gan_amd.tfrecord import masked_crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_BYTES = 48
# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"),
          6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"), 19: np.dtype("<f2")}
DT_BFLOAT16 = 14
DT_OF = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9,
         np.dtype("bool"): 10}


# ---- SSTable reader -------------------------------------------------------------------------------
def _block_handle(buf, pos):
  off, pos = _varint(buf, pos)
  size, pos = _varint(buf, pos)
  return (off, size), pos


def _read_block(raw, handle, verify):
  off, size = handle
  if off + size + 5 > len(raw):
    raise ValueError("table block [%d, +%d) runs past the end of the file" % (off, size))
  body, ctype = raw[off:off + size], raw[off + size]
  if verify:
    want = struct.unpack("<I", raw[off + size + 1:off + size + 5])[0]
    if masked_crc32c(bytes(raw[off:off + size + 1])) != want:
      raise ValueError("table block at %d: CRC mismatch" % off)
  if ctype != 0:
    raise ValueError("table block at %d is compressed (type %d): tensor bundles are written "
                     "uncompressed; snappy is not supported" % (off, ctype))
  return body


def _block_entries(body):
  """(key, value) pairs of one block: shared / unshared / value lengths as varint32, keys
  prefix-compressed against the previous key, restart array at the end."""
  if len(body) < 4:
    raise ValueError("table block too short")
  nrestarts = struct.unpack("<I", body[-4:])[0]
  end = len(body) - 4 - 4 * nrestarts
  if end < 0:
    raise ValueError("table block: bad restart count %d" % nrestarts)
  pos, key = 0, b""
  while pos < end:
    shared, pos = _varint(body, pos)
    unshared, pos = _varint(body, pos)
    vlen, pos = _varint(body, pos)
    if shared > len(key):
      raise ValueError("table block: entry shares %d bytes with a %d-byte key" % (shared, len(key)))
    key = key[:shared] + bytes(body[pos:pos + unshared])
    pos += unshared
    yield key, body[pos:pos + vlen]
    pos += vlen


def read_table(path, verify=True):
  """All (key bytes, value memoryview) entries of an SSTable file, in key order."""
  raw = memoryview(open(path, "rb").read())
  if len(raw) < FOOTER_BYTES:
    raise ValueError("%s: too short for an SSTable footer" % path)
  footer = raw[-FOOTER_BYTES:]
  if struct.unpack("<Q", footer[-8:])[0] != TABLE_MAGIC:
    raise ValueError("%s: bad table magic (not a tensor-bundle index)" % path)
  _, pos = _block_handle(footer, 0)           # metaindex (empty for bundles)
  index_handle, _ = _block_handle(footer, pos)
  out = []
  for _, handle_bytes in _block_entries(_read_block(raw, index_handle, verify)):
    handle, _ = _block_handle(handle_bytes, 0)
    out.extend((k, bytes(v)) for k, v in _block_entries(_read_block(raw, handle, verify)))
  return out


# ---- bundle ---------------------------------------------------------------------------------------
def _parse_entry(buf):
  e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
  for field, wt, val in _fields(memoryview(buf)):
    if field == 1:
      e["dtype"] = val
    elif field == 2:
      for f2, _, v2 in _fields(val):
        if f2 == 2:
          size = 0
          for f3, _, v3 in _fields(v2):
            if f3 == 1:
              size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
          e["shape"].append(size)
    elif field == 3:
      e["shard_id"] = val
    elif field == 4:
      e["offset"] = val
    elif field == 5:
      e["size"] = val
    elif field == 6:
      e["crc32c"] = struct.unpack("<I", val)[0] if wt == 5 else val
    elif field == 7:
      e["sliced"] = True
  return e


def read_index(prefix, verify=True):
  """(header dict, {variable name: entry dict}) of `<prefix>.index`."""
  header, entries = {"num_shards": 1, "endianness": 0, "version": {}}, {}
  for key, val in read_table(prefix + ".index", verify):
    if key == b"":
      for field, _, v in _fields(memoryview(val)):
        if field == 1:
          header["num_shards"] = v
        elif field == 2:
          header["endianness"] = v
      continue
    entries[key.decode("utf-8")] = _parse_entry(val)
  if header["endianness"] != 0:
    raise ValueError("%s: big-endian bundles are not supported" % prefix)
  return header, entries


def _shard_path(prefix, shard, num_shards):
  return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def read_bundle(prefix, names=None, verify_tensors=False):
  """{variable name: numpy array} of a TF checkpoint prefix (e.g. `.../model.ckpt-250000`).
  names: restrict to these variables.  verify_tensors: check every tensor's masked CRC32C (pure
  Python: slow for large checkpoints)."""
  header, entries = read_index(prefix)
  shards = {}
  out = {}
  for name, e in entries.items():
    if names is not None and name not in names:
      continue
    if e["sliced"]:
      raise ValueError("%s: %s is a partitioned (sliced) variable; not supported" % (prefix, name))
    if e["shard_id"] not in shards:
      shards[e["shard_id"]] = np.memmap(_shard_path(prefix, e["shard_id"], header["num_shards"]),
                                        dtype=np.uint8, mode="r")
    raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
    if len(raw) != e["size"]:
      raise ValueError("%s: %s runs past the end of its data shard" % (prefix, name))
    if verify_tensors and e["crc32c"] is not None and masked_crc32c(raw.tobytes()) != e["crc32c"]:
      raise ValueError("%s: CRC mismatch in the bytes of %s" % (prefix, name))
    if e["dtype"] == DT_BFLOAT16:
      arr = (np.frombuffer(raw.tobytes(), dtype="<u2").astype(np.uint32) << 16).view(np.float32)
    elif e["dtype"] in DTYPES:
      arr = np.frombuffer(raw.tobytes(), dtype=DTYPES[e["dtype"]])
    else:
      raise ValueError("%s: %s has unsupported dtype enum %d" % (prefix, name, e["dtype"]))
    n = int(np.prod(e["shape"])) if e["shape"] else 1
    if arr.size != n:
      raise ValueError("%s: %s holds %d elements, its shape %s needs %d" % (
          prefix, name, arr.size, e["shape"], n))
    out[name] = arr.reshape(e["shape"]).copy()
  return out


def latest_checkpoint(model_dir):
  """Prefix of the newest `model.ckpt-<step>` in a directory (the `checkpoint` state file is a text
  proto TF rewrites on every save; the step in the file name is the contract runner_lib uses,
  runner_lib.py:193-206), or None."""
  best = None
  for f in os.listdir(model_dir):
    if f.startswith("model.ckpt-") and f.endswith(".index"):
      try:
        step = int(f[len("model.ckpt-"):-len(".index")])
      except ValueError:
        continue
      if best is None or step > best[0]:
        best = (step, os.path.join(model_dir, f[:-len(".index")]))
  return None if best is None else best[1]


# ---- writer (tests, fixtures) ----------------------------------------------------------------------
def _emit_block(out, entries, restart_interval):
  body, restarts, last = bytearray(), [], b""
  for i, (k, v) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(body))
    else:
      while shared < min(len(last), len(k)) and last[shared] == k[shared]:
        shared += 1
    body += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
    last = k
  if not restarts:
    restarts = [0]
  for r in restarts:
    body += struct.pack("<I", r)
  body += struct.pack("<I", len(restarts))
  off = len(out)
  out += body
  out += b"\x00" + struct.pack("<I", masked_crc32c(bytes(body) + b"\x00"))
  return off, len(body)


def _enc_entry(arr, offset, crc):
  shape = b"".join(_enc_field(2, _enc_varint((1 << 3) | 0) + _enc_varint(int(d))) for d in arr.shape)
  msg = _enc_varint((1 << 3) | 0) + _enc_varint(DT_OF[arr.dtype])
  msg += _enc_field(2, shape)
  if offset:
    msg += _enc_varint((4 << 3) | 0) + _enc_varint(offset)
  msg += _enc_varint((5 << 3) | 0) + _enc_varint(arr.nbytes)
  msg += _enc_varint((6 << 3) | 5) + struct.pack("<I", crc)
  return msg


def write_bundle(prefix, tensors, block_bytes=512, restart_interval=16):
  """Writes {name: array} as a one-shard TF tensor bundle (see the module docstring)."""
  names = sorted(tensors, key=lambda n: n.encode("utf-8"))
  data, entries = bytearray(), []
  header = _enc_varint((1 << 3) | 0) + _enc_varint(1) + _enc_field(3, _enc_varint((1 << 3) | 0) + _enc_varint(1))
  entries.append((b"", bytes(header)))
  for n in names:
    arr = np.asarray(tensors[n], order="C")
    if arr.dtype not in DT_OF:
      raise ValueError("write_bundle: unsupported dtype %s of %s" % (arr.dtype, n))
    raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    entries.append((n.encode("utf-8"), bytes(_enc_entry(arr, len(data), masked_crc32c(raw)))))
    data += raw
  out, index, block, size = bytearray(), [], [], 0
  for k, v in entries:
    block.append((k, v))
    size += len(k) + len(v)
    if size >= block_bytes:
      index.append((block[-1][0], _emit_block(out, block, restart_interval)))
      block, size = [], 0
  if block:
    index.append((block[-1][0], _emit_block(out, block, restart_interval)))
  meta = _emit_block(out, [], restart_interval)
  idx = _emit_block(out, [(k, bytes(_enc_varint(o) + _enc_varint(s))) for k, (o, s) in index], 1)
  footer = bytearray(_enc_varint(meta[0]) + _enc_varint(meta[1]) + _enc_varint(idx[0]) + _enc_varint(idx[1]))
  footer += b"\x00" * (FOOTER_BYTES - 8 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
  out += footer
  with open(prefix + ".index", "wb") as f:
    f.write(bytes(out))
  with open(_shard_path(prefix, 0, 1), "wb") as f:
    f.write(bytes(data))


# ---- names ----------------------------------------------------------------------------------------
def state_key(tf_name):
  """TF variable name -> key of ModularGAN.state_dict() (None: a variable this package does not
  keep, e.g. beta1_power)."""
  base = tf_name.rsplit("/", 1)[-1]
  if base.startswith("beta1_power") or base.startswith("beta2_power"):
    return None
  for tag in ("g_opt", "d_opt"):
    if tf_name.endswith("/" + tag):
      return tf_name + "/Adam"
    if tf_name.endswith("/" + tag + "_1"):
      return tf_name[:-2] + "/Adam_1"
  return tf_name


def tf_name(state_key_):
  """Inverse of state_key (what a TF checkpoint of the reference calls this entry)."""
  for tag in ("g_opt", "d_opt"):
    if state_key_.endswith("/%s/Adam" % tag):
      return state_key_[:-len("/Adam")]
    if state_key_.endswith("/%s/Adam_1" % tag):
      return state_key_[:-len("/Adam_1")] + "_1"
  return state_key_


def export_tf_checkpoint(gan, prefix):
  """gan.state_dict() under the reference's TF names, as a tensor bundle."""
  sd = gan.state_dict()
  write_bundle(prefix, {tf_name(k): v.detach().cpu().numpy() for k, v in sd.items()})


def import_tf_checkpoint(gan, prefix, strict=True, with_optimizer=True):
  """Loads a TF-1 checkpoint of the reference into a built ModularGAN.  Every entry of
  gan.state_dict() must be present with its shape (strict) -- optimiser slots are optional when
  with_optimizer is False or the checkpoint has none (e.g. an inference export) -- and extra
  variables in the checkpoint are reported.  Returns {"loaded", "missing", "unexpected"}."""
  import torch
  want = gan.state_dict()
  _, entries = read_index(prefix)
  have = {}
  for name in entries:
    key = state_key(name)
    if key is not None:
      have[key] = name
  is_slot = lambda k: k.endswith("/Adam") or k.endswith("/Adam_1")
  missing = [k for k in want if k not in have and not (is_slot(k) and not with_optimizer)]
  slots_missing = [k for k in missing if is_slot(k)]
  if slots_missing and len(slots_missing) == sum(1 for k in want if is_slot(k)):
    missing = [k for k in missing if not is_slot(k)]      # a checkpoint without any optimiser state
  unexpected = sorted(n for k, n in have.items() if k not in want)
  if strict and missing:
    raise KeyError("%s lacks %d variable(s) of the model, e.g. %s" % (prefix, len(missing), missing[:4]))
  take = {k: n for k, n in have.items() if k in want and (with_optimizer or not is_slot(k))}
  arrays = read_bundle(prefix, names=set(take.values()))
  sd = {k: v for k, v in want.items()}
  for k, n in take.items():
    a = arrays[n]
    if tuple(a.shape) != tuple(want[k].shape):
      raise ValueError("%s: %s has shape %s, the model's %s is %s" % (prefix, n, a.shape, k,
                                                                    tuple(want[k].shape)))
    sd[k] = torch.from_numpy(a).to(want[k].dtype).to(want[k].device)
  gan.load_state_dict(sd)
  return {"loaded": sorted(take), "missing": sorted(missing), "unexpected": unexpected}
