"""TFDS TFRecord reader for the real-data path (reference: compare_gan/datasets.py:229-251:
`tfds.load(name, split, data_dir=--tfds_data_dir)` -> features {"image": uint8 HxWxC, "label"}).

No TensorFlow / TFDS here: the three layers of the storage format are decoded directly.

  * TFRecord framing: { uint64 length; uint32 masked_crc32c(length); bytes data[length];
    uint32 masked_crc32c(data) } per record, masked = rotr15(crc) + 0xa282ead8.  The length
    checksum is always verified, the payload checksum on request (a table-driven CRC in Python costs
    ~0.3 us per byte).
  * tf.train.Example: Example{Features features = 1}, Features{map<string, Feature> feature = 1},
    Feature{BytesList bytes_list = 1 | FloatList float_list = 2 | Int64List int64_list = 3}, each
    list {repeated value = 1} (numeric lists packed) -- read with the wire decoder of graphdef.py.
  * the encoded image (PNG for mnist / cifar10 / fashion_mnist, JPEG for celeb_a / lsun / imagenet):
    Pillow.

Layout of a TFDS data dir: <data_dir>/<tfds name>/[<config>/]<version>/<prefix>-<split>.tfrecord-
NNNNN-of-MMMMM.  load_split() globs for the split's shards under <data_dir>/<tfds name>, reads them
in shard order (shuffle_files=False in the reference) and returns the decoded, NOT yet cropped /
resized images -- the input of datasets.ImageDatasetV2._parse, exactly what the .npz source holds.

write_records() / make_example() emit the same format (tests, and converting other sources)."""
import glob
import io
import os
import struct

import numpy as np

from compare_gan_amd import graphdef as _pb

_MASK_DELTA = 0xA282EAD8


def _crc_table():
  poly, table = 0x82F63B78, []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ poly if c & 1 else c >> 1
    table.append(c)
  return table


_TABLE = _crc_table()


_NATIVE = [None]


def _native_crc32c():
  """cg_host_crc32c of libcgamd.so (slice-by-8, ~1 GB/s) when the library is there; False otherwise."""
  if _NATIVE[0] is None:
    try:
      from compare_gan_amd.hip import _lib
      _NATIVE[0] = _lib.load().cg_host_crc32c
    except Exception:  # pylint: disable=broad-except
      _NATIVE[0] = False
  return _NATIVE[0]


def crc32c_py(data):
  c = 0xFFFFFFFF
  for b in bytes(data):
    c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


def crc32c(data):
  fn = _native_crc32c()
  if fn:
    data = bytes(data)
    return int(fn(data, len(data), 0))
  return crc32c_py(data)


def masked_crc32c(data):
  c = crc32c(data)
  return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def read_records(path, verify_payload=False):
  """Yields the payload of every record of one TFRecord file."""
  with open(path, "rb") as f:
    while True:
      head = f.read(12)
      if not head:
        return
      if len(head) != 12:
        raise ValueError("%s: truncated record header" % path)
      length, len_crc = struct.unpack("<QI", head)
      if masked_crc32c(head[:8]) != len_crc:
        raise ValueError("%s: corrupt record length (checksum mismatch)" % path)
      data = f.read(length)
      tail = f.read(4)
      if len(data) != length or len(tail) != 4:
        raise ValueError("%s: truncated record" % path)
      if verify_payload and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
        raise ValueError("%s: corrupt record payload (checksum mismatch)" % path)
      yield data


def write_records(path, records):
  with open(path, "wb") as f:
    for data in records:
      head = struct.pack("<Q", len(data))
      f.write(head + struct.pack("<I", masked_crc32c(head)) + data +
              struct.pack("<I", masked_crc32c(data)))


def parse_example(buf):
  """tf.train.Example bytes -> {feature name: list of bytes | ndarray int64 | ndarray float32}."""
  out = {}
  for f1, _, features in _pb._fields(memoryview(buf)):   # pylint: disable=protected-access
    if f1 != 1:
      continue
    for f2, _, entry in _pb._fields(features):           # map entry  # pylint: disable=protected-access
      if f2 != 1:
        continue
      key, value = None, None
      for f3, _, v3 in _pb._fields(entry):               # pylint: disable=protected-access
        if f3 == 1:
          key = bytes(v3).decode()
        elif f3 == 2:
          for kind, _, lst in _pb._fields(v3):           # pylint: disable=protected-access
            if kind == 1:
              value = [bytes(v) for f5, _, v in _pb._fields(lst) if f5 == 1]  # pylint: disable=protected-access
            elif kind == 2:
              vals = []
              for f5, wt, v in _pb._fields(lst):         # pylint: disable=protected-access
                if f5 == 1:
                  vals += (list(np.frombuffer(bytes(v), "<f4")) if wt == 2
                           else [struct.unpack("<f", v)[0]])
              value = np.asarray(vals, dtype=np.float32)
            elif kind == 3:
              vals = []
              for f5, wt, v in _pb._fields(lst):         # pylint: disable=protected-access
                if f5 != 1:
                  continue
                if wt == 2:
                  pos = 0
                  while pos < len(v):
                    x, pos = _pb._varint(v, pos)         # pylint: disable=protected-access
                    vals.append(x)
                else:
                  vals.append(v)
              value = np.asarray([x - (1 << 64) if x >= (1 << 63) else x for x in vals],
                                 dtype=np.int64)
      if key is not None:
        out[key] = value
  return out


def _enc_len(field, payload):
  return _pb._enc_field(field, payload)   # pylint: disable=protected-access


def make_example(features):
  """{name: bytes | int | float | sequences of them} -> tf.train.Example bytes."""
  entries = b""
  for name, v in features.items():
    vals = v if isinstance(v, (list, tuple, np.ndarray)) else [v]
    first = vals[0]
    if isinstance(first, (bytes, bytearray)):
      feat = _enc_len(1, b"".join(_enc_len(1, bytes(x)) for x in vals))
    elif isinstance(first, (float, np.floating)):
      feat = _enc_len(2, _enc_len(1, np.asarray(vals, "<f4").tobytes()))
    else:
      packed = b"".join(_pb._enc_varint(int(x) & 0xFFFFFFFFFFFFFFFF) for x in vals)  # pylint: disable=protected-access
      feat = _enc_len(3, _enc_len(1, packed))
    entries += _enc_len(1, _enc_len(1, name.encode()) + _enc_len(2, feat))
  return _enc_len(1, entries)


def decode_image(data):
  """Encoded PNG / JPEG bytes -> uint8 [h, w, c] (c = 1 for greyscale)."""
  try:
    from PIL import Image
  except ImportError as e:
    raise RuntimeError("decoding TFDS image records needs Pillow") from e
  img = Image.open(io.BytesIO(data))
  if img.mode not in ("L", "RGB"):
    img = img.convert("RGB")
  a = np.asarray(img, dtype=np.uint8)
  return a[:, :, None] if a.ndim == 2 else a


# dataset name -> (TFDS name, train split, eval split, percent slice of the train records or None)
# (datasets.py:338-514; LSUN: `Split.TRAIN.subsplit([99, 1])` = records i with i % 100 < 99 / the
# others of every training shard, datasets.py:413-418)
TFDS = {
    "mnist": ("mnist", "train", "test", None),
    "fashion_mnist": ("fashion_mnist", "train", "test", None),
    "cifar10": ("cifar10", "train", "test", None),
    "celeb_a": ("celeb_a", "train", "test", None),
    "lsun-bedroom": ("lsun/bedroom", "train", "train", (99, 1)),
    "celeb_a_hq_128": ("celeb_a_hq/128", "train", "train", None),
    "imagenet_64": ("imagenet2012", "train", "validation", None),
    "imagenet_128": ("imagenet2012", "train", "validation", None),
    "imagenet_256": ("imagenet2012", "train", "validation", None),
    "imagenet_512": ("imagenet2012", "train", "validation", None),
}


def _version_key(path):
  """Sort key of a shard directory: its dotted version component (…/3.0.10 > …/3.0.9) as a tuple of
  ints, other path components as text."""
  key = []
  for part in path.split(os.sep):
    nums = part.split(".")
    if len(nums) >= 2 and all(n.isdigit() for n in nums):
      key.append((1, tuple(int(n) for n in nums), ""))
    else:
      key.append((0, (), part))
  return key


def shard_files(data_dir, tfds_name, split):
  """The split's shards under <data_dir>/<tfds name>: ONE directory -- with several installed
  versions / configs (…/3.0.9, …/3.0.10) the highest VERSION (compared as integer tuples, not as
  strings), as TFDS picks it -- never their concatenation."""
  root = os.path.join(data_dir, *tfds_name.split("/"))
  files = sorted(glob.glob(os.path.join(root, "**", "*-%s.tfrecord*" % split), recursive=True))
  files = [f for f in files if not f.endswith(".json")]
  if not files:
    return files
  last_dir = sorted(set(os.path.dirname(f) for f in files), key=_version_key)[-1]
  return [f for f in files if os.path.dirname(f) == last_dir]


def has_split(data_dir, name, training):
  if name not in TFDS:
    return False
  tname, tr, ev, _ = TFDS[name]
  return bool(shard_files(data_dir, tname, tr if training else ev))


def load_split(data_dir, name, training, max_examples=None, verify_payload=False):
  """(list of uint8 images, int32 labels) of dataset `name`'s train / eval split, file order."""
  tname, tr, ev, pct = TFDS[name]
  files = shard_files(data_dir, tname, tr if training else ev)
  if not files:
    raise ValueError("no TFRecord shards of %s (%s) under %s" % (name, tname, data_dir))
  images, labels = [], []
  seen = 0     # records of the earlier shards: the legacy mask does NOT restart at a shard boundary
  for path in files:
    # records are streamed (nothing but the decoded examples that are kept stays in memory).  The
    # legacy `subsplit([99, 1])` is a repeating 100-entry mask carried ACROSS the shards: shard s
    # starts at offset (records of shards 0..s-1) % 100 (TFDS compute_mask_offsets / _build_mask_ds),
    # so record i of shard s belongs to the first part iff (offset_s + i) % 100 < 99
    first = seen
    for i, data in enumerate(read_records(path, verify_payload)):
      seen = first + i + 1
      if pct is not None and (((first + i) % 100) < pct[0]) != bool(training):
        continue
      ex = parse_example(data)
      if "image" not in ex:
        raise ValueError("%s: record without an `image` feature (has %s)" % (path, sorted(ex)))
      images.append(decode_image(ex["image"][0]))
      lab = ex.get("label")
      labels.append(int(lab[0]) if lab is not None and len(lab) else 0)
      if max_examples is not None and len(images) >= max_examples:
        return images, np.asarray(labels, dtype=np.int32)
  return images, np.asarray(labels, dtype=np.int32)
