"""compare_gan_amd/tfrecord.py: the TFDS record path of datasets.py:229-251 without TensorFlow.
A synthetic TFDS data dir is written in the real storage format (TFRecord framing with masked
CRC32C, tf.train.Example, PNG), read back, and driven through ImageDatasetV2's train / eval
pipeline; the CRC and the Example encoding are pinned to published known answers."""
import io
import os
import struct

import numpy as np
import pytest

from compare_gan_amd import datasets
from compare_gan_amd import tfrecord


def _png(arr):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr.squeeze(-1) if arr.shape[-1] == 1 else arr).save(buf, format="PNG")
    return buf.getvalue()


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors (iSCSI CRC32C) and the classic check value
    assert tfrecord.crc32c(b"123456789") == 0xE3069283
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA
    assert tfrecord.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfrecord.crc32c(bytes(range(32))) == 0x46DD794E
    # TFRecord's mask: rotate right by 15, add the delta
    c = tfrecord.crc32c(b"123456789")
    assert tfrecord.masked_crc32c(b"123456789") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_example_round_trip_and_wire_bytes():
    ex = tfrecord.make_example({"image": b"\x89PNG...", "label": 7, "weights": [0.5, -2.0],
                                "ids": [3, -1, 1 << 40]})
    got = tfrecord.parse_example(ex)
    assert got["image"] == [b"\x89PNG..."]
    assert got["label"].tolist() == [7] and got["label"].dtype == np.int64
    assert got["ids"].tolist() == [3, -1, 1 << 40]
    np.testing.assert_array_equal(got["weights"], np.asarray([0.5, -2.0], np.float32))
    # the bytes of a one-feature Example written out by hand from example.proto / feature.proto:
    # Example{1: Features{1: entry{1: "label", 2: Feature{3: Int64List{1: packed [5]}}}}}
    #   Int64List 0A 01 05 | Feature 1A 03 .. | entry 0A 05 "label" 12 05 .. | Features 0A 0E .. | Example 0A 10 ..
    want = bytes([0x0A, 0x10, 0x0A, 0x0E, 0x0A, 0x05]) + b"label" + bytes([0x12, 0x05, 0x1A, 0x03, 0x0A, 0x01, 0x05])
    assert tfrecord.make_example({"label": 5}) == want
    assert tfrecord.parse_example(tfrecord.make_example({"label": 5}))["label"].tolist() == [5]


def test_tfds_dir_feeds_the_dataset_pipeline(tmp_path):
    rng = np.random.RandomState(0)
    root = tmp_path / "tfds"
    d = root / "cifar10" / "3.0.2"
    d.mkdir(parents=True)
    train = [(rng.randint(0, 256, size=(32, 32, 3)).astype(np.uint8), int(rng.randint(10))) for _ in range(23)]
    test = [(rng.randint(0, 256, size=(32, 32, 3)).astype(np.uint8), int(rng.randint(10))) for _ in range(9)]
    for split, items, shards in (("train", train, 3), ("test", test, 1)):
        for s in range(shards):
            part = items[s::shards] if False else items[s * len(items) // shards:(s + 1) * len(items) // shards]
            tfrecord.write_records(
                str(d / ("cifar10-%s.tfrecord-%05d-of-%05d" % (split, s, shards))),
                [tfrecord.make_example({"image": _png(im), "label": lab, "id": b"x"}) for im, lab in part])
    imgs, labs = tfrecord.load_split(str(root), "cifar10", True, verify_payload=True)
    assert len(imgs) == 23 and labs.tolist() == [l for _, l in train]
    for a, (b, _) in zip(imgs, train):
        np.testing.assert_array_equal(a, b)                      # PNG is lossless: exact pixels, file order
    # a corrupted payload is caught when verification is on
    path = str(d / "cifar10-test.tfrecord-00000-of-00001")
    raw = bytearray(open(path, "rb").read())
    raw[40] ^= 0xFF
    bad = str(tmp_path / "bad.tfrecord")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        list(tfrecord.read_records(bad, verify_payload=True))
    # through the dataset object: eval split unshuffled, train batches shuffled + repeated
    datasets.use_data_dir(str(root), shuffle_buffer_size=8)
    try:
        ds = datasets.get_dataset("cifar10", seed=3)
        ev = ds.eval_images(9)
        np.testing.assert_allclose(ev, np.stack([im for im, _ in test]).astype(np.float32) / 255.0)
        xb, yb = next(ds.train_batches(16, seed=5))
        assert xb.shape == (16, 32, 32, 3) and xb.dtype == np.float32 and yb.dtype == np.int32
        pool = {im.tobytes(): lab for im, lab in train}
        for x, y in zip(xb, yb):
            key = np.round(x * 255.0).astype(np.uint8).tobytes()
            assert pool[key] == int(y)                           # every example is a training record with its label
    finally:
        datasets.use_data_dir(None)


def test_lsun_subsplit_takes_the_tail_of_every_shard(tmp_path):
    root = tmp_path / "tfds"
    d = root / "lsun" / "bedroom" / "0.1.1"
    d.mkdir(parents=True)
    rng = np.random.RandomState(1)
    for s in range(2):
        ims = [rng.randint(0, 256, size=(140, 150, 3)).astype(np.uint8) for _ in range(100)]
        tfrecord.write_records(str(d / ("lsun-train.tfrecord-%05d-of-00002" % s)),
                               [tfrecord.make_example({"image": _png(im)}) for im in ims[:100]])
    tr, _ = tfrecord.load_split(str(root), "lsun-bedroom", True, max_examples=None)
    ev, lab = tfrecord.load_split(str(root), "lsun-bedroom", False)
    assert len(tr) == 198 and len(ev) == 2 and lab.tolist() == [0, 0]
    assert tr[0].shape == (140, 150, 3)


def test_lsun_subsplit_mask_repeats_every_100_records_and_one_version_is_read(tmp_path):
    """Legacy `Split.TRAIN.subsplit([99, 1])` (datasets.py:413-418) is a repeating 100-record mask per
    shard: records 99 and 199 of a 250-record shard are the evaluation part.  With two installed
    versions of a dataset only the highest one is read (never their concatenation)."""
    root = tmp_path / "tfds"
    rng = np.random.RandomState(2)
    for version, n in (("0.1.0", 120), ("0.1.1", 250)):
        d = root / "lsun" / "bedroom" / version
        d.mkdir(parents=True)
        ims = [np.full((8, 8, 3), i % 256, dtype=np.uint8) for i in range(n)]
        tfrecord.write_records(str(d / "lsun-train.tfrecord-00000-of-00001"),
                               [tfrecord.make_example({"image": _png(im)}) for im in ims])
    del rng
    tr, _ = tfrecord.load_split(str(root), "lsun-bedroom", True)
    ev, _ = tfrecord.load_split(str(root), "lsun-bedroom", False)
    assert len(tr) == 248 and len(ev) == 2
    assert [int(im[0, 0, 0]) for im in ev] == [99, 199]
    # max_examples stops the decoding early (the evaluation only takes the first N)
    few, _ = tfrecord.load_split(str(root), "lsun-bedroom", True, max_examples=5)
    assert len(few) == 5 and [int(im[0, 0, 0]) for im in few] == [0, 1, 2, 3, 4]


def test_lsun_subsplit_mask_is_carried_across_shards_and_versions_sort_numerically(tmp_path):
    """legacy TFDS carries the 100-entry `subsplit([99, 1])` mask across shard
    boundaries -- shard s starts at offset (records of the earlier shards) % 100
    (compute_mask_offsets / _build_mask_ds) -- so with shards of 130 and 90 records the evaluation
    part is the GLOBAL records 99 and 199 (local index 69 of the second shard), not local index 99
    of each shard.  Version directories compare as integer tuples: 3.0.10 is newer than 3.0.9."""
    root = tmp_path / "tfds"
    for version, sizes in (("3.0.9", (40,)), ("3.0.10", (130, 90))):
        d = root / "lsun" / "bedroom" / version
        d.mkdir(parents=True)
        g = 0
        for s, n in enumerate(sizes):
            ims = [np.full((8, 8, 3), (g + i) % 256, dtype=np.uint8) for i in range(n)]
            g += n
            tfrecord.write_records(str(d / ("lsun-train.tfrecord-%05d-of-%05d" % (s, len(sizes)))),
                                   [tfrecord.make_example({"image": _png(im)}) for im in ims])
    files = tfrecord.shard_files(str(root), "lsun/bedroom", "train")
    assert len(files) == 2 and all("3.0.10" in f for f in files)
    tr, _ = tfrecord.load_split(str(root), "lsun-bedroom", True)
    ev, _ = tfrecord.load_split(str(root), "lsun-bedroom", False)
    assert len(tr) == 218 and [int(im[0, 0, 0]) for im in ev] == [99, 199]
    assert 99 not in [int(im[0, 0, 0]) for im in tr] and 199 not in [int(im[0, 0, 0]) for im in tr]
