"""TF-1 tensor bundles (compare_gan_amd/tf_checkpoint.py): the checkpoint format either side of the
hot path (modular_gan.py:266-285, runner_lib.py:193-206).  CPU: table / bundle wire
format, checksums, names, the committed fixture.  GPU: export -> import of a trained ModularGAN."""
import os
import struct

import numpy as np
import pytest
import torch

from compare_gan_amd import tf_checkpoint as tfc
from compare_gan_amd import tfrecord

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_checkpoint", "model.ckpt-7")


def test_crc32c_known_answers_and_native_equals_python():
    # RFC 3720 B.4 check values of CRC32C (Castagnoli)
    assert tfrecord.crc32c(b"123456789") == 0xE3069283
    assert tfrecord.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert tfrecord.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfrecord.crc32c(bytes(range(32))) == 0x46DD794E
    rng = np.random.RandomState(3)
    for n in (0, 1, 7, 8, 9, 63, 64, 1000, 4097):
        data = rng.randint(0, 256, size=n).astype(np.uint8).tobytes()
        assert tfrecord.crc32c(data) == tfrecord.crc32c_py(data), n
    # the masked form TFRecords and table blocks store (leveldb crc32c::Mask)
    c = tfrecord.crc32c(b"abc")
    assert tfrecord.masked_crc32c(b"abc") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _many_tensors():
    rng = np.random.RandomState(5)
    t = {}
    for i in range(70):   # long common prefixes: real prefix compression, several blocks, restarts
        t["discriminator/B%d/same_conv%d/kernel" % (i // 8, i % 8)] = rng.randn(3, 3, 2, i % 5 + 1).astype(np.float32)
        t["discriminator/B%d/same_conv%d/kernel/d_opt" % (i // 8, i % 8)] = rng.randn(3, 3, 2, i % 5 + 1).astype(np.float32)
    t["global_step"] = np.asarray(123456789012, dtype=np.int64)
    t["flag"] = np.asarray([True, False, True])
    t["generator/fc_noise/bias"] = rng.randn(17).astype(np.float64)
    t["counts"] = rng.randint(-5, 5, size=(2, 3)).astype(np.int32)
    t["empty"] = np.zeros((0, 4), dtype=np.float32)
    return t


def test_bundle_round_trip_and_table_structure(tmp_path):
    prefix = str(tmp_path / "model.ckpt-5")
    want = _many_tensors()
    tfc.write_bundle(prefix, want, block_bytes=300)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57        # leveldb kTableMagicNumber
    table = tfc.read_table(prefix + ".index")
    keys = [k for k, _ in table]
    assert keys == sorted(keys) and keys[0] == b"" and len(keys) == len(want) + 1
    header, entries = tfc.read_index(prefix)
    assert header["num_shards"] == 1 and set(entries) == set(want)
    assert entries["global_step"]["shape"] == [] and entries["counts"]["dtype"] == 3
    got = tfc.read_bundle(prefix, verify_tensors=True)
    assert set(got) == set(want)
    for n in want:
        assert got[n].dtype == want[n].dtype and got[n].shape == want[n].shape, n
        np.testing.assert_array_equal(got[n], want[n])
    only = tfc.read_bundle(prefix, names={"counts"})
    assert list(only) == ["counts"]
    assert tfc.latest_checkpoint(str(tmp_path)) == prefix
    # a flipped bit in a data block is caught by the block checksum; one in the tensor bytes by the
    # entry's checksum
    bad = bytearray(raw)
    bad[10] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="CRC"):
        tfc.read_index(prefix)
    open(prefix + ".index", "wb").write(raw)
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[5] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="CRC"):
        tfc.read_bundle(prefix, verify_tensors=True)
    # not a table at all
    open(str(tmp_path / "junk.index"), "wb").write(b"\x00" * 100)
    with pytest.raises(ValueError, match="magic"):
        tfc.read_table(str(tmp_path / "junk.index"))


def test_committed_fixture_reads_back():
    """tests/golden/tf_checkpoint (scripts/make_tf_checkpoint_fixture.py): values are a function of
    the name; a change of the reader or of the format constants shows up here."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_fixture", os.path.join(os.path.dirname(GOLDEN), "..", "..", "..", "scripts",
                                     "make_tf_checkpoint_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = mod.tensors()
    got = tfc.read_bundle(GOLDEN, verify_tensors=True)
    assert set(got) == set(want) and len(got) == 23
    for n in want:
        np.testing.assert_array_equal(got[n], want[n])
    assert got["global_step"].shape == () and int(got["global_step"]) == 7 and int(got["global_step_disc"]) == 35


def test_variable_names_map_onto_the_state_dict():
    """tf.train.AdamOptimizer(lr, name="d_opt") names its slots <var>/d_opt and <var>/d_opt_1
    (modular_gan.py:607,613); everything else keeps its name; the power accumulators are dropped."""
    k = tfc.state_key
    assert k("discriminator/B1/down_conv2/kernel") == "discriminator/B1/down_conv2/kernel"
    assert k("discriminator/B1/down_conv2/kernel/d_opt") == "discriminator/B1/down_conv2/kernel/d_opt/Adam"
    assert k("discriminator/B1/down_conv2/kernel/d_opt_1") == "discriminator/B1/down_conv2/kernel/d_opt/Adam_1"
    assert k("generator/fc_noise/bias/g_opt_1") == "generator/fc_noise/bias/g_opt/Adam_1"
    assert k("generator/fc_noise/kernel/ExponentialMovingAverage") == "generator/fc_noise/kernel/ExponentialMovingAverage"
    assert k("discriminator/B1/down_conv2/kernel/u_var") == "discriminator/B1/down_conv2/kernel/u_var"
    assert k("beta1_power") is None and k("beta2_power_1") is None
    for name in ("global_step", "generator/fc_noise/kernel/g_opt", "discriminator/x/bias/d_opt_1",
                 "generator/B1/bn1/accu/accu_mean"):
        assert tfc.tf_name(tfc.state_key(name)) == name


@pytest.mark.gpu
def test_export_import_of_a_trained_gan(tmp_path):
    """A ModularGAN after one train step -> TF-named bundle -> a freshly built ModularGAN with other
    initial weights: every entry of state_dict() (variables, power-iteration vectors, moving
    averages, Adam slots, step counters) bit-identical, the next train step bit-identical, and the
    bundle carries exactly the reference's variable names (no `/Adam` spellings)."""
    from tests import gan_util as U
    dev = torch.device("cuda:0")
    config, bsz = "resnet_cifar10.gin", 8
    gan, options, dataset = U.build_product(config, bsz, dev, seed=3)
    nsub = options["disc_iters"] + 1
    rng = np.random.RandomState(1)
    batches = [(torch.from_numpy(rng.uniform(size=(nsub * bsz, 32, 32, 3)).astype(np.float32)).to(dev),
                torch.ones((nsub * bsz,), dtype=torch.int32, device=dev)) for _ in range(2)]
    gan.train_step(*batches[0])
    prefix = str(tmp_path / "model.ckpt-1")
    tfc.export_tf_checkpoint(gan, prefix)
    _, entries = tfc.read_index(prefix)
    assert not any(n.endswith("/Adam") or n.endswith("/Adam_1") for n in entries)
    assert "discriminator/B1/down_conv2/kernel/d_opt_1" in entries and "global_step_disc" in entries
    assert entries["generator/fc_noise/kernel"]["shape"] == [128, 4096]       # [in, out], no transposes
    assert entries["discriminator/B1/down_conv2/kernel"]["shape"] == [3, 3, 128, 128]   # HWIO
    other, _, _ = U.build_product(config, bsz, dev, seed=11)
    report = tfc.import_tf_checkpoint(other, prefix)
    assert not report["missing"] and not report["unexpected"]
    a, b = gan.state_dict(), other.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    oa, ob = gan.train_step(*batches[1]), other.train_step(*batches[1])
    assert torch.equal(oa["g_loss"], ob["g_loss"])
    a, b = gan.state_dict(), other.state_dict()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # a checkpoint without a variable of the model is refused
    partial = {tfc.tf_name(k): v.cpu().numpy() for k, v in a.items() if "fc_noise/kernel" not in k}
    tfc.write_bundle(str(tmp_path / "model.ckpt-2"), partial)
    with pytest.raises(KeyError):
        tfc.import_tf_checkpoint(other, str(tmp_path / "model.ckpt-2"))
