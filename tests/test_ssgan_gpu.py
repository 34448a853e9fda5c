"""SSGAN (gans/ssgan.py:39-226) on the MI355X against the oracle: losses, rotation losses and the
gradients of a D sub-step and a G sub-step, then the reference's own smoke test -- one training
step for each architecture of ssgan_test.py:35 (ssgan_test.py:41-66)."""
import numpy as np
import pytest
import torch

from tests import gan_util as U

pytestmark = pytest.mark.gpu
SEED = 3


def _build(dev, bsz, extra=()):
    from compare_gan_amd.gans import ssgan  # noqa: F401  (registers SSGAN)
    bind = ("options.gan_class = @SSGAN", "SSGAN.rotated_batch_size = 4",
            "options.disc_iters = 1") + tuple(extra)
    return U.build_product("resnet_cifar10.gin", bsz, dev, seed=SEED, bindings=bind)


def test_rotate_images_matches_the_reference_definition(dev):
    from compare_gan_amd.gans import ssgan
    from oracle import modular_gan as omg
    x = torch.rand((3, 8, 8, 3), device=dev)
    got = ssgan.rotate_images(x, (1, 2, 3))
    ref = omg.rotate_images(x.cpu(), (1, 2, 3))
    assert torch.equal(got.cpu(), ref) and tuple(got.shape) == (9, 8, 8, 3)
    assert np.array_equal(got[:3].cpu().numpy(), np.rot90(x.cpu().numpy(), 1, axes=(1, 2)))


def test_ssgan_losses_and_gradients(dev):
    from compare_gan_amd.architectures import arch_ops as ops
    from oracle import modular_gan as omg
    bsz = 4
    gan, options, dataset = _build(dev, bsz)
    assert type(gan).__name__ == "SSGAN"
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    kw = dict(U.ORACLE_CONFIGS["resnet_cifar10.gin"])
    kw.update(disc_iters=1)
    kw["g_cfg"], kw["d_cfg"] = kw["g_cfg"](), kw["d_cfg"]()
    arch = kw.pop("architecture")
    ora = omg.OracleSSGAN(vs, arch, rotated_batch_size=4, **kw)
    rng = np.random.RandomState(9)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    labels = torch.zeros((bsz,), dtype=torch.int32)
    z = U.host_uniform((bsz, 128), "z/0", -1.0, 1.0, SEED, 0)
    with torch.no_grad():
        gen_o = ora.G(z.double(), None)
    gen_in = gen_o.float()

    # ---- D sub-step ----
    feats = {"images": images.to(dev), "generated": gen_in.to(dev)}
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    with ops.use_store(gan.store):
        gan.create_loss(feats, labels.to(dev))
    gan.d_loss.backward()
    d_loss_o, _, _ = ora.create_loss(images.double(), gen_in.double(), None, None)
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    print("ssgan d_loss", float(gan.d_loss.detach()), float(d_loss_o.detach()), "c_real",
          float(gan.c_real_loss), ora.c_real_loss)
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 2e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    assert abs(float(gan.c_real_loss) - ora.c_real_loss) <= 2e-2 * max(1.0, ora.c_real_loss)
    named = [(n, dict(gan.store.trainable_variables())[n]) for n in ora.d_var_names()]
    assert any(n.startswith("discriminator_rotation/") for n, _ in named)
    _check(named, grads_o, "ssgan D-step")

    # ---- G sub-step: the rotation loss on the rotated FAKE images reaches the generator ----
    gan._set_requires_grad(gan.d_opt, False)
    gan._set_requires_grad(gan.g_opt, True)
    gan._zero_grads(gan.g_opt)
    with ops.use_store(gan.store):
        zd = gan.z_generator([bsz, 128], name="z/0")
        feats = {"images": images.to(dev), "_generator_step": True,
                 "generated": gan.generator(zd, y=None, is_training=True)}
        gan.create_loss(feats, labels.to(dev))
    gan.g_loss.backward()
    gen_o2 = ora.G(z.double(), None)
    _, g_loss_o, _ = ora.create_loss(images.double(), gen_o2, None, None, with_penalty=False)
    ggrads_o = torch.autograd.grad(g_loss_o, ora.g_vars())
    print("ssgan g_loss", float(gan.g_loss.detach()), float(g_loss_o.detach()), "c_fake",
          float(gan.c_fake_loss), ora.c_fake_loss)
    assert abs(float(gan.g_loss.detach()) - float(g_loss_o.detach())) <= 2e-2 * max(
        1.0, abs(float(g_loss_o.detach())))
    _check(gan.store.trainable_variables("generator"), ggrads_o, "ssgan G-step")


def _check(named, grads_o, what, cos_min=0.98, rel_max=0.2):
    big = max(float(g.norm()) for g in grads_o)
    worst = (1.0, None)
    for (name, p), go in zip(named, grads_o):
        assert p.grad is not None, "%s: %s has no gradient" % (what, name)
        if float((p.grad.detach().double().cpu().reshape(-1) - go.reshape(-1)).norm()) <= 2e-3 * big:
            continue
        c, r = U.cosine(p.grad, go), U.rel_l2(p.grad, go)
        worst = min(worst, (c, name))
        assert c >= cos_min and r <= rel_max, "%s: grad of %s cosine %.5f rel-L2 %.4f" % (
            what, name, c, r)
    print(what, "worst gradient cosine", worst)


@pytest.mark.parametrize("arch,dataset", [("resnet_cifar_arch", "cifar10"),
                                          ("sndcgan_arch", "cifar10"),
                                          ("resnet5_arch", "cifar10")])
def test_ssgan_single_training_step(dev, arch, dataset):
    """ssgan_test.py:41-66: batch 2, rotated_batch_size 4, hinge loss; the step runs, moves both
    networks and the rotation head, and leaves finite losses."""
    bsz = 2
    gan, options, ds = _build(dev, bsz, ('options.architecture = "%s"' % arch,
                                         'dataset.name = "%s"' % dataset, "loss.fn = @hinge"))
    before = {n: v.detach().clone() for n, v in gan.store.trainable_variables()}
    it = ds.train_batches(2 * bsz, seed=1)
    images, labels = next(it)
    out = gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    assert np.isfinite(float(out["g_loss"])) and np.isfinite(float(out["d_losses"][0]))
    # every kernel moves (a bias may legitimately see a zero gradient: in front of a batch norm, or
    # the last bias under the hinge loss when every example is inside the margin)
    moved = {n: not torch.equal(v, before[n]) for n, v in gan.store.trainable_variables()
             if n.endswith("/kernel")}
    assert moved["discriminator_rotation/score_classify/kernel"]
    assert all(moved.values()), [n for n, m in moved.items() if not m][:5]
    assert int(gan.global_step.item()) == 1
