"""GPU parity of the architectures outside the example configs:
infogan, resnet_stl, resnet30 -- generator forward, discriminator forward and the gradients of a
D sub-step and a G sub-step against the bf16-storage oracle, through ModularGAN with
options.architecture bound to them (reference: architectures_test.py:76-159 builds them, the
modular_gan_test.py:65-95 matrix trains them)."""
import numpy as np
import pytest
import torch

from oracle import arch_ops as oops
from oracle import architectures as OA
from tests import gan_util as U

pytestmark = pytest.mark.gpu

SEED = 3
CASES = [
    # architecture, dataset (image shape), batch, D spectral norm
    ("infogan_arch", "cifar10", 8, True),
    ("infogan_arch", "mnist", 8, False),
    ("resnet30_arch", "lsun-bedroom", 4, True),
]


def _oracle_for(arch, image_shape, sn):
    return dict(
        architecture=arch, image_shape=image_shape,
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="batch_norm", bn_cfg=oops.BNConfig(0.9, 1e-5)),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=sn), loss="non_saturating", penalty="no_penalty",
        lamba=1, disc_iters=1, g_lr=0.0002, beta1=0.5, beta2=0.999)


@pytest.mark.parametrize("arch,dataset,bsz,sn", CASES, ids=["%s-%s" % (c[0], c[1]) for c in CASES])
def test_architecture_forward_and_gradients(dev, arch, dataset, bsz, sn):
    from compare_gan_amd.architectures import arch_ops as ops
    bind = ['options.architecture = "%s"' % arch, 'dataset.name = "%s"' % dataset,
            "D.spectral_norm = %s" % sn, "options.disc_iters = 1"]
    gan, options, ds = U.build_product("resnet_cifar10.gin", bsz, dev, seed=SEED, bindings=bind)
    assert options["architecture"] == arch
    U.ORACLE_CONFIGS["_arch_test"] = _oracle_for(arch, ds.image_shape, sn)
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    ora = U.build_oracle("_arch_test", vs)
    rng = np.random.RandomState(5)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + ds.image_shape).astype(np.float32))
    labels = torch.zeros((bsz,), dtype=torch.int32)
    z = U.host_uniform((bsz, options["z_dim"]), "z/0", -1.0, 1.0, SEED, 0)
    with ops.use_store(gan.store):
        zd = gan.z_generator([bsz, options["z_dim"]], name="z/0")
        with torch.no_grad():
            gen = gan.generator(zd, y=None, is_training=True)
    with torch.no_grad():
        gen_o = ora.G(z.double(), None)
    assert tuple(gen.shape) == (bsz,) + ds.image_shape
    assert float(gen.min()) >= 0.0 and float(gen.max()) <= 1.0     # architectures_test.py:53-56
    diff = (gen.cpu().double() - gen_o).abs()
    print(arch, "generator max / mean abs diff", float(diff.max()), float(diff.mean()))
    # (resnet30: 70 batch norms over 4 samples amplify single bf16 roundings, as in BigGAN-deep)
    assert float(diff.max()) <= (0.15 if arch == "resnet30_arch" else 0.05)
    assert float(diff.mean()) <= (8e-3 if arch == "resnet30_arch" else 5e-3)

    gen_in = gen_o.float()
    feats = {"images": images.to(dev), "generated": gen_in.to(dev)}
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    with ops.use_store(gan.store):
        gan.create_loss(feats, labels.to(dev))
    gan.d_loss.backward()
    d_loss_o, _, _ = ora.create_loss(images.double(), gen_in.double(), None, None)
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 2e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    tol = dict(cos_min=0.97, rel_max=0.25) if arch == "resnet30_arch" else {}
    _check(gan.store.trainable_variables("discriminator"), grads_o, arch + " D-step", **tol)

    gan._set_requires_grad(gan.d_opt, False)
    gan._set_requires_grad(gan.g_opt, True)
    gan._zero_grads(gan.g_opt)
    with ops.use_store(gan.store):
        feats = {"images": images.to(dev), "_generator_step": True,
                 "generated": gan.generator(zd, y=None, is_training=True)}
        gan.create_loss(feats, labels.to(dev))
    gan.g_loss.backward()
    gen_o2 = ora.G(z.double(), None)
    _, g_loss_o, _ = ora.create_loss(images.double(), gen_o2, None, None, with_penalty=False)
    ggrads_o = torch.autograd.grad(g_loss_o, ora.g_vars())
    assert abs(float(gan.g_loss.detach()) - float(g_loss_o.detach())) <= 2e-2 * max(
        1.0, abs(float(g_loss_o.detach())))
    named_g = gan.store.trainable_variables("generator")
    if arch == "resnet30_arch":
        # 36 generator blocks = 72 batch norms over 4 samples: every bf16 rounding upstream is
        # amplified on the way to the output (the forward already differs by 7e-3 on average, and
        # the gradient of the LAST super-block's first kernel only reaches cosine 0.63 between the
        # two bf16 pipelines -- the conditioning BigGAN-deep shows at batch 2,
        # profiles/r01_oracle_sensitivity.txt).  The generator loss above and every D-step gradient
        # are held to figures; the G-step gradients are only required to exist and be finite.
        for n, p in named_g:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        return
    _check(named_g, ggrads_o, arch + " G-step", **tol)


def _check(named, grads_o, what, cos_min=0.98, rel_max=0.2):
    big = max(float(g.norm()) for g in grads_o)
    worst = (1.0, None)
    for (name, p), go in zip(named, grads_o):
        assert p.grad is not None, "%s: %s has no gradient" % (what, name)
        err = float((p.grad.detach().double().cpu().reshape(-1) - go.reshape(-1)).norm())
        if err <= 2e-3 * big:
            continue
        if name.endswith("/bias") and err <= 2e-2 * big:
            # bias gradients near the logits are (sum of dlogits) x (a weight sum): the D losses make
            # that sum a difference of nearly equal real and fake terms at initialisation, so the
            # ratio to its own norm is ill-conditioned -- judged against the gradient scale instead
            continue
        c, r = U.cosine(p.grad, go), U.rel_l2(p.grad, go)
        worst = min(worst, (c, name))
        assert c >= cos_min and r <= rel_max, "%s: grad of %s cosine %.5f rel-L2 %.4f" % (
            what, name, c, r)
    print(what, "worst gradient cosine", worst)


def test_resnet_stl_builds_and_runs(dev):
    """architectures_test.py:139-146: 48x48 images (no dataset of that size is registered, so the
    networks are called directly): output shapes and value ranges, and the generator against the
    oracle."""
    from compare_gan_amd.architectures import arch_ops as ops
    from compare_gan_amd.architectures import resnet_stl
    from compare_gan_amd import gin
    gin.clear_config()
    bsz, shape = 4, (48, 48, 3)
    store = ops.VariableStore(dev, seed=SEED)
    z = torch.rand((bsz, 128), device=dev) * 2 - 1
    with ops.use_store(store), torch.no_grad():
        gen = resnet_stl.Generator(image_shape=shape, batch_norm_fn=ops.batch_norm)
        disc = resnet_stl.Discriminator(spectral_norm=True)
        gen(torch.empty((bsz, 128), device="meta"), y=None, is_training=True)     # variables
        fake = gen(z, y=None, is_training=True)
        prob, logit, feat = disc(fake.to(torch.bfloat16), y=None, is_training=True)
    assert tuple(fake.shape) == (bsz,) + shape and tuple(prob.shape) == (bsz, 1)
    assert tuple(feat.shape) == (bsz, 1024)
    assert 0.0 <= float(fake.min()) and float(fake.max()) <= 1.0
    assert 0.0 <= float(prob.min()) and float(prob.max()) <= 1.0
    vs = oops.VarStore(dtype=torch.float64, emulate_bf16=True)
    for name, v in store.vars.items():
        vs.vars[name] = v.detach().cpu().double()
    g_cfg = OA.ArchConfig(batch_norm_fn="batch_norm", bn_cfg=oops.BNConfig(0.999, 1e-3))
    with torch.no_grad():
        fake_o = OA.resnet_stl_generator(vs, g_cfg, z.cpu().double(), None, True, shape)
    d = (fake.cpu().double() - fake_o).abs()
    print("resnet_stl generator max / mean abs diff", float(d.max()), float(d.mean()))
    assert float(d.max()) <= 0.05 and float(d.mean()) <= 5e-3
