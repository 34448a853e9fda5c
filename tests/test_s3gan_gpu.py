"""S3GAN (gans/s3gan.py:39-321) on the MI355X: losses and gradients of a D sub-step and a G
sub-step against the oracle with every head active (projection on inferred labels, predictor,
rotation; some real examples unlabelled), then the reference's own smoke test over its five head
configurations (s3gan_test.py:38-72)."""
import numpy as np
import pytest
import torch

from oracle import arch_ops as oops
from oracle import architectures as OA
from tests import gan_util as U

pytestmark = pytest.mark.gpu
SEED = 3


def _build(dev, bsz, dataset, use_predictor, project_y, self_supervision, ch=16):
    from compare_gan_amd.gans import s3gan  # noqa: F401  (registers S3GAN)
    bind = ("options.gan_class = @S3GAN", "S3GAN.rotated_batch_fraction = 2",
            "S3GAN.use_predictor = %s" % use_predictor, "S3GAN.project_y = %s" % project_y,
            'S3GAN.self_supervision = "%s"' % self_supervision, 'dataset.name = "%s"' % dataset,
            "resnet_biggan.Generator.ch = %d" % ch, "resnet_biggan.Discriminator.ch = %d" % ch,
            "loss.fn = @hinge", "options.disc_iters = 1")
    return U.build_product("biggan_imagenet128.gin", bsz, dev, seed=SEED, bindings=bind)


def test_s3gan_labels_and_weighted_xent_kernels(dev):
    from compare_gan_amd.hip import kernels as K
    g = torch.Generator().manual_seed(4)
    n, k = 6, 10
    aux = torch.randn(n, k, generator=g)
    lab = torch.tensor([3, -1, 7, -1, 0, -1])
    y = torch.zeros(n, k)
    y[lab >= 0, lab[lab >= 0]] = 1.0
    for soft in (False, True):
        y_out, avail = K.s3gan_labels(aux.to(dev), y.to(torch.bfloat16).to(dev), soft)
        pred = torch.softmax(aux, 1) if soft else torch.nn.functional.one_hot(aux.argmax(1), k).float()
        want = torch.where((lab >= 0).unsqueeze(1), y, pred)
        assert torch.equal(avail.cpu(), (lab >= 0).float())
        assert float((y_out.float().cpu() - want).abs().max()) <= (4e-3 if soft else 0.0)
    w = (lab >= 0).float()
    logits = aux.clone().requires_grad_(True)
    ce = -(y * torch.log_softmax(logits, 1)).sum(1)
    ref = (w * ce).sum() / 3.0
    ref.backward()
    loss, dl = K.softmax_xent_weighted(aux.to(dev), y.to(torch.bfloat16).to(dev), w.to(dev))
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, float(ref))
    assert float((dl.cpu() - logits.grad).abs().max()) <= 1e-6
    loss0, _ = K.softmax_xent_weighted(aux.to(dev), y.to(torch.bfloat16).to(dev),
                                       torch.zeros(n, device=dev))
    assert float(loss0) == 0.0      # no labelled example: div_no_nan


def test_s3gan_losses_and_gradients(dev):
    from compare_gan_amd.architectures import arch_ops as ops
    from oracle import modular_gan as omg
    bsz = 8
    gan, options, dataset = _build(dev, bsz, "cifar10", True, True, "rotation")
    assert type(gan).__name__ == "S3GAN" and gan.conditional
    vs = U.mirror_to_oracle(gan, emulate_bf16=True)
    sn = oops.SNConfig(singular_value="auto")
    ora = omg.OracleS3GAN(
        vs, "resnet_biggan_arch",
        g_cfg=OA.ArchConfig(batch_norm_fn="conditional_batch_norm", spectral_norm=True,
                            bn_cfg=oops.BNConfig(0.9, 1e-5, use_moving_averages=False), sn_cfg=sn,
                            hierarchical_z=True, embed_y=True, ch=16),
        d_cfg=OA.ArchConfig(spectral_norm=True, sn_cfg=sn, project_y=True, ch=16),
        image_shape=dataset.image_shape, loss="hinge", penalty="no_penalty", lamba=1, disc_iters=1,
        conditional=True, num_classes=dataset.num_classes, g_lr=0.0001, d_lr=0.0005, beta1=0.0,
        beta2=0.999, g_use_ema=True, project_y=True, use_predictor=True,
        self_supervision="rotation", rotated_batch_fraction=2)
    rng = np.random.RandomState(9)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    labels = torch.tensor([3, -1, 7, 1, -1, 0, 9, -1], dtype=torch.int32)   # three unlabelled
    sampled = torch.tensor([5, 2, 8, 0, 4, 4, 1, 6], dtype=torch.int32)
    z = U.host_normal((bsz, options["z_dim"]), "z/0", 0.0, 1.0, SEED, 0)
    with torch.no_grad():
        gen_o = ora.G(z.double(), ora.one_hot(sampled))
    gen_in = gen_o.float()

    feats = {"images": images.to(dev), "generated": gen_in.to(dev), "sampled_labels": sampled.to(dev)}
    gan._set_requires_grad(gan.g_opt, False)
    gan._zero_grads(gan.d_opt)
    with ops.use_store(gan.store):
        gan.create_loss(feats, labels.to(dev))
    gan.d_loss.backward()
    d_loss_o, _, _ = ora.create_loss(images.double(), gen_in.double(), labels, sampled)
    grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
    print("s3gan d_loss", float(gan.d_loss.detach()), float(d_loss_o.detach()), "rot",
          float(gan.rot_real_loss), ora.rot_real_loss, "class", float(gan.class_loss_real),
          ora.class_loss_real)
    assert abs(float(gan.d_loss.detach()) - float(d_loss_o.detach())) <= 3e-2 * max(
        1.0, abs(float(d_loss_o.detach())))
    assert abs(float(gan.class_loss_real) - ora.class_loss_real) <= 3e-2 * max(1.0, ora.class_loss_real)
    named = [(n, dict(gan.store.trainable_variables())[n]) for n in ora.d_var_names()]
    heads = [n for n, _ in named if n.startswith("discriminator_")]
    assert len(heads) == 5, heads       # rotation kernel+bias, predictor kernel+bias, projection
    _check(named, grads_o, "s3gan D-step")

    gan._set_requires_grad(gan.d_opt, False)
    gan._set_requires_grad(gan.g_opt, True)
    gan._zero_grads(gan.g_opt)
    with ops.use_store(gan.store):
        zd = gan.z_generator([bsz, options["z_dim"]], name="z/0")
        sy = gan._get_one_hot_labels(sampled.to(dev))
        feats = {"images": images.to(dev), "_generator_step": True, "sampled_labels": sampled.to(dev),
                 "generated": gan.generator(zd, y=sy, is_training=True)}
        gan.create_loss(feats, labels.to(dev))
    gan.g_loss.backward()
    gen_o2 = ora.G(z.double(), ora.one_hot(sampled))
    _, g_loss_o, _ = ora.create_loss(images.double(), gen_o2, labels, sampled, with_penalty=False)
    ggrads_o = torch.autograd.grad(g_loss_o, ora.g_vars(), allow_unused=True)
    print("s3gan g_loss", float(gan.g_loss.detach()), float(g_loss_o.detach()))
    assert abs(float(gan.g_loss.detach()) - float(g_loss_o.detach())) <= 3e-2 * max(
        1.0, abs(float(g_loss_o.detach())))
    named_g = [(n, p) for (n, p), go in zip(gan.store.trainable_variables("generator"), ggrads_o)
               if go is not None and p.grad is not None]
    _check(named_g, [go for go in ggrads_o if go is not None], "s3gan G-step", cos_min=0.95,
           rel_max=0.35)


def _check(named, grads_o, what, cos_min=0.98, rel_max=0.2):
    big = max(float(g.norm()) for g in grads_o)
    worst = (1.0, None)
    for (name, p), go in zip(named, grads_o):
        assert p.grad is not None, "%s: %s has no gradient" % (what, name)
        err = float((p.grad.detach().double().cpu().reshape(-1) - go.reshape(-1)).norm())
        if err <= 2e-3 * big or (name.endswith("/bias") and err <= 2e-2 * big):
            continue
        c, r = U.cosine(p.grad, go), U.rel_l2(p.grad, go)
        worst = min(worst, (c, name))
        assert c >= cos_min and r <= rel_max, "%s: grad of %s cosine %.5f rel-L2 %.4f" % (
            what, name, c, r)
    print(what, "worst gradient cosine", worst)


@pytest.mark.parametrize("use_predictor,project_y,self_supervision", [
    (False, False, "none"),        # unsupervised
    (False, True, "none"),         # fully supervised
    (True, True, "none"),          # only the predictor
    (True, True, "rotation"),      # predictor + self-supervision
    (False, True, "rotation"),     # only self-supervision
])
def test_s3gan_single_training_step(dev, use_predictor, project_y, self_supervision):
    """s3gan_test.py:38-72: resnet_biggan_arch on (fake) imagenet_128, batch 8, hinge loss,
    rotated_batch_fraction 2 -- here at width ch = 32 (the attention kernel needs ch / 8 % 4 == 0)."""
    bsz = 8
    gan, options, ds = _build(dev, bsz, "imagenet_128", use_predictor, project_y, self_supervision,
                              ch=32)
    before = {n: v.detach().clone() for n, v in gan.store.trainable_variables()}
    images, labels = next(ds.train_batches(2 * bsz, seed=1))
    labels = np.random.RandomState(2).randint(0, ds.num_classes, size=labels.shape).astype(np.int32)
    out = gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    assert np.isfinite(float(out["g_loss"])) and np.isfinite(float(out["d_losses"][0]))
    moved = {n: not torch.equal(v, before[n]) for n, v in gan.store.trainable_variables()
             if n.endswith("/kernel")}
    extra = sorted(n for n in moved if n.startswith("discriminator_"))
    want = []
    if project_y:
        want.append("discriminator_projection/kernel")
    if use_predictor:
        want.append("discriminator_predictor/predictor_linear/kernel")
    if self_supervision == "rotation":
        want.append("discriminator_rotation/score_classify/kernel")
    assert extra == sorted(want)
    still = [n for n, m in moved.items() if not m and "non_local_block" not in n]
    assert not still, still[:5]      # (attention kernels sit behind sigma = 0 at initialisation)
    assert int(gan.global_step.item()) == 1
