"""Helpers shared by the model-level parity tests: build the product GAN from an example config,
mirror its variables into the oracle, and reproduce the step's random draws on the host."""
import hashlib
import os

import numpy as np
import torch

from oracle import arch_ops as oops
from oracle import architectures as OA
from oracle import modular_gan as omg
from oracle import rng as orng

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_configs")


def build_product(config, batch_size, device, seed=3, bindings=()):
    from compare_gan_amd import datasets, gin, runner_lib
    from compare_gan_amd.gans import modular_gan  # noqa: F401  (registers configurables)
    from compare_gan_amd import eval_gan_lib  # noqa: F401
    gin.clear_config()
    gin.parse_config_files_and_bindings([os.path.join(CONFIG_DIR, config)], list(bindings))
    options = runner_lib.get_options_dict()
    dataset = datasets.get_dataset()
    gan = options["gan_class"](dataset=dataset, parameters=options, model_dir="/tmp/cg_test")
    gan.build(batch_size=batch_size, device=device, seed=seed)
    return gan, options, dataset


def mirror_to_oracle(gan, dtype=torch.float64, device="cpu", **store_kwargs):
    """The product's variables as an oracle VarStore.  device != "cpu": the restatement runs in fp64
    on plain torch ops on that device (oracle/arch_ops.py VarStore.device) -- the checker for the
    parity tests at the benchmark's batch sizes."""
    vs = oops.VarStore(dtype=dtype, device=device, **store_kwargs)
    for name, v in gan.store.vars.items():
        t = v.detach().to(device).to(dtype).clone()
        if name in gan.store.trainable:
            t.requires_grad_(True)
            vs.trainable.append(name)
        vs.vars[name] = t
    return vs


def op_id(name):
    return int(hashlib.sha512(name.encode("utf-8")).hexdigest(), 16) % (2 ** 31 - 1)


def host_uniform(shape, name, lo, hi, seed, step, replica=0):
    n = int(np.prod(shape))
    return torch.from_numpy(orng.uniform(n, lo, hi, seed, op_id(name), replica, step)).reshape(shape)


def host_normal(shape, name, mean, std, seed, step, replica=0):
    n = int(np.prod(shape))
    return torch.from_numpy(orng.normal(n, mean, std, seed, op_id(name), replica, step)).reshape(shape)


def host_labels(n, k, name, seed, step, replica=0):
    return torch.from_numpy(orng.labels(n, k, seed, op_id(name), replica, step))


def cosine(a, b):
    a = a.detach().double().reshape(-1).cpu()
    b = b.detach().double().reshape(-1).cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def rel_l2(a, b):
    a = a.detach().double().reshape(-1).cpu()
    b = b.detach().double().reshape(-1).cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


ORACLE_CONFIGS = {
    # config file -> kwargs of the oracle that restate its gin bindings
    "resnet_cifar10.gin": dict(
        architecture="resnet_cifar_arch", image_shape=(32, 32, 3),
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="batch_norm", bn_cfg=oops.BNConfig(0.9, 1e-5)),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=True), loss="non_saturating",
        penalty="no_penalty", lamba=1, disc_iters=5, g_lr=0.0002, beta1=0.5, beta2=0.999),
    "dcgan_celeba64.gin": dict(
        architecture="dcgan_arch", image_shape=(64, 64, 3),
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="batch_norm", bn_cfg=oops.BNConfig(0.9, 1e-5)),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=False), loss="non_saturating",
        penalty="no_penalty", lamba=1, disc_iters=1, g_lr=0.0002, beta1=0.5, beta2=0.999),
    "sndcgan_celebahq128.gin": dict(
        architecture="sndcgan_arch", image_shape=(128, 128, 3),
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="batch_norm", bn_cfg=oops.BNConfig(0.9, 1e-5)),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=True), loss="non_saturating",
        penalty="no_penalty", lamba=1, disc_iters=1, g_lr=0.0002, beta1=0.5, beta2=0.999),
    "resnet_lsun-bedroom128.gin": dict(
        architecture="resnet5_arch", image_shape=(128, 128, 3),
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="batch_norm", bn_cfg=oops.BNConfig(0.9, 1e-5)),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=False), loss="wasserstein",
        penalty="wgangp_penalty", lamba=10, disc_iters=5, g_lr=0.0001, beta1=0.5, beta2=0.9),
    "biggan_imagenet128.gin": dict(
        architecture="resnet_biggan_arch", image_shape=(128, 128, 3),
        g_cfg=lambda: OA.ArchConfig(batch_norm_fn="conditional_batch_norm", spectral_norm=True,
                                    bn_cfg=oops.BNConfig(0.9, 1e-5, use_moving_averages=False),
                                    sn_cfg=oops.SNConfig(singular_value="auto"),
                                    hierarchical_z=True, embed_y=True),
        d_cfg=lambda: OA.ArchConfig(spectral_norm=True,
                                    sn_cfg=oops.SNConfig(singular_value="auto"), project_y=True),
        loss="hinge", penalty="no_penalty", lamba=1, disc_iters=2, conditional=True,
        num_classes=1000, g_lr=0.0001, d_lr=0.0005, beta1=0.0, beta2=0.999, g_use_ema=True),
}


def build_oracle(config, vs, **overrides):
    kw = dict(ORACLE_CONFIGS[config])
    kw.update(overrides)
    kw["g_cfg"] = kw["g_cfg"]()
    kw["d_cfg"] = kw["d_cfg"]()
    arch = kw.pop("architecture")
    return omg.OracleGAN(vs, arch, **kw)


def resync_oracle(gan, ora):
    """Copies the product's complete training state into the oracle (variables, spectral-norm
    vectors, moving averages, Adam slots, step counters), so that the NEXT step starts from
    identical states on both sides: Adam's first updates move every weight by ~lr whatever the
    size of its gradient, which decorrelates weights with ~0 gradients after one step and makes a
    free-running comparison of later steps meaningless."""
    with torch.no_grad():
        for name, v in gan.store.vars.items():
            ov = ora.vs.vars[name]
            ov.copy_(v.detach().to(ov.device).to(ov.dtype))
    ora._ensure_opts()   # pylint: disable=protected-access
    pairs = ((gan.g_opt, ora.g_opt, [n for n in ora.vs.trainable if n.startswith("generator")]),
             (gan.d_opt, ora.d_opt, ora.d_var_names()))
    for popt, oopt, onames in pairs:
        idx = {n: i for i, n in enumerate(popt.names)}
        with torch.no_grad():
            for j, n in enumerate(onames):
                oopt.m[j].copy_(popt.m[idx[n]].detach().to(oopt.m[j].device).to(oopt.m[j].dtype))
                oopt.v[j].copy_(popt.v[idx[n]].detach().to(oopt.v[j].device).to(oopt.v[j].dtype))
    ora.g_opt.t = ora.global_step = int(gan.global_step.item())
    ora.d_opt.t = ora.global_step_disc = int(gan.global_step_disc.item())


class stepwise_parity(object):
    """One unrolled train_step() of the product against the oracle, sub-step by sub-step FROM
    IDENTICAL STATES.

    Why not the whole step against a free-running oracle: Adam's first updates are sign-like
    (update = -lr * sign(g) whatever |g|), so the ~0.5 % of discriminator weights whose gradient is
    rounding noise move by +-lr on either side after the FIRST sub-step, and the GAN dynamics
    amplify that: over eight seeds the generator loss of the product and of the bf16-storage
    oracle -- and of the bf16-storage and the exact oracle, two restatements of the reference
    that differ in storage rounding alone -- end up 1e-4 ... 2e-1 (one seed: 2.3) apart
    (profiles/r05_gloss_spread.txt), while every sub-step taken from identical states agrees to
    3e-5 (D losses), 2.3e-4 (generator loss) and cosine >= 0.99 (updates)
    (profiles/r05_gloss_stepwise.txt).

    Usage: install as gan.sub_step_hook, call gan.train_step(images, labels), then finish(out).
    Before every sub-step the oracle takes over the product's complete state (resync_oracle) and
    runs the same sub-step (oracle/modular_gan.py train_step, taken apart); finish() compares:
      d_losses / g_loss      relative tol_loss (default 2e-3; measured <= 3e-4)
      the GRADIENT of every sub-step -- recovered from the product's first-moment slots, g = (m_after -
      beta1 m_before) / (1 - beta1), the states being identical before the sub-step -- against the
      oracle's: cosine >= cos_grad (default 0.999 for BOTH networks; measured >= 0.99969 D, 0.99998 G).  This is the
      magnitude-weighted check a 2 % direction error of a kernel cannot pass;
      the UPDATE of every sub-step: cosine >= cos_min (discriminator, default 0.98; measured >= 0.9906)
      / cos_min_g (generator, default 0.95; measured 0.979).  Adam's first update is lr * sign(g) for
      every element, so this cosine is 1 - 2 x (share of elements whose gradient SIGN differs), i.e. a
      count of coin flips among the small-gradient elements: round 5 blamed the generator's 0.979 on
      the biases in front of a batch norm (mathematically zero gradient); masking every element whose
      oracle gradient is below 1e-4 of the network's rms (0.2 % of G) moves it from 0.97879 to 0.97919
      only -- the flips are spread over ~1 % of ALL elements, which is what bf16 kernels against an
      fp64 restatement with bf16 storage give.  Kept as a sanity band; the gradient cosine is the test,
      and per element |update_p - update_o| <= 3.5 lr (opposite signs of a ~0 gradient)."""

    def __init__(self, gan, ora, subs, lr_d, lr_g=None, tol_loss=2e-3, cos_min=0.98, cos_min_g=0.95,
                 cos_grad=0.999):
        self.gan, self.ora, self.subs = gan, ora, subs
        self.lr_d, self.lr_g = lr_d, lr_g if lr_g is not None else lr_d
        self.tol_loss, self.cos_min, self.cos_min_g = tol_loss, cos_min, cos_min_g
        self.cos_grad = cos_grad
        self.d_o, self.g_o = [], None
        self.pending = None      # (net, names, before, oracle_after, lr, oracle grads) in flight
        self.rows = []
        self._joint = None

    @staticmethod
    def _m_before(popt, names):
        idx = {n: i for i, n in enumerate(popt.names)}
        return {n: popt.m[idx[n]].detach().cpu().double().clone() for n in names}

    def _names(self, net):
        return [n for n, _ in self.gan.store.trainable_variables(net)]

    def _close(self):
        if self.pending is None:
            return
        net, names, before, after_o, lr, grads_o, m_before = self.pending
        popt = self.gan.g_opt if net == "generator" else self.gan.d_opt
        idx = {n: i for i, n in enumerate(popt.names)}
        beta1 = float(popt.opt.beta1)
        ups, uos, gos, gps = [], [], [], []
        for n in names:
            up = self.gan.store.vars[n].detach().cpu().double() - before[n]
            uo = after_o[n].cpu() - before[n]
            assert float((up - uo).abs().max()) <= 2.2 * lr * 1.6, (net, n, float((up - uo).abs().max()) / lr)
            ups.append(up.reshape(-1))
            uos.append(uo.reshape(-1))
            gos.append(grads_o[n].cpu().reshape(-1))
            m_after = popt.m[idx[n]].detach().cpu().double()
            gps.append(((m_after - beta1 * m_before[n]) / (1.0 - beta1)).reshape(-1))
        c = cosine(torch.cat(ups), torch.cat(uos))
        cg = cosine(torch.cat(gps), torch.cat(gos))
        self.rows.append((net, c, cg))
        floor = self.cos_min_g if net == "generator" else self.cos_min
        assert cg >= self.cos_grad, "gradient of the %s: cosine %.5f" % (net, cg)
        assert c >= floor, "update of the %s: cosine %.5f" % (net, c)
        self.pending = None

    def __call__(self, phase, index):
        self._close()
        gan, ora, s = self.gan, self.ora, self.subs[index]
        resync_oracle(gan, ora)
        ora._ensure_opts()   # pylint: disable=protected-access
        sy = ora.one_hot(s["sampled_labels"]) if ora.conditional else None
        if phase == "d":
            names = self._names("discriminator")
            before = {n: gan.store.vars[n].detach().cpu().double().clone() for n in names}
            with torch.no_grad():
                if ora.joint_gen_for_disc:
                    if self._joint is None:
                        z = torch.cat([t["z"] for t in self.subs[:ora.disc_iters]], dim=0)
                        self._joint = torch.chunk(ora.G(z, None), ora.disc_iters, dim=0)
                    generated = self._joint[index]
                else:
                    generated = ora.G(s["z"], sy)
            d_loss, _, _ = ora.create_loss(s["images"], generated, s.get("labels"),
                                           s.get("sampled_labels"), s.get("alpha"))
            dgr = torch.autograd.grad(d_loss, ora.d_vars())
            grads_o = {n: g.detach().double() for n, g in zip(ora.d_var_names(), dgr)}
            ora.d_opt.step(dgr)
            ora.global_step_disc += 1
            self.d_o.append(float(d_loss.detach()))
            after = {n: ora.vs.vars[n].detach().double().clone() for n in names}
            self.pending = ("discriminator", names, before, after, self.lr_d, grads_o,
                            self._m_before(gan.d_opt, names))
        else:
            names = self._names("generator")
            before = {n: gan.store.vars[n].detach().cpu().double().clone() for n in names}
            generated = ora.G(s["z"], sy)
            _, g_loss, _ = ora.create_loss(s["images"], generated, s.get("labels"),
                                           s.get("sampled_labels"), with_penalty=False)
            ggr = torch.autograd.grad(g_loss, ora.g_vars())
            gnames = [n for n in ora.vs.trainable if n.startswith("generator")]
            grads_o = {n: g.detach().double() for n, g in zip(gnames, ggr)}
            ora.g_opt.step(ggr)
            ora.global_step += 1
            self.g_o = float(g_loss.detach())
            after = {n: ora.vs.vars[n].detach().double().clone() for n in names}
            self.pending = ("generator", names, before, after, self.lr_g, grads_o,
                            self._m_before(gan.g_opt, names))

    def finish(self, out):
        """out: what train_step() returned.  Returns (d_losses oracle, g_loss oracle)."""
        self._close()
        d_p = [float(x) for x in out["d_losses"]]
        g_p = float(out["g_loss"])
        for i, (a, b) in enumerate(zip(d_p, self.d_o)):
            assert abs(a - b) <= self.tol_loss * max(1.0, abs(b)), ("d_loss", i, a, b)
        assert abs(g_p - self.g_o) <= self.tol_loss * max(1.0, abs(self.g_o)), ("g_loss", g_p, self.g_o)
        # state carried between the sub-steps: the step counters after the whole step (the Adam slots
        # are covered by the recovered gradients above)
        assert int(self.gan.global_step.item()) == self.ora.global_step, "global_step"
        assert int(self.gan.global_step_disc.item()) == self.ora.global_step_disc, "global_step_disc"
        return self.d_o, self.g_o
