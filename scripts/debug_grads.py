"""Diagnostic (not a test): per-variable gradient cosines product-vs-oracle, and small
double-backward probes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gan_util as U
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.hip import functional as Fn, kernels as K
from oracle import arch_ops as oops

dev = torch.device("cuda:0")

def probe_double_backward():
    torch.manual_seed(0)
    N, H, C = 2, 8, 16
    for case in ["conv", "relu_conv", "conv_pool", "two_branch", "conv_relu_conv_mean_fc"]:
        x = torch.rand(N, H, H, 3)
        w1 = torch.randn(3, 3, 3, C) * 0.2
        w2 = torch.randn(3, 3, C, C) * 0.1
        wf = torch.randn(C, 1) * 0.3
        xb = x.to(torch.bfloat16)
        # oracle
        xo = xb.double().requires_grad_(True)
        w1o, w2o, wfo = [t.to(torch.bfloat16).double().requires_grad_(True) for t in (w1, w2, wf)]
        def D_o(xx):
            if case == "conv":
                return oops.conv2d_same(xx, w1o, 1).sum(dim=(1, 2, 3))
            if case == "relu_conv":
                return oops.conv2d_same(torch.relu(xx - 0.5), w1o, 1).sum(dim=(1, 2, 3))
            if case == "conv_pool":
                return oops.avg_pool2(oops.conv2d_same(xx, w1o, 1)).sum(dim=(1, 2, 3))
            if case == "two_branch":
                return (oops.conv2d_same(xx, w1o, 1) + oops.conv2d_same(torch.relu(xx - 0.5), w1o * 0.5, 1)).sum(dim=(1, 2, 3))
            h = oops.conv2d_same(xx, w1o, 1)
            h = oops.conv2d_same(torch.relu(h), w2o, 1)
            return (torch.relu(h).mean(dim=(1, 2)) @ wfo).reshape(-1)
        lo = D_o(xo)
        go, = torch.autograd.grad(lo.sum(), xo, create_graph=True)
        Po = ((torch.sqrt(1e-4 + (go ** 2).sum(dim=(1, 2, 3))) - 1) ** 2).mean()
        refs = torch.autograd.grad(Po, [w1o, w2o, wfo], allow_unused=True)
        # product
        xd = (xb.to(dev) if case not in ("relu_conv", "two_branch") else xb.to(dev))
        xd = xd.clone().requires_grad_(True)
        w1d, w2d, wfd = [t.to(torch.bfloat16).float().to(dev).requires_grad_(True) for t in (w1, w2, wf)]
        g1 = K.geom_conv_same(N, H, H, 3, C, 3, 3, 1, 1)
        g2 = K.geom_conv_same(N, H, H, C, C, 3, 3, 1, 1)
        def conv(xx, w, geom, slope=None, gate=None):
            return Fn.gconv(xx, w, None, None, gate, None, Fn.ConvSpec(geom, slope_in=slope), True)
        def total(t):  # sum over h,w,c -> [N] via spatial reduce + colsum-like linear
            n, c = t.shape[0], t.shape[-1]
            s = Fn.SpatialReduceFn.apply(t, None, 1.0)          # [N,C] bf16
            ones = torch.ones(c, 1, device=dev)
            gg = K.make_geom(n, 1, 1, c, 1, 1, 1, 1, 1)
            return Fn.gconv(s.reshape(n, 1, 1, c), ones.reshape(1, 1, c, 1), None, None, None, None,
                            Fn.ConvSpec(gg, out_f32=True), False).reshape(-1)
        if case == "conv":
            lp = total(conv(xd, w1d, g1))
        elif case == "relu_conv":
            sh = K.cast_f32_to_bf16(K.cast_bf16_to_f32(xd.detach()), 1.0, -0.5)  # x-0.5 as gate source
            # emulate relu(x-0.5): feed shifted tensor (requires grad through a leaf alias)
            xs = sh.clone().requires_grad_(True)
            lp = total(conv(xs, w1d, g1, 0.0, xs.detach()))
            xd = xs
        elif case == "conv_pool":
            lp = total(Fn.avg_pool2(conv(xd, w1d, g1)))
        elif case == "two_branch":
            continue
        else:
            h = conv(xd, w1d, g1)
            h = conv(h, w2d, g2, 0.0, h.detach())
            feat = Fn.SpatialReduceFn.apply(h, h.detach(), 1.0 / (H * H))
            gg = K.make_geom(N, 1, 1, C, 1, 1, 1, 1, 1)
            lp = Fn.gconv(feat.reshape(N, 1, 1, C), wfd.reshape(1, 1, C, 1), None, None, None, None,
                          Fn.ConvSpec(gg, out_f32=True), False).reshape(-1)
        ones = torch.ones_like(lp)
        with Fn.only_input_grads():
            gp, = torch.autograd.grad(lp, [xd], grad_outputs=ones, create_graph=True)
        if gp.dtype != torch.float32:
            gp = Fn.ToF32Fn.apply(gp)
        Pp = Fn.GradientPenaltyFn.apply(gp)
        if case == "relu_conv":
            # oracle x shifted equivalently
            pass
        Pp.backward()
        print("probe", case, "logits", U.rel_l2(lp, lo), "g", U.rel_l2(gp, go) if case != "relu_conv" else None,
              "P", float(Pp), float(Po))
        for nm, pd, ro in (("w1", w1d, refs[0]), ("w2", w2d, refs[1]), ("wf", wfd, refs[2])):
            if ro is not None and pd.grad is not None:
                print("    d/d%s cosine %.5f rel %.4f" % (nm, U.cosine(pd.grad, ro), U.rel_l2(pd.grad, ro)))

def per_variable(config, bsz, what, emulate=False):
    gan, options, dataset = U.build_product(config, bsz, dev, seed=3)
    vs = U.mirror_to_oracle(gan, emulate_bf16=emulate)
    print("=== emulate_bf16 =", emulate)
    ora = U.build_oracle(config, vs)
    rng = np.random.RandomState(7)
    images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
    z = U.host_uniform((bsz, options["z_dim"]), "z/0", -1.0, 1.0, 3, 0)
    with torch.no_grad():
        gen_o = ora.G(z.double(), None)
    if what == "D":
        alpha = U.host_uniform((bsz,), "wgangp_penalty/alpha", 0.0, 1.0, 3, 0)
        gan._set_requires_grad(gan.g_opt, False)
        gan._zero_grads(gan.d_opt)
        with ops.use_store(gan.store):
            gan.create_loss({"images": images.to(dev), "generated": gen_o.float().to(dev)}, None)
        gan.d_loss.backward()
        d_loss_o, _, _ = ora.create_loss(images.double(), gen_o, None, None, alpha.double())
        grads_o = torch.autograd.grad(d_loss_o, ora.d_vars())
        print(config, "d_loss", float(gan.d_loss), float(d_loss_o))
        named = gan.store.trainable_variables("discriminator")
    else:
        gan._set_requires_grad(gan.d_opt, False)
        gan._zero_grads(gan.g_opt)
        with ops.use_store(gan.store):
            zd = gan.z_generator([bsz, options["z_dim"]], name="z/0")
            feats = {"images": images.to(dev), "_generator_step": True,
                     "generated": gan.generator(zd, y=None, is_training=True)}
            gan.create_loss(feats, None)
        gan.g_loss.backward()
        gen_o2 = ora.G(z.double(), None)
        _, g_loss_o, _ = ora.create_loss(images.double(), gen_o2, None, None, with_penalty=False)
        grads_o = torch.autograd.grad(g_loss_o, ora.g_vars())
        print(config, "g_loss", float(gan.g_loss), float(g_loss_o))
        named = gan.store.trainable_variables("generator")
    for (name, p), go in zip(named, grads_o):
        print("   %-50s cos %.5f rel %.4f |g| %.3e" % (name, U.cosine(p.grad, go), U.rel_l2(p.grad, go), float(go.norm())))

if __name__ == "__main__":
    per_variable("resnet_lsun-bedroom128.gin", 2, "D", True)
    per_variable("resnet_cifar10.gin", 8, "G", True)
    per_variable("resnet_cifar10.gin", 8, "D", True)
    per_variable("dcgan_celeba64.gin", 4, "D", True)
    per_variable("dcgan_celeba64.gin", 4, "G", True)
    per_variable("sndcgan_celebahq128.gin", 2, "D", True)
    per_variable("sndcgan_celebahq128.gin", 2, "G", True)
