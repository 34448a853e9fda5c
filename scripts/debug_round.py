"""Diagnostics (not tests).  usage: python scripts/debug_round.py [crash|gp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gan_util as U
dev = torch.device("cuda:0")


def crash(bsz=64):
    gan, options, dataset = U.build_product("resnet_cifar10.gin", bsz, dev, seed=3)
    nsub = options["disc_iters"] + 1
    batches = dataset.train_batches(bsz * nsub, seed=547)
    images, labels = next(batches)
    out = gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    torch.cuda.synchronize()
    print("eager step ok", float(out["g_loss"]))


def gp():
    from compare_gan_amd.architectures import arch_ops as ops
    from compare_gan_amd.gans import penalty_lib
    config, bsz = "resnet_lsun-bedroom128.gin", 2
    for emulate in (True,):
        gan, options, dataset = U.build_product(config, bsz, dev, seed=3)
        vs = U.mirror_to_oracle(gan, emulate_bf16=emulate)
        ora = U.build_oracle(config, vs)
        rng = np.random.RandomState(7)
        images = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
        fake = torch.from_numpy(rng.uniform(size=(bsz,) + dataset.image_shape).astype(np.float32))
        alpha = U.host_uniform((bsz,), "wgangp_penalty/alpha", 0.0, 1.0, 3, 0)
        named = gan.store.trainable_variables("discriminator")
        gan._set_requires_grad(gan.g_opt, False)
        # ---- penalty only ----
        gan._zero_grads(gan.d_opt)
        with ops.use_store(gan.store):
            pen = penalty_lib.get_penalty_loss(x=images.to(dev), x_fake=fake.to(dev), y=None,
                                               is_training=True, discriminator=gan.discriminator)
        pen.backward()
        from oracle import gan as ogan
        pen_o = ogan.wgangp_penalty(lambda x, yy, t: ora.D(x, yy, t), images.double(), fake.double(),
                                    None, True, alpha.double().reshape(-1, 1, 1, 1))
        g_o = torch.autograd.grad(pen_o, ora.d_vars(), allow_unused=True)
        print("penalty", float(pen), float(pen_o))
        for (name, p), go in zip(named, g_o):
            if go is None or p.grad is None:
                print("  PEN %-46s oracle %s product %s" % (name, go is None, p.grad is None)); continue
            print("  PEN %-46s cos %.5f rel %.4f |g_o| %.3e |g_p| %.3e" % (
                name, U.cosine(p.grad, go), U.rel_l2(p.grad, go), float(go.norm()), float(p.grad.norm())))
        
        # ---- wasserstein only ----
        gan._zero_grads(gan.d_opt)
        with ops.use_store(gan.store):
            gan.create_loss({"images": images.to(dev), "generated": fake.to(dev), "_generator_step": True}, None)
        gan.d_loss.backward()
        d_o, _, _ = ora.create_loss(images.double(), fake.double(), None, None, with_penalty=False)
        g_o2 = torch.autograd.grad(d_o, ora.d_vars())
        print("wasserstein", float(gan.d_loss), float(d_o))
        for (name, p), go in zip(named, g_o2):
            print("  WAS %-46s cos %.5f rel %.4f |g_o| %.3e" % (
                name, U.cosine(p.grad, go), U.rel_l2(p.grad, go), float(go.norm())))


def graph(mode, bsz=64):
    from compare_gan_amd.architectures import arch_ops as ops
    from compare_gan_amd.hip import kernels as K
    from compare_gan_amd.gans import modular_gan as mg
    variant = os.environ.get("VARIANT", "")
    if "noadam" in variant:
        mg._OptimizerState.apply_gradients = lambda self, step, **kw: None
    if "nocounter" in variant:
        K.counter_add = lambda c, inc=1: None
    bind = ["D.spectral_norm = False"] if "nosn" in variant else []
    gan, options, dataset = U.build_product("resnet_cifar10.gin", bsz, dev, seed=3, bindings=bind)
    nsub = options["disc_iters"] + 1
    images, labels = next(dataset.train_batches(bsz * nsub, seed=547))
    images = torch.from_numpy(images).to(dev); labels = torch.from_numpy(labels).to(dev)
    f, l = gan._preprocess(images[:bsz], labels[:bsz], 0)

    def body():
        with ops.use_store(gan.store):
            if mode == "gfwd":
                with torch.no_grad():
                    return gan.generator(f["z"], y=None, is_training=True)
            if mode == "dfwd":
                with torch.no_grad():
                    gan.create_loss({"images": f["images"], "generated": f["images"]}, l)
                    return gan.d_loss
            if mode == "dfwdbwd":
                gan._set_requires_grad(gan.g_opt, False)
                gan._zero_grads(gan.d_opt)
                gan.create_loss({"images": f["images"], "generated": f["images"]}, l)
                gan.d_loss.backward()
                return gan.d_loss.detach()
            if mode == "dstep":
                ff = dict(f); ff["generated"] = f["images"]
                return gan._train_discriminator(ff, l)
            if mode == "gstep":
                return gan._train_generator(f, l)
            if mode == "full":
                return gan.train_step(images, labels)["g_loss"]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(mode, "warmup ok", flush=True)
    gan.d_opt.reserve_tables(8); gan.g_opt.reserve_tables(2)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    for opt in (gan.g_opt, gan.d_opt):
        for t in opt.captured_tables:
            t.flush()
    torch.cuda.synchronize()
    print(mode, "capture ok", flush=True)
    import ctypes
    from compare_gan_amd.hip import _lib
    segs = []
    for seg in torch.cuda.memory_snapshot():
        for b in seg["blocks"]:
            pass
        segs.append((seg["address"], seg["address"] + seg["total_size"]))
    def inside(ptr):
        return any(a <= ptr < b for a, b in segs)
    for opt in (gan.g_opt, gan.d_opt):
        for t in opt.captured_tables:
            back = t.table.cpu().numpy().tobytes()
            assert back == bytes(t.entries), "device table differs from host table"
            for e in t.entries:
                for f in ("param", "grad", "m", "v"):
                    ptr = getattr(e, f)
                    if not inside(ptr):
                        print("BAD POINTER", f, hex(ptr or 0), flush=True)
    print("tables verified", len(gan.g_opt.captured_tables), len(gan.d_opt.captured_tables), flush=True)
    def wsum():
        return [float(p.detach().double().abs().sum()) for p in (gan.g_opt.params[0], gan.d_opt.params[0], gan.d_opt.params[-2])]
    print("weights", wsum(), flush=True)
    if "eageradam" in variant:
        for opt in (gan.g_opt, gan.d_opt):
            for t in opt.captured_tables:
                o = opt.opt
                t.adam(o.learning_rate, o.beta1, o.beta2, o.epsilon, 1.0, gan.global_step_disc)
                torch.cuda.synchronize()
                print("eager adam on captured table ok", wsum(), flush=True)
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("replay", i, "ok", wsum(), flush=True)
    print(mode, "replay ok", float(out.float().mean()), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "graph":
        graph(sys.argv[2])
    else:
        {"crash": crash, "gp": gp}[sys.argv[1]]()
