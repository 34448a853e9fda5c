#!/usr/bin/env python
"""Where does the product's generator loss leave the oracle's (-1.6 ... -2.9 %
after ONE unrolled resnet_cifar10.gin step while the D losses agree to 1.5e-3; the exact and the
bf16-storage oracle agree to 1e-4 with each other, scripts/gloss_spread.py)?  The oracle's G
sub-step is evaluated twice: with its own discriminator weights and with the PRODUCT's updated
discriminator weights copied in.  Then the discriminator updates are compared variable by variable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests import gan_util as U

config, bsz, seed = "resnet_cifar10.gin", int(os.environ.get("BSZ", "8")), int(os.environ.get("SEED", "3"))
dev = torch.device("cuda:0")
gan, options, dataset = U.build_product(config, bsz, dev, seed=seed)
vs = U.mirror_to_oracle(gan, emulate_bf16=True)
ora = U.build_oracle(config, vs)
before = {n: v.detach().clone().cpu().double() for n, v in gan.store.trainable_variables()}
rng = np.random.RandomState(500)
images = rng.uniform(size=(6 * bsz, 32, 32, 3)).astype(np.float32)
subs = [{"images": torch.from_numpy(images[i * bsz:(i + 1) * bsz]).double(),
         "z": U.host_uniform((bsz, 128), "z/%d" % i, -1.0, 1.0, seed, 0).double()} for i in range(6)]
out = gan.train_step(torch.from_numpy(images).to(dev), torch.ones((6 * bsz,), dtype=torch.int32, device=dev))
torch.cuda.synchronize()
d_p, g_p = [float(x) for x in out["d_losses"]], float(out["g_loss"])

# the oracle's five D sub-steps (oracle/modular_gan.py train_step), then its G-step loss twice
ora._ensure_opts()
d_o = []
for i in range(5):
    s = subs[i]
    with torch.no_grad():
        generated = ora.G(s["z"], None)
    d_loss, _, _ = ora.create_loss(s["images"], generated, None, None, None)
    ora.d_opt.step(torch.autograd.grad(d_loss, ora.d_vars()))
    ora.global_step_disc += 1
    d_o.append(float(d_loss.detach()))
print("d_losses product", ["%.6f" % x for x in d_p])
print("d_losses oracle ", ["%.6f" % x for x in d_o])
saved_u = {n: v.detach().clone() for n, v in vs.vars.items() if n.endswith("u_var")}
saved_mov = {n: v.detach().clone() for n, v in vs.vars.items() if "moving_" in n}


def g_loss_now():
    with torch.no_grad():
        for n, v in saved_u.items():
            vs.vars[n].copy_(v)
        for n, v in saved_mov.items():
            vs.vars[n].copy_(v)
    gen = ora.G(subs[5]["z"], None)
    _, g_loss, _ = ora.create_loss(subs[5]["images"], gen, None, None, with_penalty=False)
    return float(g_loss.detach())


g_o = g_loss_now()
d_names = [n for n, _ in gan.store.trainable_variables("discriminator")]
own = {n: vs.vars[n].detach().clone() for n in d_names}
with torch.no_grad():
    for n in d_names:
        vs.vars[n].copy_(gan.store.vars[n].detach().cpu().double())
g_o_pd = g_loss_now()
print("g_loss: product %.6f | oracle %.6f | oracle with the PRODUCT's updated D weights %.6f" % (g_p, g_o, g_o_pd))
print("        product - oracle %+.2e ; (oracle+product D) - oracle %+.2e" % (g_p - g_o, g_o_pd - g_o))
lr = 2e-4
print("%-52s %9s %9s %9s %9s %8s" % ("discriminator variable", "|up_p|/lr", "|up_o|/lr", "maxdiff/lr", "cosine", "n"))
for n in d_names:
    up = gan.store.vars[n].detach().cpu().double() - before[n]
    uo = own[n] - before[n]
    print("%-52s %9.3f %9.3f %9.3f %9.4f %8d" % (n[-52:], float(up.abs().mean()) / lr, float(uo.abs().mean()) / lr,
                                               float((up - uo).abs().max()) / lr, U.cosine(up, uo), up.numel()))
