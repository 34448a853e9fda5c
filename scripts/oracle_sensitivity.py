"""CPU-only: how far do the G-step gradients of an architecture move when ONLY the bf16 storage
rounding points change (exact fp64 oracle vs. bf16-storage oracle, identical weights and inputs)?
That distance is the floor any bf16 pipeline can be held to against the exact oracle, and a scale
for the product-vs-emulated-oracle figures of the GPU parity tests.
usage: python scripts/oracle_sensitivity.py [resnet_biggan_deep_arch|resnet_biggan_arch] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import architectures as OA
from oracle import arch_ops as oops

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet_biggan_deep_arch"
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(0)
sn = oops.SNConfig(singular_value="auto")
extra_g = dict(embed_y=True, ch=32) if "deep" in arch else dict(hierarchical_z=True, embed_y=True, ch=32)
g_cfg = OA.ArchConfig(batch_norm_fn="conditional_batch_norm", spectral_norm=True,
                      bn_cfg=oops.BNConfig(0.9, 1e-5, use_moving_averages=False), sn_cfg=sn, **extra_g)
d_cfg = OA.ArchConfig(spectral_norm=True, sn_cfg=sn, project_y=True, ch=32)
G, D = OA.GENERATORS[arch], OA.DISCRIMINATORS[arch]
z = torch.randn(bsz, 120, dtype=torch.float64)
y = torch.zeros(bsz, 1000, dtype=torch.float64)
for i in range(bsz):
    y[i, (37 * i + 5) % 1000] = 1.0


def run(vs):
    img = G(vs, g_cfg, z, y, True, (128, 128, 3))
    _, logit, _ = D(vs, d_cfg, vs.q(img), y, True)
    g_loss = -logit.mean()                         # hinge generator loss (loss_lib.py:134-148)
    names = [n for n in vs.trainable if n.startswith("generator/")]
    grads = torch.autograd.grad(g_loss, [vs.vars[n] for n in names], allow_unused=True)
    return float(g_loss), dict(zip(names, grads))


exact = oops.VarStore(dtype=torch.float64, seed=1, weights_initializer="orthogonal")
l0, g0 = run(exact)                                # creates the variables
emu = oops.VarStore(dtype=torch.float64, seed=1, weights_initializer="orthogonal", emulate_bf16=True)
exact2 = oops.VarStore(dtype=torch.float64, seed=1, weights_initializer="orthogonal")
l1, g1 = run(exact2)
assert abs(l0 - l1) < 1e-12                        # the oracle itself is deterministic
le, ge = run(emu)
print("%s batch %d: g_loss exact %.6f  bf16-storage %.6f" % (arch, bsz, l0, le))
rows = []
big = max(float(g.norm()) for g in g0.values() if g is not None)
for n in g0:
    if g0[n] is None or ge[n] is None:
        continue
    a, b = g0[n].reshape(-1), ge[n].reshape(-1)
    if float(a.norm()) <= 1e-6 * big:      # biases in front of a batch norm, attention at sigma = 0
        continue
    cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
    rel = float((a - b).norm() / (a.norm() + 1e-300))
    rows.append((cos, rel, n))
rows.sort()
for cos, rel, n in rows[:25]:
    print("  %-62s cos %.5f rel %.4f" % (n, cos, rel))
print("  ... %d variables, median cos %.5f" % (len(rows), sorted(r[0] for r in rows)[len(rows) // 2]))
