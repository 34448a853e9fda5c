"""Pure write / pure read / copy rates of the box with plain torch ops (diagnostic: which side of
HBM bounds the write-heavy kernels -- the RGB stems write 21x what they read)."""
import torch
dev = torch.device("cuda:0")
n = 1 << 30
a = torch.zeros(n // 4, dtype=torch.float32, device=dev)
b = torch.zeros(n // 4, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for name, fn, nbytes in (("fill (write only)", lambda: a.fill_(1.0), n),
                         ("sum (read only)", lambda: a.sum(), n),
                         ("copy (read + write)", lambda: b.copy_(a), 2 * n)):
    t = timed(fn)
    print("%-22s %.1f GB/s (%.0f us for %d MiB)" % (name, nbytes / t * 1e-9, t * 1e6, nbytes >> 20))
