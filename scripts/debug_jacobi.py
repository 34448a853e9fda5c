"""Block vs scalar Jacobi on the matrices of test_fid_matches_oracle[300-512] (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compare_gan_amd.hip import kernels as K
from oracle import fid as ofid

dev = torch.device("cuda:0")
n, d = 300, 512
rng = np.random.RandomState(n + d)
a = (rng.randn(n, d) * rng.rand(d) * 2 + rng.randn(d)).astype(np.float32)
b = (rng.randn(n, d) * rng.rand(d) * 3 + 0.5).astype(np.float32)
m, sigma = K.mean_cov_f64(torch.from_numpy(a).to(dev))
mv, sigma_v = K.mean_cov_f64(torch.from_numpy(b).to(dev))
S = sigma.cpu().numpy()
ref_w = np.linalg.eigvalsh(S)
for sweeps in (18, 30, 60):
    w, v = K.syevj_f64(sigma.clone(), max_sweeps=sweeps, tol=1e-12)
    w_, v_ = w.cpu().numpy(), v.cpu().numpy()
    err_w = np.abs(np.sort(w_) - ref_w).max()
    rec = np.abs(v_.T @ np.diag(w_) @ v_ - S).max()
    orth = np.abs(v_ @ v_.T - np.eye(d)).max()
    print("sweeps %d: eig err %.3e (scale %.3e), reconstruction %.3e, orthogonality %.3e, min |w| %.3e, #neg %d" % (
        sweeps, err_w, ref_w.max(), rec, orth, np.abs(w_).min(), int((w_ < 0).sum())))
    f, _ = K.spectral_sqrt_f64(w, 1e-10)
    sq = K.gemm_f64(v, K.rowscale_f64(v, f), ta=True)
    inner = K.gemm_f64(K.gemm_f64(sq, sigma_v), sq)
    I = inner.cpu().numpy()
    print("   inner asymmetry %.3e" % np.abs(I - I.T).max())
    w2, _ = K.syevj_f64(inner.clone(), max_sweeps=sweeps, tol=1e-12)
    w2_ = w2.cpu().numpy()
    ref2 = np.linalg.eigvalsh((I + I.T) / 2)
    print("   inner eig err %.3e (scale %.3e); sum sqrt: got %.9f ref %.9f" % (
        np.abs(np.sort(w2_) - ref2).max(), ref2.max(),
        float(np.sum(np.sign(w2_) * np.where(np.abs(w2_) < 1e-10, np.abs(w2_), np.sqrt(np.abs(w2_))))),
        float(np.sum(np.sign(ref2) * np.where(np.abs(ref2) < 1e-10, np.abs(ref2), np.sqrt(np.abs(ref2)))))))
print("oracle fid", ofid.frechet_distance(a, b))
from compare_gan_amd.metrics import fid_score
print("product fid", fid_score.frechet_distance(a, b, device=dev))
