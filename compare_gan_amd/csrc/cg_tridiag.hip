// Eigenvalues of a symmetric fp64 matrix WITHOUT eigenvectors: Householder tridiagonalisation + Sturm
// bisection (gfx950).
//
// What it is for: the second matrix square root of the Frechet distance only contributes its TRACE
// (metrics/fid_score.py:58-75 through tfgan's trace_sqrt_product: trace(sqrtm(sqrt(sigma) sigma_v
// sqrt(sigma)))), i.e. a sum over the spectrum.  The values-only Jacobi solve (cg_syevj_f64, v = NULL)
// spent ~27 sweeps x 127 rounds = 0.33 s on it at d = 2048 (profiles/r06_fid_stats.txt).  Here:
//
//   n - 1 Householder steps (LAPACK dsytd2's arithmetic), each ONE pass over the trailing matrix:
//     S(k)  one workgroup   w_{k-1} = tau p - (tau^2 (p.v) / 2) v from the raw product p_{k-1} = A v_{k-1};
//                           column k with the pending rank-2 update applied on the fly; d[k]; the next
//                           reflector v_k and e[k] (dlarfg)
//     P(k)  256 workgroups  A <- A - v_{k-1} w_{k-1}^T - w_{k-1} v_{k-1}^T on [k+1, n)^2 -- full storage,
//                           every row belongs to one wave, no symmetric scatter -- and p_k = A v_k in
//                           the SAME pass (the three vectors sit in LDS): 16 bytes of HBM / MALL traffic
//                           per element and step, n^3 / 3 * 16 B = 46 GB at n = 2048
//   the step index lives on the device, so one captured hipGraph of TD_CHUNK (S, P) pairs is replayed
//   ceil(n / TD_CHUNK) times (4094 dependent launches otherwise);
//   then every eigenvalue by multisection on the Sturm count of (d, e), one wave each.
//
// Accuracy: backward stable -- the computed values are the exact eigenvalues of A + E, |E|_F a modest
// multiple of u |A|_F (measured against LAPACK: <= 17 u |A|_F in the Frobenius sense for n <= 2048,
// tests/test_kernels_gpu.py::test_tridiagonal_eigenvalues) -- an ABSOLUTE statement, not the
// per-eigenvalue relative accuracy of Jacobi.  cg_spectral_sqrt_bound_f64 turns a bound on |E|_F into a
// bound on the sum the caller needs; metrics/fid_score.py accepts the result only under that
// certificate and falls back to cg_syevj_f64.
#include "cg_common.h"

#include <float.h>
#include <math.h>

namespace {
constexpr int TD_CHUNK = 128;
constexpr int TD_MAX_N = 4096;     // the three vectors of P(k) in LDS: 3 * 8 * n <= 96 KiB
constexpr int TD_P_BLOCKS = 256;   // one workgroup per CU

__device__ __forceinline__ double td_block_sum(double v, double* sm) {   // blockDim.x == 1024
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += sm[i];
  return t;
}

__global__ void td_init_kernel(int* ctr, int n) {
  ctr[0] = 0;   // step of the next S
  ctr[1] = n;   // step of the next P (none until an S has run)
}

// vb: [2][n] reflectors (v_k in vb[k & 1], defined on [k+1, n), v_k[k+1] = 1); wv: [n] pending w_{k-1}
// (on [k, n)); pv: [n] raw products; tau: [2]; col: [n] scratch.
__global__ __launch_bounds__(1024) void td_s_kernel(const double* __restrict__ a, int n,
                                                    int* __restrict__ ctr, double* __restrict__ vb,
                                                    double* __restrict__ wv,
                                                    const double* __restrict__ pv,
                                                    double* __restrict__ tau,
                                                    double* __restrict__ col,
                                                    double* __restrict__ dd,
                                                    double* __restrict__ ee) {
  __shared__ double sm[16];
  const int k = ctr[0];
  if (k >= n) return;
  const int t = threadIdx.x;
  const double* vp = vb + (size_t)((k + 1) & 1) * n;   // v_{k-1}
  double* vc = vb + (size_t)(k & 1) * n;               // v_k
  double wk = 0.0, vpk = 0.0;
  if (k >= 1) {
    const double tp = tau[(k + 1) & 1];
    double dot = 0.0;
    for (int i = k + t; i < n; i += 1024) dot += pv[i] * vp[i];
    dot = td_block_sum(dot, sm);
    const double alpha = -0.5 * tp * tp * dot;
    for (int i = k + t; i < n; i += 1024) wv[i] = tp * pv[i] + alpha * vp[i];
    wk = tp * pv[k] + alpha * vp[k];
    vpk = vp[k];
  }
  // column k (rows k .. n-1) with the pending update applied: a_ik - v_i w_k - w_i v_k
  for (int i = k + t; i < n; i += 1024) {
    double c = a[(size_t)i * n + k];
    if (k >= 1) c -= vp[i] * wk + wv[i] * vpk;   // wv[i]: this thread's own store above
    col[i] = c;
  }
  __syncthreads();   // col[] of this block is visible
  if (t == 0) dd[k] = col[k];
  if (k == n - 1) {
    if (t == 0) { ctr[0] = k + 1; ctr[1] = k; }
    return;
  }
  const double alpha0 = col[k + 1];
  double x2 = 0.0;
  for (int i = k + 2 + t; i < n; i += 1024) x2 += col[i] * col[i];
  x2 = td_block_sum(x2, sm);
  double beta, tk, scale;
  if (x2 == 0.0) {
    beta = alpha0; tk = 0.0; scale = 0.0;
  } else {
    beta = -copysign(hypot(alpha0, sqrt(x2)), alpha0);
    tk = (beta - alpha0) / beta;
    scale = 1.0 / (alpha0 - beta);
  }
  for (int i = k + 1 + t; i < n; i += 1024) vc[i] = (i == k + 1) ? 1.0 : col[i] * scale;
  if (t == 0) {
    ee[k] = beta;
    tau[k & 1] = tk;
    ctr[0] = k + 1;
    ctr[1] = k;
  }
}

__global__ __launch_bounds__(256) void td_p_kernel(double* __restrict__ a, int n,
                                                   const int* __restrict__ ctr,
                                                   const double* __restrict__ vb,
                                                   const double* __restrict__ wv,
                                                   double* __restrict__ pv) {
  extern __shared__ double lds[];
  const int k = ctr[1];
  if (k > n - 2) return;
  const int lo = k + 1, m = n - lo;
  double* sv = lds;            // v_{k-1} on [lo, n)
  double* sw = lds + m;        // w_{k-1}
  double* sc = lds + 2 * m;    // v_k
  const bool prev = k >= 1;
  const double* vp = vb + (size_t)((k + 1) & 1) * n;
  const double* vc = vb + (size_t)(k & 1) * n;
  for (int j = threadIdx.x; j < m; j += 256) {
    sv[j] = prev ? vp[lo + j] : 0.0;
    sw[j] = prev ? wv[lo + j] : 0.0;
    sc[j] = vc[lo + j];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int r = gw; r < m; r += nw) {
    double* row = a + (size_t)(lo + r) * n + lo;
    const double vi = sv[r], wi = sw[r];
    double acc = 0.0;
    if (prev) {
      for (int j = lane; j < m; j += 64) {
        const double x = row[j] - (vi * sw[j] + wi * sv[j]);
        row[j] = x;
        acc += x * sc[j];
      }
    } else {
      for (int j = lane; j < m; j += 64) acc += row[j] * sc[j];
    }
    acc = wave_sum_d(acc);
    if (lane == 0) pv[lo + r] = acc;
  }
}

// Gershgorin interval, pivmin and |T|_F = |A|_F of (d, e): bnd = {gl, gu, pivmin, fro}.
__global__ __launch_bounds__(256) void td_bounds_kernel(const double* __restrict__ dd,
                                                        const double* __restrict__ ee, int n,
                                                        double* __restrict__ bnd) {
  __shared__ double sm[4][4];
  double gl = DBL_MAX, gu = -DBL_MAX, emax2 = 0.0, fro2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double el = i > 0 ? fabs(ee[i - 1]) : 0.0, er = i < n - 1 ? fabs(ee[i]) : 0.0;
    gl = fmin(gl, dd[i] - el - er);
    gu = fmax(gu, dd[i] + el + er);
    emax2 = fmax(emax2, er * er);
    fro2 += dd[i] * dd[i] + 2.0 * er * er;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gl = fmin(gl, __shfl_xor(gl, o, 64));
    gu = fmax(gu, __shfl_xor(gu, o, 64));
    emax2 = fmax(emax2, __shfl_xor(emax2, o, 64));
  }
  fro2 = wave_sum_d(fro2);
  if ((threadIdx.x & 63) == 0) {
    const int wv = threadIdx.x >> 6;
    sm[0][wv] = gl; sm[1][wv] = gu; sm[2][wv] = emax2; sm[3][wv] = fro2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bnd[0] = fmin(fmin(sm[0][0], sm[0][1]), fmin(sm[0][2], sm[0][3]));
    bnd[1] = fmax(fmax(sm[1][0], sm[1][1]), fmax(sm[1][2], sm[1][3]));
    bnd[2] = DBL_MIN * fmax(1.0, fmax(fmax(sm[2][0], sm[2][1]), fmax(sm[2][2], sm[2][3])));
    bnd[3] = sqrt((sm[3][0] + sm[3][1]) + (sm[3][2] + sm[3][3]));
  }
}

// Sturm-count multisection (LAPACK dstebz's recurrence): ONE WAVE per eigenvalue j -- its 64 lanes count
// the eigenvalues below 64 interior points of the current interval, the sub-interval in which the count
// crosses j + 1 is the next one: 6 bits per pass instead of bisection's one, ~9 passes of n dependent
// divisions instead of ~56 (a thread per eigenvalue measured 25 ms at n = 2048: a latency-bound chain on
// one wave per CU).  The loop index is wave-uniform, so d[i] / e[i]^2 arrive through the scalar cache.
// Invariant: count(lo) < j + 1 <= count(hi).
__global__ __launch_bounds__(64) void td_bisect_kernel(const double* __restrict__ dd,
                                                       const double* __restrict__ ee, int n,
                                                       const double* __restrict__ bnd,
                                                       double* __restrict__ w,
                                                       double* __restrict__ fro_out) {
  const int j = blockIdx.x, lane = threadIdx.x;
  const double gl = bnd[0], gu = bnd[1], pivmin = bnd[2];
  if (fro_out && j == 0 && lane == 0) *fro_out = bnd[3];
  const double tn = fmax(fabs(gl), fabs(gu));
  double lo = gl - 2.0 * DBL_EPSILON * n * tn - 2.0 * pivmin;
  double hi = gu + 2.0 * DBL_EPSILON * n * tn + 2.0 * pivmin;
  const int want = j + 1;
  for (int pass = 0; pass < 24; ++pass) {
    if (!(hi - lo > 2.0 * DBL_EPSILON * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin)) break;
    const double x = lo + (hi - lo) * ((double)(lane + 1) / 65.0);
    double q = dd[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    int cnt = q < 0.0;
    for (int i = 1; i < n; ++i) {
      const double e2 = ee[i - 1] * ee[i - 1];
      q = dd[i] - x - e2 / q;
      if (fabs(q) < pivmin) q = -pivmin;
      cnt += q < 0.0;
    }
    // the points ascend with the lane and the count is monotone in x (up to rounding: the FIRST lane
    // whose count reaches j + 1 is taken, as bisection would)
    const unsigned long long ge = __ballot(cnt >= want);
    if (ge == 0ull) {
      lo = __shfl(x, 63, 64);
    } else {
      const int first = __ffsll((long long)ge) - 1;
      const double nhi = __shfl(x, first, 64);
      const double nlo = __shfl(x, first > 0 ? first - 1 : 0, 64);
      if (first > 0) lo = nlo;
      hi = nhi;
    }
    if (!(hi > lo)) break;   // the interval collapsed to neighbouring floating-point numbers
  }
  if (lane == 0) w[j] = 0.5 * (lo + hi);
}

// out[0] = sum_i f(|w_i|), f(s) = s < eps ? s : sqrt(s) (tfgan's _symmetric_matrix_square_root rule);
// out[1] = bound on the error of that sum when the w_i are the exact eigenvalues of A + E with
// |E|_F <= delta = delta_f_rel * *fro and |E|_2 <= d2 = delta_2_rel * *fro.  Weyl: every w_i is within
// d2 of its eigenvalue; Hoffman-Wielandt: sum_i (w_i - lambda_i)^2 <= |E|_F^2, so with g_i = sup |f'|
// over [s_i - d2, s_i + d2] (1 / (2 sqrt(s_i - d2)) above the cut-off, 1 below) Cauchy-Schwarz gives
// |sum f(w_i) - sum f(lambda_i)| <= delta * sqrt(sum g_i^2); a value within d2 of the cut-off, where f
// jumps from eps to sqrt(eps), is charged sqrt(eps + 2 d2) on its own.
__global__ __launch_bounds__(256) void spectral_sqrt_bound_kernel(const double* __restrict__ w, int n,
                                                                  double eps, double delta_f_rel,
                                                                  double delta_2_rel,
                                                                  const double* __restrict__ fro,
                                                                  double* __restrict__ out) {
  __shared__ double sm[3][4];
  const double delta = delta_f_rel * *fro, d2 = delta_2_rel * *fro;
  double s = 0.0, g2 = 0.0, amb = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double a = fabs(w[i]);
    s += a < eps ? a : sqrt(a);
    if (a - d2 >= eps) g2 += 1.0 / (4.0 * (a - d2));
    else if (a + d2 < eps) g2 += 1.0;
    else amb += sqrt(eps + 2.0 * d2);
  }
  s = wave_sum_d(s);
  g2 = wave_sum_d(g2);
  amb = wave_sum_d(amb);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = s;
    sm[1][threadIdx.x >> 6] = g2;
    sm[2][threadIdx.x >> 6] = amb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
    out[1] = delta * sqrt(sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3]) +
             (sm[2][0] + sm[2][1] + sm[2][2] + sm[2][3]);
  }
}

size_t td_align(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

extern "C" size_t cg_sytrd_eigvals_workspace_bytes(int n) {
  if (n <= 0) return 0;
  // ctr (256 B) + vb [2n] + wv [n] + pv [n] + col [n] + dd [n] + ee [n] + tau [2] + bnd [4]
  return 256 + td_align((size_t)(7 * (size_t)n + 6) * sizeof(double));
}

extern "C" int cg_sytrd_eigvals_f64(double* a, int n, double* w, double* fro_out, void* ws,
                                    size_t ws_bytes, cgStream stream) {
  if (!a || !w || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_sytrd_eigvals_f64: bad argument");
  if (n > TD_MAX_N)
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_sytrd_eigvals_f64: n = %d exceeds %d", n, TD_MAX_N);
  if (!ws || ws_bytes < cg_sytrd_eigvals_workspace_bytes(n))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_sytrd_eigvals_f64: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int* ctr = (int*)ws;
  double* vb = reinterpret_cast<double*>((char*)ws + 256);
  double* wv = vb + 2 * (size_t)n;
  double* pv = wv + n;
  double* col = pv + n;
  double* dd = col + n;
  double* ee = dd + n;
  double* tau = ee + n;
  double* bnd = tau + 2;
  const size_t lds = (size_t)3 * n * sizeof(double);
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute((const void*)td_p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              3 * TD_MAX_N * (int)sizeof(double));
    return true;
  }();
  (void)attr_set;
  td_init_kernel<<<1, 1, 0, st>>>(ctr, n);
  CG_CHECK_LAUNCH("cg_sytrd_eigvals_f64(init)");
  auto launch_chunk = [&](hipStream_t q, int steps) {
    for (int s = 0; s < steps; ++s) {
      td_s_kernel<<<1, 1024, 0, q>>>(a, n, ctr, vb, wv, pv, tau, col, dd, ee);
      td_p_kernel<<<TD_P_BLOCKS, 256, lds, q>>>(a, n, ctr, vb, wv, pv);
    }
  };
  const int chunk = n < TD_CHUNK ? n : TD_CHUNK;
  const int replays = (n + chunk - 1) / chunk;
  static const int graph_env = []() {
    const char* e = getenv("CGAMD_TRIDIAG_GRAPH");
    return e ? atoi(e) : 1;
  }();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  hipStreamIsCapturing(st, &cap);
  bool done = false;
  if (graph_env && cap == hipStreamCaptureStatusNone && replays > 1) {
    // captured on a private stream (the caller's may be the legacy default stream, which cannot be
    // captured), ordered behind / ahead of the caller's stream by events -- as cg_syevj_f64 does
    static hipStream_t js = nullptr;
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!js) {
      hipStreamCreateWithFlags(&js, hipStreamNonBlocking);
      hipEventCreateWithFlags(&e0, hipEventDisableTiming);
      hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEventRecord(e0, st);
    hipStreamWaitEvent(js, e0, 0);
    if (hipStreamBeginCapture(js, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      launch_chunk(js, chunk);
      if (hipStreamEndCapture(js, &graph) == hipSuccess && graph &&
          hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        for (int r = 0; r < replays; ++r) hipGraphLaunch(exec, js);
        hipStreamSynchronize(js);   // the graph objects are destroyed below
        done = true;
      }
    }
    (void)hipGetLastError();
    if (exec) hipGraphExecDestroy(exec);
    if (graph) hipGraphDestroy(graph);
    hipEventRecord(e1, js);
    hipStreamWaitEvent(st, e1, 0);
    if (!done) {   // a failed capture may have left the counters anywhere: start over
      td_init_kernel<<<1, 1, 0, st>>>(ctr, n);
    }
  }
  if (!done) launch_chunk(st, n);
  CG_CHECK_LAUNCH("cg_sytrd_eigvals_f64(steps)");
  td_bounds_kernel<<<1, 256, 0, st>>>(dd, ee, n, bnd);
  td_bisect_kernel<<<n, 64, 0, st>>>(dd, ee, n, bnd, w, fro_out);
  CG_CHECK_LAUNCH("cg_sytrd_eigvals_f64(bisection)");
  return CG_OK;
}

extern "C" int cg_spectral_sqrt_bound_f64(const double* w, int n, double eps, double delta_f_rel,
                                          double delta_2_rel, const double* fro, double* out2,
                                          cgStream stream) {
  if (!w || !fro || !out2 || n <= 0 || !(delta_f_rel >= 0.0) || !(delta_2_rel >= 0.0))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_spectral_sqrt_bound_f64: bad argument");
  spectral_sqrt_bound_kernel<<<1, 256, 0, (hipStream_t)stream>>>(w, n, eps, delta_f_rel, delta_2_rel,
                                                                fro, out2);
  CG_CHECK_LAUNCH("cg_spectral_sqrt_bound_f64");
  return CG_OK;
}
