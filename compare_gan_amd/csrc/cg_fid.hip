// FID / Inception-score statistics in fp64 (metrics/fid_score.py:44-75,
// metrics/inception_score.py:39-48; arithmetic restated from tensorflow_gan):
// centred covariance GEMM, general fp64 GEMM, parallel one-sided Jacobi eigensolver for the
// symmetric square roots, and the classifier score.  v1 kernels are LDS-tiled vector-ALU fp64.
#include "cg_common.h"
#include <stdlib.h>

namespace {

// ---- column means in fp64 ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void colmean_part_kernel(const float* __restrict__ x, int64_t n,
                                                           int d, int64_t rows_per_split,
                                                           double* __restrict__ part) {
  __shared__ double sm[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split, r1 = min(n, r0 + rows_per_split);
  double s = 0.0;
  if (c < d)
    for (int64_t r = r0 + w; r < r1; r += 4) s += (double)x[r * d + c];
  sm[w][l] = s;
  __syncthreads();
  if (w == 0 && c < d) part[(int64_t)blockIdx.y * d + c] = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
}
__global__ void colmean_final_kernel(const double* __restrict__ part, int splits, int d,
                                     double inv_n, double* __restrict__ mean) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  double s = 0.0;
  for (int z = 0; z < splits; ++z) s += part[(int64_t)z * d + c];
  mean[c] = s * inv_n;
}

// ---- tiled fp64 GEMM: C[m,n] = scale * sum_k A(i,k) B(k,j) ------------------------------------
// MODE 0: A,B fp64 with optional transposes.  MODE 1: covariance: A = B = (x - mean)^T from fp32 x.
constexpr int GT = 64, GK = 16;
template <int MODE>
__global__ __launch_bounds__(256) void gemm_f64_kernel(const void* __restrict__ ap,
                                                       const void* __restrict__ bp,
                                                       const double* __restrict__ mean,
                                                       double* __restrict__ c, int M, int N,
                                                       int64_t K, int ta, int tb, double scale,
                                                       double eye) {
  __shared__ double As[GK][GT + 1];
  __shared__ double Bs[GK][GT + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int64_t k0 = 0; k0 < K; k0 += GK) {
    // load A tile [GK][GT] (k-major) and B tile [GK][GT]
    for (int e = threadIdx.x; e < GK * GT; e += 256) {
      double av = 0.0, bv = 0.0;
      if (MODE == 1) {
        const int kk = e / GT, ii = e - kk * GT;  // consecutive threads -> consecutive columns
        const int64_t k = k0 + kk;
        const float* x = (const float*)ap;
        if (k < K && i0 + ii < M) av = (double)x[k * M + i0 + ii] - mean[i0 + ii];
        if (k < K && j0 + ii < N) bv = (double)x[k * N + j0 + ii] - mean[j0 + ii];
        As[kk][ii] = av;
        Bs[kk][ii] = bv;
      } else {
        const double* A = (const double*)ap;
        const double* B = (const double*)bp;
        {
          int kk, ii;
          if (ta) { kk = e / GT; ii = e - kk * GT; } else { ii = e / GK; kk = e - ii * GK; }
          const int64_t k = k0 + kk;
          if (k < K && i0 + ii < M) av = ta ? A[k * M + i0 + ii] : A[(int64_t)(i0 + ii) * K + k];
          As[kk][ii] = av;
        }
        {
          int kk, jj;
          if (tb) { jj = e / GK; kk = e - jj * GK; } else { kk = e / GT; jj = e - kk * GT; }
          const int64_t k = k0 + kk;
          if (k < K && j0 + jj < N) bv = tb ? B[(int64_t)(j0 + jj) * K + k] : B[k * N + j0 + jj];
          Bs[kk][jj] = bv;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = As[kk][ty * 4 + u];
        b[u] = Bs[kk][tx * 4 + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + ty * 4 + u, j = j0 + tx * 4 + v;
      if (i < M && j < N) c[(int64_t)i * N + j] = acc[u][v] * scale + (i == j ? eye : 0.0);
    }
}

// ---- one-sided Jacobi (Hestenes) on the ROWS of G (= A, symmetric) and V (= I) ---------------
// Round r of a sweep pairs rows by the circle method; one block per pair.
__global__ void jacobi_init_kernel(double* __restrict__ v, int d, int* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v && i < (int64_t)d * d) v[i] = (i / d == i % d) ? 1.0 : 0.0;
  if (i == 0) {
    flags[0] = 0;  // converged
    flags[1] = 0;  // rotations in current sweep
    *reinterpret_cast<unsigned long long*>(flags + 2) = 0ull;   // largest squared row norm
  }
}
// largest squared row norm of A (ordered as an integer: the values are non-negative doubles)
__global__ __launch_bounds__(256) void jacobi_scale_kernel(const double* __restrict__ a, int d,
                                                           int* __restrict__ flags) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int k = threadIdx.x; k < d; k += 256) {
    const double x = a[(int64_t)blockIdx.x * d + k];
    s += x * x;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(flags + 2),
              (unsigned long long)__double_as_longlong(sm[0] + sm[1] + sm[2] + sm[3]));
}
// rows whose squared norm fell below this fraction of the largest one are numerically zero (the
// null space of a rank-deficient matrix): rotating them is noise and would keep the sweeps from
// ever reporting convergence
constexpr double JACOBI_NULL_ROW = 1e-26;
__global__ __launch_bounds__(256) void jacobi_round_kernel(double* __restrict__ g,
                                                           double* __restrict__ v, int d, int np,
                                                           int round, double tol,
                                                           int* __restrict__ flags) {
  if (flags[0]) return;
  __shared__ double sm[3][4];
  __shared__ double s_c, s_s;
  __shared__ int s_rot;
  const int i = blockIdx.x;
  const int m = np - 1;  // np even (padded)
  int p, q;
  if (i == 0) {
    p = np - 1;
    q = round % m;
  } else {
    p = (round + i) % m;
    q = (round - i + m) % m;
  }
  if (p >= d || q >= d) return;  // dummy player
  if (p > q) { const int t = p; p = q; q = t; }
  double* gp = g + (int64_t)p * d;
  double* gq = g + (int64_t)q * d;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int k = threadIdx.x; k < d; k += 256) {
    const double x = gp[k], y = gq[k];
    a += x * x;
    b += y * y;
    c += x * y;
  }
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = a;
    sm[1][threadIdx.x >> 6] = b;
    sm[2][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double al = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
    const double be = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    const double ga = sm[2][0] + sm[2][1] + sm[2][2] + sm[2][3];
    int rot = 0;
    double cs = 1.0, sn = 0.0;
    const double thr = JACOBI_NULL_ROW * __longlong_as_double(
        (long long)*reinterpret_cast<const unsigned long long*>(flags + 2));
    if (fabs(ga) > tol * sqrt(al * be) && fabs(ga) > 1e-300 && al > thr && be > thr) {
      const double zeta = (be - al) / (2.0 * ga);
      const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      cs = 1.0 / sqrt(1.0 + t * t);
      sn = cs * t;
      rot = 1;
    }
    s_c = cs;
    s_s = sn;
    s_rot = rot;
    if (rot) atomicAdd(&flags[1], 1);
  }
  __syncthreads();
  if (!s_rot) return;
  const double cs = s_c, sn = s_s;
  double* vp = v + (int64_t)p * d;
  double* vq = v + (int64_t)q * d;
  for (int k = threadIdx.x; k < d; k += 256) {
    const double x = gp[k], y = gq[k];
    gp[k] = cs * x - sn * y;
    gq[k] = sn * x + cs * y;
    const double vx = vp[k], vy = vq[k];
    vp[k] = cs * vx - sn * vy;
    vq[k] = sn * vx + cs * vy;
  }
}
// ---- block form of the same method --------------------------------------------------------------
// The scalar round above moves all of G and V (2 x d^2 doubles, read + write) for ONE rotation per
// row pair: bandwidth bound, ~27 us per round, 2047 rounds per sweep at d = 2048.  Here the rows
// form nb = d / 16 blocks; a round pairs the BLOCKS by the circle method (nb - 1 rounds per sweep)
// and one workgroup handles a pair of blocks = 32 rows:
//   1. Gram matrix M = X X^T of its 32 rows (32 x 32, the inner products every rotation needs);
//   2. one cyclic Jacobi sweep over all 496 row pairs ON M (two-sided rotations M <- R M R^T, the
//      exact image of rotating the rows), accumulating the product J of the rotations;
//   3. X <- J X for the rows of G and of V (one pass).
// One pass over the data now carries 496 rotations instead of 16, the traffic per sweep drops 16x.
// Same rotation rule and threshold as the scalar kernel; deterministic.
constexpr int JB = 16;            // rows per block
constexpr int JR = 2 * JB;        // rows per workgroup
constexpr int JCW = 64;           // columns per staged chunk
__global__ __launch_bounds__(256) void bjacobi_round_kernel(double* __restrict__ g,
                                                            double* __restrict__ v, int d, int nb,
                                                            int round, double tol,
                                                            int* __restrict__ flags) {
  if (flags[0]) return;
  __shared__ double Tt[JCW][JR + 2];       // chunk, column-major: Tt[k][row]
  __shared__ double Mp[4][JR][JR + 1];     // per-wave Gram partials
  __shared__ double M[JR][JR + 1];
  __shared__ double J[JR][JR + 1];
  __shared__ double s_c[JB], s_s[JB];
  __shared__ int s_p[JB], s_q[JB], s_rot;
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  // circle method over the nb blocks (nb even)
  const int i = blockIdx.x, m = nb - 1;
  int P, Q;
  if (i == 0) { P = nb - 1; Q = round % m; }
  else { P = (round + i) % m; Q = (round - i + m) % m; }
  if (P > Q) { const int x = P; P = Q; Q = x; }
  auto grow = [&](int r) -> int64_t { return (int64_t)(r < JB ? P * JB + r : Q * JB + r - JB); };

  // ---- 1. Gram matrix: thread (ti, tj) of every wave owns a 4 x 4 block, waves split the columns
  {
    const int ti = l >> 3, tj = l & 7;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    // the next chunk's loads are in flight while this one is multiplied (only 64 workgroups are
    // resident: a round is bound by the latency of its serial chunk loads, not by bandwidth)
    double cur[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = t + 256 * e;
      cur[e] = g[grow(idx >> 6) * d + (idx & 63)];
    }
    for (int c0 = 0; c0 < d; c0 += JCW) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = t + 256 * e;
        Tt[idx & 63][idx >> 6] = cur[e];
      }
      __syncthreads();
      if (c0 + JCW < d) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int idx = t + 256 * e;
          cur[e] = g[grow(idx >> 6) * d + c0 + JCW + (idx & 63)];
        }
      }
      for (int k = w; k < JCW; k += 4) {
        double a[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = Tt[k][4 * ti + e]; b[e] = Tt[k][4 * tj + e]; }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[x][y] += a[x] * b[y];
      }
      __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) Mp[w][4 * ti + x][4 * tj + y] = acc[x][y];
  }
  if (t == 0) s_rot = 0;
  __syncthreads();
  for (int e = t; e < JR * JR; e += 256) {
    const int r = e / JR, c = e - r * JR;
    M[r][c] = Mp[0][r][c] + Mp[1][r][c] + Mp[2][r][c] + Mp[3][r][c];
    J[r][c] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();

  // ---- 2. one cyclic sweep on M (circle method over the 32 rows: 31 rounds of 16 disjoint pairs)
  for (int ir = 0; ir < JR - 1; ++ir) {
    if (t < JB) {
      const int mm = JR - 1;
      int p, q;
      if (t == 0) { p = JR - 1; q = ir % mm; }
      else { p = (ir + t) % mm; q = (ir - t + mm) % mm; }
      if (p > q) { const int x = p; p = q; q = x; }
      const double al = M[p][p], be = M[q][q], ga = M[p][q];
      double cs = 1.0, sn = 0.0;
      const double thr = JACOBI_NULL_ROW * __longlong_as_double(
          (long long)*reinterpret_cast<const unsigned long long*>(flags + 2));
      if (fabs(ga) > tol * sqrt(al * be) && fabs(ga) > 1e-300 && al > thr && be > thr) {
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        cs = 1.0 / sqrt(1.0 + tt * tt);
        sn = cs * tt;
        atomicAdd(&s_rot, 1);
      }
      s_c[t] = cs; s_s[t] = sn; s_p[t] = p; s_q[t] = q;
    }
    __syncthreads();
    {   // rows p, q of M and J <- R (rows)
      const int pi = t >> 4, p = s_p[pi], q = s_q[pi];
      const double cs = s_c[pi], sn = s_s[pi];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = (t & 15) + 16 * h;
        const double mp = M[p][c], mq = M[q][c];
        M[p][c] = cs * mp - sn * mq;
        M[q][c] = sn * mp + cs * mq;
        const double jp = J[p][c], jq = J[q][c];
        J[p][c] = cs * jp - sn * jq;
        J[q][c] = sn * jp + cs * jq;
      }
    }
    __syncthreads();
    {   // columns p, q of M <- (M) R^T
      const int pi = t >> 4, p = s_p[pi], q = s_q[pi];
      const double cs = s_c[pi], sn = s_s[pi];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = (t & 15) + 16 * h;
        const double mp = M[r][p], mq = M[r][q];
        M[r][p] = cs * mp - sn * mq;
        M[r][q] = sn * mp + cs * mq;
      }
    }
    __syncthreads();
  }
  if (s_rot == 0) return;          // nothing rotated: rows already orthogonal to the threshold
  if (t == 0) atomicAdd(&flags[1], s_rot);

  // ---- 3. rows <- J rows, for G and V; thread tile 4 rows x 2 columns of a 32 x 64 chunk
  const int ti = t >> 5, tj = t & 31;
  // chunks of G then of V as one stream of 2 * d / JCW chunks, the next one prefetched
  const int nchunks = 2 * (d / JCW);
  auto chunk_ptr = [&](int ci) -> double* { return (ci < d / JCW ? g : v); };
  auto chunk_col = [&](int ci) -> int { return (ci < d / JCW ? ci : ci - d / JCW) * JCW; };
  double cur[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int idx = t + 256 * e;
    cur[e] = g[grow(idx >> 6) * d + (idx & 63)];
  }
  for (int ci = 0; ci < nchunks; ++ci) {
    double* __restrict__ x = chunk_ptr(ci);
    const int c0 = chunk_col(ci);
    {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = t + 256 * e;
        Tt[idx & 63][idx >> 6] = cur[e];
      }
      __syncthreads();
      if (ci + 1 < nchunks) {
        const double* __restrict__ xn = chunk_ptr(ci + 1);
        const int cn = chunk_col(ci + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int idx = t + 256 * e;
          cur[e] = xn[grow(idx >> 6) * d + cn + (idx & 63)];
        }
      }
      double acc[4][2];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a][0] = acc[a][1] = 0.0;
#pragma unroll 8
      for (int sI = 0; sI < JR; ++sI) {
        const double b0 = Tt[2 * tj][sI], b1 = Tt[2 * tj + 1][sI];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const double ja = J[4 * ti + a][sI];
          acc[a][0] += ja * b0;
          acc[a][1] += ja * b1;
        }
      }
      __syncthreads();             // every read of this chunk is done before it is overwritten
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double* o = x + grow(4 * ti + a) * d + c0 + 2 * tj;
        o[0] = acc[a][0];
        o[1] = acc[a][1];
      }
    }
  }
}
// ---- the block round as three kernels ---------------------------------------------------------
// The fused round above runs nb / 2 = 64 workgroups at d = 2048 (a quarter of the CUs), each a
// serial chain of 32 + 64 chunk loads: 88 us per round, 15 k rounds per FID
// (profiles/r02_fid10k_kernel_stats.csv).  Split by phase the same arithmetic fills the chip:
//   gram  : (pair, column split) -> partial Gram matrices of the pair's 32 rows   [nb/2 x GS groups]
//   sweep : pair -> sum of the partials, one cyclic sweep on it, rotation product J [nb/2 groups]
//   apply : (pair, chunk range of G | V) -> rows <- J rows                          [nb/2 x AS groups]
// Sums run in the same order as in the fused kernel within a split; the partials of the GS splits
// are added in a fixed order: deterministic.
__device__ __forceinline__ void bj_pair(int i, int nb, int round, int* P, int* Q) {
  const int m = nb - 1;
  int p, q;
  if (i == 0) { p = nb - 1; q = round % m; }
  else { p = (round + i) % m; q = (round - i + m) % m; }
  if (p > q) { const int x = p; p = q; q = x; }
  *P = p; *Q = q;
}
__global__ __launch_bounds__(256) void bj_gram_kernel(const double* __restrict__ g, int d, int nb,
                                                      int round, int gs,
                                                      double* __restrict__ mpart,
                                                      const int* __restrict__ flags) {
  if (flags[0]) return;
  __shared__ double Tt[JCW][JR + 2];
  __shared__ double Mp[4][JR][JR + 1];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  int P, Q;
  bj_pair(blockIdx.x, nb, round, &P, &Q);
  auto grow = [&](int r) -> int64_t { return (int64_t)(r < JB ? P * JB + r : Q * JB + r - JB); };
  const int cbeg = blockIdx.y * (d / gs), cend = cbeg + d / gs;
  const int ti = l >> 3, tj = l & 7;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  double cur[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int idx = t + 256 * e;
    cur[e] = g[grow(idx >> 6) * d + cbeg + (idx & 63)];
  }
  for (int c0 = cbeg; c0 < cend; c0 += JCW) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = t + 256 * e;
      Tt[idx & 63][idx >> 6] = cur[e];
    }
    __syncthreads();
    if (c0 + JCW < cend) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = t + 256 * e;
        cur[e] = g[grow(idx >> 6) * d + c0 + JCW + (idx & 63)];
      }
    }
    for (int k = w; k < JCW; k += 4) {
      double a[4], b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = Tt[k][4 * ti + e]; b[e] = Tt[k][4 * tj + e]; }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] += a[x] * b[y];
    }
    __syncthreads();
  }
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) Mp[w][4 * ti + x][4 * tj + y] = acc[x][y];
  __syncthreads();
  double* out = mpart + ((int64_t)blockIdx.x * gs + blockIdx.y) * (JR * JR);
  for (int e = t; e < JR * JR; e += 256) {
    const int r = e / JR, c = e - r * JR;
    out[e] = Mp[0][r][c] + Mp[1][r][c] + Mp[2][r][c] + Mp[3][r][c];
  }
}
__global__ __launch_bounds__(256) void bj_sweep_kernel(const double* __restrict__ mpart, int gs,
                                                       double tol, double* __restrict__ jg,
                                                       int* __restrict__ rot,
                                                       int* __restrict__ flags) {
  if (flags[0]) return;
  __shared__ double M[JR][JR + 1];
  __shared__ double J[JR][JR + 1];
  __shared__ double s_c[JB], s_s[JB];
  __shared__ int s_p[JB], s_q[JB], s_rot;
  const int t = threadIdx.x;
  if (t == 0) s_rot = 0;
  const double* mp = mpart + (int64_t)blockIdx.x * gs * (JR * JR);
  for (int e = t; e < JR * JR; e += 256) {
    const int r = e / JR, c = e - r * JR;
    double sum = mp[e];
    for (int k = 1; k < gs; ++k) sum += mp[(int64_t)k * (JR * JR) + e];
    M[r][c] = sum;
    J[r][c] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();
  // (two or three cyclic sweeps over the pair's Gram matrix per round do NOT save outer sweeps: FID-10k
  // statistics 0.241 -> 0.306 -> 0.385 s, profiles/r06_fid_stats.txt)
  for (int ir = 0; ir < JR - 1; ++ir) {
    if (t < JB) {
      const int mm = JR - 1;
      int p, q;
      if (t == 0) { p = JR - 1; q = ir % mm; }
      else { p = (ir + t) % mm; q = (ir - t + mm) % mm; }
      if (p > q) { const int x = p; p = q; q = x; }
      const double al = M[p][p], be = M[q][q], ga = M[p][q];
      double cs = 1.0, sn = 0.0;
      const double thr = JACOBI_NULL_ROW * __longlong_as_double(
          (long long)*reinterpret_cast<const unsigned long long*>(flags + 2));
      if (fabs(ga) > tol * sqrt(al * be) && fabs(ga) > 1e-300 && al > thr && be > thr) {
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        cs = 1.0 / sqrt(1.0 + tt * tt);
        sn = cs * tt;
        atomicAdd(&s_rot, 1);
      }
      s_c[t] = cs; s_s[t] = sn; s_p[t] = p; s_q[t] = q;
    }
    __syncthreads();
    {
      const int pi = t >> 4, p = s_p[pi], q = s_q[pi];
      const double cs = s_c[pi], sn = s_s[pi];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = (t & 15) + 16 * h;
        const double mpv = M[p][c], mq = M[q][c];
        M[p][c] = cs * mpv - sn * mq;
        M[q][c] = sn * mpv + cs * mq;
        const double jp = J[p][c], jq = J[q][c];
        J[p][c] = cs * jp - sn * jq;
        J[q][c] = sn * jp + cs * jq;
      }
    }
    __syncthreads();
    {
      const int pi = t >> 4, p = s_p[pi], q = s_q[pi];
      const double cs = s_c[pi], sn = s_s[pi];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = (t & 15) + 16 * h;
        const double mpv = M[r][p], mq = M[r][q];
        M[r][p] = cs * mpv - sn * mq;
        M[r][q] = sn * mpv + cs * mq;
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    rot[blockIdx.x] = s_rot;
    if (s_rot) atomicAdd(&flags[1], s_rot);
  }
  if (s_rot == 0) return;
  double* out = jg + (int64_t)blockIdx.x * (JR * JR);
  for (int e = t; e < JR * JR; e += 256) out[e] = J[e / JR][e % JR];
}
__global__ __launch_bounds__(256) void bj_apply_kernel(double* __restrict__ g,
                                                       double* __restrict__ v, int d, int nb,
                                                       int round, int as,
                                                       const double* __restrict__ jg,
                                                       const int* __restrict__ rot,
                                                       const int* __restrict__ flags) {
  if (flags[0] || rot[blockIdx.x] == 0) return;
  __shared__ double Tt[JCW][JR + 2];
  __shared__ double J[JR][JR + 1];
  const int t = threadIdx.x;
  int P, Q;
  bj_pair(blockIdx.x, nb, round, &P, &Q);
  auto grow = [&](int r) -> int64_t { return (int64_t)(r < JB ? P * JB + r : Q * JB + r - JB); };
  for (int e = t; e < JR * JR; e += 256) J[e / JR][e % JR] = jg[(int64_t)blockIdx.x * (JR * JR) + e];
  const int ti = t >> 5, tj = t & 31;
  const int per = (v ? 2 : 1) * (d / JCW) / as;  // chunks of this workgroup (G chunks, then V chunks)
  const int cbeg = blockIdx.y * per, cendc = cbeg + per;
  auto chunk_ptr = [&](int ci) -> double* { return (ci < d / JCW ? g : v); };
  auto chunk_col = [&](int ci) -> int { return (ci < d / JCW ? ci : ci - d / JCW) * JCW; };
  double cur[8];
  {
    const double* __restrict__ x0 = chunk_ptr(cbeg);
    const int c0 = chunk_col(cbeg);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = t + 256 * e;
      cur[e] = x0[grow(idx >> 6) * d + c0 + (idx & 63)];
    }
  }
  for (int ci = cbeg; ci < cendc; ++ci) {
    double* __restrict__ x = chunk_ptr(ci);
    const int c0 = chunk_col(ci);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = t + 256 * e;
      Tt[idx & 63][idx >> 6] = cur[e];
    }
    __syncthreads();               // (also publishes J on the first pass)
    if (ci + 1 < cendc) {
      const double* __restrict__ xn = chunk_ptr(ci + 1);
      const int cn = chunk_col(ci + 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = t + 256 * e;
        cur[e] = xn[grow(idx >> 6) * d + cn + (idx & 63)];
      }
    }
    double acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a][0] = acc[a][1] = 0.0;
#pragma unroll 8
    for (int sI = 0; sI < JR; ++sI) {
      const double b0 = Tt[2 * tj][sI], b1 = Tt[2 * tj + 1][sI];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double ja = J[4 * ti + a][sI];
        acc[a][0] += ja * b0;
        acc[a][1] += ja * b1;
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double* o = x + grow(4 * ti + a) * d + c0 + 2 * tj;
      o[0] = acc[a][0];
      o[1] = acc[a][1];
    }
  }
}
__global__ void jacobi_sweep_end_kernel(int* flags) {
  if (flags[0]) return;
  if (flags[1] == 0) flags[0] = 1;
  flags[1] = 0;
}
// eigenvalue_i = g_i . v_i  (g_i = A v_i = lambda_i v_i, |v_i| = 1)
__global__ __launch_bounds__(256) void jacobi_eigvals_kernel(const double* __restrict__ g,
                                                             const double* __restrict__ v, int d,
                                                             double* __restrict__ w) {
  __shared__ double sm[4];
  const int i = blockIdx.x;
  double s = 0.0;
  if (v) {
    for (int k = threadIdx.x; k < d; k += 256) s += g[(int64_t)i * d + k] * v[(int64_t)i * d + k];
  } else {   // without the vectors: |lambda_i| = |g_i| (the singular values)
    for (int k = threadIdx.x; k < d; k += 256) s += g[(int64_t)i * d + k] * g[(int64_t)i * d + k];
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) w[i] = v ? sm[0] + sm[1] + sm[2] + sm[3] : sqrt(sm[0] + sm[1] + sm[2] + sm[3]);
}

// ---- inception score ----------------------------------------------------------------------------
// pass 1: q partial sums of softmax rows; pass 2: KL sums.
__global__ __launch_bounds__(256) void is_pass_kernel(const float* __restrict__ logits, int64_t n,
                                                      int k, int64_t rows_per_block, int pass,
                                                      const double* __restrict__ logq,
                                                      double* __restrict__ part) {
  // one wave per row at a time; 4 waves per block
  extern __shared__ double sq[];  // [4][k] for pass 1
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  if (pass == 1)
    for (int c = l; c < k; c += 64) sq[w * k + c] = 0.0;
  double klsum = 0.0;
  for (int64_t r = r0 + w; r < r1; r += 4) {
    const float* row = logits + r * k;
    double mx = -1e300;
    for (int c = l; c < k; c += 64) mx = fmax(mx, (double)row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    double se = 0.0;
    for (int c = l; c < k; c += 64) se += exp((double)row[c] - mx);
    se = wave_sum_d(se);
    const double lse = mx + log(se);
    if (pass == 1) {
      for (int c = l; c < k; c += 64) sq[w * k + c] += exp((double)row[c] - lse);
    } else {
      double kl = 0.0;
      for (int c = l; c < k; c += 64) {
        const double lp = (double)row[c] - lse;
        kl += exp(lp) * (lp - logq[c]);
      }
      klsum += wave_sum_d(kl);
    }
  }
  __syncthreads();
  if (pass == 1) {
    for (int c = threadIdx.x; c < k; c += 256)
      part[(int64_t)blockIdx.x * k + c] = sq[c] + sq[k + c] + sq[2 * k + c] + sq[3 * k + c];
  } else {
    __shared__ double skl[4];
    if (l == 0) skl[w] = klsum;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = skl[0] + skl[1] + skl[2] + skl[3];
  }
}
__global__ void is_logq_kernel(const double* __restrict__ part, int blocks, int k, double inv_n,
                               double* __restrict__ logq) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += part[(int64_t)b * k + c];
  logq[c] = log(s * inv_n);
}
__global__ void is_final_kernel(const double* __restrict__ part, int blocks, double inv_n,
                                double* __restrict__ score) {
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += part[b];
  *score = exp(s * inv_n);
}

inline int mean_splits(int64_t n, int d) {
  const int ct = cdiv(d, 64);
  int s = cdiv(1024, ct);
  const int64_t maxs = n / 16 > 0 ? n / 16 : 1;
  if (s > maxs) s = (int)maxs;
  if (s < 1) s = 1;
  return s;
}
inline int is_blocks(int64_t n) {
  int64_t b = (n + 15) / 16;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t cg_mean_cov_workspace_bytes(int64_t n, int d) {
  if (n <= 0 || d <= 0) return 0;
  return align_up((size_t)mean_splits(n, d) * d * sizeof(double), 256);
}

extern "C" int cg_mean_cov_f64(const float* x, int64_t n, int d, double* mean, double* cov,
                               void* ws, size_t ws_bytes, cgStream stream) {
  if (!x || !mean || !cov || n <= 1 || d <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_mean_cov_f64: bad argument (need n >= 2)");
  if (!ws || ws_bytes < cg_mean_cov_workspace_bytes(n, d))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_mean_cov_f64: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int splits = mean_splits(n, d);
  const int64_t rps = (n + splits - 1) / splits;
  dim3 g1(cdiv(d, 64), splits);
  colmean_part_kernel<<<g1, 256, 0, st>>>(x, n, d, rps, (double*)ws);
  CG_CHECK_LAUNCH("cg_mean_cov_f64(mean part)");
  colmean_final_kernel<<<cdiv(d, 256), 256, 0, st>>>((const double*)ws, splits, d, 1.0 / (double)n,
                                                     mean);
  CG_CHECK_LAUNCH("cg_mean_cov_f64(mean)");
  dim3 g2(cdiv(d, GT), cdiv(d, GT));
  gemm_f64_kernel<1><<<g2, 256, 0, st>>>(x, x, mean, cov, d, d, n, 1, 0, 1.0 / (double)(n - 1), 0.0);
  CG_CHECK_LAUNCH("cg_mean_cov_f64(cov)");
  return CG_OK;
}

extern "C" int cg_gemm_f64(const double* a, const double* b, double* c, int m, int n, int k,
                           int ta, int tb, cgStream stream) {
  if (!a || !b || !c || m <= 0 || n <= 0 || k <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gemm_f64: bad argument");
  dim3 grid(cdiv(n, GT), cdiv(m, GT));
  gemm_f64_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(a, b, nullptr, c, m, n, k, ta, tb, 1.0,
                                                            0.0);
  CG_CHECK_LAUNCH("cg_gemm_f64");
  return CG_OK;
}

extern "C" int cg_gemm_f64_ex(const double* a, const double* b, double* c, int m, int n, int k,
                              int ta, int tb, double alpha, double beta_eye, cgStream stream) {
  if (!a || !b || !c || m <= 0 || n <= 0 || k <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gemm_f64_ex: bad argument");
  dim3 grid(cdiv(n, GT), cdiv(m, GT));
  gemm_f64_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(a, b, nullptr, c, m, n, k, ta, tb, alpha,
                                                            beta_eye);
  CG_CHECK_LAUNCH("cg_gemm_f64_ex");
  return CG_OK;
}

// out = alpha * a + beta_eye * I  (n x n, fp64)
__global__ void axpby_eye_f64_kernel(const double* __restrict__ a, double alpha, double eye,
                                     double* __restrict__ out, int n) {
  const int64_t total = (int64_t)n * n, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
    out[i] = alpha * a[i] + (r == c ? eye : 0.0);
  }
}

extern "C" int cg_axpby_eye_f64(const double* a, double alpha, double beta_eye, double* out, int n,
                                cgStream stream) {
  if (!a || !out || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_axpby_eye_f64: bad argument");
  int64_t b = ((int64_t)n * n + 255) / 256;
  if (b > 4096) b = 4096;
  axpby_eye_f64_kernel<<<(int)b, 256, 0, (hipStream_t)stream>>>(a, alpha, beta_eye, out, n);
  CG_CHECK_LAUNCH("cg_axpby_eye_f64");
  return CG_OK;
}

// out[0] = trace(a), out[1] = sum of squares of a, out[2] = smallest diagonal entry (n x n, fp64);
// deterministic two-stage sums
constexpr int MS_BLOCKS = 256;
__global__ __launch_bounds__(256) void mat_stats_part_kernel(const double* __restrict__ a, int n,
                                                             double* __restrict__ part) {
  __shared__ double sm[3][4];
  const int64_t total = (int64_t)n * n, stride = (int64_t)gridDim.x * blockDim.x;
  double tr = 0.0, sq = 0.0, dmin = 1e300;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const double v = a[i];
    sq += v * v;
    if (i % (n + 1) == 0) {
      tr += v;
      dmin = fmin(dmin, v);
    }
  }
  tr = wave_sum_d(tr);
  sq = wave_sum_d(sq);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmin = fmin(dmin, __shfl_xor(dmin, o, 64));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[0][w] = tr;
    sm[1][w] = sq;
    sm[2][w] = dmin;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 3 + 0] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
    part[blockIdx.x * 3 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    part[blockIdx.x * 3 + 2] = fmin(fmin(sm[2][0], sm[2][1]), fmin(sm[2][2], sm[2][3]));
  }
}
__global__ void mat_stats_final_kernel(const double* __restrict__ part, double* __restrict__ out) {
  double tr = 0.0, sq = 0.0, dmin = 1e300;
  for (int b = 0; b < MS_BLOCKS; ++b) {
    tr += part[b * 3 + 0];
    sq += part[b * 3 + 1];
    dmin = fmin(dmin, part[b * 3 + 2]);
  }
  out[0] = tr;
  out[1] = sq;
  out[2] = dmin;
}

extern "C" size_t cg_mat_stats_workspace_bytes(void) { return MS_BLOCKS * 3 * sizeof(double); }
extern "C" int cg_mat_stats_f64(const double* a, int n, double* out2, void* ws, size_t ws_bytes,
                                cgStream stream) {   // out2: three doubles
  if (!a || !out2 || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_mat_stats_f64: bad argument");
  if (!ws || ws_bytes < cg_mat_stats_workspace_bytes())
    CG_FAIL(CG_ERR_WORKSPACE, "cg_mat_stats_f64: workspace too small");
  mat_stats_part_kernel<<<MS_BLOCKS, 256, 0, (hipStream_t)stream>>>(a, n, (double*)ws);
  mat_stats_final_kernel<<<1, 1, 0, (hipStream_t)stream>>>((const double*)ws, out2);
  CG_CHECK_LAUNCH("cg_mat_stats_f64");
  return CG_OK;
}

__global__ void rowscale_f64_kernel(const double* __restrict__ a, const double* __restrict__ sc,
                                    double* __restrict__ out, int64_t total, int cols) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    out[i] = a[i] * sc[i / cols];
}

extern "C" int cg_rowscale_f64(const double* a, const double* scale, double* out, int rows,
                               int cols, cgStream stream) {
  if (!a || !scale || !out || rows <= 0 || cols <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_rowscale_f64: bad argument");
  const int64_t total = (int64_t)rows * cols;
  int64_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  rowscale_f64_kernel<<<(int)b, 256, 0, (hipStream_t)stream>>>(a, scale, out, total, cols);
  CG_CHECK_LAUNCH("cg_rowscale_f64");
  return CG_OK;
}

// flags (256 B) + for the split block rounds: partial Gram matrices [nb/2][4][32 x 32], rotation
// products [nb/2][32 x 32], rotation counts [nb/2]
extern "C" size_t cg_syevj_workspace_bytes(int d) {
  if (d <= 0) return 0;
  const size_t pairs = (size_t)(d / JB + 1) / 2 + 1;
  return 256 + pairs * 5 * JR * JR * sizeof(double) + align_up(pairs * sizeof(int), 256);
}

extern "C" int cg_syevj_f64(double* a, int d, double* w, double* v, int max_sweeps, double tol,
                            void* ws, size_t ws_bytes, cgStream stream) {
  if (!a || !w || d <= 0 || max_sweeps <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_syevj_f64: bad argument");
  if (!ws || ws_bytes < cg_syevj_workspace_bytes(d))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_syevj_f64: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int* flags = (int*)ws;
  jacobi_init_kernel<<<cdiv((int64_t)d * d, 256), 256, 0, st>>>(v, d, flags);
  jacobi_scale_kernel<<<d, 256, 0, st>>>(a, d, flags);
  CG_CHECK_LAUNCH("cg_syevj_f64(init)");
  const int np = (d + 1) & ~1;
  static const int block_min = []() {
    const char* e = getenv("CGAMD_JACOBI_BLOCK_MIN");   // smallest d for the block form (0 = off)
    return e ? atoi(e) : 256;
  }();
  const bool block_form = block_min > 0 && d >= block_min && d % JCW == 0;
  if (!v && !block_form)
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_syevj_f64: values-only (v = NULL) needs d >= %d, d %% %d == 0",
            block_min, JCW);
  if (block_form) {   // whole 64-column chunks, nb even
    const int nb = d / JB;   // even
    static const int split_env0 = []() {
      const char* e = getenv("CGAMD_JACOBI_SPLIT");   // 0: the fused round kernel (A/B)
      return e ? atoi(e) : 1;
    }();
    const int split_env = v ? split_env0 : 1;   // (the fused round kernel always carries V)
    const int chunks = d / JCW;
    const int gs = (chunks % 4 == 0) ? 4 : ((chunks % 2 == 0) ? 2 : 1);
    const int ac = (v ? 2 : 1) * chunks;        // chunks the apply pass walks: G (and V)
    const int as = ac % 8 == 0 ? 8 : (ac % 4 == 0 ? 4 : (ac % 2 == 0 ? 2 : 1));
    double* mpart = reinterpret_cast<double*>((char*)ws + 256);
    double* jg = mpart + (size_t)(nb / 2) * 4 * JR * JR;
    int* rot = reinterpret_cast<int*>(jg + (size_t)(nb / 2) * JR * JR);
    auto launch_sweep = [&](hipStream_t q) {
      for (int r = 0; r < nb - 1; ++r) {
        if (split_env) {
          bj_gram_kernel<<<dim3(nb / 2, gs), 256, 0, q>>>(a, d, nb, r, gs, mpart, flags);
          bj_sweep_kernel<<<nb / 2, 256, 0, q>>>(mpart, gs, tol, jg, rot, flags);
          bj_apply_kernel<<<dim3(nb / 2, as), 256, 0, q>>>(a, v, d, nb, r, as, jg, rot, flags);
        } else {
          bjacobi_round_kernel<<<nb / 2, 256, 0, q>>>(a, v, d, nb, r, tol, flags);
        }
      }
      jacobi_sweep_end_kernel<<<1, 1, 0, q>>>(flags);
    };
    // A sweep is 3 (nb - 1) + 1 dependent launches of ~10 us kernels: replayed as ONE hipGraph per
    // sweep (captured on a private stream: the caller's may be the legacy default stream, which
    // cannot be captured), and the host looks at the convergence flag every other sweep from the
    // sixth on instead of queueing all max_sweeps sweeps of no-op launches behind the last useful one.
    static const int graph_env = []() {
      const char* e = getenv("CGAMD_JACOBI_GRAPH");
      return e ? atoi(e) : 1;
    }();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(st, &cap);
    bool done = false;
    if (graph_env && cap == hipStreamCaptureStatusNone) {
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
        launch_sweep(js);
        if (hipStreamEndCapture(js, &graph) == hipSuccess && graph &&
            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          int hflags[2] = {0, 0};
          for (int s = 0; s < max_sweeps; ++s) {
            hipGraphLaunch(exec, js);
            if ((s >= 5 && (s & 1)) || s + 1 == max_sweeps) {
              hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, js);
              hipStreamSynchronize(js);
              if (hflags[0]) break;
            }
          }
          done = true;
        }
      }
      (void)hipGetLastError();
      if (exec) hipGraphExecDestroy(exec);
      if (graph) hipGraphDestroy(graph);
      hipEventRecord(e1, js);
      hipStreamWaitEvent(st, e1, 0);
    }
    if (!done)
      for (int s = 0; s < max_sweeps; ++s) launch_sweep(st);
    CG_CHECK_LAUNCH("cg_syevj_f64(block sweeps)");
  } else if (d > 1) {
    for (int s = 0; s < max_sweeps; ++s) {
      for (int r = 0; r < np - 1; ++r) {
        jacobi_round_kernel<<<np / 2, 256, 0, st>>>(a, v, d, np, r, tol, flags);
      }
      jacobi_sweep_end_kernel<<<1, 1, 0, st>>>(flags);
    }
    CG_CHECK_LAUNCH("cg_syevj_f64(sweeps)");
  }
  jacobi_eigvals_kernel<<<d, 256, 0, st>>>(a, v, d, w);
  CG_CHECK_LAUNCH("cg_syevj_f64(eigvals)");
  return CG_OK;
}

extern "C" size_t cg_inception_score_workspace_bytes(int64_t n, int k) {
  if (n <= 0 || k <= 0) return 0;
  return align_up(((size_t)is_blocks(n) * k + (size_t)k + (size_t)is_blocks(n)) * sizeof(double),
                  256);
}

extern "C" int cg_inception_score_f64(const float* logits, int64_t n, int k, double* score,
                                      void* ws, size_t ws_bytes, cgStream stream) {
  if (!logits || !score || n <= 0 || k <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_inception_score_f64: bad argument");
  if (!ws || ws_bytes < cg_inception_score_workspace_bytes(n, k))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_inception_score_f64: workspace too small");
  if ((size_t)4 * k * sizeof(double) > 60000)
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_inception_score_f64: k=%d too large", k);
  hipStream_t st = (hipStream_t)stream;
  const int nb = is_blocks(n);
  const int64_t rpb = (n + nb - 1) / nb;
  double* part = (double*)ws;
  double* logq = part + (size_t)nb * k;
  double* klp = logq + k;
  is_pass_kernel<<<nb, 256, 4 * k * sizeof(double), st>>>(logits, n, k, rpb, 1, nullptr, part);
  CG_CHECK_LAUNCH("cg_inception_score_f64(pass1)");
  is_logq_kernel<<<cdiv(k, 256), 256, 0, st>>>(part, nb, k, 1.0 / (double)n, logq);
  CG_CHECK_LAUNCH("cg_inception_score_f64(logq)");
  is_pass_kernel<<<nb, 256, 4 * k * sizeof(double), st>>>(logits, n, k, rpb, 2, logq, klp);
  CG_CHECK_LAUNCH("cg_inception_score_f64(pass2)");
  is_final_kernel<<<1, 1, 0, st>>>(klp, nb, 1.0 / (double)n, score);
  CG_CHECK_LAUNCH("cg_inception_score_f64(final)");
  return CG_OK;
}

// ---- FID scalar assembly on the device (no host round trip between the two eigen-solves) ----
namespace {
// f[i] = sign(w) * (|w| < eps ? |w| : sqrt|w|): tfgan's _symmetric_matrix_square_root rule applied
// to the spectrum; sum_out (optional) = sum_i f[i], one block, fixed order.
__global__ __launch_bounds__(256) void spectral_sqrt_kernel(const double* __restrict__ w, int n,
                                                            double eps, double* __restrict__ f,
                                                            double* __restrict__ sum_out) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double x = w[i], a = fabs(x);
    const double r = (x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0)) * (a < eps ? a : sqrt(a));
    if (f) f[i] = r;
    s += r;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && sum_out) *sum_out = sm[0] + sm[1] + sm[2] + sm[3];
}
// d[i] = f(s_i) / s_i^2 for s_i = |w_i| > tiny, else 0 (f as above): the weights that rebuild the
// symmetric square root from the rows g_i = lambda_i v_i the one-sided Jacobi solver leaves in its
// matrix argument -- sum_i f(s_i) v_i v_i^T = G^T diag(d) G -- without accumulating V.
__global__ __launch_bounds__(256) void spectral_root_scale_kernel(const double* __restrict__ w, int n,
                                                                  double eps,
                                                                  double* __restrict__ d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double a = fabs(w[i]);
  d[i] = a > 1e-150 ? (a < eps ? a : sqrt(a)) / (a * a) : 0.0;
}
// out = tr(sigma) + tr(sigma_v) - 2 * sqrt_trace + |m - m_v|^2   (fid_score.py:58-75 via tfgan)
__global__ __launch_bounds__(256) void fid_combine_kernel(const double* __restrict__ sigma,
                                                          const double* __restrict__ sigma_v,
                                                          const double* __restrict__ m,
                                                          const double* __restrict__ m_v, int d,
                                                          const double* __restrict__ sqrt_trace,
                                                          double* __restrict__ out) {
  __shared__ double sm[2][4];
  double t = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < d; i += 256) {
    t += sigma[(int64_t)i * d + i] + sigma_v[(int64_t)i * d + i];
    const double dm = m[i] - m_v[i];
    q += dm * dm;
  }
  t = wave_sum_d(t);
  q = wave_sum_d(q);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = t;
    sm[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    *out = (sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3]) - 2.0 * *sqrt_trace +
           (sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3]);
}
}  // namespace

extern "C" int cg_spectral_sqrt_f64(const double* w, int n, double eps, double* f, double* sum_out,
                                    cgStream stream) {
  if (!w || n <= 0 || (!f && !sum_out))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_spectral_sqrt_f64: bad argument");
  spectral_sqrt_kernel<<<1, 256, 0, (hipStream_t)stream>>>(w, n, eps, f, sum_out);
  CG_CHECK_LAUNCH("cg_spectral_sqrt_f64");
  return CG_OK;
}

extern "C" int cg_spectral_root_scale_f64(const double* w, int n, double eps, double* d,
                                          cgStream stream) {
  if (!w || !d || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_spectral_root_scale_f64: bad argument");
  spectral_root_scale_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(w, n, eps, d);
  CG_CHECK_LAUNCH("cg_spectral_root_scale_f64");
  return CG_OK;
}

extern "C" int cg_fid_combine_f64(const double* sigma, const double* sigma_v, const double* mean,
                                  const double* mean_v, int d, const double* sqrt_trace,
                                  double* out, cgStream stream) {
  if (!sigma || !sigma_v || !mean || !mean_v || !sqrt_trace || !out || d <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_fid_combine_f64: bad argument");
  fid_combine_kernel<<<1, 256, 0, (hipStream_t)stream>>>(sigma, sigma_v, mean, mean_v, d,
                                                         sqrt_trace, out);
  CG_CHECK_LAUNCH("cg_fid_combine_f64");
  return CG_OK;
}

// ---- KID (metrics/kid_score.py:44-149): sum and trace of the cubic polynomial kernel
// (gram / dim + 1)^3 of one block, fp64, two-stage deterministic reduction ----
constexpr int KID_BLOCKS = 256;
__global__ __launch_bounds__(256) void poly3_part_kernel(const double* __restrict__ g, int m, int n,
                                                         double inv_dim, double* __restrict__ part) {
  __shared__ double sm[2][4];
  double s = 0.0, t = 0.0;
  const int64_t total = (int64_t)m * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const double k1 = g[i] * inv_dim + 1.0;
    const double k3 = k1 * k1 * k1;
    s += k3;
    const int64_t r = i / n, c = i - r * n;
    if (r == c) t += k3;
  }
  s = wave_sum_d(s);
  t = wave_sum_d(t);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[0][w] = s;
    sm[1][w] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
    part[blockIdx.x * 2 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
  }
}
__global__ void poly3_final_kernel(const double* __restrict__ part, int nb, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0, t = 0.0;
    for (int i = 0; i < nb; ++i) {
      s += part[i * 2];
      t += part[i * 2 + 1];
    }
    out[0] = s;
    out[1] = t;
  }
}

extern "C" size_t cg_poly3_kernel_workspace_bytes(void) { return KID_BLOCKS * 2 * sizeof(double); }

extern "C" int cg_poly3_kernel_sums_f64(const double* gram, int m, int n, double inv_dim,
                                        double* out2, void* ws, size_t ws_bytes, cgStream stream) {
  if (!gram || !out2 || m <= 0 || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_poly3_kernel_sums_f64: bad argument");
  if (!ws || ws_bytes < cg_poly3_kernel_workspace_bytes())
    CG_FAIL(CG_ERR_WORKSPACE, "cg_poly3_kernel_sums_f64: workspace too small");
  const int64_t total = (int64_t)m * n;
  int nb = (int)((total + 255) / 256);
  if (nb > KID_BLOCKS) nb = KID_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
  poly3_part_kernel<<<nb, 256, 0, st>>>(gram, m, n, inv_dim, (double*)ws);
  poly3_final_kernel<<<1, 64, 0, st>>>((const double*)ws, nb, out2);
  CG_CHECK_LAUNCH("cg_poly3_kernel_sums_f64");
  return CG_OK;
}
