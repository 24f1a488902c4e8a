// R5 for TRAINING: the reference's Lanczos layer (model/ada_lanczos_net.py:139-247) on an fp64
// Laplacian, forward with the state its backward needs, and the backward itself — the reverse sweep
// of the recurrence that torch autograd would run through ~1500 small fp64 launches.
//
//   lnz_ada_lanczos_layer_f64           Le [B,N,N] f64, mask, q1 -> T [B,K,K] f64, Q [B,N,K] f64, ws
//   lnz_ada_lanczos_layer_f64_backward  Le, ws, dT, dQ           -> dLe [B,N,N] f64
//
// One wavefront per molecule, lane = node row (N <= 32), the row of Le in registers, the basis in
// LDS, every inner product an fp64 wave reduction (DPP tree, identical in all lanes).
//
// Forward per step ii (q_0 = 0, beta_0 = 0):
//   z0 = Le q_ii ; alpha = <q_ii, z0> ; z1 = z0 - alpha q_ii - beta_{ii-1} q_{ii-1}
//   twice, for j = 1 .. ii-1 in order:  s = <y, q_j> ; y <- y - c_j s q_j ,  c_j = 1 / (<q_j,q_j> + EPS)
//   beta = |y| ; valid_ii = valid_{ii-1} [beta >= 1e-4] ; q_{ii+1} = y valid_ii / (beta + EPS)
// then the quirks of :209-237 (idx = min(sum valid, sum mask); alpha, beta, columns of Q masked by
// valid and by step < idx, rows of Q >= idx zeroed) — masks are constants for the gradient.
// The backward keeps q_j, alpha, beta, <q_j,q_j>, the valid flags and every coefficient s of the
// Gram-Schmidt sweeps (<= 2 x 19 x 20 / 2 scalars): a projection y' = y - c s q is undone exactly by
// y = y' + c s q, so the reverse sweep rebuilds each intermediate vector from the final one.
//   d y   = d y' - c <d y', q> q
//   d q  += -c s d y' - c <d y', q> y + 2 c^2 s <d y', q> q          (c depends on q)
// Le is symmetric to rounding (D^-1/2 A D^-1/2 of a symmetric A), so Le^T d z0 is taken as Le d z0;
// dLe = sum_ii d z0 (x) q_ii is NOT symmetrised (autograd through the Laplacian does the rest).
#include "common.hpp"

namespace {

constexpr double kEps = 1.1920928955078125e-07;  // np.finfo(np.float32).eps (ada_lanczos_net.py:8)
constexpr int NM = 32;                            // rows / steps per molecule
// workspace per molecule, in doubles
constexpr int WS_Q = 0;                           // q_0 .. q_33: [34][32]
constexpr int WS_ALPHA = WS_Q + 34 * NM;          // [33]   (1-based step)
constexpr int WS_BETA = WS_ALPHA + 33;            // [33]
constexpr int WS_QQ = WS_BETA + 33;               // [34]
constexpr int WS_VALID = WS_QQ + 34;              // [33]   cumulative valid flag
constexpr int WS_VM = WS_VALID + 33;              // [33]   valid x [step < idx]
constexpr int WS_IDX = WS_VM + 33;                // [1]
constexpr int WS_S = WS_IDX + 1;                  // [2][33][32] Gram-Schmidt coefficients
constexpr int WS_TOTAL = WS_S + 2 * 33 * NM;

__device__ inline double dpp_add_d(double v, const int sel) {
  union { double d; int i[2]; } x, y;
  x.d = v;
  if (sel == 0) {
    y.i[0] = __builtin_amdgcn_update_dpp(0, x.i[0], 0xB1, 0xF, 0xF, false);
    y.i[1] = __builtin_amdgcn_update_dpp(0, x.i[1], 0xB1, 0xF, 0xF, false);
  } else if (sel == 1) {
    y.i[0] = __builtin_amdgcn_update_dpp(0, x.i[0], 0x4E, 0xF, 0xF, false);
    y.i[1] = __builtin_amdgcn_update_dpp(0, x.i[1], 0x4E, 0xF, 0xF, false);
  } else if (sel == 2) {
    y.i[0] = __builtin_amdgcn_update_dpp(0, x.i[0], 0x141, 0xF, 0xF, false);
    y.i[1] = __builtin_amdgcn_update_dpp(0, x.i[1], 0x141, 0xF, 0xF, false);
  } else {
    y.i[0] = __builtin_amdgcn_update_dpp(0, x.i[0], 0x140, 0xF, 0xF, false);
    y.i[1] = __builtin_amdgcn_update_dpp(0, x.i[1], 0x140, 0xF, 0xF, false);
  }
  return v + y.d;
}
__device__ inline double readlane_d(double v, int l) {
  union { double d; int i[2]; } x, y;
  x.d = v;
  y.i[0] = __builtin_amdgcn_readlane(x.i[0], l);
  y.i[1] = __builtin_amdgcn_readlane(x.i[1], l);
  return y.d;
}
// wave-wide sum, identical in every lane, fixed tree (the same as ada_lanczos.hip)
__device__ inline double wave_sum(double v) {
  v = dpp_add_d(v, 0);
  v = dpp_add_d(v, 1);
  v = dpp_add_d(v, 2);
  v = dpp_add_d(v, 3);
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// y[lane] = sum_c ar[c] x[c]: x broadcast lane by lane
__device__ inline double matvec_row(const double (&ar)[NM], const double x, const int N) {
  double z = 0.0;
#pragma unroll
  for (int c = 0; c < NM; ++c) {
    const double xc = readlane_d(x, c);
    if (c < N) z = fma(ar[c], xc, z);
  }
  return z;
}

__global__ __launch_bounds__(64) void ada_lanczos_f64_forward_kernel(
    const double* __restrict__ A, const uint8_t* __restrict__ mask, const float* __restrict__ q1,
    int N, int K, double* __restrict__ T, double* __restrict__ Q, double* __restrict__ ws) {
  __shared__ double Qs[34 * NM];
  __shared__ double alpha_s[33], beta_s[33], qq_s[34], valid_s[33];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int Tit = N < K ? N : K;
  const bool row = lane < N;
  double* w = ws + (int64_t)b * WS_TOTAL;
  double ar[NM];
#pragma unroll
  for (int c = 0; c < NM; ++c) ar[c] = (row && c < N) ? A[((int64_t)b * N + lane) * N + c] : 0.0;
  const double mk = (row && (mask == nullptr || mask[(int64_t)b * N + lane] != 0)) ? 1.0 : 0.0;
  double q = row ? (double)q1[(int64_t)b * N + lane] * mk : 0.0;   // :161-165
  q = q / sqrt(wave_sum(q * q));                                    // :167
  if (lane < NM) {
    Qs[lane] = 0.0;
    Qs[NM + lane] = q;
  }
  if (lane == 0) qq_s[0] = 0.0;
  double q_prev = 0.0, beta_prev = 0.0, valid_prev = 1.0;
  const double nmask = wave_sum(mk);
  __syncthreads();
  for (int ii = 1; ii <= Tit; ++ii) {
    double z = matvec_row(ar, q, N);                    // :173
    const double alpha = wave_sum(q * z);               // :174
    z = z - alpha * q - beta_prev * q_prev;             // :175
    if (ii > 1) {                                       // :177-189
      for (int pass = 0; pass < 2; ++pass) {
        for (int jj = 1; jj < ii; ++jj) {
          const double qj = lane < NM ? Qs[jj * NM + lane] : 0.0;
          const double s = wave_sum(z * qj);
          if (lane == 0) w[WS_S + (pass * 33 + ii) * NM + jj] = s;
          z = z - s / (qq_s[jj] + kEps) * qj;
        }
      }
    }
    const double beta = sqrt(wave_sum(z * z));          // :191
    const double ok = beta >= 1.0e-4 ? 1.0 : 0.0;       // :195
    const double valid = ii == 1 ? ok : valid_prev * ok;
    const double qn = (z * valid) / (beta + kEps);      // :202
    const double qq_cur = wave_sum(q * q);
    if (lane == 0) {
      alpha_s[ii] = alpha;
      beta_s[ii] = beta;
      valid_s[ii] = valid;
      qq_s[ii] = qq_cur;
    }
    if (lane < NM) Qs[(ii + 1) * NM + lane] = qn;
    q_prev = q;
    q = qn;
    beta_prev = beta;
    valid_prev = valid;
    __syncthreads();
  }
  // idx = min(sum(valid), sum(mask)) (:209-211); masks of :213-237
  double vsum = 0.0;
  for (int ii = 1; ii <= Tit; ++ii) vsum += valid_s[ii];
  int idx = (int)vsum;
  if (mask != nullptr) idx = idx < (int)nmask ? idx : (int)nmask;
  auto vm = [&](int step1) { return (step1 - 1 < idx) ? valid_s[step1] : 0.0; };   // 1-based step
  double* Tb = T + (int64_t)b * K * K;
  for (int e = lane; e < K * K; e += 64) {
    const int i = e / K, jn = e - i * K;
    double v = 0.0;
    if (i < Tit && jn < Tit) {
      if (i == jn) v = alpha_s[i + 1] * vm(i + 1);
      else if (jn == i + 1 && i < Tit - 1) v = beta_s[i + 1] * vm(i + 1);
      else if (i == jn + 1 && jn < Tit - 1) v = beta_s[jn + 1] * vm(jn + 1);
    }
    Tb[e] = v;
  }
  double* Qb = Q + (int64_t)b * N * K;
  for (int e = lane; e < N * K; e += 64) {
    const int r = e / K, k = e - r * K;
    double v = 0.0;
    if (k < Tit) {
      const double rowkeep = (idx < N && r >= idx) ? 0.0 : 1.0;
      v = Qs[(k + 1) * NM + r] * (vm(k + 1) * rowkeep);
    }
    Qb[e] = v;
  }
  // state for the backward
  for (int e = lane; e < 34 * NM; e += 64) w[WS_Q + e] = e < (Tit + 2) * NM ? Qs[e] : 0.0;
  if (lane <= Tit && lane >= 1) {
    w[WS_ALPHA + lane] = alpha_s[lane];
    w[WS_BETA + lane] = beta_s[lane];
    w[WS_VALID + lane] = valid_s[lane];
    w[WS_VM + lane] = vm(lane);
  }
  if (lane <= Tit) w[WS_QQ + lane] = qq_s[lane];
  if (lane == 0) {
    w[WS_IDX] = (double)idx;
    w[WS_BETA] = 0.0;
  }
}

__global__ __launch_bounds__(64) void ada_lanczos_f64_backward_kernel(
    const double* __restrict__ A, int N, int K, const double* __restrict__ ws,
    const double* __restrict__ dT, const double* __restrict__ dQ, double* __restrict__ dA) {
  __shared__ double Qs[34 * NM], dQs[34 * NM];
  __shared__ double S[2 * 33 * NM];
  __shared__ double alpha_s[33], beta_s[33], qq_s[34], valid_s[33], vm_s[33], dbeta_s[33];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int Tit = N < K ? N : K;
  const bool row = lane < N;
  const double* w = ws + (int64_t)b * WS_TOTAL;
  for (int e = lane; e < 34 * NM; e += 64) Qs[e] = w[WS_Q + e];
  for (int e = lane; e < 2 * 33 * NM; e += 64) S[e] = w[WS_S + e];
  if (lane < 33) {
    alpha_s[lane] = w[WS_ALPHA + lane];
    beta_s[lane] = w[WS_BETA + lane];
    valid_s[lane] = w[WS_VALID + lane];
    vm_s[lane] = w[WS_VM + lane];
  }
  if (lane < 34) qq_s[lane] = w[WS_QQ + lane];
  const int idx = (int)w[WS_IDX];
  __syncthreads();
  // gradients arriving at the outputs: T (:218-226) and Q (:229-237), masks as constants
  const double* dTb = dT + (int64_t)b * K * K;
  const double* dQb = dQ + (int64_t)b * N * K;
  if (lane < 33) {
    double v = 0.0;
    if (lane >= 1 && lane <= Tit - 1)
      v = (dTb[(lane - 1) * K + lane] + dTb[lane * K + (lane - 1)]) * vm_s[lane];
    dbeta_s[lane] = v;
  }
  for (int e = lane; e < 34 * NM; e += 64) {
    const int j = e / NM, r = e - j * NM;   // dq_j[r]
    double v = 0.0;
    if (j >= 1 && j <= Tit && r < N) {
      const double rowkeep = (idx < N && r >= idx) ? 0.0 : 1.0;
      v = dQb[(int64_t)r * K + (j - 1)] * vm_s[j] * rowkeep;
    }
    dQs[e] = v;
  }
  double ar[NM], da[NM];
#pragma unroll
  for (int c = 0; c < NM; ++c) {
    ar[c] = (row && c < N) ? A[((int64_t)b * N + lane) * N + c] : 0.0;
    da[c] = 0.0;
  }
  __syncthreads();
  for (int ii = Tit; ii >= 1; --ii) {
    const double v = valid_s[ii];
    const double q = lane < NM ? Qs[ii * NM + lane] : 0.0;
    double dq_ii = 0.0;   // accumulates the gradient on q_ii from this step (added to dQs at the end)
    double dz1 = 0.0, dalpha = dTb[(ii - 1) * K + (ii - 1)] * vm_s[ii];
    if (v != 0.0) {
      const double beta = beta_s[ii];
      const double dqn = lane < NM ? dQs[(ii + 1) * NM + lane] : 0.0;
      double y = (lane < NM ? Qs[(ii + 1) * NM + lane] : 0.0) * (beta + kEps);   // the final z of the step
      const double dbeta = dbeta_s[ii] - wave_sum(dqn * y) / ((beta + kEps) * (beta + kEps));
      double dy = dqn / (beta + kEps) + dbeta * y / beta;
      if (ii > 1) {
        for (int pass = 1; pass >= 0; --pass) {
          for (int jj = ii - 1; jj >= 1; --jj) {
            const double qj = lane < NM ? Qs[jj * NM + lane] : 0.0;
            const double c = 1.0 / (qq_s[jj] + kEps);
            const double s = S[(pass * 33 + ii) * NM + jj];
            const double t = wave_sum(dy * qj);
            const double yp = y + c * s * qj;   // the vector before this projection
            if (lane < NM) dQs[jj * NM + lane] += -c * s * dy - c * t * yp + 2.0 * c * c * s * t * qj;
            dy = dy - c * t * qj;
            y = yp;
          }
        }
      }
      dz1 = dy;
    }
    // z1 = z0 - alpha q_ii - beta_{ii-1} q_{ii-1} ; alpha = <q_ii, z0> ; z0 = Le q_ii
    dalpha -= wave_sum(dz1 * q);
    dq_ii += -alpha_s[ii] * dz1;
    if (ii > 1) {
      const double qm = lane < NM ? Qs[(ii - 1) * NM + lane] : 0.0;
      const double db = -wave_sum(dz1 * qm);
      if (lane == 0) dbeta_s[ii - 1] += db;
      if (lane < NM) dQs[(ii - 1) * NM + lane] += -beta_s[ii - 1] * dz1;
    }
    const double dz0 = dz1 + dalpha * q;
    const double z0 = matvec_row(ar, q, N);
    dq_ii += dalpha * z0 + matvec_row(ar, dz0, N);
#pragma unroll
    for (int c = 0; c < NM; ++c) {
      const double qc = readlane_d(q, c);
      da[c] = fma(dz0, qc, da[c]);
    }
    if (lane < NM) dQs[ii * NM + lane] += dq_ii;
    __syncthreads();
  }
  if (row) {
    double* out = dA + ((int64_t)b * N + lane) * N;
#pragma unroll
    for (int c = 0; c < NM; ++c)
      if (c < N) out[c] = da[c];
  }
}

// ---------------------------------------------------------------------------------------------
// R4 for training: the learned Laplacian (model/ada_lanczos_net.py:101-137) in fp64 and its
// backward.  One workgroup per molecule.
//   dist2_ij = |x_j - x_i|^2 ; sigma2 = mean over ALL N^2 pairs ; A = exp(-dist2 / sigma2) adj ;
//   rs_i = sum_j A_ij ; g_i = (rs_i + [rs_i == 0])^-1/2 ; Le_ij = g_i A_ij g_j
// backward, G = dLoss/dLe:
//   dg_i = sum_j (G_ij A_ij + G_ji A_ji) g_j ; drs_i = -1/2 dg_i g_i^3 ; dA_ij = G_ij g_i g_j + drs_i
//   ddist2_ij = -dA_ij A_ij / sigma2 + dsigma2 / N^2 ,  dsigma2 = sum_ij dA_ij A_ij dist2_ij / sigma2^2
//   W = ddist2 + ddist2^T ;  dX = 2 (diag(W 1) - W) X
// saved by the forward: A, dist2 [N,N], g [N], sigma2 (fp64).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ada_laplacian_f64_forward_kernel(
    const float* __restrict__ X, int D, const float* __restrict__ L0, int64_t sb, int64_t sr,
    int64_t sc, int N, double* __restrict__ Le, double* __restrict__ sv) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* xs = dsm;               // [N][D]
  double* d2 = xs + N * D;        // [N][N] -> A
  double* g = d2 + N * N;         // [N]
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < N * D; i += 256) xs[i] = (double)X[(int64_t)b * N * D + i];
  __syncthreads();
  double* sA = sv + (int64_t)b * (2 * N * N + N + 1);
  double* sD2 = sA + N * N;
  double* sG = sD2 + N * N;
  double part = 0.0;
  for (int p = tid; p < N * N; p += 256) {
    const int i = p / N, j = p - i * N;
    double s = 0.0;
    for (int f = 0; f < D; ++f) {
      const double df = xs[j * D + f] - xs[i * D + f];
      s = fma(df, df, s);
    }
    d2[p] = s;
    sD2[p] = s;
    part += s;
  }
  red[tid] = part;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const double sigma2 = red[0] / (double)(N * N);
  const float* Lb = L0 + (int64_t)b * sb;
  for (int p = tid; p < N * N; p += 256) {
    const int i = p / N, j = p - i * N;
    const double adj = Lb[i * sr + j * sc] != 0.0f ? 1.0 : 0.0;
    const double a = exp(-d2[p] / sigma2) * adj;
    d2[p] = a;
    sA[p] = a;
  }
  __syncthreads();
  for (int i = tid; i < N; i += 256) {
    double rs = 0.0;
    for (int j = 0; j < N; ++j) rs += d2[i * N + j];
    const double gi = 1.0 / sqrt(rs + (rs == 0.0 ? 1.0 : 0.0));
    g[i] = gi;
    sG[i] = gi;
  }
  if (tid == 0) sG[N] = sigma2;
  __syncthreads();
  double* out = Le + (int64_t)b * N * N;
  for (int p = tid; p < N * N; p += 256) {
    const int i = p / N, j = p - i * N;
    out[p] = (g[i] * d2[p]) * g[j];
  }
}

__global__ __launch_bounds__(256) void ada_laplacian_f64_backward_kernel(
    const float* __restrict__ X, int D, int N, const double* __restrict__ sv,
    const double* __restrict__ G, double* __restrict__ dX) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* xs = dsm;               // [N][D]
  double* A = xs + N * D;         // [N][N]
  double* dd = A + N * N;         // [N][N]: dA, then ddist2, then W
  double* g = dd + N * N;         // [N]
  double* aux = g + N;            // [N]: drs, then row sums of W
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* sA = sv + (int64_t)b * (2 * N * N + N + 1);
  const double* sD2 = sA + N * N;
  const double* sG = sD2 + N * N;
  const double* Gb = G + (int64_t)b * N * N;
  for (int i = tid; i < N * D; i += 256) xs[i] = (double)X[(int64_t)b * N * D + i];
  for (int p = tid; p < N * N; p += 256) A[p] = sA[p];
  for (int i = tid; i < N; i += 256) g[i] = sG[i];
  const double sigma2 = sG[N];
  __syncthreads();
  for (int i = tid; i < N; i += 256) {
    double dg = 0.0;
    for (int j = 0; j < N; ++j) dg += (Gb[i * N + j] * A[i * N + j] + Gb[j * N + i] * A[j * N + i]) * g[j];
    aux[i] = -0.5 * dg * g[i] * g[i] * g[i];
  }
  __syncthreads();
  double part = 0.0;
  for (int p = tid; p < N * N; p += 256) {
    const int i = p / N, j = p - i * N;
    const double dA = Gb[p] * g[i] * g[j] + aux[i];
    dd[p] = dA;
    part += dA * A[p] * sD2[p];
  }
  red[tid] = part;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const double dsig = red[0] / (sigma2 * sigma2) / (double)(N * N);
  for (int p = tid; p < N * N; p += 256) dd[p] = -dd[p] * A[p] / sigma2 + dsig;
  __syncthreads();
  // W = dd + dd^T (in A: no longer needed), its row sums
  for (int p = tid; p < N * N; p += 256) {
    const int i = p / N, j = p - i * N;
    A[p] = dd[p] + dd[j * N + i];
  }
  __syncthreads();
  for (int i = tid; i < N; i += 256) {
    double r = 0.0;
    for (int j = 0; j < N; ++j) r += A[i * N + j];
    aux[i] = r;
  }
  __syncthreads();
  double* out = dX + (int64_t)b * N * D;
  for (int e = tid; e < N * D; e += 256) {
    const int k = e / D, f = e - k * D;
    double acc = xs[e] * aux[k];
    for (int i = 0; i < N; ++i) acc -= A[k * N + i] * xs[i * D + f];
    out[e] = 2.0 * acc;
  }
}

// ---------------------------------------------------------------------------------------------
// T powers (model/ada_lanczos_net.py:262-270) of an fp64 T, keeping every power for the backward:
//   P_1 = T, P_i = P_{i-1} T ; Tcat[b][r][s K + c] = (float) P_{p_s}[r][c]
// backward: R_i = dLoss/dP_i; for i = pmax .. 2: R_{i-1} += R_i T^T, dT += P_{i-1}^T R_i; dT += R_1.
// ---------------------------------------------------------------------------------------------
struct PowArr64 {
  int32_t v[16];
};

__global__ __launch_bounds__(256) void ada_t_powers_f64_forward_kernel(
    const double* __restrict__ T, int K, PowArr64 dist, int S, int pmax, float* __restrict__ Tcat,
    double* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* Ts = dsm;
  double* TT = Ts + K * K;
  double* TN = TT + K * K;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < K * K; i += 256) {
    const double v = T[(int64_t)b * K * K + i];
    Ts[i] = v;
    TT[i] = v;
  }
  __syncthreads();
  float* out = Tcat + (int64_t)b * K * S * K;
  double* Pb = P + (int64_t)b * pmax * K * K;
  for (int ii = 1; ii <= pmax; ++ii) {
    for (int i = tid; i < K * K; i += 256) Pb[(int64_t)(ii - 1) * K * K + i] = TT[i];
    for (int s = 0; s < S; ++s) {
      if (dist.v[s] == ii) {
        for (int i = tid; i < K * K; i += 256) {
          const int r = i / K, c = i - r * K;
          out[(int64_t)r * S * K + s * K + c] = (float)TT[i];
        }
      }
    }
    if (ii == pmax) break;
    for (int i = tid; i < K * K; i += 256) {
      const int r = i / K, c = i - r * K;
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc = fma(TT[r * K + k], Ts[k * K + c], acc);
      TN[i] = acc;
    }
    __syncthreads();
    double* t = TT;
    TT = TN;
    TN = t;
  }
}

__global__ __launch_bounds__(256) void ada_t_powers_f64_backward_kernel(
    const double* __restrict__ T, int K, PowArr64 dist, int S, int pmax,
    const float* __restrict__ dTcat, const double* __restrict__ P, double* __restrict__ dT) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* Ts = dsm;             // T
  double* R = Ts + K * K;       // R_i
  double* RN = R + K * K;       // R_{i-1}
  double* Pm = RN + K * K;      // P_{i-1}
  double* acc = Pm + K * K;     // dT
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* gin = dTcat + (int64_t)b * K * S * K;
  const double* Pb = P + (int64_t)b * pmax * K * K;
  for (int i = tid; i < K * K; i += 256) {
    Ts[i] = T[(int64_t)b * K * K + i];
    R[i] = 0.0;
    acc[i] = 0.0;
  }
  __syncthreads();
  for (int ii = pmax; ii >= 1; --ii) {
    for (int s = 0; s < S; ++s) {
      if (dist.v[s] == ii) {
        for (int i = tid; i < K * K; i += 256) {
          const int r = i / K, c = i - r * K;
          R[i] += (double)gin[(int64_t)r * S * K + s * K + c];
        }
      }
    }
    if (ii == 1) break;
    for (int i = tid; i < K * K; i += 256) Pm[i] = Pb[(int64_t)(ii - 2) * K * K + i];
    __syncthreads();
    for (int i = tid; i < K * K; i += 256) {
      const int r = i / K, c = i - r * K;
      double a = 0.0, d = 0.0;
      for (int k = 0; k < K; ++k) {
        a = fma(R[r * K + k], Ts[c * K + k], a);     // (R_i T^T)[r][c]
        d = fma(Pm[k * K + r], R[k * K + c], d);     // (P_{i-1}^T R_i)[r][c]
      }
      RN[i] = a;
      acc[i] += d;
    }
    __syncthreads();
    double* t = R;
    R = RN;
    RN = t;
  }
  __syncthreads();
  for (int i = tid; i < K * K; i += 256) dT[(int64_t)b * K * K + i] = acc[i] + R[i];
}

}  // namespace

static int pow_args(const int32_t* dist_host, int S, PowArr64* d) {
  int pmax = 0;
  for (int i = 0; i < 16; ++i) {
    d->v[i] = i < S ? dist_host[i] : -1;
    if (i < S && dist_host[i] > pmax) pmax = dist_host[i];
  }
  return pmax;
}

extern "C" int64_t lnz_ada_laplacian_f64_state_doubles(int B, int N) {
  return (int64_t)(B > 0 ? B : 0) * (2 * (int64_t)N * N + N + 1);
}

extern "C" int lnz_ada_graph_laplacian_f64(const float* X, int D, const float* L0, int64_t stride_b,
                                           int64_t stride_r, int64_t stride_c, int B, int N,
                                           double* Le, double* state, lnz_stream_t stream) {
  LNZ_REQUIRE(X && L0 && Le && state && B > 0 && N > 0 && D > 0, LNZ_EINVAL,
              "lnz_ada_graph_laplacian_f64: bad arguments (B=%d N=%d D=%d)", B, N, D);
  const size_t lds = ((size_t)N * D + (size_t)N * N + N) * sizeof(double);
  LNZ_REQUIRE(lds <= 96 * 1024, LNZ_ENOTSUP, "lnz_ada_graph_laplacian_f64: N=%d, D=%d too large", N, D);
  LNZ_DYNAMIC_LDS(ada_laplacian_f64_forward_kernel, lds, "ada_lanczos_grad.hip");
  hipLaunchKernelGGL(ada_laplacian_f64_forward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, X, D,
                     L0, stride_b, stride_r, stride_c, N, Le, state);
  return lnz::check_launch("lnz_ada_graph_laplacian_f64");
}

extern "C" int lnz_ada_graph_laplacian_f64_backward(const float* X, int D, int B, int N,
                                                    const double* state, const double* dLe, double* dX,
                                                    lnz_stream_t stream) {
  LNZ_REQUIRE(X && state && dLe && dX && B > 0 && N > 0 && D > 0, LNZ_EINVAL,
              "lnz_ada_graph_laplacian_f64_backward: bad arguments (B=%d N=%d D=%d)", B, N, D);
  const size_t lds = ((size_t)N * D + 2 * (size_t)N * N + 2 * N) * sizeof(double);
  LNZ_REQUIRE(lds <= 96 * 1024, LNZ_ENOTSUP, "lnz_ada_graph_laplacian_f64_backward: N=%d, D=%d too large",
              N, D);
  LNZ_DYNAMIC_LDS(ada_laplacian_f64_backward_kernel, lds, "ada_lanczos_grad.hip");
  hipLaunchKernelGGL(ada_laplacian_f64_backward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, X, D,
                     N, state, dLe, dX);
  return lnz::check_launch("lnz_ada_graph_laplacian_f64_backward");
}

extern "C" int lnz_ada_t_powers_f64(const double* T, int B, int K, const int32_t* dist_host, int S,
                                    float* Tcat, double* P, lnz_stream_t stream) {
  LNZ_REQUIRE(T && dist_host && Tcat && P && B > 0 && K > 0 && S > 0, LNZ_EINVAL,
              "lnz_ada_t_powers_f64: bad arguments");
  LNZ_REQUIRE(S <= 16 && K <= 64, LNZ_ENOTSUP, "lnz_ada_t_powers_f64: S=%d K=%d out of range", S, K);
  PowArr64 d;
  const int pmax = pow_args(dist_host, S, &d);
  LNZ_REQUIRE(pmax >= 1 && pmax <= 4096, LNZ_EINVAL, "lnz_ada_t_powers_f64: bad exponents");
  const size_t lds = (size_t)3 * K * K * sizeof(double);
  LNZ_DYNAMIC_LDS(ada_t_powers_f64_forward_kernel, lds, "ada_lanczos_grad.hip");
  hipLaunchKernelGGL(ada_t_powers_f64_forward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, T, K, d,
                     S, pmax, Tcat, P);
  return lnz::check_launch("lnz_ada_t_powers_f64");
}

extern "C" int lnz_ada_t_powers_f64_backward(const double* T, int B, int K, const int32_t* dist_host,
                                             int S, const float* dTcat, const double* P, double* dT,
                                             lnz_stream_t stream) {
  LNZ_REQUIRE(T && dist_host && dTcat && P && dT && B > 0 && K > 0 && S > 0, LNZ_EINVAL,
              "lnz_ada_t_powers_f64_backward: bad arguments");
  LNZ_REQUIRE(S <= 16 && K <= 64, LNZ_ENOTSUP, "lnz_ada_t_powers_f64_backward: S=%d K=%d out of range", S, K);
  PowArr64 d;
  const int pmax = pow_args(dist_host, S, &d);
  LNZ_REQUIRE(pmax >= 1 && pmax <= 4096, LNZ_EINVAL, "lnz_ada_t_powers_f64_backward: bad exponents");
  const size_t lds = (size_t)5 * K * K * sizeof(double);
  LNZ_DYNAMIC_LDS(ada_t_powers_f64_backward_kernel, lds, "ada_lanczos_grad.hip");
  hipLaunchKernelGGL(ada_t_powers_f64_backward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, T, K, d,
                     S, pmax, dTcat, P, dT);
  return lnz::check_launch("lnz_ada_t_powers_f64_backward");
}

extern "C" int64_t lnz_ada_lanczos_f64_workspace_doubles(int B) {
  return (int64_t)(B > 0 ? B : 0) * WS_TOTAL;
}

extern "C" int lnz_ada_lanczos_layer_f64(const double* A, const uint8_t* mask, const float* q1, int B,
                                         int N, int K, double* T, double* Q, double* ws,
                                         lnz_stream_t stream) {
  LNZ_REQUIRE(A && q1 && T && Q && ws && B > 0 && N > 0 && K > 0, LNZ_EINVAL,
              "lnz_ada_lanczos_layer_f64: bad arguments (B=%d N=%d K=%d)", B, N, K);
  LNZ_REQUIRE(N <= NM && K <= NM, LNZ_ENOTSUP, "lnz_ada_lanczos_layer_f64: N=%d, K=%d exceed %d", N, K,
              NM);
  hipLaunchKernelGGL(ada_lanczos_f64_forward_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, A, mask,
                     q1, N, K, T, Q, ws);
  return lnz::check_launch("lnz_ada_lanczos_layer_f64");
}

extern "C" int lnz_ada_lanczos_layer_f64_backward(const double* A, int B, int N, int K,
                                                  const double* ws, const double* dT, const double* dQ,
                                                  double* dA, lnz_stream_t stream) {
  LNZ_REQUIRE(A && ws && dT && dQ && dA && B > 0 && N > 0 && K > 0, LNZ_EINVAL,
              "lnz_ada_lanczos_layer_f64_backward: bad arguments (B=%d N=%d K=%d)", B, N, K);
  LNZ_REQUIRE(N <= NM && K <= NM, LNZ_ENOTSUP,
              "lnz_ada_lanczos_layer_f64_backward: N=%d, K=%d exceed %d", N, K, NM);
  hipLaunchKernelGGL(ada_lanczos_f64_backward_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, A, N, K,
                     ws, dT, dQ, dA);
  return lnz::check_launch("lnz_ada_lanczos_layer_f64_backward");
}
