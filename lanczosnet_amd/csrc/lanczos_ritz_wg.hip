// R2 + R6 for graphs of 33..192 nodes: full-length Lanczos -> tridiagonal eigensolve -> Ritz
// select, ONE WORKGROUP (4 wavefronts) per graph.  This is the regime of the reference's own
// synthetic-graph configuration (config/graph_lanczos_net.yaml with dataset/get_graph_data.py:15-49:
// n in [20, 100]; (D, V) = np.linalg.eigh + |lambda| sort, utils/data_helper.py:197-223) and of
// SURVEY.md §8(d)'s "A fits on chip" band (N <= ~180).
//
// Same algorithm as lanczos_ritz.hip (fp64; m = n steps; classical Gram-Schmidt twice against ALL
// previous vectors; restart from the unit vector of largest residual on breakdown; implicit-shift
// QL with the rotations applied to the basis so it ends up holding V = Q B; |lambda| ordering and
// sign convention) — what changes is where things live and who works on them:
//   * A (fp32, n x n, row pitch N|1 floats) is staged ONCE into LDS: HBM traffic per graph stays
//     the algorithmic 4 n^2 (A) + 4 K (D) + 4 N K (V) bytes;
//   * the Krylov basis Qt[i][r] = q_i[r] (fp64, row pitch N|1 doubles) lives in LDS next to A while
//     12 N (N|1) + 14 KB <= 160 KB, i.e. N <= 111 (every graph of the reference generator); for
//     111 < N <= 192 A alone takes up to 148 KB and the basis moves to a caller-provided workspace
//     (N (N|1) 8 bytes per graph, L2 / MALL resident: it is written and re-read by the one CU that
//     owns the graph) — template parameter QG;
//   * every reduction of a step (A w, the j+1 Gram-Schmidt dot products, the update w -= Q c) is
//     split over all 256 threads as (output row) x (segment of the reduction range); partials are
//     combined through LDS in a fixed order, so alpha / beta are bit-identical in every thread and
//     the breakdown / restart control flow stays workgroup-uniform.  No atomics, no shuffles;
//   * QL runs barrier-free: thread r owns element r of every basis vector (the rotation of rows
//     i, i+1 touches only its own two words), each wavefront carries a PRIVATE copy of T's diagonal
//     and off-diagonal (in the then dead A region) and runs the scalar recurrences redundantly.
#include "common.hpp"

namespace {

constexpr double kBreakdownTol = 1e-8;  // see lanczos_ritz.hip
constexpr double kEps = 2.220446049250313e-16;
constexpr int kNT = 256;     // threads per workgroup
constexpr int kWaves = kNT / 64;
constexpr int kNMax = 192;   // largest graph one workgroup owns
constexpr int kLdsMax = 160 * 1024;

struct WgFixed {  // fixed part of the LDS block
  double zb[kNMax];    // broadcast of the current vector
  double cb[kNMax];    // Gram-Schmidt coefficients
  double part[kNT];    // (segment, output) partial sums
  double dd[kNMax];    // T diagonal / eigenvalues
  double ee[kNMax];    // T off-diagonal
  double pn[8];        // partial squared norms
  int perm[kNMax];
  float sgn[kNMax];
};

__host__ __device__ inline size_t wg_a_bytes(int N) {
  size_t a = (size_t)N * (size_t)(N | 1) * sizeof(float);
  const size_t ql = (size_t)kWaves * 2 * (size_t)N * sizeof(double);  // QL's per-wave (d, e) copies
  a = a < ql ? ql : a;
  return (a + 15) & ~(size_t)15;
}

inline size_t wg_lds_bytes(int N, bool qg) {
  return sizeof(WgFixed) + (qg ? 0 : (size_t)N * (size_t)(N | 1) * sizeof(double)) + wg_a_bytes(N);
}

template <bool QG>
__global__ __launch_bounds__(kNT) void lanczos_ritz_wg_kernel(
    const float* __restrict__ A, int64_t sb, int64_t sr, int64_t sc,
    const int32_t* __restrict__ n_nodes, int N, int K, float* __restrict__ D,
    float* __restrict__ V, int32_t* __restrict__ info, double* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  WgFixed& sm = *reinterpret_cast<WgFixed*>(smem_raw);
  const int LD = N | 1;  // doubles per basis row: odd -> "lane i reads row i" is conflict free
  const int LA = N | 1;  // floats per row of A
  double* Qt;
  float* As;
  if constexpr (QG) {
    Qt = ws + (int64_t)blockIdx.x * N * LD;
    As = reinterpret_cast<float*>(smem_raw + sizeof(WgFixed));
  } else {
    Qt = reinterpret_cast<double*>(smem_raw + sizeof(WgFixed));
    As = reinterpret_cast<float*>(Qt + (size_t)N * LD);
  }
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = n_nodes[b];
  n = n < 0 ? 0 : (n > N ? N : n);
  const int kk = K < n ? K : n;  // number of non-padded eigen slots

  // ---- stage the n x n block of A: wave per row, lanes along the row (coalesced when sc == 1)
  {
    const float* Ab = A + (int64_t)b * sb;
    for (int r = wave; r < n; r += kWaves)
      for (int c = lane; c < n; c += 64) As[r * LA + c] = Ab[r * sr + c * sc];
  }
  if (tid < kNMax) {
    sm.dd[tid] = 0.0;
    sm.ee[tid] = 0.0;
  }
  __syncthreads();

  int nrestart = 0;
#ifdef LNZ_PROFILE_PHASES
  long long tp0 = clock64(), tp1 = tp0, tp2 = tp0;
#endif
  if (n > 0) {
    // (output row, segment) split of the length-n reductions with n outputs (A w; w -= Q c)
    const int row_n = tid % n, seg_n = tid / n;
    int nss = kNT / n;
    {
      int cap = n >> 3;
      cap = cap < 1 ? 1 : (cap > 8 ? 8 : cap);
      nss = nss < cap ? nss : cap;
    }
    const int sc0 = seg_n * n / nss, sc1 = (seg_n + 1) * n / nss;

    // x <- (I - Q Q^T)^2 x over basis vectors 0..cnt-1 (thread tid < n owns x[tid]); returns the
    // accumulated coefficient on vector jidx
    auto cgs2 = [&](double& x, const int cnt, const int jidx) -> double {
      double coef = 0.0;
      // dot products: cnt outputs, reduction over the n node rows
      int nsd = kNT / cnt;
      {
        int cap = n >> 3;
        cap = cap < 1 ? 1 : cap;
        nsd = nsd < cap ? nsd : cap;
      }
      const int di = tid % cnt, ds = tid / cnt;
      const int dc0 = ds * n / nsd, dc1 = (ds + 1) * n / nsd;
      // update: n outputs, reduction over the cnt basis vectors
      int nsu = kNT / n;
      {
        int cap = cnt >> 3;
        cap = cap < 1 ? 1 : cap;
        nsu = nsu < cap ? nsu : cap;
      }
      const int ui0 = seg_n * cnt / nsu, ui1 = (seg_n + 1) * cnt / nsu;
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        if (tid < n) sm.zb[tid] = x;
        __syncthreads();
        if (ds < nsd) {
          const double* qi = Qt + (size_t)di * LD;
          double p0 = 0.0, p1 = 0.0;
          int c = dc0;
          for (; c + 1 < dc1; c += 2) {
            p0 = fma(qi[c], sm.zb[c], p0);
            p1 = fma(qi[c + 1], sm.zb[c + 1], p1);
          }
          if (c < dc1) p0 = fma(qi[c], sm.zb[c], p0);
          sm.part[ds * cnt + di] = p0 + p1;
        }
        __syncthreads();
        if (tid < cnt) {
          double c = sm.part[tid];
          for (int s = 1; s < nsd; ++s) c += sm.part[s * cnt + tid];
          sm.cb[tid] = c;
        }
        __syncthreads();
        coef += sm.cb[jidx];
        if (seg_n < nsu) {
          double p0 = 0.0, p1 = 0.0;
          int i = ui0;
          for (; i + 1 < ui1; i += 2) {
            p0 = fma(Qt[(size_t)i * LD + row_n], sm.cb[i], p0);
            p1 = fma(Qt[(size_t)(i + 1) * LD + row_n], sm.cb[i + 1], p1);
          }
          if (i < ui1) p0 = fma(Qt[(size_t)i * LD + row_n], sm.cb[i], p0);
          if (nsu == 1) x -= p0 + p1;
          else sm.part[seg_n * n + row_n] = p0 + p1;
        }
        if (nsu > 1) {
          __syncthreads();
          if (tid < n) {
            double p = sm.part[tid];
            for (int s = 1; s < nsu; ++s) p += sm.part[s * n + tid];
            x -= p;
          }
        }
      }
      return coef;
    };

    // deterministic, strictly positive, non-symmetric start vector (as lanczos_ritz.hip)
    double w = 0.0;
    if (tid < n) {
      unsigned hsh = (unsigned)(tid + 1) * 2654435761u;
      w = 1.0 + (double)((hsh >> 8) & 0xffff) * (1.0 / 65536.0);
    }
    bool fresh = true;  // w is a start / restart vector: its norm is not a coupling beta
    for (int j = 0; j < n; ++j) {
      double beta, u;
      for (;;) {
        // ---- one broadcast of w serves beta = |w| and u = A w
        if (tid < n) sm.zb[tid] = w;
        __syncthreads();
        if (seg_n < nss) {
          const float* ar = As + row_n * LA;
          double u0 = 0.0, u1 = 0.0, s0 = 0.0, s1 = 0.0;
          int c = sc0;
          for (; c + 1 < sc1; c += 2) {
            const double z0 = sm.zb[c], z1 = sm.zb[c + 1];
            u0 = fma((double)ar[c], z0, u0);
            u1 = fma((double)ar[c + 1], z1, u1);
            s0 = fma(z0, z0, s0);
            s1 = fma(z1, z1, s1);
          }
          if (c < sc1) {
            const double z0 = sm.zb[c];
            u0 = fma((double)ar[c], z0, u0);
            s0 = fma(z0, z0, s0);
          }
          sm.part[seg_n * n + row_n] = u0 + u1;
          if (row_n == 0) sm.pn[seg_n] = s0 + s1;
        }
        __syncthreads();
        u = 0.0;
        if (tid < n) {
          u = sm.part[tid];
          for (int s = 1; s < nss; ++s) u += sm.part[s * n + tid];
        }
        double nn = sm.pn[0];
        for (int s = 1; s < nss; ++s) nn += sm.pn[s];
        beta = sqrt(nn);
        if (fresh || beta > kBreakdownTol) break;
        // breakdown: span(q_0..q_{j-1}) is A-invariant.  Restart from the unit vector with the
        // largest residual against the basis (residual^2 >= (n-j)/n > 0); T[j-1][j] stays 0.
        ++nrestart;
        __syncthreads();  // zb is rewritten below
        if (tid < n) {
          double s = 0.0;
          for (int i = 0; i < j; ++i) {
            const double qv = Qt[(size_t)i * LD + tid];
            s = fma(qv, qv, s);
          }
          sm.zb[tid] = 1.0 - s;
        }
        __syncthreads();
        int cand = 0;
        double best = sm.zb[0];
        for (int r = 1; r < n; ++r) {
          const double v = sm.zb[r];
          if (v > best) {
            best = v;
            cand = r;
          }
        }
        __syncthreads();
        w = (tid == cand) ? 1.0 : 0.0;
        (void)cgs2(w, j, 0);
        fresh = true;
      }
      if (!fresh && tid == 0) sm.ee[j - 1] = beta;
      fresh = false;
      const double binv = 1.0 / beta;
      double x = u * binv;  // A q_j
      if (tid < n) Qt[(size_t)j * LD + tid] = w * binv;
      // (the first barrier inside cgs2 orders this store before the basis reads)
      const double alpha = cgs2(x, j + 1, j);
      if (tid == 0) sm.dd[j] = alpha;
      w = x;
    }
    __syncthreads();
#ifdef LNZ_PROFILE_PHASES
    tp1 = clock64();
#endif

    // ---- implicit-shift QL (EISPACK tql2 recurrences) on per-wave copies of (d, e); the
    //      rotations are applied to rows i, i+1 of Qt at this thread's own element
    double* wd = reinterpret_cast<double*>(As) + (size_t)wave * 2 * N;
    double* we = wd + N;
    for (int i = lane; i < n; i += 64) {
      wd[i] = sm.dd[i];
      we[i] = sm.ee[i];
    }
    __syncthreads();
    if (wave * 64 < n) {
      const bool own = tid < n;
      double f = 0.0, tst1 = 0.0;
      for (int l = 0; l < n; ++l) {
        tst1 = fmax(tst1, fabs(wd[l]) + fabs(we[l]));
        // first m >= l with a negligible coupling e_m (m = n-1 at the latest): 64 candidates per
        // pass, one per lane, instead of a dependent LDS read per index
        int m = n - 1;
        for (int base = l; base < n - 1; base += 64) {
          const int idx = base + lane;
          const bool small = idx < n - 1 && !(fabs(we[idx]) > kEps * tst1);
          const unsigned long long mk = __ballot(small);
          if (mk) {
            m = base + __builtin_ctzll(mk);
            break;
          }
        }
        if (m > l) {
          int iter = 0;
          double el;
          do {
            ++iter;
            double g = wd[l];
            el = we[l];
            double p = (wd[l + 1] - g) / (2.0 * el);
            double rr = sqrt(p * p + 1.0);
            if (p < 0) rr = -rr;
            const double dl = el / (p + rr), dl1 = el * (p + rr), hh = g - dl;
            const double el1 = we[l + 1];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // lanes share the shift of the remaining diagonal (LDS ops of one wave run in order)
            for (int i = l + 2 + lane; i < n; i += 64) wd[i] -= hh;
            if (lane == 0) {
              wd[l] = dl;
              wd[l + 1] = dl1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f += hh;
            p = wd[m];
            double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
            double carry = own ? Qt[(size_t)m * LD + tid] : 0.0;
            double z0 = own ? Qt[(size_t)(m - 1) * LD + tid] : 0.0;
            double ei = we[m - 1], di = wd[m - 1];
            for (int i = m - 1; i >= l; --i) {
              // prefetch the next rotation's inputs: none of them is written by this rotation
              const int ip = i > l ? i - 1 : l;
              const double znext = own ? Qt[(size_t)ip * LD + tid] : 0.0;
              const double ei_n = we[ip], di_n = wd[ip];
              c3 = c2;
              c2 = c;
              s2 = s;
              g = c * ei;
              const double hp = c * p;
              const double tt = fma(p, p, ei * ei);
              const double num = fma(p, di, -(ei * g));  // (p d_i - e_i g): off the rsqrt chain
              // 1/sqrt(tt): hardware seed + two Newton steps (tt is a normal double here:
              // |e_i| > eps * tst1 for l <= i < m).  The IEEE sqrt + divide expand to ~40 dependent
              // fp64 instructions on the rotation-to-rotation critical path (n = 100: 4 ms per graph);
              // v_rsq_f64's seed is good to ~2^-26, each step squares that.
              double y = __builtin_amdgcn_rsq(tt);
              {
                const double hy = 0.5 * y;
                const double er = fma(-(tt * y), hy, 0.5);
                y = fma(y, er, y);
              }
              {
                const double hy = 0.5 * y;
                const double er = fma(-(tt * y), hy, 0.5);
                y = fma(y, er, y);
              }
              const double rad = tt * y;
              const double e_next = s * rad;
              s = ei * y;
              c = p * y;
              p = y * num;  // = c d_i - s g
              const double d_next = hp + s * (c * g + s * di);
              // ONE lane stores (64 lanes storing one address serialise in the LDS); the LDS
              // operations of a wave execute in order, so every lane's later reads see the value
              if (lane == 0) {
                we[i + 1] = e_next;
                wd[i + 1] = d_next;
              }
              if (own) Qt[(size_t)(i + 1) * LD + tid] = s * z0 + c * carry;
              carry = c * z0 - s * carry;
              z0 = znext;
              ei = ei_n;
              di = di_n;
            }
            if (own) Qt[(size_t)l * LD + tid] = carry;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            p = -s * s2 * c3 * el1 * we[l] / dl1;
            el = s * p;
            if (lane == 0) {
              we[l] = el;
              wd[l] = c * p;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          } while (fabs(el) > kEps * tst1 && iter < 60);
        }
        const double dfin = wd[l] + f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) {
          wd[l] = dfin;
          we[l] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    __syncthreads();
#ifdef LNZ_PROFILE_PHASES
    tp2 = clock64();
#endif
    if (tid < n) sm.dd[tid] = wd[tid];  // wave 0's copy (identical in every active wave)
    __syncthreads();

    // ---- order by descending |lambda| (ties: ascending lambda, then index)
    // = np.argsort(-|eig|, kind='mergesort') on eigh's ascending output (utils/data_helper.py:218-223)
    if (tid < n) {
      const double di = sm.dd[tid], ai = fabs(di);
      int rank = 0;
      for (int jj = 0; jj < n; ++jj) {
        const double dj = sm.dd[jj], aj = fabs(dj);
        const bool before = (aj > ai) || (aj == ai && (dj < di || (dj == di && jj < tid)));
        rank += before ? 1 : 0;
      }
      sm.perm[rank] = tid;
    }
    __syncthreads();
    // sign convention: largest-magnitude component (first one on ties) is positive
    if (tid < kk) {
      const double* v = Qt + (size_t)sm.perm[tid] * LD;
      double best = 0.0;
      float sg = 1.0f;
      for (int r = 0; r < n; ++r) {
        const double a = fabs(v[r]);
        if (a > best) {
          best = a;
          sg = v[r] < 0 ? -1.0f : 1.0f;
        }
      }
      sm.sgn[tid] = sg;
    }
    __syncthreads();
  }

  // ---- write D [K] and V [N, K] (dataset/graph_data.py:262-287: zero rows >= n, zero slots >= n)
  for (int k = tid; k < K; k += kNT)
    D[(int64_t)b * K + k] = k < kk ? (float)sm.dd[sm.perm[k]] : 0.0f;
  float* Vb = V + (int64_t)b * N * K;
  {
    const int dq = kNT / K, dr = kNT - dq * K;
    int rr = tid / K, k = tid - rr * K;
    for (int idx = tid; idx < N * K; idx += kNT) {
      float v = 0.0f;
      if (rr < n && k < kk) v = sm.sgn[k] * (float)Qt[(size_t)sm.perm[k] * LD + rr];
      Vb[idx] = v;
      rr += dq;
      k += dr;
      if (k >= K) k -= K, ++rr;
    }
  }
  if (info && tid == 0) info[b] = nrestart;
#ifdef LNZ_PROFILE_PHASES
  __syncthreads();
  if (tid == 0 && K >= 4) {  // cycles: Lanczos, QL, ordering + output; n
    const long long tp3 = clock64();
    D[(int64_t)b * K + 0] = (float)(tp1 - tp0);
    D[(int64_t)b * K + 1] = (float)(tp2 - tp1);
    D[(int64_t)b * K + 2] = (float)(tp3 - tp2);
    D[(int64_t)b * K + 3] = (float)n;
  }
#endif
}

}  // namespace

extern "C" int64_t lnz_lanczos_ritz_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 64 || N > kNMax) return 0;
  if (wg_lds_bytes(N, false) <= (size_t)kLdsMax) return 0;
  return (int64_t)B * N * (N | 1) * (int64_t)sizeof(double);
}

// Shared by lnz_lanczos_ritz (lanczos_ritz.hip) and lnz_lanczos_ritz_ws.
// flags: bit 0 = the basis goes to the workspace even if it would fit in LDS (testing).
int lnz_launch_ritz_wg(const float* A, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                       const int32_t* n_nodes, int B, int N, int K, float* D, float* V,
                       int32_t* info, void* workspace, int64_t workspace_bytes, int flags,
                       hipStream_t s) {
  LNZ_REQUIRE(N <= kNMax, LNZ_ENOTSUP,
              "lnz_lanczos_ritz: N=%d > %d: use lnz_lanczos_ritz_large / _sym (streamed kernels)",
              N, kNMax);
  const bool qg = (flags & 1) || wg_lds_bytes(N, false) > (size_t)kLdsMax;
  const size_t lds = wg_lds_bytes(N, qg);
  LNZ_REQUIRE(lds <= (size_t)kLdsMax, LNZ_ENOTSUP, "lnz_lanczos_ritz: N=%d needs %zu B of LDS", N,
              lds);
  if (qg) {
    const int64_t need = (int64_t)B * N * (N | 1) * (int64_t)sizeof(double);
    void* owned = nullptr;
    if (!workspace) {
      // no caller workspace: a stream-ordered allocation that lives for this launch only
      hipError_t e = hipMallocAsync(&owned, (size_t)need, s);
      LNZ_REQUIRE(e == hipSuccess, LNZ_ELAUNCH, "lnz_lanczos_ritz: workspace of %lld B: %s",
                  (long long)need, hipGetErrorString(e));
      workspace = owned;
    } else {
      LNZ_REQUIRE(workspace_bytes >= need, LNZ_EINVAL,
                  "lnz_lanczos_ritz: workspace %lld B < %lld B (lnz_lanczos_ritz_workspace_bytes)",
                  (long long)workspace_bytes, (long long)need);
    }
    auto kfn = lanczos_ritz_wg_kernel<true>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(kfn, dim3(B), dim3(kNT), lds, s, A, stride_b, stride_r, stride_c, n_nodes, N,
                       K, D, V, info, (double*)workspace);
    const int rc = lnz::check_launch("lnz_lanczos_ritz");
    if (owned) (void)hipFreeAsync(owned, s);
    return rc;
  }
  auto kfn = lanczos_ritz_wg_kernel<false>;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(B), dim3(kNT), lds, s, A, stride_b, stride_r, stride_c, n_nodes, N, K,
                     D, V, info, (double*)nullptr);
  return lnz::check_launch("lnz_lanczos_ritz");
}

extern "C" int lnz_lanczos_ritz_ws(const float* A, int64_t stride_b, int64_t stride_r,
                                   int64_t stride_c, const int32_t* n_nodes, int B, int N, int K,
                                   float* D, float* V, int32_t* info, void* workspace,
                                   int64_t workspace_bytes, int flags, lnz_stream_t stream) {
  LNZ_REQUIRE(A && n_nodes && D && V && B > 0 && N > 0 && K > 0, LNZ_EINVAL,
              "lnz_lanczos_ritz_ws: bad arguments (B=%d N=%d K=%d)", B, N, K);
  return lnz_launch_ritz_wg(A, stride_b, stride_r, stride_c, n_nodes, B, N, K, D, V, info, workspace,
                            workspace_bytes, flags, (hipStream_t)stream);
}
