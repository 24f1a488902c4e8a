// R2 + R6: batched Lanczos tridiagonalisation -> tridiagonal eigensolve -> Ritz select, for the
// QM8 regime N <= 32: ONE WAVEFRONT per molecule (lanczos_ritz32_*), fused with the batch plan and
// the Laplacian pack into one launch (prepare_batch_*).  Graphs of 33..192 nodes take the
// workgroup-per-graph kernel of lanczos_ritz_wg.hip (same algorithm, same entry point).
//
// The whole problem (A rows in registers, Krylov basis, T) stays on chip: nothing but A is read from
// HBM and nothing but (D, V) is written:
//   algorithmic bytes / molecule = 4 n^2 (A) + 4 K (D) + 4 N K (V)        
//
// Numerics: fp64 throughout; full-length (m = n) Lanczos, classical
// Gram-Schmidt applied twice against ALL previous vectors (this subsumes the three-term
// recurrence; alpha_j is the coefficient on q_j), restart with the unit vector of largest
// residual on breakdown; eigendecomposition of T by Sturm-count section search + twisted
// factorisation (implicit-shift QL as the fallback); ordering by descending |lambda|.
#include "common.hpp"
#include "prep.hpp"
#include "gains_body.hpp"

namespace {

// |A| <= 1 for L4.  Dropping a coupling beta < tol perturbs T by tol; keeping it leaves q_{j+1}
// orthogonal only to eps/beta after CGS2 — both errors balance at sqrt(eps) ~ 1e-8.
constexpr double kBreakdownTol = 1e-8;
constexpr double kEps = 2.220446049250313e-16;

// =========================================================================================
// Fast path, N <= 32 (QM8): same algorithm, scheduled for latency.
//   * lane = r + 32*h : both wave halves own node row r; every length-32 reduction is split
//     into two 16-element halves (h = 0/1) held in REGISTERS (loaded with back-to-back
//     ds_read_b128, one wait) and combined with one v_permlane32_swap pair — no LDS loop with a
//     dependent load per element;
//   * norm and SpMV share one broadcast of the residual:  beta = |w|, A q = (A w) / beta;
//   * T's diagonal / off-diagonal live in lane registers during QL (uniform reads via
//     v_readlane, writes by lane-select), and the rotated eigenvector row is carried in a
//     register, so the LDS eigenvector update is off the scalar sqrt/rsqrt critical path.
// =========================================================================================
struct Ritz32Smem {
  static constexpr int LD = 34;   // doubles per basis row: b128 row reads conflict free
  double Qt[32 * LD];
  double za[32];
  double zb[32];
  double ca[32];
  double cb[32];
  double dd[32];
  int perm[32];
  float sgn[32];
};

__device__ inline double xhalf_sum(double x) {
  // x(lower half lane) + x(upper half lane), identical (bitwise) in both halves
  unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  double a = __hiloint2double((int)rh[0], (int)rl[0]);  // lower-half value, in every lane
  double b = __hiloint2double((int)rh[1], (int)rl[1]);  // upper-half value, in every lane
  return a + b;
}

__device__ inline double readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__device__ inline void load16(const double* p, double (&v)[16]) {
  // p is 16-byte aligned: 8 x ds_read_b128
#pragma unroll
  for (int t = 0; t < 16; t += 2) {
    double2 x = *reinterpret_cast<const double2*>(p + t);
    v[t] = x.x;
    v[t + 1] = x.y;
  }
}

__device__ inline double dot16(const double (&a)[16], const double (&b)[16]) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int t = 0; t < 16; t += 4) {
    s0 = fma(a[t], b[t], s0);
    s1 = fma(a[t + 1], b[t + 1], s1);
    s2 = fma(a[t + 2], b[t + 2], s2);
    s3 = fma(a[t + 3], b[t + 3], s3);
  }
  return (s0 + s1) + (s2 + s3);
}

// x <- (I - Q Q^T)^2 x over all stored basis rows (rows not yet written are zero); returns the
// accumulated coefficient on basis vector j.  r = lane & 31, h = lane >> 5.
// The second pass runs only where the first one removed more than 99 % of the vector's squared
// length (sum of the squared coefficients against its squared norm, both from the 16-element
// blocks the pass has in registers anyway; every lane computes the same two numbers, so the
// decision is wave uniform): what is left then has lost two digits to cancellation and is
// re-orthogonalised ("twice is enough").  LNZ_RITZ32_CGS2_ALWAYS restores the unconditional form.
__device__ inline double cgs2_32(Ritz32Smem& sm, double& x, int j, int r, int h) {
  constexpr int LD = Ritz32Smem::LD;
  double qrow[16], qcol[16], v[16];
  double coef = 0.0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    double* zbuf = pass ? sm.zb : sm.za;
    double* cbuf = pass ? sm.cb : sm.ca;
    zbuf[r] = x;
    __syncthreads();
    load16(zbuf + 16 * h, v);
    if (pass == 0) {
      load16(&sm.Qt[r * LD + 16 * h], qrow);  // half of basis vector r
#pragma unroll
      for (int t = 0; t < 16; ++t) qcol[t] = sm.Qt[(16 * h + t) * LD + r];  // element r of vectors
    }
    double c = xhalf_sum(dot16(qrow, v));  // <q_r, x>
#ifndef LNZ_RITZ32_CGS2_ALWAYS
    const double xx = pass == 0 ? xhalf_sum(dot16(v, v)) : 0.0;  // |x|^2
#endif
    cbuf[r] = c;
    coef += readlane_f64(c, j);
    __syncthreads();
    load16(cbuf + 16 * h, v);
    x -= xhalf_sum(dot16(qcol, v));
#ifndef LNZ_RITZ32_CGS2_ALWAYS
    if (pass == 0) {
      const double cc2 = xhalf_sum(dot16(v, v));  // sum of the squared coefficients
      if (cc2 <= 0.99 * xx) break;
    }
#endif
  }
  return coef;
}

// -----------------------------------------------------------------------------------------
// Eigendecomposition of the Lanczos tridiagonal T (n <= 32) with every lane working, instead of the
// implicit-QL sweep whose ~n^2 rotations form one serial chain that all 64 lanes recompute
// (80 % of a 26-atom molecule's time):
//   1. T splits where Lanczos restarted (off-diagonal exactly 0) or an off-diagonal is negligible;
//      lane k owns the (k - s)-th eigenvalue of its unreduced block [s, t] — simple there, while
//      equal eigenvalues of different blocks get eigenvectors with disjoint support;
//   2. eigenvalue: section search on Sturm counts (LAPACK dstebz's recurrence), the two lane
//      halves probing the interval at 1/3 and 2/3 — the interval shrinks 3x per pass;
//   3. eigenvector of the block: twisted factorisation (the dlar1v / MRRR vector: stationary and
//      progressive qd transforms of T - lambda, twist at min |gamma|), one lane per vector, no
//      pivoting, no iteration.  Orthogonality of twisted vectors degrades like eps / relative
//      gap, so a block with two eigenvalues closer than ~1e-8 |T| makes the function return
//      false (early, before anything is overwritten) and the caller falls back to the QL sweep;
//   4. Ritz vectors V = Q S: lane (k, h) accumulates 16 node rows of column k from broadcast
//      reads of the basis, then the basis rows are overwritten by the Ritz vectors.
// On exit sm.dd[k] / sm.Qt[k][.] hold eigenpair k (any order), like after the QL sweep.
// -----------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  return fma(y, fma(-x, y, 1.0), y);
}

// Eigenvalue of rank (r - s) of the lane's block (see tridiag_eig_parallel).  Kept out of line:
// inlined, its two 32-entry register arrays stay allocated next to those of the eigenvector stage
// and push the kernel past 256 registers.
__device__ __attribute__((noinline)) double tridiag_eigenvalue(const Ritz32Smem& sm, const int n,
                                                               const int r, const int h, const int s,
                                                               const int t, const int jloc,
                                                               const double gsc, bool* bail,
                                                               double* blo, double* bhi,
                                                               const bool may_bail) {
  // may_bail: first call (start from the spectral bound, stop at pass 12 if a cluster shows);
  // otherwise resume from the bracket the first call returned in *blo / *bhi
  const bool act = r < n;
    // ---- 2. eigenvalue.  Sturm count in product form — p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2},
    //      one dependent FMA per row instead of a division; the count is the number of sign
    //      changes, collected as sign bits.  The lane's rows outside its block are replaced by a
    //      diagonal above the spectrum (never a sign change) with no coupling, so the 32-row
    //      recurrence is straight-line code with no predicate.  Each lane carries two probe
    //      points, the two lane halves four: the bracket shrinks 5x per pass.
    double dv[32], e2[32];
    const double pad = 2.0 * gsc + 1.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const bool in = act && i >= s && i <= t;
      dv[i] = in ? sm.za[i] : pad;
      e2[i] = (in && i < t) ? sm.ca[i] : 0.0;
    }
    double lo = may_bail ? -gsc : *blo, hi = may_bail ? gsc : *bhi;
    bool cluster = false;
    for (int it = may_bail ? 0 : 13; it < 30; ++it) {
      const double w = (hi - lo) * 0.2;
      const double xa = lo + w * (h ? 3.0 : 1.0), xb = lo + w * (h ? 4.0 : 2.0);
      double a1 = 1.0, a2 = 0.0, b1 = 1.0, b2 = 0.0;
      unsigned ma = 0u, mb = 0u;  // sign bits of p_0 .. p_31, row 31 in bit 0
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const double ep = i > 0 ? e2[i > 0 ? i - 1 : 0] : 0.0;
        const double pa = fma(dv[i] - xa, a1, -(ep * a2));
        const double pb = fma(dv[i] - xb, b1, -(ep * b2));
        ma = (ma << 1) | ((unsigned)__double2hiint(pa) >> 31);
        mb = (mb << 1) | ((unsigned)__double2hiint(pb) >> 31);
        a2 = a1, a1 = pa, b2 = b1, b1 = pb;
        if ((i & 7) == 7) {  // keep |p| inside the exponent range
          const double fa = fabs(a1), fb = fabs(b1);
          const double ka = fa < 1e-100 ? 1e100 : (fa > 1e100 ? 1e-100 : 1.0);
          const double kb = fb < 1e-100 ? 1e100 : (fb > 1e100 ? 1e-100 : 1.0);
          a1 *= ka, a2 *= ka, b1 *= kb, b2 *= kb;
        }
      }
      // sign changes between consecutive p (p_{-1} = 1 > 0)
      const int ca = __popc(ma ^ (ma >> 1)), cb = __popc(mb ^ (mb >> 1));
      const int oa = __shfl_xor(ca, 32, 64), ob = __shfl_xor(cb, 32, 64);
      // eigenvalues of the block below lo + w, lo + 2w, lo + 3w, lo + 4w
      const int c1 = h ? oa : ca, c2 = h ? ob : cb, c3 = h ? ca : oa, c4 = h ? cb : ob;
      const double x1 = lo + w, x2 = lo + 2.0 * w, x3 = lo + 3.0 * w, x4 = lo + 4.0 * w;
      if (c1 > jloc) {
        hi = x1;
      } else if (c2 > jloc) {
        lo = x1, hi = x2;
      } else if (c3 > jloc) {
        lo = x2, hi = x3;
      } else if (c4 > jloc) {
        lo = x3, hi = x4;
      } else {
        lo = x4;
      }
      if (may_bail && it == 12) {
        // Bracket width is now ~1e-8 |T|.  Two eigenvalues of ONE block still sharing a bracket:
        // a degenerate eigenvalue whose second copy crept into the Krylov space through round-off
        // instead of a clean breakdown — the twisted vectors of such a pair would coincide.
        // Rare (about one molecule per thousand): give up early, the caller runs the QL sweep.
        const double lo_n = __shfl_up(lo, 1, 64);
        const int s_n = __shfl_up(s, 1, 64);
        cluster = act && (r & 31) > 0 && s_n == s && lo_n == lo;
        if (__any(cluster)) {
          *bail = cluster;  // per lane: shares its bracket with the lane below
          *blo = lo, *bhi = hi;
          return 0.0;
        }
      }
      // LAPACK dstebz's stopping rule: relative to |lambda| but never below ulp * |T| — an
      // eigenvalue at (or next to) zero would otherwise keep every lane of the wave in the loop
      // for all 30 passes without gaining anything the fp32 outputs or the twisted vectors need
      const bool done = !act || (hi - lo) <= 4.0 * kEps * fmax(fmax(fabs(lo), fabs(hi)), 0.125 * gsc);
      if (__all(done)) break;
    }
    return 0.5 * (lo + hi);
}

__device__ inline bool tridiag_eig_parallel(Ritz32Smem& sm, const double dreg, const double ereg,
                                            const int n, const int r, const int h, long long* ts = nullptr) {
  constexpr int LD = Ritz32Smem::LD;
  if (ts) ts[0] = clock64();
  // ---- d, e -> LDS (broadcast reads); negligible couplings split the matrix
  const double dn = __shfl_down(dreg, 1, 64);
  const bool live = r < n - 1 && fabs(ereg) > kEps * (fabs(dreg) + fabs(dn));
  if (h == 0) {
    sm.za[r] = r < n ? dreg : 0.0;
    sm.zb[r] = live ? ereg : 0.0;
    sm.ca[r] = live ? ereg * ereg : 0.0;
  }
  __syncthreads();
  const bool act = r < n;
  // block [s, t] of this lane's index; the lane owns the (r - s)-th eigenvalue of that block.
  // Bit i of `cut` = coupling e_i (between rows i and i+1) is dead: s is one past the highest cut
  // below r, t the lowest cut at or above r (bit n-1 is always set) — bit scans, not LDS walks.
  const unsigned cut = (unsigned)__ballot(h == 0 && !live);
  int s = r, t = r;
  if (act) {
    const unsigned below = cut & ((1u << r) - 1u);          // cuts at e_0 .. e_{r-1}
    s = below ? 32 - __clz(below) : 0;
    const unsigned above = cut & ~((1u << r) - 1u);         // cuts at e_r ..
    t = above ? __ffs(above) - 1 : n - 1;
    t = t > n - 1 ? n - 1 : t;
  }
  // Gershgorin bound of the spectrum: one row per lane, wave maximum
  const double e_live = live ? fabs(ereg) : 0.0;
  const double e_prev = __shfl_up(e_live, 1, 64);
  double gmax = act ? fabs(dreg) + ((r & 31) > 0 ? e_prev : 0.0) + e_live : 0.0;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, off, 64));
  const double gsc = gmax > 0.0 ? gmax : 1.0;
  bool bail = false;
  double blo = 0.0, bhi = 0.0;
  double lam = tridiag_eigenvalue(sm, n, r, h, s, t, r - s, gsc, &bail, &blo, &bhi, true);
  // twist window: where this lane looks for the twist index (its whole block unless it is part
  // of a cluster, see below)
  int ws = s, wt = t;
  bool member = false;
  if (__any(bail)) {
    // Rescue of a cluster (rare, ~2 molecules per thousand).  The second copy of a degenerate
    // eigenvalue enters the Krylov space through couplings of round-off origin that stayed above
    // the breakdown threshold, so the block holds the eigenvalue twice (to ~1e-9) with one copy
    // living mostly on either side of a weak coupling.  The twisted factorisation of the FULL block
    // yields an accurate eigenvector for EVERY twist index where gamma is tiny — one such index on
    // each side.  The cluster's lanes therefore cut their block at its couplings below 1e-3 |T|
    // into windows, take the window that the rank inside the shared bracket selects (Sturm counts
    // of the window's own sub-matrix), look for their twist index only there, and the upper lane
    // is finally orthogonalised against the lower one.  If the windows cannot separate the
    // copies, the caller runs the QL sweep.
    const int upper = __shfl_down(bail ? 1 : 0, 1, 64);  // all lanes take part in the shuffle
    member = bail || (upper != 0 && (r & 31) < 31);
    auto count = [&](double x, int a, int b) {  // eigenvalues of rows a..b (no outside coupling) below x
      int c = 0;
      double q = 1.0;
      for (int i = a; i <= b; ++i) {
        const double e2p = i > a ? sm.ca[i - 1] : 0.0;
        q = (sm.za[i] - x) - e2p / q;
        q = fabs(q) < 1e-290 ? -1e-290 : q;
        c += q < 0.0 ? 1 : 0;
      }
      return c;
    };
    bool ok = true;
    if (member && act) {
      const double cut = 1e-3 * gsc, wd = 1e-5 * gsc;
      const double xl = blo - wd, xh = bhi + wd;
      int g = (r - s) - count(xl, s, t);  // rank inside the widened bracket
      int a = s;
      bool found = false;
      for (int i = s; i <= t && !found; ++i) {
        if (i == t || fabs(sm.zb[i]) <= cut) {
          const int inside = count(xh, a, i) - count(xl, a, i);
          if (g < inside) {
            ws = a, wt = i;
            found = true;
            ok = inside == 1;  // two copies in one window: not separable this way
          } else {
            g -= inside;
            a = i + 1;
          }
        }
      }
      ok = ok && found;
    }
    if (__any(!ok)) return false;
    bool bail2 = false;
    lam = tridiag_eigenvalue(sm, n, r, h, s, t, r - s, gsc, &bail2, &blo, &bhi, false);
  }
  if (ts) ts[1] = clock64();
  // ---- 3. eigenvector of the block: twisted factorisation of T - lam
  double z[32];
  {
    double Dm[32];
    const double tiny = kEps * gsc;
    // stationary transform, top down: D+_{i+1} = (d_{i+1} - lam) - e_i^2 / D+_i   (parked in z)
    double Dp = 1.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      z[i] = 0.0;
      if (i < n) {
        const bool in = act && i >= s && i <= t;
        const double di = sm.za[i] - lam;
        const double e2p = i > 0 ? sm.ca[i > 0 ? i - 1 : 0] : 0.0;
        double dnew = (i > s) ? di - e2p * rcp_nr(Dp) : di;
        dnew = fabs(dnew) < tiny ? (dnew < 0.0 ? -tiny : tiny) : dnew;
        Dp = in ? dnew : Dp;
        z[i] = in ? dnew : 0.0;
      }
    }
    // progressive transform, bottom up: D-_i = (d_i - lam) - e_i^2 / D-_{i+1};
    // gamma_i = D+_i + D-_i - (d_i - lam); twist where |gamma| is smallest
    double Dn = 1.0, gbest = 1e300;
    int tw = s;
#pragma unroll
    for (int i = 31; i >= 0; --i) {
      Dm[i] = 1.0;
      if (i < n) {
        const bool in = act && i >= s && i <= t;
        const double di = sm.za[i] - lam;
        const double e2 = sm.ca[i];
        double dnew = (i < t) ? di - e2 * rcp_nr(Dn) : di;
        dnew = fabs(dnew) < tiny ? (dnew < 0.0 ? -tiny : tiny) : dnew;
        Dn = in ? dnew : Dn;
        Dm[i] = in ? dnew : 1.0;
        const double g = fabs(z[i] + dnew - di);
        if (in && i >= ws && i <= wt && g < gbest) {
          gbest = g;
          tw = i;
        }
      }
    }
    // z_tw = 1; upward z_i = -(e_i / D+_i) z_{i+1} (D+_i is read from z[i] before it is replaced);
    // downward z_{i+1} = -(e_i / D-_{i+1}) z_i
#pragma unroll
    for (int i = 31; i >= 0; --i) {
      if (i < n) {
        const bool in = act && i >= s && i <= t;
        const double up = i < 31 ? -(sm.zb[i] * rcp_nr(z[i])) * z[i < 31 ? i + 1 : 31] : 0.0;
        z[i] = !in ? 0.0 : (i == tw ? 1.0 : (i < tw ? up : 0.0));
      }
    }
#pragma unroll
    for (int i = 1; i < 32; ++i) {
      if (i < n) {
        const bool in = act && i >= s && i <= t;
        z[i] = (in && i > tw) ? -(sm.zb[i - 1] * rcp_nr(Dm[i])) * z[i - 1] : z[i];
      }
    }
    double nn = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) nn = fma(z[i], z[i], nn);
    const double sc = act ? rsqrt(nn) : 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) z[i] *= sc;
  }

  if (__any(member)) {
    // cluster lanes: modified Gram-Schmidt in lane order.  pos = how many consecutive lower lanes
    // belong to the same cluster; the lane at position p is orthogonalised against the (final)
    // vectors of the p lanes below it.
    int pos = 0;
    {
      bool chain = true;
      for (int d = 1; d <= 3; ++d) {
        const int s_lo = __shfl_up(s, d, 64);
        const int m_lo = __shfl_up(member ? 1 : 0, d, 64);
        const double l_lo = __shfl_up(lam, d, 64);
        chain = chain && act && member && m_lo != 0 && (r & 31) >= d && s_lo == s &&
                fabs(lam - l_lo) <= 1e-6 * gsc;
        pos += chain ? 1 : 0;
      }
    }
    for (int p = 1; p <= 3; ++p) {
      if (!__any(pos >= p)) break;
      for (int d = 1; d <= p; ++d) {
        const bool fix = pos == p;
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) dot = fma(z[i], __shfl_up(z[i], d, 64), dot);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const double zl = __shfl_up(z[i], d, 64);
          z[i] = fix ? fma(-dot, zl, z[i]) : z[i];
        }
      }
      double nn = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) nn = fma(z[i], z[i], nn);
      const double sc = (pos == p && nn > 0.0) ? rsqrt(nn) : 1.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) z[i] *= sc;
    }
  }

  if (ts) ts[2] = clock64();
  // ---- 4. V = Q S: column k, node rows 16h .. 16h+15
  double acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    if (i < n) {
      double q[16];
      load16(&sm.Qt[i * LD + 16 * h], q);
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] = fma(q[u], z[i], acc[u]);
    }
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int u = 0; u < 16; ++u) sm.Qt[r * LD + 16 * h + u] = acc[u];
    if (h == 0) sm.dd[r] = lam;
  }
  __syncthreads();
  if (ts) ts[3] = clock64();
  return true;
}

// body of one molecule's wavefront (lane = 0..63); the caller owns the LDS block
__device__ __forceinline__ void lanczos_ritz32_body(
    const float* __restrict__ A, int64_t sb, int64_t sr, int64_t sc,
    const int32_t* __restrict__ n_nodes, int N, int K, float* __restrict__ D,
    float* __restrict__ V, int32_t* __restrict__ info, const int b, const int lane,
    Ritz32Smem& sm) {
  constexpr int LD = Ritz32Smem::LD;
  const int r = lane & 31, h = lane >> 5;
  int n = n_nodes[b];
  n = n < 0 ? 0 : (n > N ? N : n);
  const int kk = K < n ? K : n;

  // this lane's half row of A (fp64 registers), zero outside the n x n block
  double arow[16];
  {
    const float* Ab = A + (int64_t)b * sb + (int64_t)r * sr;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      int c = 16 * h + t;
      arow[t] = (r < n && c < n) ? (double)Ab[c * sc] : 0.0;
    }
  }
  // zero the basis (rows beyond the current step must read as zero)
  for (int idx = lane; idx < 32 * LD / 2; idx += 64)
    reinterpret_cast<double2*>(sm.Qt)[idx] = make_double2(0.0, 0.0);
  __syncthreads();
#ifdef LNZ_PROFILE_PHASES
  long long tp0 = clock64(), tp1 = tp0, tp2 = tp0;
  long long tsx[4] = {0, 0, 0, 0};
#endif

  int nrestart = 0;
  double dreg = 0.0, ereg = 0.0;  // lane r holds T[r][r], T[r][r+1]
  if (n > 0) {
    double w = 0.0;
    if (r < n) {
      unsigned hsh = (unsigned)(r + 1) * 2654435761u;
      w = 1.0 + (double)((hsh >> 8) & 0xffff) * (1.0 / 65536.0);
    }
    bool fresh = true;  // w is a start/restart vector: its norm is not a coupling beta
    for (int j = 0; j < n; ++j) {
      double beta, u;
      for (;;) {
        // ---- one broadcast of w: beta = |w| and u = A w
        sm.za[r] = w;
        __syncthreads();
        double v[16];
        load16(sm.za + 16 * h, v);
        double nn = xhalf_sum(dot16(v, v));
        u = xhalf_sum(dot16(arow, v));
        beta = sqrt(nn);
        __syncthreads();
        if (fresh || beta > kBreakdownTol) break;
        // breakdown: span(q_0..q_{j-1}) is A-invariant -> restart from the unit vector with the
        // largest residual against the basis; T[j-1][j] stays 0
        ++nrestart;
        double colsq[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) colsq[t] = sm.Qt[(16 * h + t) * LD + r];
        const double res = 1.0 - xhalf_sum(dot16(colsq, colsq));
        // arg max over the rows (lowest index on ties), by lane shuffles inside each half
        double best = r < n ? res : -1.0;
        int cand = r;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
          const double ob = __shfl_xor(best, off, 64);
          const int oc = __shfl_xor(cand, off, 64);
          const bool take = ob > best || (ob == best && oc < cand);
          best = take ? ob : best;
          cand = take ? oc : cand;
        }
        w = (r == cand) ? 1.0 : 0.0;
        (void)cgs2_32(sm, w, 0, r, h);
        fresh = true;
      }
      if (!fresh && r == j - 1) ereg = beta;
      fresh = false;
      const double binv = 1.0 / beta;
      const double q = w * binv;
      double x = u * binv;  // A q
      if (h == 0) sm.Qt[j * LD + r] = q;
      // (the __syncthreads inside cgs2_32 orders this store before the basis reads)
      const double alpha = cgs2_32(sm, x, j, r, h);
      if (r == j) dreg = alpha;
      w = x;
    }
    __syncthreads();
#ifdef LNZ_PROFILE_PHASES
    tp1 = clock64();
#endif

#ifndef LNZ_RITZ32_QL
#ifdef LNZ_PROFILE_PHASES
    const bool solved = tridiag_eig_parallel(sm, dreg, ereg, n, r, h, tsx);
#else
    const bool solved = tridiag_eig_parallel(sm, dreg, ereg, n, r, h);
#endif
#else
    const bool solved = false;
#endif
    if (!solved) {
    nrestart += 256;  // diagnostic: the QL fallback ran (info = restarts + 256)
    // ---- implicit-shift QL (tql2 recurrences); d/e in lane registers, eigenvectors in Qt ----
    double f = 0.0, tst1 = 0.0;
    for (int l = 0; l < n; ++l) {
      tst1 = fmax(tst1, fabs(readlane_f64(dreg, l)) + fabs(readlane_f64(ereg, l)));
      const unsigned long long small =
          __ballot(h == 0 && r >= l && r < n && (r == n - 1 || fabs(ereg) <= kEps * tst1));
      const int m = __builtin_ctzll(small);
      if (m > l) {
        int iter = 0;
        double el;
        do {
          ++iter;
          double g = readlane_f64(dreg, l);
          el = readlane_f64(ereg, l);
          double p = (readlane_f64(dreg, l + 1) - g) / (2.0 * el);
          double rr = sqrt(p * p + 1.0);
          if (p < 0) rr = -rr;
          const double dl = el / (p + rr);
          const double dl1 = el * (p + rr);
          const double hh = g - dl;
          dreg = (r == l) ? dl : (r == l + 1) ? dl1 : (r >= l + 2 && r < n) ? dreg - hh : dreg;
          f += hh;
          p = readlane_f64(dreg, m);
          double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
          const double el1 = readlane_f64(ereg, l + 1);
          double carry = sm.Qt[m * LD + r];
          double z0 = sm.Qt[(m - 1) * LD + r];
          double ei = readlane_f64(ereg, m - 1);
          double di = readlane_f64(dreg, m - 1);
#pragma unroll 2
          for (int i = m - 1; i >= l; --i) {
            // prefetch the next rotation's inputs: none of them is written by this rotation
            const int ip = i > l ? i - 1 : l;
            const double znext = sm.Qt[ip * LD + r];
            const double ei_n = readlane_f64(ereg, ip);
            const double di_n = readlane_f64(dreg, ip);
            c3 = c2;
            c2 = c;
            s2 = s;
            g = c * ei;
            const double hp = c * p;
            const double tt = fma(p, p, ei * ei);
            const double num = fma(p, di, -(ei * g));  // (p d_i - e_i g): off the rsqrt chain
            // 1/sqrt(tt): hardware seed + ONE Newton step (tt is a normal double here:
            // |e_i| > eps * tst1 for l <= i < m).  v_rsq_f64's seed is good to ~2^-26, one step
            // squares that; the outputs of this kernel are fp32 (a second step changed D by at
            // most one fp32 ulp and no Ritz residual on 1024 molecules, and costs 4 % of the kernel:
            // the three dependent FMAs sit on the rotation-to-rotation critical path).
            double y = __builtin_amdgcn_rsq(tt);
            {
              double hy = 0.5 * y;
              double er = fma(-(tt * y), hy, 0.5);
              y = fma(y, er, y);
            }
            const double rad = tt * y;
            const double e_next = s * rad;
            s = ei * y;
            c = p * y;
            p = y * num;  // = c d_i - s g
            const double d_next = hp + s * (c * g + s * di);
            ereg = (r == i + 1) ? e_next : ereg;
            dreg = (r == i + 1) ? d_next : dreg;
            sm.Qt[(i + 1) * LD + r] = s * z0 + c * carry;  // both halves hold the same value
            carry = c * z0 - s * carry;
            z0 = znext;
            ei = ei_n;
            di = di_n;
          }
          if (h == 0) sm.Qt[l * LD + r] = carry;
          p = -s * s2 * c3 * el1 * readlane_f64(ereg, l) / dl1;
          el = s * p;
          ereg = (r == l) ? el : ereg;
          dreg = (r == l) ? c * p : dreg;
          __syncthreads();  // rows written by half 0 are re-read by both halves next sweep
        } while (fabs(el) > kEps * tst1 && iter < 60);
      }
      dreg = (r == l) ? dreg + f : dreg;
      ereg = (r == l) ? 0.0 : ereg;
    }
    if (h == 0) sm.dd[r] = dreg;
    __syncthreads();
    }
#ifdef LNZ_PROFILE_PHASES
    tp2 = clock64();
#endif

    // ---- order by descending |lambda| (ties: ascending lambda, then index) — see generic kernel
    {
      // every lane compares its eigenvalue with the others' through lane reads (no LDS walk)
      const double di = lane < n ? sm.dd[lane] : 0.0, ai = fabs(di);
      int rank = 0;
      for (int jj = 0; jj < n; ++jj) {
        const double dj = readlane_f64(di, jj), aj = fabs(dj);
        const bool before = (aj > ai) || (aj == ai && (dj < di || (dj == di && jj < lane)));
        rank += before ? 1 : 0;
      }
      if (lane < n) sm.perm[rank] = lane;
    }
    __syncthreads();
    if (lane < kk) {
      const double* v = &sm.Qt[sm.perm[lane] * LD];
      double best = 0.0;
      float sg = 1.0f;
      for (int x = 0; x < n; ++x) {
        double av = fabs(v[x]);
        if (av > best) {
          best = av;
          sg = v[x] < 0 ? -1.0f : 1.0f;
        }
      }
      sm.sgn[lane] = sg;
    }
    __syncthreads();
  }

  for (int k = lane; k < K; k += 64) D[(int64_t)b * K + k] = k < kk ? (float)sm.dd[sm.perm[k]] : 0.0f;
  float* Vb = V + (int64_t)b * N * K;
  {
    // idx = rr * K + k walks by 64 per pass: one division up front, then carries
    const int dq = 64 / K, dr = 64 - dq * K;
    int rr = lane / K, k = lane - rr * K;
    for (int idx = lane; idx < N * K; idx += 64) {
      float v = 0.0f;
      if (rr < n && k < kk) v = sm.sgn[k] * (float)sm.Qt[sm.perm[k] * LD + rr];
      Vb[idx] = v;
      rr += dq;
      k += dr;
      if (k >= K) k -= K, ++rr;
    }
  }
  if (info && lane == 0) info[b] = nrestart;
#ifdef LNZ_PROFILE_PHASES
  __syncthreads();
  if (lane == 0) {
    long long tp3 = clock64();
    D[(int64_t)b * K + 0] = (float)(tp1 - tp0);
    D[(int64_t)b * K + 1] = (float)(tp2 - tp1);
    D[(int64_t)b * K + 2] = (float)(tp3 - tp2);
    D[(int64_t)b * K + 3] = (float)n;
    D[(int64_t)b * K + 4] = (float)(tsx[1] - tsx[0]);
    D[(int64_t)b * K + 5] = (float)(tsx[2] - tsx[1]);
    D[(int64_t)b * K + 6] = (float)(tsx[3] - tsx[2]);
  }
#endif
}

__global__ __launch_bounds__(64) void lanczos_ritz32_kernel(
    const float* __restrict__ A, int64_t sb, int64_t sr, int64_t sc,
    const int32_t* __restrict__ n_nodes, int N, int K, float* __restrict__ D,
    float* __restrict__ V, int32_t* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) Ritz32Smem sm;
  lanczos_ritz32_body(A, sb, sr, sc, n_nodes, N, K, D, V, info, blockIdx.x, threadIdx.x, sm);
}

// Everything the fused forward needs from a collated batch besides the gains, in ONE launch of
// 256-thread workgroups: workgroup 0 plans the batch (tile plan + live eigen slots), workgroups
// 1..B are the Lanczos/QL wavefronts (one live wave each — the other three exit at once, and a
// terminated wave does not take part in barriers), workgroups B+1..2B pack the Laplacian tiles.
// The Ritz wavefronts are the long pole (latency bound, one wave per SIMD); dispatched first, they
// leave most of the machine idle, and the two byte movers run in that shadow instead of in front.
__global__ __launch_bounds__(256) void prepare_batch_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int N, int C,
    float4* __restrict__ Lp, const uint8_t* __restrict__ mask, int B, int n_cu, int allow_pairs,
    int wg_cap, int32_t* __restrict__ plan, int32_t* __restrict__ n_wg, int K,
    int32_t* __restrict__ gain_rows, int32_t* __restrict__ n_gain_rows,
    const int32_t* __restrict__ n_nodes, float* __restrict__ D, float* __restrict__ V,
    int32_t* __restrict__ info, uint32_t* __restrict__ ident, int32_t* __restrict__ strips,
    int32_t* __restrict__ n_strips) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ __attribute__((aligned(16))) Ritz32Smem sm;
  const int blk = blockIdx.x;
  if (blk == 0) {
    plan_tiles_body(mask, B, N, n_cu, allow_pairs, wg_cap, plan, n_wg, K, gain_rows, n_gain_rows);
    if (strips) {  // (this kernel is launched with a staging tile > kPrepLds >= kStripScratch)
      __syncthreads();
      plan_strips_body(mask, B, N, n_cu, strips, n_strips, reinterpret_cast<unsigned char*>(tile));
    }
  } else if (blk <= B) {
    if (threadIdx.x >= 64) return;
    // channel 0 of L is the simple-graph Laplacian the Ritz pairs belong to (dataset/qm8.py:262)
    lanczos_ritz32_body(L, sb, sr, sc, n_nodes, N, K, D, V, info, blk - 1, threadIdx.x, sm);
  } else {
    pack_laplacian_body(L, sb, sr, sc, sch, N, C, Lp, tile, blk - 1 - B, ident);
  }
}

constexpr int kPrepLds = 20480;  // static LDS of the fused preparation launch (>= sizeof(Ritz32Smem))
static_assert(sizeof(Ritz32Smem) <= kPrepLds, "Ritz scratch must fit the shared block");
static_assert(kStripScratch <= kPrepLds, "the strip planner works in the same block");

// The same launch with ONE static LDS block per workgroup, used according to its role (Ritz
// scratch or pack staging tile), for staging tiles up to kPrepLds (QM8: 26 x 26 x 7 floats =
// 18.9 KB).  With the separate dynamic tile above every workgroup carries ~29 KB: the four
// resident Ritz workgroups of a CU (one per SIMD, the long pole) leave room for ONE more, and the
// B pack workgroups trickle through behind them instead of running in their shadow.
__global__ __launch_bounds__(256) void prepare_batch_union_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int N, int C,
    float4* __restrict__ Lp, const uint8_t* __restrict__ mask, int B, int n_cu, int allow_pairs,
    int wg_cap, int32_t* __restrict__ plan, int32_t* __restrict__ n_wg, int K,
    int32_t* __restrict__ gain_rows, int32_t* __restrict__ n_gain_rows,
    const int32_t* __restrict__ n_nodes, float* __restrict__ D, float* __restrict__ V,
    int32_t* __restrict__ info, uint32_t* __restrict__ ident, int32_t* __restrict__ strips,
    int32_t* __restrict__ n_strips) {
  __shared__ __attribute__((aligned(16))) unsigned char ubuf[kPrepLds];
  const int blk = blockIdx.x;
  if (blk == 0) {
    plan_tiles_body(mask, B, N, n_cu, allow_pairs, wg_cap, plan, n_wg, K, gain_rows, n_gain_rows);
    if (strips) {
      __syncthreads();
      plan_strips_body(mask, B, N, n_cu, strips, n_strips, ubuf);
    }
  } else if (blk <= B) {
    if (threadIdx.x >= 64) return;
    lanczos_ritz32_body(L, sb, sr, sc, n_nodes, N, K, D, V, info, blk - 1, threadIdx.x,
                        *reinterpret_cast<Ritz32Smem*>(ubuf));
  } else {
    pack_laplacian_body(L, sb, sr, sc, sch, N, C, Lp, reinterpret_cast<float*>(ubuf), blk - 1 - B,
                        ident);
  }
}

// Software pipeline over a stream of batches, ONE launch of 256-thread workgroups: workgroup 0
// plans batch k+1, workgroups 1..B are its Lanczos / eigensolve wavefronts (one live wave each,
// raised issue priority), the next n_cons workgroups run the spectral-gains MLP of batch k (whose
// D and live-row list were completed by the previous launch: no dependency inside this one), the
// last B workgroups pack batch k+1's Laplacian tiles.  The Ritz wavefronts are latency bound and
// leave the matrix pipes idle; the MLP is matrix-pipe work: one gains wave runs next to each Ritz
// wave (2 waves per SIMD at 248 VGPRs), two once the Ritz waves are gone.
// (A same-batch variant, gains consumers spinning on per-molecule `done` flags, was built and
// measured: 253 us with the QL eigensolver, 244 us with the parallel one — slower than the plain
// launches once all molecules finish within a narrow window; it was removed.)
__global__ __launch_bounds__(256, 2) void prepare_batch_gains_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int N, int C,
    float4* __restrict__ Lp, const uint8_t* __restrict__ mask, int B, int n_cu, int allow_pairs,
    int wg_cap, int32_t* __restrict__ plan, int32_t* __restrict__ n_wg, int K,
    int32_t* __restrict__ gain_rows, int32_t* __restrict__ n_gain_rows,
    const int32_t* __restrict__ n_nodes, float* __restrict__ D, float* __restrict__ V,
    lnz_gains::DistArr dist, int S, int num_layer,
    const float* __restrict__ mlp_pack, float* __restrict__ G, uint32_t* __restrict__ ident,
    int n_cons, const float* __restrict__ Dg, const int32_t* __restrict__ rows_g,
    const int32_t* __restrict__ n_rows_g, int Bg, int32_t* __restrict__ strips,
    int32_t* __restrict__ n_strips) {
  // One static LDS block per workgroup, used according to its role (Ritz scratch, pack staging
  // tile): with a separate dynamic tile every workgroup carried 32 KB and the four resident Ritz
  // workgroups of a CU left room for ONE more — the pack workgroups trickled through and the
  // consumers started 50 us late.
  __shared__ __attribute__((aligned(16))) unsigned char ubuf[kPrepLds];
  Ritz32Smem& sm = *reinterpret_cast<Ritz32Smem*>(ubuf);
  float* tile = reinterpret_cast<float*>(ubuf);
  const int blk = blockIdx.x;
  if (blk == 0) {
    plan_tiles_body(mask, B, N, n_cu, allow_pairs, wg_cap, plan, n_wg, K, gain_rows, n_gain_rows);
    if (strips) {
      __syncthreads();
      plan_strips_body(mask, B, N, n_cu, strips, n_strips, ubuf);
    }
  } else if (blk <= B) {
    if (threadIdx.x >= 64) return;
    // the Lanczos / eigensolve chain is the critical path of the launch: it wins every issue
    // arbitration against the gains wave that shares its SIMD
    __builtin_amdgcn_s_setprio(3);
    lanczos_ritz32_body(L, sb, sr, sc, n_nodes, N, K, D, V, nullptr, blk - 1, threadIdx.x, sm);
  } else {
    // behind the Ritz workgroups: the gains workgroups, then the pack (needed by the next launch
    // only).  Alternating the two measured slower (229 vs 182 us).
    const int p = blk - B - 1;
    if (p >= n_cons) {
      pack_laplacian_body(L, sb, sr, sc, sch, N, C, Lp, tile, p - n_cons, ident);
      return;
    }
    const int lane = threadIdx.x & 63;
    const int gw = p * 4 + (threadIdx.x >> 6);
    const int t = gw / num_layer, l = gw - t * num_layer;
    const int R = *n_rows_g;
    if (32 * t >= R) return;
    const int idx = 32 * t + (lane & 31);
    const bool valid = idx < R;
    const int row = valid ? rows_g[idx] : 0;
    lnz_gains::gains_mlp_tile(Dg, row, valid, l, lane, Bg, K, dist, S, mlp_pack, G);
  }
}

}  // namespace

extern "C" int lnz_prepare_batch(const float* L, int64_t stride_b, int64_t stride_r,
                                 int64_t stride_c, int64_t stride_ch, int B, int N, int C,
                                 float* Lp, const uint8_t* mask, const int32_t* n_nodes, int n_cu,
                                 int allow_pairs, int32_t* plan, int32_t* n_wg, int K,
                                 int32_t* gain_rows, int32_t* n_gain_rows, float* D, float* V,
                                 int32_t* info, uint32_t* ident, int32_t* strips,
                                 int32_t* n_strips, lnz_stream_t stream) {
  LNZ_REQUIRE(L && Lp && mask && n_nodes && plan && n_wg && D && V && B > 0 && C > 0 &&
                  C <= LNZ_MAX_CHANNELS && n_cu > 0 && K > 0,
              LNZ_EINVAL, "lnz_prepare_batch: bad arguments (B=%d C=%d K=%d)", B, C, K);
  LNZ_REQUIRE(N > 0 && N <= LNZ_TILE, LNZ_ENOTSUP, "lnz_prepare_batch: N=%d > %d", N, LNZ_TILE);
  LNZ_REQUIRE(!gain_rows || n_gain_rows, LNZ_EINVAL, "lnz_prepare_batch: gain_rows needs n_gain_rows");
  size_t lds = (size_t)N * N * C * sizeof(float);
  LNZ_REQUIRE(lds <= 40 * 1024, LNZ_ENOTSUP,
              "lnz_prepare_batch: N*N*C*4 = %zu B exceeds the 40 KiB staging tile", lds);
  LNZ_REQUIRE(!strips || n_strips, LNZ_EINVAL, "lnz_prepare_batch: strips need n_strips");
  if (lds <= (size_t)kPrepLds)
    hipLaunchKernelGGL(prepare_batch_union_kernel, dim3(2 * B + 1), dim3(256), 0,
                       (hipStream_t)stream, L, stride_b, stride_r, stride_c, stride_ch, N, C,
                       (float4*)Lp, mask, B, n_cu, allow_pairs, lnz_plan_wg_cap(B, n_cu), plan, n_wg,
                       K, gain_rows, n_gain_rows, n_nodes, D, V, info, ident, strips, n_strips);
  else
    hipLaunchKernelGGL(prepare_batch_kernel, dim3(2 * B + 1), dim3(256), lds, (hipStream_t)stream,
                       L, stride_b, stride_r, stride_c, stride_ch, N, C, (float4*)Lp, mask, B, n_cu,
                       allow_pairs, lnz_plan_wg_cap(B, n_cu), plan, n_wg, K, gain_rows, n_gain_rows,
                       n_nodes, D, V, info, ident, strips, n_strips);
  return lnz::check_launch("lnz_prepare_batch");
}

extern "C" int lnz_prepare_batch_prev_gains(
    const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c, int64_t stride_ch, int B,
    int N, int C, float* Lp, const uint8_t* mask, const int32_t* n_nodes, int n_cu, int allow_pairs,
    int32_t* plan, int32_t* n_wg, int K, int32_t* gain_rows, int32_t* n_gain_rows, float* D,
    float* V, uint32_t* ident, const float* D_prev, int B_prev, const int32_t* rows_prev,
    const int32_t* n_rows_prev, const int32_t* dist_host, int S, int num_layer,
    const float* mlp_pack, float* G_prev, int32_t* strips, int32_t* n_strips,
    lnz_stream_t stream) {
  LNZ_REQUIRE(L && Lp && mask && n_nodes && plan && n_wg && gain_rows && n_gain_rows && D && V &&
                  D_prev && rows_prev && n_rows_prev && dist_host && mlp_pack && G_prev && B > 0 &&
                  B_prev > 0 && C > 0 && C <= LNZ_MAX_CHANNELS && n_cu > 0 && K > 0 && num_layer > 0,
              LNZ_EINVAL, "lnz_prepare_batch_prev_gains: bad arguments (B=%d C=%d K=%d)", B, C, K);
  LNZ_REQUIRE(N > 0 && N <= LNZ_TILE, LNZ_ENOTSUP, "lnz_prepare_batch_prev_gains: N=%d > %d", N,
              LNZ_TILE);
  LNZ_REQUIRE(S >= 1 && S <= lnz_gains::SMAX, LNZ_ENOTSUP, "lnz_prepare_batch_prev_gains: S=%d", S);
  size_t lds = (size_t)N * N * C * sizeof(float);
  LNZ_REQUIRE(lds <= (size_t)kPrepLds, LNZ_ENOTSUP,
              "lnz_prepare_batch_prev_gains: N*N*C*4 = %zu B exceeds the %d B staging tile", lds,
              kPrepLds);
  lnz_gains::DistArr dist;
  for (int i = 0; i < lnz_gains::SMAX; ++i) dist.v[i] = i < S ? dist_host[i] : 0;
  const int64_t tiles = ((int64_t)B_prev * K + 31) / 32;
  const int64_t n_cons = (tiles * num_layer + 3) / 4;
  const int64_t grid = 2 * (int64_t)B + 1 + n_cons;
  LNZ_REQUIRE(grid < (1ll << 31), LNZ_ENOTSUP, "lnz_prepare_batch_prev_gains: batch too large");
  hipLaunchKernelGGL(prepare_batch_gains_kernel, dim3((unsigned)grid), dim3(256), 0,
                     (hipStream_t)stream, L, stride_b, stride_r, stride_c, stride_ch, N, C,
                     (float4*)Lp, mask, B, n_cu, allow_pairs, lnz_plan_wg_cap(B, n_cu), plan, n_wg,
                     K, gain_rows, n_gain_rows, n_nodes, D, V, dist, S, num_layer, mlp_pack, G_prev,
                     ident, (int)n_cons, D_prev, rows_prev, n_rows_prev, B_prev,
                     (strips && n_strips) ? strips : nullptr, n_strips);
  return lnz::check_launch("lnz_prepare_batch_prev_gains");
}

int lnz_launch_ritz_wg(const float* A, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                       const int32_t* n_nodes, int B, int N, int K, float* D, float* V,
                       int32_t* info, void* workspace, int64_t workspace_bytes, int flags,
                       hipStream_t s);  // lanczos_ritz_wg.hip

extern "C" int lnz_lanczos_ritz(const float* A, int64_t stride_b, int64_t stride_r,
                                int64_t stride_c, const int32_t* n_nodes, int B, int N, int K,
                                float* D, float* V, int32_t* info, lnz_stream_t stream) {
  LNZ_REQUIRE(A && n_nodes && D && V && B > 0 && N > 0 && K > 0, LNZ_EINVAL,
              "lnz_lanczos_ritz: bad arguments (B=%d N=%d K=%d)", B, N, K);
  hipStream_t s = (hipStream_t)stream;
  // 32 < N <= 192: one workgroup per graph (lanczos_ritz_wg.hip; measured 2.3x - 2.8x faster than
  // the one-wavefront kernel this file used to run for 32 < N <= 64); beyond 192 only the K-step
  // streamed kernels apply (a different function)
  if (N > 32)
    return lnz_launch_ritz_wg(A, stride_b, stride_r, stride_c, n_nodes, B, N, K, D, V, info, nullptr,
                              0, 0, s);
  hipLaunchKernelGGL(lanczos_ritz32_kernel, dim3(B), dim3(64), 0, s, A, stride_b, stride_r,
                     stride_c, n_nodes, N, K, D, V, info);
  return lnz::check_launch("lnz_lanczos_ritz");
}

// ---------------------------------------------------------------------------------------------
// R6 standalone: eigendecomposition of a batch of symmetric tridiagonal matrices (the step the
// reference leaves to LAPACK / ARPACK).  One wavefront per matrix, M <= 64; same tql2 recurrences
// as the fused kernels, eigenvectors accumulated from the identity.  Output ascending (LAPACK
// convention): R [B,M], Bm [B,M,M] with columns = eigenvectors (Bm[b][i][k] = component i of k).
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void tridiag_eigh_kernel(const double* __restrict__ diag,
                                                          const double* __restrict__ offd, int M,
                                                          double* __restrict__ R,
                                                          double* __restrict__ Bm) {
  constexpr int NMAX = 64, LD = NMAX + 2;
  __shared__ double Qt[NMAX * LD];
  __shared__ double dd[NMAX], ee[NMAX];
  __shared__ int perm[NMAX];
  const int b = blockIdx.x, lane = threadIdx.x, n = M;
  for (int idx = lane; idx < NMAX * LD; idx += 64) {
    int i = idx / LD, r = idx - i * LD;
    Qt[idx] = (i == r) ? 1.0 : 0.0;
  }
  dd[lane] = lane < n ? diag[(int64_t)b * M + lane] : 0.0;
  ee[lane] = lane < n - 1 ? offd[(int64_t)b * (M - 1) + lane] : 0.0;
  __syncthreads();
  double f = 0.0, tst1 = 0.0;
  for (int l = 0; l < n; ++l) {
    tst1 = fmax(tst1, fabs(dd[l]) + fabs(ee[l]));
    int m = l;
    while (m < n - 1 && fabs(ee[m]) > kEps * tst1) ++m;
    if (m > l) {
      int iter = 0;
      double el;
      do {
        ++iter;
        double g = dd[l];
        el = ee[l];
        double p = (dd[l + 1] - g) / (2.0 * el);
        double rr = sqrt(p * p + 1.0);
        if (p < 0) rr = -rr;
        const double dl = el / (p + rr), dl1 = el * (p + rr), hh = g - dl;
        __syncthreads();
        if (lane == 0) {
          dd[l] = dl;
          dd[l + 1] = dl1;
        }
        if (lane >= l + 2 && lane < n) dd[lane] -= hh;
        __syncthreads();
        f += hh;
        p = dd[m];
        double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
        const double el1 = ee[l + 1];
        double carry = Qt[m * LD + lane];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          const double ei = ee[i], di = dd[i];
          g = c * ei;
          const double hp = c * p;
          const double tt = fma(p, p, ei * ei);
          const double rinv = rsqrt(tt), rad = tt * rinv;
          const double e_next = s * rad;
          s = ei * rinv;
          c = p * rinv;
          p = c * di - s * g;
          const double d_next = hp + s * (c * g + s * di);
          ee[i + 1] = e_next;  // every lane stores the same value
          dd[i + 1] = d_next;
          const double z0 = Qt[i * LD + lane];
          Qt[(i + 1) * LD + lane] = s * z0 + c * carry;
          carry = c * z0 - s * carry;
        }
        Qt[l * LD + lane] = carry;
        p = -s * s2 * c3 * el1 * ee[l] / dl1;
        el = s * p;
        __syncthreads();
        if (lane == 0) {
          ee[l] = el;
          dd[l] = c * p;
        }
        __syncthreads();
      } while (fabs(el) > kEps * tst1 && iter < 60);
    }
    __syncthreads();
    if (lane == 0) {
      dd[l] = dd[l] + f;
      ee[l] = 0.0;
    }
    __syncthreads();
  }
  if (lane < n) {  // ascending order
    double di = dd[lane];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (dd[j] < di || (dd[j] == di && j < lane)) ? 1 : 0;
    perm[rank] = lane;
  }
  __syncthreads();
  if (lane < n) R[(int64_t)b * M + lane] = dd[perm[lane]];
  for (int idx = lane; idx < n * n; idx += 64) {
    int i = idx / n, k = idx - i * n;
    Bm[(int64_t)b * M * M + idx] = Qt[perm[k] * LD + i];
  }
}
}  // namespace

extern "C" int lnz_tridiag_eigh(const double* diag, const double* offdiag, int B, int M, double* R,
                                double* Bm, lnz_stream_t stream) {
  LNZ_REQUIRE(diag && R && Bm && B > 0 && M > 0 && (offdiag || M == 1), LNZ_EINVAL,
              "lnz_tridiag_eigh: bad arguments (B=%d M=%d)", B, M);
  LNZ_REQUIRE(M <= 64, LNZ_ENOTSUP, "lnz_tridiag_eigh: M=%d > 64", M);
  hipLaunchKernelGGL(tridiag_eigh_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, diag, offdiag,
                     M, R, Bm);
  return lnz::check_launch("lnz_tridiag_eigh");
}
