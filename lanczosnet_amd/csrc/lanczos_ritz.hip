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
struct Ritz32Smem {  // a VIEW (pointers in registers) onto the wavefront's dynamic LDS block
  static constexpr int LD = 34;   // doubles per basis row: b128 row reads conflict free
  double* Qt;     // [NR][LD] Krylov basis (transposed), then the Ritz vectors
  double* za;     // [32] broadcast buffers of the Lanczos steps; the eigensolver keeps d here,
  double* zb;     // [32] ... e,
  double* ca;     // [32] ... e^2
  double* cb;     // [32]
  double* dd;     // [32] eigenvalues
  double* Dw;     // [2][NR][32] twisted factorisation: lane-private columns [row][eigen lane]
  double2* de;    // [10 + 2 NR] eigenvalue search: row operands {d_i, e_{i-1}^2}, a pad row behind every block
  int* perm;      // [32]
  float* sgn;     // [32]
  int NR;         // rows of Dw = the batch's tile size N (n <= N <= 32)
  __device__ __forceinline__ double& dw(int hh, int i, int rr) const { return Dw[(hh * NR + i) * 32 + rr]; }
};

// bytes of the block for tile size N.  Layout: the small arrays, the basis, the factorisation
// columns, the search rows.  The basis is zeroed (and read) over its full 32-row extent whatever N:
// rows N..31 lie in the arrays BEHIND it, which nothing writes before the eigensolver — the block
// is never smaller than that extent.
__host__ __device__ constexpr size_t ritz32_lds_bytes(int N) {
  const size_t small_arrays = 5 * 32 * 8 + 256;
  const size_t need = (size_t)(N * Ritz32Smem::LD + 64 * N + 2 * (10 + 2 * N)) * 8;
  const size_t zeroed = (size_t)32 * Ritz32Smem::LD * 8;
  return small_arrays + (need > zeroed ? need : zeroed);
}

__device__ __forceinline__ Ritz32Smem ritz32_view(unsigned char* base, int N) {
  Ritz32Smem v;
  double* p = reinterpret_cast<double*>(base);
  v.za = p, p += 32;
  v.zb = p, p += 32;
  v.ca = p, p += 32;
  v.cb = p, p += 32;
  v.dd = p, p += 32;
  v.perm = reinterpret_cast<int*>(p);
  v.sgn = reinterpret_cast<float*>(v.perm + 32);
  p += 32;
  v.Qt = p, p += N * Ritz32Smem::LD;
  v.Dw = p, p += 64 * N;
  v.de = reinterpret_cast<double2*>(p);
  v.NR = N;
  return v;
}

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
__device__ inline double cgs2_32(const Ritz32Smem& sm, double& x, int j, int r, int h) {
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

// ---- 2. eigenvalue search.  Sturm count in product form — p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2},
// one dependent FMA per row instead of a division; the count is the number of sign changes of the
// sequence, collected as sign bits (one v_alignbit per row and probe).  ALL blocks of T are
// searched at once: the row operands {d_i, e_{i-1}^2} sit in LDS with one pad row (diagonal above
// the spectrum, no coupling: never a sign change) behind every block, lane r walks the rows of
// ITS block from P(s) on and stays on its pad row once it is through — address = min(running
// address, pad address), one integer instruction per row — so the recurrence is predicate-free
// straight-line code over ceil(longest block / 8) * 8 rows: its cost follows the molecule's size,
// not the 32-row tile, and restarts (more blocks) cost nothing.  The operands of the next eight
// rows are in flight while the current eight are worked on.  The value p_len(x) of the
// characteristic polynomial comes out of the same recurrence (renormalised every eight rows,
// exponent tracked) and drives the probe placement.
struct EigState {
  double lo, hi;    // bracket of the lane's eigenvalue
  double flo, fhi;  // p(lo), p(hi): mantissas ...
  int elo, ehi;     // ... and binary exponents
  int clo, chi;     // eigenvalues of the block below lo / below hi
  bool done, sect;
};

__device__ __forceinline__ double both_halves(double x, double& upper) {
  // -> value of the lower-half lane (returned) and of the upper-half lane, in both halves
  unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  upper = __hiloint2double((int)rh[1], (int)rl[1]);
  return __hiloint2double((int)rh[0], (int)rl[0]);
}

__device__ __forceinline__ int both_halves(int x, int& upper) {
  auto rr = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
  upper = (int)rr[1];
  return (int)rr[0];
}

// value of the neighbouring lanes (lane - 1, lane + 1) by DPP wave shifts (0 at the wave's ends)
__device__ __forceinline__ void lane_neighbours(double x, double& below, double& above) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  below = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false),   // wave_shr:1
                           __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false));
  above = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false),   // wave_shl:1
                           __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false));
}

struct SturmRows {
  double2 v[8];
};

// the eight row operands from byte address a0 on (lane-private walk, clamped to the pad row)
__device__ __forceinline__ void sturm_load(const Ritz32Smem& sm, const unsigned a0,
                                           const unsigned (&lim)[8], SturmRows& rows) {
  const char* base = reinterpret_cast<const char*>(sm.de);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const unsigned a = a0 < lim[u] ? a0 : lim[u];  // min(a0 + 16 u, pad) - 16 u
    rows.v[u] = *reinterpret_cast<const double2*>(base + a + 16 * u);
  }
}

struct SturmChain {
  double p1, p2;
  unsigned m;
  int k;
};

__device__ __forceinline__ void sturm_rows(const SturmRows& rows, const double xa, const double xb,
                                           SturmChain& A, SturmChain& B) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double pa = fma(rows.v[u].x - xa, A.p1, -(rows.v[u].y * A.p2));
    const double pb = fma(rows.v[u].x - xb, B.p1, -(rows.v[u].y * B.p2));
    A.m = __builtin_amdgcn_alignbit(A.m, (unsigned)__double2hiint(pa), 31);
    B.m = __builtin_amdgcn_alignbit(B.m, (unsigned)__double2hiint(pb), 31);
    A.p2 = A.p1, A.p1 = pa, B.p2 = B.p1, B.p1 = pb;
  }
  // keep |p| inside the exponent range: p = mantissa * 2^k, 0.5 <= |mantissa| < 1
  const int qa = __builtin_amdgcn_frexp_exp(A.p1), qb = __builtin_amdgcn_frexp_exp(B.p1);
  A.p1 = __builtin_ldexp(A.p1, -qa), A.p2 = __builtin_ldexp(A.p2, -qa), A.k += qa;
  B.p1 = __builtin_ldexp(B.p1, -qb), B.p2 = __builtin_ldexp(B.p2, -qb), B.k += qb;
}

// Sturm counts and polynomial values of the lane's block at its two probe points.  a_base: byte
// offset of the block's first row in sm.de, lim[u] = (pad row's offset) - 16 u.
__device__ __forceinline__ void sturm_pair(const Ritz32Smem& sm, const unsigned a_base,
                                           const unsigned (&lim)[8], const int maxlen,
                                           const double xa, const double xb, int& ca, int& cb,
                                           double& fa, double& fb, int& ea, int& eb) {
  SturmChain A = {1.0, 0.0, 0u, 0}, B = {1.0, 0.0, 0u, 0};
  SturmRows r0, r1;
  sturm_load(sm, a_base, lim, r0);
  if (maxlen > 8) sturm_load(sm, a_base + 128u, lim, r1);
  sturm_rows(r0, xa, xb, A, B);
  if (maxlen > 8) {
    if (maxlen > 16) sturm_load(sm, a_base + 256u, lim, r0);
    sturm_rows(r1, xa, xb, A, B);
    if (maxlen > 16) {
      if (maxlen > 24) sturm_load(sm, a_base + 384u, lim, r1);
      sturm_rows(r0, xa, xb, A, B);
      if (maxlen > 24) sturm_rows(r1, xa, xb, A, B);
    }
  }
  // sign changes between consecutive p (p_{-1} = 1 > 0)
  ca = __popc(A.m ^ (A.m >> 1)), cb = __popc(B.m ^ (B.m >> 1));
  fa = A.p1, fb = B.p1, ea = A.k, eb = B.k;
}

// Passes it0 .. 29 of the search.  Every lane carries two probe points, the two lane halves four.
// Placement:
//   pass 0    the bracket's ends and its thirds (so that every later end point carries p);
//   section   lo + k w / 5: the bracket shrinks 5x per pass;
//   isolated  (exactly one eigenvalue inside, p changes sign): the secant point x* of the two end
//             values, probes at x* -+ d1 and x* -+ d2 with d1 ~ the secant's error w^2 / gap (gap:
//             distance to the neighbouring lanes' brackets) and d2 = 16 d1: the bracket collapses
//             quadratically — the lanes of a well separated spectrum are through after 8..11
//             passes instead of 23..25; a pass that gains less than 4x is followed by a section pass.
// The brackets are always updated from the COUNTS (the values only place the probes).
// may_bail: at pass 12 (bracket width ~ 3e-9 |T|) two lanes of one block that still share a
// bracket are reported in `bail` (a numerically multiple eigenvalue, see the caller) and the
// search of their block stops; the caller resumes it with it0 = 13.
__device__ __forceinline__ void eigenvalue_search(const Ritz32Smem& sm, EigState& st, const int n,
                                                  const int r, const int h, const int s, const int t,
                                                  const unsigned a_base, const unsigned a_pad,
                                                  const int maxlen, const double gsc, const int it0,
                                                  const bool may_bail, bool& bail) {
  const bool act = r < n;
  const int jloc = r - s;
  const bool has_dn = act && r > s, has_up = act && r < t;
  unsigned lim[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) lim[u] = a_pad - 16u * u;
  bool halt = false;  // the lane's block holds a cluster: wait for the caller
  for (int it = it0; it < 30; ++it) {
    if (__all(st.done || halt)) break;
    const double lo = st.lo, hi = st.hi, w = hi - lo;
    const double tol = 4.0 * kEps * fmax(fmax(fabs(lo), fabs(hi)), 0.125 * gsc);
    // the neighbouring lanes' brackets
    double m_dn, m_up;
    lane_neighbours(0.5 * (lo + hi), m_dn, m_up);
    const bool iso = it > 0 && !st.sect && st.chi - st.clo == 1 && st.flo != 0.0 &&
                     ((__double2hiint(st.flo) ^ __double2hiint(st.fhi)) < 0);
    // section placement (pass 0: thirds and both ends)
    const double step = it == 0 ? w * (1.0 / 3.0) : w * 0.2;
    double x1 = it == 0 ? lo : lo + step, x2 = x1 + step, x3 = x2 + step, x4 = it == 0 ? hi : x3 + step;
    if (iso) {
      int de_ = st.ehi - st.elo;
      de_ = de_ < -1000 ? -1000 : (de_ > 1000 ? 1000 : de_);
      // x* = lo + w p(lo) / (p(lo) - p(hi)), the values of opposite sign
      const double fh = __builtin_ldexp(st.fhi, de_);
      const double xs = fma(w, st.flo * rcp_nr(st.flo - fh), lo);
      double g = gsc;
      g = has_dn ? fmin(g, fabs(xs - m_dn)) : g;
      g = has_up ? fmin(g, fabs(m_up - xs)) : g;
      double d1 = 0.5 * w * w * __builtin_amdgcn_rcp(fmax(g, 1e-300));
      d1 = fmin(d1, 0.125 * w);
      d1 = fmax(d1, 0.45 * tol);
      const double d2 = fmin(0.25 * w, 16.0 * d1);
      x1 = fmin(fmax(xs - d2, lo), hi), x2 = fmin(fmax(xs - d1, lo), hi);
      x3 = fmin(fmax(xs + d1, lo), hi), x4 = fmin(fmax(xs + d2, lo), hi);
    }
    int ca, cb, ea, eb;
    double fa, fb;
    sturm_pair(sm, a_base, lim, maxlen, h ? x3 : x1, h ? x4 : x2, ca, cb, fa, fb, ea, eb);
    // all four probes in both halves (count and exponent travel in one word)
    int k3, k4;
    const int k1 = both_halves((ea << 8) | ca, k3), k2 = both_halves((eb << 8) | cb, k4);
    double f3, f4;
    const double f1 = both_halves(fa, f3), f2 = both_halves(fb, f4);
    // the first probe with more than jloc eigenvalues below it closes the bracket from above
    const bool b1 = (k1 & 255) > jloc, b2 = (k2 & 255) > jloc, b3 = (k3 & 255) > jloc, b4 = (k4 & 255) > jloc;
    const bool upd = !st.done && !halt;
    const bool nl = upd && !b1, nh = upd && (b1 || b2 || b3 || b4);  // a new lower / upper end
    // lower end: the last probe in front of that one
    const bool l4 = !b1 && !b2 && !b3 && !b4, l3 = !b1 && !b2 && !b3 && b4, l2 = !b1 && !b2 && b3;
    const double nlo = l4 ? x4 : (l3 ? x3 : (l2 ? x2 : x1));
    const double nflo = l4 ? f4 : (l3 ? f3 : (l2 ? f2 : f1));
    const int nklo = l4 ? k4 : (l3 ? k3 : (l2 ? k2 : k1));
    const double nhi = b1 ? x1 : (b2 ? x2 : (b3 ? x3 : x4));
    const double nfhi = b1 ? f1 : (b2 ? f2 : (b3 ? f3 : f4));
    const int nkhi = b1 ? k1 : (b2 ? k2 : (b3 ? k3 : k4));
    st.lo = nl ? nlo : st.lo, st.flo = nl ? nflo : st.flo;
    st.clo = nl ? (nklo & 255) : st.clo, st.elo = nl ? (nklo >> 8) : st.elo;
    st.hi = nh ? nhi : st.hi, st.fhi = nh ? nfhi : st.fhi;
    st.chi = nh ? (nkhi & 255) : st.chi, st.ehi = nh ? (nkhi >> 8) : st.ehi;
    const double nw = st.hi - st.lo;
    st.sect = nw > 0.25 * w;
    // LAPACK dstebz's stopping rule: relative to |lambda| but never below ulp * |T|.  (Stopping the
    // lanes of a cluster at 1e-12 |T| was tried: 5 passes fewer, and one molecule in 30 k whose
    // cluster vectors lost orthogonality to 3e-5.)
    st.done = st.done || nw <= 4.0 * kEps * fmax(fmax(fabs(st.lo), fabs(st.hi)), 0.125 * gsc);
    if (may_bail && it == 12) {
      // Two eigenvalues of ONE block still sharing a bracket: a degenerate eigenvalue whose
      // second copy crept into the Krylov space through round-off instead of a clean breakdown
      // — the twisted vectors of such a pair would coincide.  Rare (about two molecules per
      // thousand): the caller assigns twist windows (or runs the QL sweep).
      const double lo_n = __shfl_up(st.lo, 1, 64);
      const bool cluster = has_dn && !st.done && lo_n == st.lo;
      const unsigned cm = (unsigned)__ballot(cluster && h == 0);
      if (cm) {
        bail = bail || cluster;  // per lane: shares its bracket with the lane below
        // every lane of a block with a cluster stops here (blocks are lane ranges [s, t])
        const unsigned range = (t >= 31 ? 0xffffffffu : ((2u << t) - 1u)) & ~((1u << s) - 1u);
        halt = act && (cm & range) != 0u;
      }
    }
  }
}

__device__ inline bool tridiag_eig_parallel(const Ritz32Smem& sm, const double dreg, const double ereg,
                                            const int n, const int r, const int h, long long* ts = nullptr,
                                            float* dbg = nullptr) {
  constexpr int LD = Ritz32Smem::LD;
  if (ts) ts[0] = clock64();
  // ---- d, e -> LDS (broadcast reads); negligible couplings split the matrix
  const double dn = __shfl_down(dreg, 1, 64);
  const bool live = r < n - 1 && fabs(ereg) > kEps * (fabs(dreg) + fabs(dn));
  const double e_live = live ? fabs(ereg) : 0.0;
  const double e_prev = __shfl_up(e_live, 1, 64);
  const bool act = r < n;
  // Gershgorin bound of the spectrum: one row per lane, wave maximum
  double gmax = act ? fabs(dreg) + ((r & 31) > 0 ? e_prev : 0.0) + e_live : 0.0;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, off, 64));
  const double gsc = gmax > 0.0 ? gmax : 1.0;
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
  // row i of T sits at entry P(i) = 8 + i + (blocks in front of it) of sm.de, the pad row of a
  // block behind its last row (see eigenvalue_search)
  const unsigned nlow = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
  auto P = [&](int i) { return 8 + i + __popc(cut & nlow & ((1u << i) - 1u)); };
  const int last_pad = P(n - 1) + 1;
  if (h == 0 && act) {
    sm.za[r] = dreg;
    sm.zb[r] = live ? ereg : 0.0;
    sm.ca[r] = live ? ereg * ereg : 0.0;
    const int p = P(r);
    sm.de[p] = make_double2(dreg, (r & 31) > 0 ? e_prev * e_prev : 0.0);
    if ((cut >> r) & 1u) sm.de[p + 1] = make_double2(2.0 * gsc + 1.0, 0.0);
  }
  __syncthreads();
  const unsigned a_base = 16u * (unsigned)(act ? P(s) : last_pad);
  const unsigned a_pad = 16u * (unsigned)(act ? P(t) + 1 : last_pad);
  // longest block (wave uniform): the loop bound of the recurrences below
  int maxlen = 1;
  for (int sb = 0; sb < n;) {
    const unsigned above = cut & ~((1u << sb) - 1u);
    int tb = above ? __ffs(above) - 1 : n - 1;
    tb = tb > n - 1 ? n - 1 : tb;
    maxlen = tb - sb + 1 > maxlen ? tb - sb + 1 : maxlen;
    sb = tb + 1;
  }
  EigState st;
  st.lo = -gsc, st.hi = gsc, st.flo = st.fhi = 0.0, st.elo = st.ehi = 0, st.clo = 0, st.chi = t - s + 1;
  st.done = !act, st.sect = true;
  if (act && t == s) st.lo = st.hi = dreg, st.done = true;  // a 1 x 1 block: its diagonal entry
  bool bail = false;
  eigenvalue_search(sm, st, n, r, h, s, t, a_base, a_pad, maxlen, gsc, 0, true, bail);
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
    // An eigenvalue of the block within 1e-7 |T| of a member belongs to the cluster as well.  The
    // window of a member is chosen by its RANK inside the widened bracket, which counts every
    // eigenvalue in there: a simple eigenvalue 5e-9 next to a double one (r05 fuzz: two molecules
    // in 65 k) took a rank without taking part, one member got ITS window and the two vectors
    // came out equal.  Three rounds reach a chain of four.
    {
      const double midv = 0.5 * (st.lo + st.hi);
      for (int rep = 0; rep < 3; ++rep) {
        const double m_dn = __shfl_up(midv, 1, 64), m_up = __shfl_down(midv, 1, 64);
        const int mem_dn = __shfl_up(member ? 1 : 0, 1, 64), mem_up = __shfl_down(member ? 1 : 0, 1, 64);
        const bool join = act && ((r > s && mem_dn != 0 && fabs(midv - m_dn) <= 1e-7 * gsc) ||
                                  (r < t && mem_up != 0 && fabs(m_up - midv) <= 1e-7 * gsc));
        member = member || join;
      }
    }
    // eigenvalues of rows a..b (no outside coupling) below xa / below xb: the two recurrences
    // side by side (each is one chain of reciprocals: latency bound)
    auto count2 = [&](double xa, double xb, int a, int b, int& na, int& nb) {
      na = nb = 0;
      double qa = 1.0, qb = 1.0;
      for (int i = a; i <= b; ++i) {
        const double e2p = i > a ? sm.ca[i - 1] : 0.0, di = sm.za[i];
        qa = fma(-e2p, rcp_nr(qa), di - xa);
        qb = fma(-e2p, rcp_nr(qb), di - xb);
        qa = fabs(qa) < 1e-290 ? -1e-290 : qa;
        qb = fabs(qb) < 1e-290 ? -1e-290 : qb;
        na += qa < 0.0 ? 1 : 0;
        nb += qb < 0.0 ? 1 : 0;
      }
    };
    bool ok = true;
    if (member && act) {
      const double cutoff = 1e-3 * gsc, wd = 1e-5 * gsc;
      const double xl = st.lo - wd, xh = st.hi + wd;
      int below_l, below_h;
      count2(xl, xh, s, t, below_l, below_h);
      int g = (r - s) - below_l;  // rank inside the widened bracket
      int a = s;
      bool found = false;
      for (int i = s; i <= t && !found; ++i) {
        if (i == t || fabs(sm.zb[i]) <= cutoff) {
          int cl, ch;
          count2(xl, xh, a, i, cl, ch);
          const int inside = ch - cl;
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
    eigenvalue_search(sm, st, n, r, h, s, t, a_base, a_pad, maxlen, gsc, 13, false, bail2);
  }
  const double lam = 0.5 * (st.lo + st.hi);
  if (ts) ts[1] = clock64();
  // ---- 3. eigenvector of the block: twisted factorisation of T - lam, both lane halves at work.
  //      Half 0 runs the stationary transform top down, D+_{i+1} = (d_{i+1} - lam) - e_i^2 / D+_i,
  //      half 1 the progressive one bottom up, D-_i = (d_i - lam) - e_i^2 / D-_{i+1} — the same
  //      instruction stream on mirrored rows; both park their pivots in the lane's LDS columns
  //      Dw[0] / Dw[1] ([row][eigen lane]: conflict free).  gamma_i = D+_i + D-_i - (d_i - lam);
  //      twist where |gamma| is smallest; then half 0 solves upwards from the twist index, half 1
  //      downwards, and the vector replaces D+ in Dw[0].  Loop bounds: the longest block.
  const double tiny = kEps * gsc;
  const int len = act ? t - s : -1;
  {
    // pivot chains: one reciprocal per row is the critical path (rcp, Newton step, FMA, clamp);
    // the row operands are read one step ahead, the store sits behind the chain
    double Dprev = 1.0;
    const int i0 = len >= 0 ? (h ? t : s) : 0, step = h ? -1 : 1;
    double dcur = sm.za[i0], ecur = 0.0;  // (no coupling into the first row of the sweep)
#pragma unroll 2
    for (int k = 0; k < maxlen; ++k) {
      const bool in = k <= len;
      const int i = i0 + step * (in ? k : 0);
      const int i_n = i0 + step * (k + 1 <= len ? k + 1 : 0);
      const double dnx = sm.za[i_n], enx = sm.ca[h ? i_n : (i_n > 0 ? i_n - 1 : 0)];
      double dnew = fma(-ecur, rcp_nr(Dprev), dcur - lam);
      dnew = fabs(dnew) < tiny ? (dnew < 0.0 ? -tiny : tiny) : dnew;
      Dprev = in ? dnew : Dprev;
      if (in) sm.dw(h, i, r) = dnew;
      dcur = dnx, ecur = enx;
    }
  }
  __syncthreads();
  if (ts) ts[4] = clock64();
  int tw = s;
  {
    // gamma over the rows of the window, the halves on alternate rows, four rows in flight
    double gbest = 1e300;
    for (int k0 = 0; k0 < maxlen; k0 += 8) {
      double g[4];
      int ii[4];
      bool inn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 2 * u + h;
        ii[u] = s + k;
        inn[u] = k <= len && ii[u] >= ws && ii[u] <= wt;
        const int ic = inn[u] ? ii[u] : 0;
        g[u] = fabs((sm.dw(0, ic, r) + sm.dw(1, ic, r)) - (sm.za[ic] - lam));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // ties: the highest index (the order of a downward scan with a strict comparison)
        const bool take = inn[u] && (g[u] < gbest || (g[u] == gbest && ii[u] > tw));
        gbest = take ? g[u] : gbest;
        tw = take ? ii[u] : tw;
      }
    }
    double g_up;
    int t_up;
    const double g_lo = both_halves(gbest, g_up);
    const int t_lo = both_halves(tw, t_up);
    const bool take_up = g_up < g_lo || (g_up == g_lo && t_up > t_lo);
    tw = take_up ? t_up : t_lo;
    tw = act ? tw : 0;
  }
  __syncthreads();  // (every gamma has been read before the vectors overwrite D+)
  if (ts) ts[5] = clock64();
  double sc;
  {
    // half 0: z_i = -(e_i / D+_i) z_{i+1} upwards from the twist index; half 1:
    // z_i = -(e_{i-1} / D-_i) z_{i-1} downwards.  The multipliers do not depend on the chain:
    // four rows at a time, loads and reciprocals issued together, branch free; the chain itself
    // is one multiplication per row.
    double zprev = 1.0, nn = h ? 0.0 : 1.0;
    if (act && h == 0) sm.dw(0, tw, r) = 1.0;
    for (int k0 = 1; k0 < maxlen; k0 += 4) {
      double Dv[4], ev[4];
      int ix[4];
      bool inx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = h ? tw + k0 + u : tw - k0 - u;
        inx[u] = act && (h ? i <= t : i >= s);
        ix[u] = inx[u] ? i : 1;
        Dv[u] = sm.dw(h, ix[u], r);
        ev[u] = sm.zb[h ? ix[u] - 1 : ix[u]];
      }
      double mul[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) mul[u] = -(ev[u] * rcp_nr(Dv[u]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double znew = mul[u] * zprev;
        zprev = inx[u] ? znew : zprev;
        const double zz = inx[u] ? znew : 0.0;
        nn = fma(zz, zz, nn);
        if (inx[u]) sm.dw(0, ix[u], r) = znew;
      }
    }
    nn = xhalf_sum(nn);
    sc = act ? rsqrt(nn) : 0.0;
  }
  __syncthreads();

  if (__any(member)) {
    // cluster lanes: modified Gram-Schmidt in lane order, on the LDS columns.  pos = how many
    // consecutive lower lanes belong to the same cluster; the lane at position p is
    // orthogonalised against the (final) vectors of the p lanes below it.
    if (member && act) {  // normalised column, zero outside the block (both halves write the same)
      for (int i = 0; i < n; ++i) sm.dw(0, i, r) = (i >= s && i <= t) ? sm.dw(0, i, r) * sc : 0.0;
      sc = 1.0;
    }
    __syncthreads();
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
      if (pos == p) {
        for (int d = 1; d <= p; ++d) {
          double dot = 0.0;
#pragma unroll 4
          for (int i = 0; i < n; ++i) dot = fma(sm.dw(0, i, r), sm.dw(0, i, r - d), dot);
#pragma unroll 4
          for (int i = 0; i < n; ++i) sm.dw(0, i, r) = fma(-dot, sm.dw(0, i, r - d), sm.dw(0, i, r));
        }
        double nn = 0.0;
#pragma unroll 4
        for (int i = 0; i < n; ++i) nn = fma(sm.dw(0, i, r), sm.dw(0, i, r), nn);
        const double s2 = nn > 0.0 ? rsqrt(nn) : 1.0;
#pragma unroll 4
        for (int i = 0; i < n; ++i) sm.dw(0, i, r) *= s2;
      }
      __syncthreads();
    }
  }

#ifdef LNZ_RITZ_DEBUG  // per-lane eigensolver state of molecule LNZ_RITZ_DEBUG -> dbg (16 floats per lane)
  if (dbg && act && h == 0) {
    float* o = dbg + 16 * r;
    o[0] = (float)lam, o[1] = (float)((lam - (double)(float)lam) * 1e9), o[2] = (float)s, o[3] = (float)t;
    o[4] = member ? 1.0f : 0.0f, o[5] = (float)ws, o[6] = (float)wt, o[7] = (float)tw;
    o[8] = (float)((st.hi - st.lo) / gsc * 1e12), o[9] = (float)st.clo, o[10] = (float)st.chi, o[11] = bail ? 1.0f : 0.0f;
    o[12] = (float)dreg, o[13] = (float)(dreg - (double)(float)dreg);
    o[14] = (float)ereg, o[15] = (float)(ereg - (double)(float)ereg);
    o[1] = (float)(lam - (double)(float)lam);
  }
#endif
  if (ts) ts[2] = clock64();
  // ---- 4. V = Q S: column k, node rows 16h .. 16h+15
  double acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.0;
#pragma unroll 2
  for (int i = 0; i < n; ++i) {
    double q[16];
    load16(&sm.Qt[i * LD + 16 * h], q);
    const double zi = (act && i >= s && i <= t) ? sm.dw(0, i, r) : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = fma(q[u], zi, acc[u]);
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int u = 0; u < 16; ++u) sm.Qt[r * LD + 16 * h + u] = acc[u] * sc;
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
    const Ritz32Smem& sm) {
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
  // zero the basis (rows beyond the current step must read as zero) over its full 32-row extent:
  // rows NR..31 lie in the factorisation columns behind it, which nothing writes before the
  // eigensolver (8.7 KB < the block's size for every N >= 1)
  for (int idx = lane; idx < 32 * LD / 2; idx += 64)
    reinterpret_cast<double2*>(sm.Qt)[idx] = make_double2(0.0, 0.0);
  __syncthreads();
#ifdef LNZ_PROFILE_PHASES
  long long tp0 = clock64(), tp1 = tp0, tp2 = tp0;
  long long tsx[6] = {0, 0, 0, 0, 0, 0};
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
      double beta, u, binv;
      for (;;) {
        // ---- one broadcast of w: beta = |w| and u = A w
        sm.za[r] = w;
        __syncthreads();
        double v[16];
        load16(sm.za + 16 * h, v);
        double nn = xhalf_sum(dot16(v, v));
        u = xhalf_sum(dot16(arow, v));
        {
          // 1 / sqrt(nn) by the hardware seed + two Newton steps, beta = nn / sqrt(nn): one chain
          // of 9 instructions instead of a square root and a division (about 25)
          double y = __builtin_amdgcn_rsq(nn);
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const double hy = 0.5 * y;
            const double er = fma(-(nn * y), hy, 0.5);
            y = fma(y, er, y);
          }
          binv = y;
          beta = nn > 0.0 ? nn * y : 0.0;
        }
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
#ifdef LNZ_RITZ_DEBUG  // (the caller's info buffer holds B ints + 32 x 16 floats)
    float* dbg_ = b == LNZ_RITZ_DEBUG ? reinterpret_cast<float*>(info) + gridDim.x : nullptr;
    const bool solved = tridiag_eig_parallel(sm, dreg, ereg, n, r, h, nullptr, dbg_);
#elif defined(LNZ_PROFILE_PHASES)
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
    {
      // sign convention: the entry of largest magnitude is positive (the first one on ties).  Lane
      // (k, h) scans nodes 16h .. 16h+15 of Ritz vector k from registers; the halves combine.
      const bool on = r < kk;
      double v[16];
      load16(&sm.Qt[(on ? sm.perm[r] : 0) * LD + 16 * h], v);
      double best = 0.0;
      float sg = 1.0f;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const double av = fabs(v[x]);
        const bool take = av > best;
        sg = take ? (v[x] < 0 ? -1.0f : 1.0f) : sg;
        best = take ? av : best;
      }
      double best_up;
      const double best_lo = both_halves(best, best_up);
      int sg_up;
      const int sg_lo = both_halves(__float_as_int(sg), sg_up);
      if (on && h == 0) sm.sgn[r] = __int_as_float(best_up > best_lo ? sg_up : sg_lo);
    }
    __syncthreads();
  }

  for (int k = lane; k < K; k += 64) D[(int64_t)b * K + k] = k < kk ? (float)sm.dd[sm.perm[k]] : 0.0f;
  float* Vb = V + (int64_t)b * N * K;
  if (K <= 64) {
    // lane -> (node row of the pass, slot): the slot is fixed per lane, 64 / K node rows per pass
    // (K = 20: three rows = 240 contiguous bytes per store)
    const int rpp = 64 / K;
    const int rr0 = lane / K, k = lane - rr0 * K;
    const bool lane_on = rr0 < rpp, col_on = lane_on && k < kk;
    const double* col = &sm.Qt[(col_on ? sm.perm[k] : 0) * LD];
    const float sg = col_on ? sm.sgn[k] : 0.0f;
#pragma unroll 4
    for (int r0 = 0; r0 < N; r0 += rpp) {
      const int rr = r0 + rr0;
      if (lane_on && rr < N) Vb[rr * K + k] = (col_on && rr < n) ? sg * (float)col[rr] : 0.0f;
    }
  } else {
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
    D[(int64_t)b * K + 7] = (float)(tsx[4] - tsx[1]);   // pivot sweeps
    D[(int64_t)b * K + 8] = (float)(tsx[5] - tsx[4]);   // twist index
  }
#endif
}

#ifdef LNZ_RITZ_VGPR160   // (probe: a wave that fits next to two 176-register forward waves on a SIMD)
#define LNZ_RITZ32_ATTR __attribute__((amdgpu_num_vgpr(80)))
#else
#define LNZ_RITZ32_ATTR
#endif
__global__ __launch_bounds__(64) LNZ_RITZ32_ATTR void lanczos_ritz32_kernel(
    const float* __restrict__ A, int64_t sb, int64_t sr, int64_t sc,
    const int32_t* __restrict__ n_nodes, int N, int K, float* __restrict__ D,
    float* __restrict__ V, int32_t* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ubuf[];
  lanczos_ritz32_body(A, sb, sr, sc, n_nodes, N, K, D, V, info, blockIdx.x, threadIdx.x,
                      ritz32_view(ubuf, N));
}

// Everything the fused forward needs from a collated batch besides the gains, in ONE launch of
// 256-thread workgroups: workgroup 0 plans the batch (tile plan + live eigen slots + strips),
// workgroups 1..B are the Lanczos / eigensolve wavefronts (one live wave each — the other three
// exit at once, and a terminated wave does not take part in barriers), workgroups B+1..2B pack
// the Laplacian tiles.  The Ritz wavefronts are the long pole (latency bound, one wave per SIMD);
// dispatched first, they leave most of the machine idle, and the two byte movers run in that
// shadow instead of in front.
// ONE dynamic LDS block per workgroup, used according to its role (Ritz scratch, pack staging
// tile, planner scratch): max of the three — QM8 (N = 26, C = 7): 24.5 KB, six workgroups per
// compute unit.  168 registers (three wave slots per SIMD): next to the four resident Ritz
// wavefronts of a compute unit, two pack workgroups run at a time (with the 248 registers of the
// r04 body it was one, and the B pack workgroups trickled through behind the Ritz waves).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
void prepare_batch_union_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int N, int C,
    float4* __restrict__ Lp, const uint8_t* __restrict__ mask, int B, int n_cu, int allow_pairs,
    int wg_cap, int32_t* __restrict__ plan, int32_t* __restrict__ n_wg, int K,
    int32_t* __restrict__ gain_rows, int32_t* __restrict__ n_gain_rows,
    const int32_t* __restrict__ n_nodes, float* __restrict__ D, float* __restrict__ V,
    int32_t* __restrict__ info, uint32_t* __restrict__ ident, int32_t* __restrict__ strips,
    int32_t* __restrict__ n_strips) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ubuf[];
  const int blk = blockIdx.x;
#ifdef LNZ_PREP_X_NOPLAN   // probes of the launch's parts (tools/experiments/prep_parts.py)
  if (blk < 2) return;
#endif
#ifdef LNZ_PREP_X_EMPTYPACK
  if (blk > B + 1) return;
#endif
  if (blk < 2) {
    // The two planners — tile plan + live eigen slots (workgroup 0), strip plan (workgroup 1) —
    // are independent serial chains of scans over the mask.  As ONE workgroup behind each other,
    // sharing its compute unit with four Ritz wavefronts, they were the long pole of the launch
    // (r05 probes, B = 1024: 92 us with the planner, 75 without).  Side by side and at raised issue
    // priority they finish under the Ritz wavefronts.
    __builtin_amdgcn_s_setprio(3);
    if (blk == 0)
      plan_tiles_body(mask, B, N, n_cu, allow_pairs, wg_cap, plan, n_wg, K, gain_rows, n_gain_rows);
    else if (strips)
      plan_strips_body(mask, B, N, n_cu, strips, n_strips, ubuf);
  } else if (blk <= B + 1) {
    if (threadIdx.x >= 64) return;
    // channel 0 of L is the simple-graph Laplacian the Ritz pairs belong to (dataset/qm8.py:262)
    lanczos_ritz32_body(L, sb, sr, sc, n_nodes, N, K, D, V, info, blk - 2, threadIdx.x,
                        ritz32_view(ubuf, N));
  } else {
    pack_laplacian_body(L, sb, sr, sc, sch, N, C, Lp, reinterpret_cast<float*>(ubuf), blk - 2 - B,
                        ident);
  }
}

// dynamic LDS of the preparation launches: the largest of the three roles' blocks
static size_t prep_lds_bytes(int N, int C) {
  size_t lds = (size_t)N * N * C * sizeof(float);
  lds = lds > ritz32_lds_bytes(N) ? lds : ritz32_lds_bytes(N);
  return lds > (size_t)kStripScratch ? lds : (size_t)kStripScratch;
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
  // One dynamic LDS block per workgroup, used according to its role (Ritz scratch, pack staging
  // tile, planner scratch): with a separate tile every workgroup carried 32 KB and the four
  // resident Ritz workgroups of a CU left room for ONE more — the pack workgroups trickled
  // through and the consumers started 50 us late.
  extern __shared__ __attribute__((aligned(16))) unsigned char ubuf[];
  const Ritz32Smem sm = ritz32_view(ubuf, N);
  float* tile = reinterpret_cast<float*>(ubuf);
  const int blk = blockIdx.x;
  if (blk < 2) {
    // the two planners side by side (see prepare_batch_union_kernel)
    __builtin_amdgcn_s_setprio(3);
    if (blk == 0)
      plan_tiles_body(mask, B, N, n_cu, allow_pairs, wg_cap, plan, n_wg, K, gain_rows, n_gain_rows);
    else if (strips)
      plan_strips_body(mask, B, N, n_cu, strips, n_strips, ubuf);
  } else if (blk <= B + 1) {
    if (threadIdx.x >= 64) return;
    // the Lanczos / eigensolve chain is the critical path of the launch: it wins every issue
    // arbitration against the gains wave that shares its SIMD
    __builtin_amdgcn_s_setprio(3);
    lanczos_ritz32_body(L, sb, sr, sc, n_nodes, N, K, D, V, nullptr, blk - 2, threadIdx.x, sm);
  } else {
    // behind the Ritz workgroups: the gains workgroups, then the pack (needed by the next launch
    // only).  Alternating the two measured slower (229 vs 182 us).
    const int p = blk - B - 2;
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
  LNZ_REQUIRE(L && mask && n_nodes && plan && n_wg && D && V && B > 0 && C > 0 &&
                  C <= LNZ_MAX_CHANNELS && n_cu > 0 && K > 0,
              LNZ_EINVAL, "lnz_prepare_batch: bad arguments (B=%d C=%d K=%d)", B, C, K);
  LNZ_REQUIRE(Lp || !ident, LNZ_EINVAL, "lnz_prepare_batch: ident needs Lp (the pack writes both)");
  LNZ_REQUIRE(N > 0 && N <= LNZ_TILE, LNZ_ENOTSUP, "lnz_prepare_batch: N=%d > %d", N, LNZ_TILE);
  LNZ_REQUIRE(!gain_rows || n_gain_rows, LNZ_EINVAL, "lnz_prepare_batch: gain_rows needs n_gain_rows");
  const size_t tile = (size_t)N * N * C * sizeof(float);
  LNZ_REQUIRE(tile <= 40 * 1024, LNZ_ENOTSUP,
              "lnz_prepare_batch: N*N*C*4 = %zu B exceeds the 40 KiB staging tile", tile);
  LNZ_REQUIRE(!strips || n_strips, LNZ_EINVAL, "lnz_prepare_batch: strips need n_strips");
  // without Lp: plan + Ritz pairs only (no pack workgroups in the grid)
  hipLaunchKernelGGL(prepare_batch_union_kernel, dim3(Lp ? 2 * B + 2 : B + 2), dim3(256),
                     Lp ? prep_lds_bytes(N, C) : prep_lds_bytes(N, 0),
                     (hipStream_t)stream, L, stride_b, stride_r, stride_c, stride_ch, N, C,
                     (float4*)Lp, mask, B, n_cu, allow_pairs, lnz_plan_wg_cap(B, n_cu), plan, n_wg,
                     K, gain_rows, n_gain_rows, n_nodes, D, V, info, ident, strips, n_strips);
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
  const size_t tile = (size_t)N * N * C * sizeof(float);
  LNZ_REQUIRE(tile <= 40 * 1024, LNZ_ENOTSUP,
              "lnz_prepare_batch_prev_gains: N*N*C*4 = %zu B exceeds the 40 KiB staging tile", tile);
  lnz_gains::DistArr dist;
  for (int i = 0; i < lnz_gains::SMAX; ++i) dist.v[i] = i < S ? dist_host[i] : 0;
  const int64_t tiles = ((int64_t)B_prev * K + 31) / 32;
  const int64_t n_cons = (tiles * num_layer + 3) / 4;
  const int64_t grid = 2 * (int64_t)B + 2 + n_cons;
  LNZ_REQUIRE(grid < (1ll << 31), LNZ_ENOTSUP, "lnz_prepare_batch_prev_gains: batch too large");
  hipLaunchKernelGGL(prepare_batch_gains_kernel, dim3((unsigned)grid), dim3(256),
                     prep_lds_bytes(N, C), (hipStream_t)stream, L, stride_b, stride_r, stride_c, stride_ch, N, C,
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
  hipLaunchKernelGGL(lanczos_ritz32_kernel, dim3(B), dim3(64), ritz32_lds_bytes(N), s, A, stride_b,
                     stride_r, stride_c, n_nodes, N, K, D, V, info);
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
