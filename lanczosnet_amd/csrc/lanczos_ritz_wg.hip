// R2 + R6 for graphs of 33..192 nodes: full-length Lanczos -> tridiagonal eigensolve -> Ritz
// select, ONE WORKGROUP (8 wavefronts) per graph.  This is the regime of the reference's own
// synthetic-graph configuration (config/graph_lanczos_net.yaml with dataset/get_graph_data.py:15-49:
// n in [20, 100]; (D, V) = np.linalg.eigh + |lambda| sort, utils/data_helper.py:197-223) and of
// SURVEY.md §8(d)'s "A fits on chip" band (N <= ~180).
//
// Same algorithm as lanczos_ritz.hip (fp64; m = n steps; classical Gram-Schmidt twice against ALL
// previous vectors; restart from the unit vector of largest residual on breakdown; implicit-shift
// QL with the rotations applied to the basis so it ends up holding V = Q B; |lambda| ordering and
// sign convention) — what changes is where things live and who works on them:
//   * A (fp32, n x n) is staged ONCE into LDS: HBM traffic per graph stays the algorithmic
//     4 n^2 (A) + 4 K (D) + 4 N K (V) bytes;
//   * the Krylov basis Qt[i][r] = q_i[r] (fp64, row pitch N|1 doubles) lives in LDS next to A while
//     12 N (N|1) + 17 KB <= 158 KB, i.e. N <= 108 (every graph of the reference generator); for
//     108 < N <= 192 A alone takes up to 148 KB and the basis moves to a caller-provided workspace
//     (N (N|1) 8 bytes per graph, L2 / MALL resident: it is written and re-read by the one CU that
//     owns the graph) — template parameter QG;
//   * basis in LDS: the Lanczos phase runs on FOUR of the eight waves (lanczos_waves: row groups x
//     parts, one row of the residual per lane, every inner product a software-pipelined walk over
//     LDS, four or five workgroup barriers per step) while the others keep the barriers company;
//   * basis in the workspace (and flags bit 2): every reduction of a step (A w, the j+1
//     Gram-Schmidt dot products, the update w -= Q c) is split over all 512 threads as (output
//     row) x (segment of the reduction range); partials are combined through LDS in a fixed
//     order.  In both forms alpha / beta are bit-identical in every thread and the breakdown /
//     restart control flow stays workgroup-uniform.  No atomics in the recurrences;
//   * QL runs barrier-free: thread r owns element r of every basis vector (the rotation of rows
//     i, i+1 touches only its own two words), each wavefront carries a PRIVATE copy of T's diagonal
//     and off-diagonal (in the then dead A region) and runs the scalar recurrences redundantly.
#include "common.hpp"

namespace {

#ifndef LNZ_WG_CGS_AGAIN_WAVES
#define LNZ_WG_CGS_AGAIN_WAVES 0.99   // the same for the wave-level form (n <= 128)
#endif
#ifndef LNZ_WG_CGS_AGAIN
// Second Gram-Schmidt pass when the first removed more than this part of |x|^2.  The eight-wave form
// (n > 128) used 0.99 until r05: forests of equal stars with 134..192 nodes — massively degenerate
// spectra, Ritz values converge within a few steps — then lost orthogonality geometrically (a factor
// ~3 per step from 1e-16 on: <q_40, q_0> = 1e-4, garbage after the first near-breakdown) and half of
// them came back with non-orthonormal V (tools/experiments/ritz_wg_forest.py).  0.5 is the classical
// "twice is enough" threshold, applied beyond 128 nodes (+ 30 % on those launches); up to 128 nodes
// both forms pass the same graphs at 0.99 (1,536 forests of 100..128 nodes) and a tighter rule
// would cost 22 % (the reference's graph configuration lives there).
#define LNZ_WG_CGS_AGAIN 0.5
#endif
#ifndef LNZ_WG_BREAKDOWN_TOL
#define LNZ_WG_BREAKDOWN_TOL 1e-8
#endif
constexpr double kBreakdownTol = LNZ_WG_BREAKDOWN_TOL;  // see lanczos_ritz.hip
constexpr double kEps = 2.220446049250313e-16;
#ifndef LNZ_RITZ_WG_THREADS
#define LNZ_RITZ_WG_THREADS 512
#endif
constexpr int kNT = LNZ_RITZ_WG_THREADS;   // threads per workgroup (the reductions of a Lanczos step are split over all of them)
constexpr int kWaves = kNT / 64;
constexpr int kNMax = 192;   // largest graph one workgroup owns
#ifndef LNZ_RITZ_EIG_THREADS
#define LNZ_RITZ_EIG_THREADS 512   // threads of the section search: P = this / n probe pairs per eigenvalue
#endif
#ifndef LNZ_RITZ_PARTS_SMALL
#define LNZ_RITZ_PARTS_SMALL 4   // parts per row group of the wave-level Lanczos phase, n <= 64
#endif
#ifndef LNZ_RITZ_PARTS_LARGE
#define LNZ_RITZ_PARTS_LARGE 2   // ... 64 < n <= 128
#endif
#ifndef LNZ_RITZ_ONE_WAVE_MAX
#define LNZ_RITZ_ONE_WAVE_MAX 128   // graphs up to this size (basis in LDS) run their Lanczos phase on one or two wavefronts
#endif
// LDS a workgroup may ask for: the CU has 160 KB; a request of 163,712 B was refused by the runtime
// (HSA_STATUS_ERROR_INVALID_ALLOCATION) where 163,020 B had launched — keep 2 KB clear
constexpr int kLdsMax = 160 * 1024 - 2048;

struct LwExtra {   // wave-level Lanczos phase (lanczos_waves; basis in LDS only): what does not fit in WgFixed::part
  double pd[256];      // partial updates [part][row]
  double cbx[3][128];  // the coefficient copies of waves 1..3 (wave 0: WgFixed::cb)
  double pn[2], px[2]; // partial |w|^2, |x|^2 by row group
  double best[4];      // restart: largest residual by wave
  int cand[4];
};
static_assert(sizeof(LwExtra) % 16 == 0, "the basis behind it starts on a 16-byte boundary");

struct WgFixed {  // fixed part of the LDS block
  double zb[kNMax];    // broadcast of the current vector
  double cb[kNMax];    // Gram-Schmidt coefficients
  double part[kNT];    // (segment, output) partial sums
  double dd[kNMax];    // T diagonal / eigenvalues
  double ee[kNMax];    // T off-diagonal
  double pn[32];       // partial squared norms (one per reduction segment)
  double pc[kWaves];   // per-wave sums of squared Gram-Schmidt coefficients
  int perm[kNMax];
  float sgn[kNMax];
};

// Floats per row of the staged A.  Eight-wave Lanczos phase: odd ("thread i reads row i" is conflict
// free for single words).  One-wave phase (basis in LDS, n <= 64): the lanes read their rows four
// columns at a time — a multiple of four that is not one of eight (sixteen lanes' ds_read_b128 then
// cover every bank once).
__host__ __device__ inline int wg_a_pitch4(int N) {
  const int p = (N + 3) & ~3;
  return (p & 7) ? p : p + 4;
}

__host__ __device__ inline size_t wg_a_bytes(int N, bool qg) {
  size_t a = (size_t)N * (size_t)(N | 1) * sizeof(float);
  if (!qg) {
    const size_t a1 = (size_t)(N < LNZ_RITZ_ONE_WAVE_MAX ? N : LNZ_RITZ_ONE_WAVE_MAX) * wg_a_pitch4(N) * sizeof(float);
    a = a < a1 ? a1 : a;
  }
  const size_t ql = (size_t)kWaves * 2 * (size_t)N * sizeof(double);  // QL's per-wave (d, e) copies
  a = a < ql ? ql : a;
  // the eigenvalue search: T's (d, e, e^2), the probes' polynomial values, the brackets' midpoints
  const size_t hs = (size_t)4 * N * sizeof(double) + 2 * LNZ_RITZ_EIG_THREADS * sizeof(float);
  a = a < hs ? hs : a;
  return (a + 15) & ~(size_t)15;
}

inline size_t wg_lds_bytes(int N, bool qg) {
  // (+ 8: A starts on a 16-byte boundary behind an odd number of basis doubles)
  return sizeof(WgFixed) + (qg ? 0 : sizeof(LwExtra) + (size_t)N * (size_t)(N | 1) * sizeof(double) + 8) +
         wg_a_bytes(N, qg);
}

__device__ __forceinline__ double rcp_nr(double x) {  // 1/x: hardware seed + one Newton step
  double y = __builtin_amdgcn_rcp(x);
  return fma(y, fma(-x, y, 1.0), y);
}

// The unreduced block [bs, bt] of T that row e belongs to, from the bit mask of the rows whose
// coupling to the next row is zero (three 64-bit words, one ballot each) — instead of every thread
// walking the off-diagonal through LDS one dependent read at a time (~12 k cycles at n = 100).
__device__ __forceinline__ void block_bounds(const unsigned long long* cutw, const int e, const int n,
                                             int& bs, int& bt) {
  bt = n - 1;
  bs = 0;
  {
    int w = e >> 6;
    unsigned long long m = cutw[w] & (~0ull << (e & 63));
    for (;;) {
      if (m) {
        bt = 64 * w + __ffsll((long long)m) - 1;
        break;
      }
      if (++w >= 3) break;
      m = cutw[w];
    }
  }
  {
    int w = e >> 6;
    unsigned long long m = (e & 63) ? cutw[w] & (~0ull >> (64 - (e & 63))) : 0ull;
    for (;;) {
      if (m) {
        bs = 64 * w + 64 - __clzll((long long)m);
        break;
      }
      if (--w < 0) break;
      m = cutw[w];
    }
  }
}

// Eigenvalues of rows s..t of T (an unreduced block: no coupling to its neighbours) below x — the
// Sturm count in product form, p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}: one dependent FMA per
// row instead of a division (LAPACK dstebz's recurrence); the count is the number of sign changes.
// Two probe points per call so that two independent chains are in flight.
__device__ __forceinline__ void sturm2(const double* __restrict__ d, const double* __restrict__ e2,
                                       const int s, const int t, const double xa, const double xb,
                                       int& ca, int& cb, float& fa, float& fb, int& ea, int& eb) {
  // fa * 2^ea, fb * 2^eb: the value p_len(x) of the block's characteristic polynomial at the two
  // probes (the recurrence is renormalised every eight rows, the exponent tracked) — it places
  // the probes of the passes behind the isolation of the eigenvalue (see the search loop)
  double a1 = 1.0, a2 = 0.0, b1 = 1.0, b2 = 0.0;
  unsigned na = 0u, nb = 0u;  // sign of the previous p (p_{-1} = 1 > 0)
  int ka = 0, kb = 0;
  ca = cb = 0;
  auto rescale = [&]() {  // keep |p| inside the exponent range: p = mantissa * 2^k, 0.5 <= |mantissa| < 1
    const int qa = __builtin_amdgcn_frexp_exp(a1), qb = __builtin_amdgcn_frexp_exp(b1);
    a1 = __builtin_ldexp(a1, -qa), a2 = __builtin_ldexp(a2, -qa), ka += qa;
    b1 = __builtin_ldexp(b1, -qb), b2 = __builtin_ldexp(b2, -qb), kb += qb;
  };
  // Rows in groups of eight whose (d, e^2) are all fetched from LDS BEFORE the dependent chain
  // runs: read in program order, every row waited ~120 cycles for its two LDS words in front of
  // ~60 cycles of fp64 chain (the section search was 0.93 M of a 100-node graph's 2.5 M cycles).
  // The search is bound by the number of vector instructions (every thread of the workgroup walks
  // the rows): a row is a subtraction, a product, an FMA and ONE v_alignbit that shifts p's sign
  // into a mask; the sign changes of a group are a popcount.
  int i = s;
  for (; i + 7 <= t; i += 8) {
    double dv[8], ev[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      dv[u] = d[i + u];
      ev[u] = (i + u) > s ? e2[i + u - 1] : 0.0;
    }
    unsigned ma = na, mb = nb;   // bit 8: the sign before the group; bits 7..0: its rows, oldest first
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double pa = fma(dv[u] - xa, a1, -(ev[u] * a2));
      const double pb = fma(dv[u] - xb, b1, -(ev[u] * b2));
      ma = __builtin_amdgcn_alignbit(ma, (unsigned)__double2hiint(pa), 31);   // (ma << 1) | sign(pa)
      mb = __builtin_amdgcn_alignbit(mb, (unsigned)__double2hiint(pb), 31);
      a2 = a1, a1 = pa, b2 = b1, b1 = pb;
    }
    ca += __popc((ma ^ (ma >> 1)) & 0xffu);
    cb += __popc((mb ^ (mb >> 1)) & 0xffu);
    na = ma & 1u, nb = mb & 1u;
    rescale();
  }
  for (; i <= t; ++i) {
    const double di = d[i], ep = i > s ? e2[i - 1] : 0.0;
    const double pa = fma(di - xa, a1, -(ep * a2));
    const double pb = fma(di - xb, b1, -(ep * b2));
    const unsigned sa = (unsigned)__double2hiint(pa) >> 31, sb = (unsigned)__double2hiint(pb) >> 31;
    ca += (int)(sa ^ na);
    cb += (int)(sb ^ nb);
    na = sa, nb = sb;
    a2 = a1, a1 = pa, b2 = b1, b1 = pb;
  }
  rescale();
  fa = (float)a1, fb = (float)b1, ea = ka, eb = kb;
}

// ---------------------------------------------------------------------------------------------
// The Lanczos phase on ONE wavefront (basis in LDS, n <= 64).  A step of the eight-wave form below
// is a chain of ~7-10 workgroup barriers, each in front of one or two LDS round trips around a few
// dozen FMAs: ~1.3 k cycles per phase whatever the work.  Here lane l owns row l
// of the residual, basis vector l in the Gram-Schmidt dot products, and nothing in a step waits
// for another wave: the only cross-lane traffic is the broadcast of a
// vector through LDS (same wave: in order, no barrier) and three wave reductions by DPP.  The
// other seven waves wait at the barrier behind the phase.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, bit-identical in every lane (symmetric pairings inside a row of 16, the
// four row sums added in a fixed order)
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);   // row_half_mirror
  v += dpp_f64<0x140>(v);   // row_mirror
  const double s0 = readlane_f64(v, 0), s1 = readlane_f64(v, 16);
  const double s2 = readlane_f64(v, 32), s3 = readlane_f64(v, 48);
  return (s0 + s1) + (s2 + s3);
}

// LDS traffic between the lanes of one wave: in order in hardware; this keeps the compiler from
// moving a load over the store it depends on
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) double* lds_cd;
typedef const __attribute__((address_space(3))) d2_t* lds_cd2;
typedef const __attribute__((address_space(3))) f4_t* lds_cf4;

// The three inner products of a step, each a walk over LDS in chunks of eight with TWO chunks in
// flight: the loads of chunk i + 1 are issued before the FMAs of chunk i — one wave has nobody to
// hide an LDS round trip behind.  Four (two) independent accumulators, combined in a fixed order.
// Not inlined: inlined six times into the step the three pipelines took 256 registers + 178 spills.

// sum_r q[r] z[r], r < n: q = this lane's own words (stride 1), z = a broadcast vector (16-byte
// aligned)
__device__ __noinline__ double walk_own(lds_cd q, lds_cd z, const int n_) {
  const int n = __builtin_amdgcn_readfirstlane(n_);   // (arguments arrive in vector registers)
  double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
  struct Ch { d2_t z[4]; double a[8]; };
  auto ld = [&](Ch& ch, const int r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) ch.z[u] = *(lds_cd2)(z + r + 2 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) ch.a[u] = q[r + u];
  };
  auto fm = [&](const Ch& ch) {
    p0 = fma(ch.a[0], ch.z[0][0], p0), p1 = fma(ch.a[1], ch.z[0][1], p1);
    p2 = fma(ch.a[2], ch.z[1][0], p2), p3 = fma(ch.a[3], ch.z[1][1], p3);
    p0 = fma(ch.a[4], ch.z[2][0], p0), p1 = fma(ch.a[5], ch.z[2][1], p1);
    p2 = fma(ch.a[6], ch.z[3][0], p2), p3 = fma(ch.a[7], ch.z[3][1], p3);
  };
  const int nc = n >> 3;
  Ch A8, B8;
  if (nc > 0) {
    // (the loop body is branch free: behind a conditional load the wait counters fall back to
    // "everything", which would wait for the chunk that was just requested)
    ld(A8, 0);
    int i = 0;
#pragma unroll 1
    for (; i + 2 < nc; i += 2) {
      ld(B8, 8 * (i + 1));
      fm(A8);
      ld(A8, 8 * (i + 2));
      fm(B8);
    }
    if (i + 1 < nc) {
      ld(B8, 8 * (i + 1));
      fm(A8);
      fm(B8);
    } else {
      fm(A8);
    }
  }
  for (int r = 8 * nc; r < n; ++r) p0 = fma(q[r], z[r], p0);
  return (p0 + p1) + (p2 + p3);
}

// sum_k q[k * ld] c[k], k < cnt: q = this lane's row of every basis vector, c = a broadcast vector
__device__ __noinline__ double walk_strided(lds_cd q, const int ld_v, lds_cd c, const int cnt_) {
  const int ld_ = __builtin_amdgcn_readfirstlane(ld_v), cnt = __builtin_amdgcn_readfirstlane(cnt_);
  double p0 = 0.0, p1 = 0.0;
  struct Ch { d2_t z[4]; double a[8]; };
  auto ld = [&](Ch& ch, const int k) {
#pragma unroll
    for (int u = 0; u < 4; ++u) ch.z[u] = *(lds_cd2)(c + k + 2 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) ch.a[u] = q[(k + u) * ld_];
  };
  auto fm = [&](const Ch& ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) p0 = fma(ch.a[2 * u], ch.z[u][0], p0), p1 = fma(ch.a[2 * u + 1], ch.z[u][1], p1);
  };
  const int nc = cnt >> 3;
  Ch A8, B8;
  if (nc > 0) {
    // (the loop body is branch free: behind a conditional load the wait counters fall back to
    // "everything", which would wait for the chunk that was just requested)
    ld(A8, 0);
    int i = 0;
#pragma unroll 1
    for (; i + 2 < nc; i += 2) {
      ld(B8, 8 * (i + 1));
      fm(A8);
      ld(A8, 8 * (i + 2));
      fm(B8);
    }
    if (i + 1 < nc) {
      ld(B8, 8 * (i + 1));
      fm(A8);
      fm(B8);
    } else {
      fm(A8);
    }
  }
  for (int k = 8 * nc; k < cnt; ++k) p0 = fma(q[k * ld_], c[k], p0);
  return p0 + p1;
}

// sum_c a[c] z[c], c < n4 (a multiple of four): a = this lane's fp32 row of A (16-byte aligned)
__device__ __noinline__ double walk_arow(const __attribute__((address_space(3))) float* a, lds_cd z,
                                         const int n4_) {
  const int n4 = __builtin_amdgcn_readfirstlane(n4_);
  double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
  struct Ch { d2_t z[4]; f4_t xa, xb; };
  auto ld = [&](Ch& ch, const int c) {
#pragma unroll
    for (int u = 0; u < 4; ++u) ch.z[u] = *(lds_cd2)(z + c + 2 * u);
    ch.xa = *(lds_cf4)(a + c), ch.xb = *(lds_cf4)(a + c + 4);
  };
  auto fm = [&](const Ch& ch) {
    p0 = fma((double)ch.xa[0], ch.z[0][0], p0), p1 = fma((double)ch.xa[1], ch.z[0][1], p1);
    p2 = fma((double)ch.xa[2], ch.z[1][0], p2), p3 = fma((double)ch.xa[3], ch.z[1][1], p3);
    p0 = fma((double)ch.xb[0], ch.z[2][0], p0), p1 = fma((double)ch.xb[1], ch.z[2][1], p1);
    p2 = fma((double)ch.xb[2], ch.z[3][0], p2), p3 = fma((double)ch.xb[3], ch.z[3][1], p3);
  };
  const int nc = n4 >> 3;
  Ch A8, B8;
  if (nc > 0) {
    // (the loop body is branch free: behind a conditional load the wait counters fall back to
    // "everything", which would wait for the chunk that was just requested)
    ld(A8, 0);
    int i = 0;
#pragma unroll 1
    for (; i + 2 < nc; i += 2) {
      ld(B8, 8 * (i + 1));
      fm(A8);
      ld(A8, 8 * (i + 2));
      fm(B8);
    }
    if (i + 1 < nc) {
      ld(B8, 8 * (i + 1));
      fm(A8);
      fm(B8);
    } else {
      fm(A8);
    }
  }
  if (n4 & 4) {   // one group of four left
    const int c = 8 * nc;
    const d2_t za = *(lds_cd2)(z + c), zc = *(lds_cd2)(z + c + 2);
    const f4_t xa = *(lds_cf4)(a + c);
    p0 = fma((double)xa[0], za[0], p0), p1 = fma((double)xa[1], za[1], p1);
    p2 = fma((double)xa[2], zc[0], p2), p3 = fma((double)xa[3], zc[1], p3);
  }
  return (p0 + p1) + (p2 + p3);
}

// G row groups x H parts = W waves.  Wave (g, h) holds row 64 g + l of the residual (every part h
// keeps its own identical copy) and walks part h of every inner product: a column range of A w, a
// vector range of the update w -= Q c, and (all W waves) a row range of the Gram-Schmidt dot
// products, whose vector l (and l + 64) is lane l's.  Partials meet in LDS and are summed in a fixed
// order by every wave that needs them; every wave keeps its own copy of the coefficients.  A step
// has four workgroup barriers (five with H > 1; + three for a second Gram-Schmidt pass) with one LDS
// round trip each, against the eight-wave form's seven to ten with two or three; <1, 1> has none.
// The waves without a part keep the barriers company (lanczos_idle_waves).
// (Two rows per lane on ONE wave were measured as well: 3 % slower than the eight-wave form at
// n = 68, 13 % at n = 100.)

// The waves without a part arrive at every barrier of the phase and leave with its last one.  The
// "last" mark alternates between two words by barrier parity: a wave reads word k & 1 behind
// barrier k, the mark for barrier k is written behind barrier k - 1 — into the word nobody reads
// then, so a wave can never see it one barrier early.
__device__ __forceinline__ void lanczos_idle_waves(WgFixed& sm) {
  for (int k = 0;; ++k) {
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(&sm.perm[k & 1]) != 0) break;
  }
}

// part idx of `parts` of the range [0, total): bounds on multiples of eight (16-byte aligned
// broadcast reads, whole chunks), the last part takes what is left
__device__ __forceinline__ void part_range(const int total, const int parts, const int idx, int& lo, int& hi) {
  const int per = ((total + parts - 1) / parts + 7) & ~7;
  lo = idx * per < total ? idx * per : total;
  hi = (idx + 1) * per < total ? (idx + 1) * per : total;
}

template <int G, int H>
__device__ __forceinline__ int lanczos_waves(WgFixed& sm, LwExtra& lw, double* __restrict__ Qt,
                                             const float* __restrict__ As, const int n,
                                             const int LD, const int LA, const int lane, const int wave,
                                             long long* tl = nullptr) {  // tl: LNZ_PROFILE_PHASES (A w, dots, sums, update)
#ifdef LNZ_PROFILE_PHASES
  long long _lt = clock64();
#define LNZ_WT(i) { const long long _l1 = clock64(); tl[i] += _l1 - _lt; _lt = _l1; }
#else
#define LNZ_WT(i)
#endif
  constexpr int W = G * H, RW = 64 * G;
  const int g = wave % G, h = wave / G;
  const int row = 64 * g + lane;
  const bool vr = row < n;
  const int n4 = (n + 3) & ~3;
  int nrestart = 0;
  double* pu = sm.part;     // [H][RW] partial A w          (dead before the dot products)
  double* pc = sm.part;     // [W][RW] partial dot products
  double* pd = lw.pd;    // [H][RW] partial updates
  double* cbw = wave == 0 ? sm.cb : lw.cbx[wave > 0 ? wave - 1 : 0];   // this wave's coefficients
  const lds_cd zb = (lds_cd)sm.zb, Ql = (lds_cd)Qt, cb = (lds_cd)cbw;
  int ac0, ac1, dr0, dr1;
  part_range(n4, H, h, ac0, ac1);      // this wave's columns of A w
  part_range(n, W, wave, dr0, dr1);    // this wave's rows of the dot products
  int kbar = 0;   // workgroup barriers of this phase so far (see lanczos_idle_waves)
  auto wg_bar = [&]() {
    __syncthreads();
    ++kbar;
  };
  auto bar = [&]() {
    if constexpr (W == 1) wave_sync();
    else wg_bar();
  };

  // x <- (I - Q Q^T) x once or twice over basis vectors 0..cnt-1 (x: this lane's row); returns the
  // accumulated coefficient on vector jidx.  The second pass runs where the first removed more
  // than 99 % of the squared length (the rule of the eight-wave form).
  auto cgs2 = [&](double& x, const int cnt, const int jidx) -> double {
    double coef = 0.0;
    const bool vk0 = lane < cnt, vk1 = lane + 64 < cnt;
    int uk0, uk1;
    part_range(cnt, H, h, uk0, uk1);   // this wave's vectors of the update
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      if (vr && h == 0) sm.zb[row] = x;
      bar();
      // ---- c_k = <q_k, x>: lane k walks its own vector over this wave's rows, x is a broadcast read
      // (lanes beyond cnt walk rows that were never written: masked)
      double c0 = walk_own(Ql + lane * LD + dr0, zb + dr0, dr1 - dr0), c1 = 0.0;
      c0 = vk0 ? c0 : 0.0;
      if (G == 2 && cnt > 64) {
        c1 = walk_own(Ql + (lane + 64) * LD + dr0, zb + dr0, dr1 - dr0);
        c1 = vk1 ? c1 : 0.0;
      }
      double xx = wave_sum_f64(x * x);   // (rows beyond n hold 0)
      if constexpr (W > 1) {
        if (h == 0 && lane == 0) lw.px[g] = xx;
        pc[wave * RW + lane] = c0;
        if (G == 2) pc[wave * RW + 64 + lane] = c1;
        wg_bar();
        c0 = pc[lane];
        if (G == 2) c1 = pc[64 + lane];
#pragma unroll
        for (int p_ = 1; p_ < W; ++p_) {
          c0 += pc[p_ * RW + lane];
          if (G == 2) c1 += pc[p_ * RW + 64 + lane];
        }
        xx = lw.px[0];
        if (G == 2) xx += lw.px[1];
      }
      LNZ_WT(1)
      if (vk0) cbw[lane] = c0;
      if (G == 2 && vk1) cbw[lane + 64] = c1;
      bool again = true;
      if (pass == 0) {
        const double cc2 = wave_sum_f64(fma(c0, c0, c1 * c1));
        again = !(cc2 <= LNZ_WG_CGS_AGAIN_WAVES * xx);
      }
      wave_sync();
      coef += cbw[jidx];
      LNZ_WT(2)
      // ---- x -= sum_k c_k q_k: lanes along the rows of this wave's vectors, c a broadcast read
      double d = walk_strided(Ql + row + uk0 * LD, LD, cb + uk0, uk1 - uk0);
      if constexpr (H > 1) {
        pd[h * RW + row] = d;
        wg_bar();
        d = pd[row];
#pragma unroll
        for (int p_ = 1; p_ < H; ++p_) d += pd[p_ * RW + row];
      }
      if (vr) x -= d;
      wave_sync();   // the coefficients are rewritten by the next pass / call (zb, pc, pd: behind a barrier)
      LNZ_WT(3)
      if (!again) break;
    }
    return coef;
  };

  // deterministic, strictly positive, non-symmetric start vector (as lanczos_ritz.hip)
  double w = 0.0;
  if (vr) {
    const unsigned hsh = (unsigned)(row + 1) * 2654435761u;
    w = 1.0 + (double)((hsh >> 8) & 0xffff) * (1.0 / 65536.0);
  }
  if (wave == 0 && lane < 4 && n + lane < n4) sm.zb[n + lane] = 0.0;   // A w reads the vector four columns at a time
  const __attribute__((address_space(3))) float* Al = (const __attribute__((address_space(3))) float*)As;
  bool fresh = true;  // w is a start / restart vector: its norm is not a coupling beta
  for (int j = 0; j < n; ++j) {
    double beta, u;
    for (;;) {
      // ---- beta = |w| and u = A w from one broadcast of w
      if (vr && h == 0) sm.zb[row] = w;
      bar();
      u = walk_arow(Al + row * LA + ac0, zb + ac0, ac1 - ac0);   // (lanes beyond n walk rows that were never staged)
      u = vr ? u : 0.0;
      double nn = wave_sum_f64(w * w);
      if constexpr (W > 1) {
        if (H > 1) pu[h * RW + row] = u;
        if (h == 0 && lane == 0) lw.pn[g] = nn;
      }
      bar();   // zb is rewritten below
      if constexpr (W > 1) {
        if (H > 1) {
          u = pu[row];
#pragma unroll
          for (int p_ = 1; p_ < H; ++p_) u += pu[p_ * RW + row];
        }
        nn = lw.pn[0];
        if (G == 2) nn += lw.pn[1];
      }
      beta = sqrt(nn);
      LNZ_WT(0)
      if (fresh || beta > kBreakdownTol) break;
      // breakdown: span(q_0..q_{j-1}) is A-invariant.  Restart from the unit vector with the
      // largest residual against the basis (residual^2 >= (n-j)/n > 0; ties: the lowest row);
      // T[j-1][j] stays 0.
      ++nrestart;
      double best = -1.0;
      int cand = row;
      if (vr) {
        double s = 0.0;
        for (int i = 0; i < j; ++i) {
          const double qv = Qt[(size_t)i * LD + row];
          s = fma(qv, qv, s);
        }
        best = 1.0 - s;
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oc = __shfl_xor(cand, off, 64);
        if (ob > best || (ob == best && oc < cand)) best = ob, cand = oc;
      }
      if constexpr (W > 1) {
        if (lane == 0) lw.best[wave] = best, lw.cand[wave] = cand;
        wg_bar();
        best = lw.best[0], cand = lw.cand[0];     // (waves 0..G-1 are the row groups' part 0)
        if (G == 2 && lw.best[1] > best) cand = lw.cand[1];   // (equal: group 0's row is the lower one)
      }
      w = (row == cand) ? 1.0 : 0.0;
      (void)cgs2(w, j, 0);
      fresh = true;
    }
    if (!fresh && wave == 0 && lane == 0) sm.ee[j - 1] = beta;
    fresh = false;
    const double binv = 1.0 / beta;
    double x = u * binv;  // A q_j
    if (vr && h == 0) Qt[(size_t)j * LD + row] = w * binv;
    // (the barrier inside cgs2 orders this store before the basis reads)
    const double alpha = cgs2(x, j + 1, j);
    if (wave == 0 && lane == 0) sm.dd[j] = alpha;
    w = x;
  }
  if constexpr (W > 1) {
    if (wave == 0 && lane == 0) sm.perm[kbar & 1] = 1;   // releases the waves that kept the barriers company
    __syncthreads();
  }
  return nrestart;
}

template <bool QG>
__global__ __launch_bounds__(kNT) void lanczos_ritz_wg_kernel(
    const float* __restrict__ A, int64_t sb, int64_t sr, int64_t sc,
    const int32_t* __restrict__ n_nodes, int N, int K, float* __restrict__ D,
    float* __restrict__ V, int32_t* __restrict__ info, double* __restrict__ ws,
    const int mode_flags,    // bit 0: the QL sweep; bit 1: the eight-wave Lanczos phase; bits 2-3: parts
    const int a_bytes) {     // size of the A region (>= wg_a_bytes(N, QG): the launcher adds what the
                             // parallel eigensolver's scratch needs beyond a small graph's A)
  const bool force_ql = (mode_flags & 1) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  WgFixed& sm = *reinterpret_cast<WgFixed*>(smem_raw);
  const int LD = N | 1;  // doubles per basis row: odd -> "lane i reads row i" is conflict free
  double* Qt;
  float* As;
  if constexpr (QG) {
    Qt = ws + (int64_t)blockIdx.x * N * LD;
    As = reinterpret_cast<float*>(smem_raw + sizeof(WgFixed));
  } else {
    Qt = reinterpret_cast<double*>(smem_raw + sizeof(WgFixed) + sizeof(LwExtra));
    As = reinterpret_cast<float*>(Qt + (((size_t)N * LD + 1) & ~(size_t)1));
  }
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = n_nodes[b];
  n = n < 0 ? 0 : (n > N ? N : n);
  // basis in LDS: the Lanczos phase on one wavefront (n <= 64) or two
  const bool one_wave = !QG && n <= LNZ_RITZ_ONE_WAVE_MAX && !(mode_flags & 2);
  const int LA = one_wave ? wg_a_pitch4(N) : (N | 1);  // floats per row of A
  const int kk = K < n ? K : n;  // number of non-padded eigen slots

  // ---- stage the n x n block of A: wave per row, lanes along the row (coalesced when sc == 1)
  {
    const float* Ab = A + (int64_t)b * sb;
    const int nc = one_wave ? (n + 3) & ~3 : n;   // (one-wave phase: rows are read four columns at a time)
    for (int r = wave; r < n; r += kWaves)
      for (int c = lane; c < nc; c += 64) As[r * LA + c] = c < n ? Ab[r * sr + c * sc] : 0.0f;
  }
  if (tid < kNMax) {
    sm.dd[tid] = 0.0;
    sm.ee[tid] = 0.0;
  }
  __syncthreads();

  int nrestart = 0;
  bool solved_out = false;  // the parallel eigensolver ran: V = Q S is formed at the output
#ifdef LNZ_PROFILE_PHASES
  long long tp0 = clock64(), tp1 = tp0, tp2 = tp0, tq0 = tp0, tq1 = tp0, tq2 = tp0, tq3 = tp0;
  long long tl_mv = 0, tl_dot = 0, tl_red = 0, tl_upd = 0;   // Lanczos step parts (thread 0's clock)
#define LNZ_LT0 long long _lt = clock64();
#define LNZ_LACC(x) { const long long _l1 = clock64(); x += _l1 - _lt; _lt = _l1; }
#else
#define LNZ_LT0
#define LNZ_LACC(x)
#endif
  long long* tlw = nullptr;
#ifdef LNZ_PROFILE_PHASES
  long long tlw_[4] = {0, 0, 0, 0};
  tlw = tlw_;
#endif
  if (n > 0 && one_wave) {
    LwExtra& lwx = *reinterpret_cast<LwExtra*>(smem_raw + sizeof(WgFixed));
    // parts per row group: mode_flags bits 2-3 (1, 2, 4; testing), else by size
    int parts = (mode_flags >> 2) & 3;
    parts = parts == 0 ? (n <= 64 ? LNZ_RITZ_PARTS_SMALL : LNZ_RITZ_PARTS_LARGE) : (parts == 3 ? 4 : parts);
    if (n > 64 && parts > 2) parts = 2;
    const int nw = (n <= 64 ? 1 : 2) * parts;   // waves with a part
    if (nw > 1) {
      if (tid == 0) sm.perm[0] = sm.perm[1] = 0;
      __syncthreads();
    }
    if (wave >= nw) {
      if (nw > 1) lanczos_idle_waves(sm);   // (one wave alone has no barrier to keep company at)
    } else if (n <= 64) {
      if (parts == 1) nrestart = lanczos_waves<1, 1>(sm, lwx, Qt, As, n, LD, LA, lane, wave, tlw);
      else if (parts == 2) nrestart = lanczos_waves<1, 2>(sm, lwx, Qt, As, n, LD, LA, lane, wave, tlw);
      else nrestart = lanczos_waves<1, 4>(sm, lwx, Qt, As, n, LD, LA, lane, wave, tlw);
    } else {
      if (parts == 1) nrestart = lanczos_waves<2, 1>(sm, lwx, Qt, As, n, LD, LA, lane, wave, tlw);
      else nrestart = lanczos_waves<2, 2>(sm, lwx, Qt, As, n, LD, LA, lane, wave, tlw);
    }
    __syncthreads();
#ifdef LNZ_PROFILE_PHASES
    tl_mv = tlw_[0], tl_dot = tlw_[1], tl_red = tlw_[2], tl_upd = tlw_[3];
#endif
  }
  if (n > 0 && !one_wave) {
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
        LNZ_LT0
        if (tid < n) sm.zb[tid] = x;
        __syncthreads();
        if (ds < nsd) {
          const double* qi = Qt + (size_t)di * LD;
          double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, q0 = 0.0, q1 = 0.0;
          int c = dc0;
          // eight elements per round, all sixteen LDS words requested before the first FMA (read
          // pair by pair, every FMA waited a full LDS round trip: the step's parts are a barrier
          // plus a few such round trips around a few dozen FMAs); four independent chains
          for (; c + 7 < dc1; c += 8) {
            double qv[8], zv_[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qv[u] = qi[c + u], zv_[u] = sm.zb[c + u];
            p0 = fma(qv[0], zv_[0], p0);
            p1 = fma(qv[1], zv_[1], p1);
            p2 = fma(qv[2], zv_[2], p2);
            p3 = fma(qv[3], zv_[3], p3);
            p0 = fma(qv[4], zv_[4], p0);
            p1 = fma(qv[5], zv_[5], p1);
            p2 = fma(qv[6], zv_[6], p2);
            p3 = fma(qv[7], zv_[7], p3);
#pragma unroll
            for (int u = 0; u < 8; u += 2) q0 = fma(zv_[u], zv_[u], q0), q1 = fma(zv_[u + 1], zv_[u + 1], q1);
          }
          for (; c < dc1; ++c) {
            const double zc_ = sm.zb[c];
            p0 = fma(qi[c], zc_, p0);
            q0 = fma(zc_, zc_, q0);
          }
          sm.part[ds * cnt + di] = (p0 + p1) + (p2 + p3);
          if (di == 0 && pass == 0) sm.pn[ds] = q0 + q1;   // |x|^2 over this segment (second-pass test)
        }
        __syncthreads();
        LNZ_LACC(tl_dot)
        {
          double c = 0.0;
          if (tid < cnt) {
            c = sm.part[tid];
            for (int s = 1; s < nsd; ++s) c += sm.part[s * cnt + tid];
            sm.cb[tid] = c;
          }
          if (pass == 0) {   // sum of the squared coefficients, per wave (fixed order)
            double c2 = c * c;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c2 += __shfl_xor(c2, off, 64);
            if (lane == 0) sm.pc[wave] = c2;
          }
        }
        __syncthreads();
        LNZ_LACC(tl_red)
        coef += sm.cb[jidx];
        // The second pass only where the first one cancelled: with an orthonormal basis
        // |x'|^2 = |x|^2 - sum c_i^2, and one classical Gram-Schmidt pass leaves components of
        // order eps |x| along the basis, i.e. eps |x| / |x'| relative to the new vector.  A Lanczos
        // vector A q_j always loses most of its length to q_j and q_{j-1} (alpha, beta), so the
        // textbook threshold 1 / sqrt(2) would never skip; the second pass (four of the ten
        // barrier phases of a step) runs when |x'| < 0.1 |x| — orthogonality 10 eps per step
        // otherwise, nine orders of magnitude inside what (D, V) are compared at.  The breakdown
        // / restart path (|x'| tiny) always takes it.  Every thread sees the same numbers: the
        // decision is workgroup uniform.
        bool again = true;
        if (pass == 0) {
          double xx = sm.pn[0], cc2 = sm.pc[0];
          for (int s = 1; s < nsd; ++s) xx += sm.pn[s];
          for (int wv = 1; wv < kWaves; ++wv) cc2 += sm.pc[wv];
          again = !(cc2 <= (n > 128 ? LNZ_WG_CGS_AGAIN : LNZ_WG_CGS_AGAIN_WAVES) * xx);
        }
        if (seg_n < nsu) {
          double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
          int i = ui0;
          for (; i + 7 < ui1; i += 8) {   // (eight rows per round, loads first: see the dot products)
            double qv[8], cv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qv[u] = Qt[(size_t)(i + u) * LD + row_n], cv[u] = sm.cb[i + u];
            p0 = fma(qv[0], cv[0], p0);
            p1 = fma(qv[1], cv[1], p1);
            p2 = fma(qv[2], cv[2], p2);
            p3 = fma(qv[3], cv[3], p3);
            p0 = fma(qv[4], cv[4], p0);
            p1 = fma(qv[5], cv[5], p1);
            p2 = fma(qv[6], cv[6], p2);
            p3 = fma(qv[7], cv[7], p3);
          }
          for (; i + 3 < ui1; i += 4) {
            p0 = fma(Qt[(size_t)i * LD + row_n], sm.cb[i], p0);
            p1 = fma(Qt[(size_t)(i + 1) * LD + row_n], sm.cb[i + 1], p1);
            p2 = fma(Qt[(size_t)(i + 2) * LD + row_n], sm.cb[i + 2], p2);
            p3 = fma(Qt[(size_t)(i + 3) * LD + row_n], sm.cb[i + 3], p3);
          }
          for (; i < ui1; ++i) p0 = fma(Qt[(size_t)i * LD + row_n], sm.cb[i], p0);
          if (nsu == 1) x -= (p0 + p1) + (p2 + p3);
          else sm.part[seg_n * n + row_n] = (p0 + p1) + (p2 + p3);
        }
        if (nsu > 1) {
          __syncthreads();
          if (tid < n) {
            double p = sm.part[tid];
            for (int s = 1; s < nsu; ++s) p += sm.part[s * n + tid];
            x -= p;
          }
        }
        LNZ_LACC(tl_upd)
        if (!again) break;
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
        LNZ_LT0
        if (tid < n) sm.zb[tid] = w;
        __syncthreads();
        if (seg_n < nss) {
          const float* ar = As + row_n * LA;
          double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0, s0 = 0.0, s1 = 0.0;
          int c = sc0;
          for (; c + 7 < sc1; c += 8) {   // (eight columns per round, loads first)
            float av[8];
            double zz[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) av[q] = ar[c + q], zz[q] = sm.zb[c + q];
            u0 = fma((double)av[0], zz[0], u0);
            u1 = fma((double)av[1], zz[1], u1);
            u2 = fma((double)av[2], zz[2], u2);
            u3 = fma((double)av[3], zz[3], u3);
            s0 = fma(zz[0], zz[0], s0);
            s1 = fma(zz[1], zz[1], s1);
            s0 = fma(zz[2], zz[2], s0);
            s1 = fma(zz[3], zz[3], s1);
            u0 = fma((double)av[4], zz[4], u0);
            u1 = fma((double)av[5], zz[5], u1);
            u2 = fma((double)av[6], zz[6], u2);
            u3 = fma((double)av[7], zz[7], u3);
            s0 = fma(zz[4], zz[4], s0);
            s1 = fma(zz[5], zz[5], s1);
            s0 = fma(zz[6], zz[6], s0);
            s1 = fma(zz[7], zz[7], s1);
          }
          for (; c + 3 < sc1; c += 4) {
            const double z0 = sm.zb[c], z1 = sm.zb[c + 1], z2 = sm.zb[c + 2], z3 = sm.zb[c + 3];
            u0 = fma((double)ar[c], z0, u0);
            u1 = fma((double)ar[c + 1], z1, u1);
            u2 = fma((double)ar[c + 2], z2, u2);
            u3 = fma((double)ar[c + 3], z3, u3);
            s0 = fma(z0, z0, s0);
            s1 = fma(z1, z1, s1);
            s0 = fma(z2, z2, s0);
            s1 = fma(z3, z3, s1);
          }
          for (; c < sc1; ++c) {
            const double z0 = sm.zb[c];
            u0 = fma((double)ar[c], z0, u0);
            s0 = fma(z0, z0, s0);
          }
          sm.part[seg_n * n + row_n] = (u0 + u1) + (u2 + u3);
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
        LNZ_LACC(tl_mv)
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
  }
  if (n > 0) {
#ifdef LNZ_PROFILE_PHASES
    tp1 = clock64();
#endif
#ifdef LNZ_RITZ_WG_DEBUG  // T of graph LNZ_RITZ_WG_DEBUG -> behind the B info words: [n] d, [n] e as doubles
    __syncthreads();
    if (b == LNZ_RITZ_WG_DEBUG && tid < n) {
      double* o = reinterpret_cast<double*>(info + ((gridDim.x + 1) & ~1));
      o[tid] = sm.dd[tid];
      o[kNMax + tid] = sm.ee[tid];
      // Gram matrix diagonal and one off-diagonal of the basis: |q_tid|^2, <q_tid, q_0>
      double s0 = 0.0, s1 = 0.0;
      for (int i = 0; i < n; ++i) {
        const double qv = Qt[(size_t)tid * LD + i];
        s0 = fma(qv, qv, s0);
        s1 = fma(qv, Qt[i], s1);
      }
      o[2 * kNMax + tid] = s0;
      o[3 * kNMax + tid] = s1;
    }
    __syncthreads();
#endif

    // ---- eigenvalues of T with every thread working (the QL sweep below is one serial chain of
    //      ~n^2 rotations at ~700 cycles each: 5 M cycles at n = 100):
    //   1. T splits where Lanczos restarted (coupling exactly 0) or a coupling is negligible; thread
    //      k owns the (k - s)-th eigenvalue of its unreduced block [s, t] — simple there, while equal
    //      eigenvalues of different blocks get eigenvectors with disjoint support;
    //   2. section search on Sturm counts, two probes per pass (the bracket shrinks 3x);
    //   3. two eigenvalues of ONE block closer than 1e-8 |T| (a degenerate eigenvalue whose second
    //      copy crept into the Krylov space through round-off instead of a clean breakdown): the
    //      twisted-factorisation vectors further down would lose orthogonality like eps / gap — the
    //      workgroup takes the QL sweep instead (info += 256; rare).
    //   T's (d, e, e^2) move to the head of the dead A region; the eigenvectors of the kk SELECTED
    //   eigenvalues are computed after the ordering (twisted factorisation, then V = Q S).
    double* Td = reinterpret_cast<double*>(As);
    double* Te = Td + N;
    double* Te2 = Te + N;
    double* zv = Te2 + N;  // [n][kk]: component i of selected vector q at zv[i * kk + q]
    unsigned long long* cutw = reinterpret_cast<unsigned long long*>(sm.pn);   // [3]
    bool solved = (size_t)3 * N * sizeof(double) + (size_t)kk * n * sizeof(double) <= (size_t)a_bytes &&
                  !force_ql;
    if (solved) {
      bool live = false;
      if (tid < n) {
        const double di = sm.dd[tid], ei = tid < n - 1 ? sm.ee[tid] : 0.0;
        const double dn = tid < n - 1 ? sm.dd[tid + 1] : 0.0;
        live = tid < n - 1 && fabs(ei) > kEps * (fabs(di) + fabs(dn));
        Td[tid] = di;
        Te[tid] = live ? ei : 0.0;
        Te2[tid] = live ? ei * ei : 0.0;
      }
      {  // rows whose coupling to the next one is zero (block_bounds)
        const unsigned long long cm = __ballot(tid < n - 1 && !live);
        if (wave < 3 && lane == 0) cutw[wave] = cm;
      }
      __syncthreads();
      // Gershgorin bound of the spectrum
      if (tid < n) sm.zb[tid] = fabs(Td[tid]) + (tid > 0 ? fabs(Te[tid - 1]) : 0.0) + fabs(Te[tid]);
      __syncthreads();
      double gmax = 0.0;
      for (int i = 0; i < n; ++i) gmax = fmax(gmax, sm.zb[i]);
      const double gsc = gmax > 0.0 ? gmax : 1.0;
      // P threads per eigenvalue (P = kNT / n, at most 8): a pass probes 2 P points that cut the
      // bracket into 2 P + 1 parts — the threads of a group exchange their Sturm counts through
      // LDS and all take the same new bracket — so a 100-node graph needs 15 passes (x 11) where
      // one thread per eigenvalue needed 33 (x 3), and every thread of the workgroup works.
      // (n <= 64: the probes of 256 threads — one wave per SIMD; 512 threads' 17 x shrink per pass
      // saves fewer passes than its second wave per SIMD costs: n = 40 / 64 -5 / -4 %, n = 100 +2 %)
      int P = (n <= 64 ? LNZ_RITZ_EIG_THREADS / 2 : LNZ_RITZ_EIG_THREADS) / n;
      P = P > 8 ? 8 : (P < 1 ? 1 : P);
      const int ev = tid / P, sub = tid - ev * P;       // eigenvalue index, probe pair of this thread
      const bool mine = ev < n;
      int bs = mine ? ev : 0, bt = bs;
      double lam = 0.0;
      double lo = -gsc * 1.0000001, hi = gsc * 1.0000001;
      int jloc = 0;
      if (mine) {
        block_bounds(cutw, ev, n, bs, bt);
        jloc = ev - bs;
      }
      // [n][2 P] Sturm counts + exponents (n P <= kNT: fits sm.part), the probes' mantissas and
      // the brackets' midpoints behind T in the dead A region; the uniform-exit vote below is the
      // barrier between a pass's reads and the next pass's writes.
      // Placement of a group's 2 P probes (the r05 hybrid of csrc/lanczos_ritz.hip): pass 0 covers
      // the bracket uniformly INCLUDING its ends (every later end point carries p); a section pass
      // cuts it into 2 P + 1 parts; once exactly one eigenvalue is inside and p changes sign the
      // probes sit in rings x* -+ d1 8^i around the secant point x* of the end values, d1 ~ the
      // secant's error w^2 / (2 gap to the neighbouring brackets): the bracket collapses
      // quadratically (a 100-node graph: 7..9 passes instead of 15).  Brackets follow the COUNTS.
      int* cnt = reinterpret_cast<int*>(sm.part);
      float* fm = reinterpret_cast<float*>(Te2 + N);
      double* mids = reinterpret_cast<double*>(fm + 2 * LNZ_RITZ_EIG_THREADS);
      const int np = 2 * P;
      bool done = !mine, sect = true;
      float flo = 0.0f, fhi = 0.0f;
      int elo = 0, ehi = 0, clo = 0, chi = bt - bs + 1;
      if (mine && sub == 0) mids[ev] = 0.0;
      __syncthreads();
      for (int it = 0; it < 48; ++it) {
        const double w = hi - lo;
        const double tol = 4.0 * kEps * fmax(fmax(fabs(lo), fabs(hi)), 0.125 * gsc);
        const bool iso = it > 0 && !sect && chi - clo == 1 && flo != 0.0f && ((flo < 0.0f) != (fhi < 0.0f));
        double xs = 0.0, d1 = 0.0;
        if (iso && !done) {
          int de_ = ehi - elo;
          de_ = de_ < -1000 ? -1000 : (de_ > 1000 ? 1000 : de_);
          const double fl = (double)flo, fh = __builtin_ldexp((double)fhi, de_);
          xs = fma(w, fl * rcp_nr(fl - fh), lo);
          double g = gsc;
          if (ev > bs) g = fmin(g, fabs(xs - mids[ev - 1]));
          if (ev < bt) g = fmin(g, fabs(mids[ev + 1] - xs));
          d1 = 0.5 * w * w * __builtin_amdgcn_rcp(fmax(g, 1e-300));
          d1 = fmin(d1, 0.125 * w);
          d1 = fmax(fmax(d1, 0.45 * tol), 4e-7 * w);   // (the probes' values travel as fp32 mantissas)
        }
        const double step0 = w / (double)(np - 1), step1 = w / (double)(np + 1);
        auto probe = [&](int k) -> double {
          if (it == 0) return k == np - 1 ? hi : lo + (double)k * step0;
          if (!iso) return lo + (double)(k + 1) * step1;
          const int ring = k < P ? P - 1 - k : k - P;
          const double off = fmin(__builtin_ldexp(d1, 3 * ring), 0.25 * w);
          const double x = k < P ? xs - off : xs + off;
          return fmin(fmax(x, lo), hi);
        };
        if (!done) {
          int ca, cb, ea, eb;
          float fa, fb;
          sturm2(Td, Te2, bs, bt, probe(2 * sub), probe(2 * sub + 1), ca, cb, fa, fb, ea, eb);
          cnt[ev * np + 2 * sub] = (ea << 8) | ca;
          cnt[ev * np + 2 * sub + 1] = (eb << 8) | cb;
          fm[ev * np + 2 * sub] = fa;
          fm[ev * np + 2 * sub + 1] = fb;
        }
        __syncthreads();
        if (!done) {
          // first probe with more than jloc eigenvalues below it: the eigenvalue lies in the part
          // in front of it (behind the last probe when there is none)
          int pidx = np;
          for (int q = np - 1; q >= 0; --q)
            if ((cnt[ev * np + q] & 255) > jloc) pidx = q;
          const double xlo = pidx > 0 ? probe(pidx - 1) : lo, xhi = pidx < np ? probe(pidx) : hi;
          if (pidx > 0) {
            const int kq_ = cnt[ev * np + pidx - 1];
            clo = kq_ & 255, elo = kq_ >> 8, flo = fm[ev * np + pidx - 1];
          }
          if (pidx < np) {
            const int kq_ = cnt[ev * np + pidx];
            chi = kq_ & 255, ehi = kq_ >> 8, fhi = fm[ev * np + pidx];
          }
          lo = xlo, hi = xhi;
          const double nw = hi - lo;
          sect = nw > 0.25 * w;
          // LAPACK dstebz's stopping rule: relative to |lambda| but never below ulp * |T|
          done = nw <= 4.0 * kEps * fmax(fmax(fabs(lo), fabs(hi)), 0.125 * gsc);
          if (sub == 0) mids[ev] = 0.5 * (lo + hi);
        }
        if (!__syncthreads_or(done ? 0 : 1)) break;
      }
      lam = 0.5 * (lo + hi);
      const bool writer = mine && sub == 0;
      __syncthreads();  // zb (Gershgorin rows) is dead: reuse for the eigenvalues
      if (writer) sm.zb[ev] = lam;
      __syncthreads();
      const bool cluster = writer && ev > bs && (lam - sm.zb[ev - 1]) <= 1e-8 * gsc;
      solved = !__syncthreads_or(cluster ? 1 : 0);
      if (solved && writer) sm.dd[ev] = lam;   // (T itself stays in Td / Te / Te2)
      __syncthreads();
    }

    // ---- fallback: implicit-shift QL (EISPACK tql2 recurrences) on per-wave copies of (d, e);
    //      the rotations are applied to rows i, i+1 of Qt at this thread's own element
    double* wd = reinterpret_cast<double*>(As) + (size_t)wave * 2 * N;
    double* we = wd + N;
    if (!solved) {
    nrestart += 256;  // diagnostic: the QL sweep ran (info = restarts + 256)
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
    if (tid < n) sm.dd[tid] = wd[tid];  // wave 0's copy (identical in every active wave)
    __syncthreads();
    }  // !solved
#ifdef LNZ_PROFILE_PHASES
    tp2 = clock64();
#endif

    // ---- order by descending |lambda| (ties: ascending lambda, then index)
    // = np.argsort(-|eig|, kind='mergesort') on eigh's ascending output (utils/data_helper.py:218-223)
    if (tid < n) {
      const double di = sm.dd[tid], ai = fabs(di);
      int rank = 0;
      for (int j0 = 0; j0 < n; j0 += 8) {   // (eight values requested before they are compared)
        double dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) dv[u] = sm.dd[min(j0 + u, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int jj = j0 + u;
          const double dj = dv[u], aj = fabs(dj);
          const bool before = (aj > ai) || (aj == ai && (dj < di || (dj == di && jj < tid)));
          rank += (jj < n && before) ? 1 : 0;
        }
      }
      sm.perm[rank] = tid;
    }
    __syncthreads();
#ifdef LNZ_PROFILE_PHASES
    tq0 = clock64();
#endif
    if (solved) {
      // ---- eigenvectors of the kk selected eigenvalues: twisted factorisation of T - lambda on the
      //      eigenvalue's block (the dlar1v / MRRR vector: stationary and progressive qd transforms,
      //      twist where |gamma| is smallest), one thread per vector, no pivoting, no iteration.
      //      One array per vector: D+ in place, D- recomputed into the part above the twist.
      if (tid < kk) {
        const int q = tid, idx = sm.perm[q];
        const double lamq = sm.dd[idx];
        int s0, t0;
        block_bounds(cutw, idx, n, s0, t0);
        double gs = 0.0;
        for (int c0 = s0; c0 <= t0; c0 += 8) {   // (a repeated last row does not change a maximum)
          double td[8], te[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) td[u] = Td[min(c0 + u, t0)], te[u] = Te[min(c0 + u, t0)];
#pragma unroll
          for (int u = 0; u < 8; ++u) gs = fmax(gs, fabs(td[u]) + fabs(te[u]));
        }
        const double tiny = kEps * (gs > 0.0 ? gs : 1.0);
        auto guard = [&](double v) { return fabs(v) < tiny ? (v < 0.0 ? -tiny : tiny) : v; };
        for (int i = 0; i < s0; ++i) zv[(size_t)i * kk + q] = 0.0;   // (zero outside the block)
        for (int i = t0 + 1; i < n; ++i) zv[(size_t)i * kk + q] = 0.0;
        // Rows in chunks of eight whose LDS words are all requested before the dependent chain runs
        // (read row by row, every step of the recurrences waited for two or three LDS round trips
        // in front of ~60 cycles of chain: 177 k of a 100-node graph's 1.5 M cycles) — the same
        // operations in the same order.
        double Dp = 1.0;
        for (int c0 = s0; c0 <= t0; c0 += 8) {   // stationary transform, top down
          double td[8], te[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = min(c0 + u, t0);
            td[u] = Td[i];
            te[u] = i > s0 ? Te2[i - 1] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = c0 + u;
            if (i <= t0) {
              const double di = td[u] - lamq;
              Dp = guard(i > s0 ? di - te[u] * rcp_nr(Dp) : di);
              zv[(size_t)i * kk + q] = Dp;
            }
          }
        }
        double Dn = 1.0, gbest = 1e300;
        int tw = s0;
        for (int c0 = t0; c0 >= s0; c0 -= 8) {   // progressive transform, bottom up: find the twist
          double td[8], te[8], zp[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = max(c0 - u, s0);
            td[u] = Td[i];
            te[u] = i < t0 ? Te2[i] : 0.0;
            zp[u] = zv[(size_t)i * kk + q];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = c0 - u;
            if (i >= s0) {
              const double di = td[u] - lamq;
              Dn = guard(i < t0 ? di - te[u] * rcp_nr(Dn) : di);
              const double g = fabs(zp[u] + Dn - di);
              if (g < gbest) gbest = g, tw = i;
            }
          }
        }
        Dn = 1.0;
        for (int c0 = t0; c0 > tw; c0 -= 8) {    // D- again, kept where D+ is no longer needed
          double td[8], te[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = max(c0 - u, s0);
            td[u] = Td[i];
            te[u] = i < t0 ? Te2[i] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = c0 - u;
            if (i > tw) {
              const double di = td[u] - lamq;
              Dn = guard(i < t0 ? di - te[u] * rcp_nr(Dn) : di);
              zv[(size_t)i * kk + q] = Dn;
            }
          }
        }
        double zc = 1.0, nn = 1.0;
        zv[(size_t)tw * kk + q] = 1.0;
        for (int c0 = tw - 1; c0 >= s0; c0 -= 8) {   // z_i = -(e_i / D+_i) z_{i+1}
          double rr[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = max(c0 - u, s0);
            rr[u] = Te[i] * rcp_nr(zv[(size_t)i * kk + q]);   // (off the chain)
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = c0 - u;
            if (i >= s0) {
              zc = -rr[u] * zc;
              zv[(size_t)i * kk + q] = zc;
              nn = fma(zc, zc, nn);
            }
          }
        }
        zc = 1.0;
        for (int c0 = tw + 1; c0 <= t0; c0 += 8) {   // z_i = -(e_{i-1} / D-_i) z_{i-1}
          double rr[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = min(c0 + u, t0);
            rr[u] = Te[i - 1] * rcp_nr(zv[(size_t)i * kk + q]);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = c0 + u;
            if (i <= t0) {
              zc = -rr[u] * zc;
              zv[(size_t)i * kk + q] = zc;
              nn = fma(zc, zc, nn);
            }
          }
        }
        const double sc = rsqrt(nn);
        for (int c0 = s0; c0 <= t0; c0 += 8) {
          double zz[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) zz[u] = zv[(size_t)min(c0 + u, t0) * kk + q];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (c0 + u <= t0) zv[(size_t)(c0 + u) * kk + q] = zz[u] * sc;
        }
      }
      __syncthreads();
    }
    // sign convention: largest-magnitude component (first one on ties) is positive
    if (!solved && tid < kk) {
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
    solved_out = solved;
#ifdef LNZ_PROFILE_PHASES
    tq1 = clock64();
#endif
  }

  // ---- write D [K] and V [N, K] (dataset/graph_data.py:262-287: zero rows >= n, zero slots >= n)
  for (int k = tid; k < K; k += kNT)
    D[(int64_t)b * K + k] = k < kk ? (float)sm.dd[sm.perm[k]] : 0.0f;
  float* Vb = V + (int64_t)b * N * K;
  if (solved_out) {
    // ---- V = Q S for the selected vectors, fp32 to the output; the sign convention (largest
    //      magnitude component positive, first one on ties) is applied in place afterwards
    const double* zvp = reinterpret_cast<const double*>(As) + 3 * (size_t)N;
    for (int idx = tid; idx < N * K; idx += kNT) {
      const int rr = idx / K, k = idx - rr * K;
      if (rr >= n || k >= kk) Vb[idx] = 0.0f;
    }
    // (the sign of a column = the sign of its largest fp32 component, the first one on ties: an
    // LDS maximum over keys |v| bits : 2^31 - 1 - row : sign bit, taken where the values are in
    // registers — a scan of the column just written to global memory was 67 k of a 100-node
    // graph's 1.5 M cycles, one L2 round trip per row)
    unsigned long long* skey = reinterpret_cast<unsigned long long*>(sm.cb);
    for (int k = tid; k < kk; k += kNT) skey[k] = 0ull;
    __syncthreads();
    const int row = tid % n, seg = tid / n, nseg = kNT / n;
    if (seg < nseg) {
      for (int q = seg; q < kk; q += nseg) {
        double a0 = 0.0, a1 = 0.0;
        int i = 0;
        for (; i + 1 < n; i += 2) {
          a0 = fma(Qt[(size_t)i * LD + row], zvp[(size_t)i * kk + q], a0);
          a1 = fma(Qt[(size_t)(i + 1) * LD + row], zvp[(size_t)(i + 1) * kk + q], a1);
        }
        if (i < n) a0 = fma(Qt[(size_t)i * LD + row], zvp[(size_t)i * kk + q], a0);
        const float v = (float)(a0 + a1);
        Vb[(size_t)row * K + q] = v;
        const unsigned long long key = ((unsigned long long)__float_as_uint(fabsf(v)) << 32) |
                                       ((unsigned long long)(0x7fffffffu - (unsigned)row) << 1) |
                                       (v < 0.0f ? 1ull : 0ull);
        atomicMax(&skey[q], key);
      }
    }
    __syncthreads();
#ifdef LNZ_PROFILE_PHASES
    tq2 = tq3 = clock64();
#endif
    if (seg < nseg)   // (every thread re-reads what it wrote itself)
      for (int q = seg; q < kk; q += nseg)
        if (skey[q] & 1ull) Vb[(size_t)row * K + q] = -Vb[(size_t)row * K + q];
  } else {
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
    if (K >= 8) {
      D[(int64_t)b * K + 4] = (float)tl_mv;
      D[(int64_t)b * K + 5] = (float)tl_dot;
      D[(int64_t)b * K + 6] = (float)tl_red;
      D[(int64_t)b * K + 7] = (float)tl_upd;
    }
    if (K >= 13) {  // ordering | twisted vectors | V = Q S | sign scan | sign apply
      D[(int64_t)b * K + 8] = (float)(tq0 - tp2);
      D[(int64_t)b * K + 9] = (float)(tq1 - tq0);
      D[(int64_t)b * K + 10] = (float)(tq2 - tq1);
      D[(int64_t)b * K + 11] = (float)(tq3 - tq2);
      D[(int64_t)b * K + 12] = (float)(tp3 - tq3);
    }
  }
#endif
}

}  // namespace

extern "C" int64_t lnz_lanczos_ritz_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 32 || N > kNMax) return 0;
  if (wg_lds_bytes(N, false) <= (size_t)kLdsMax) return 0;
  return (int64_t)B * N * (N | 1) * (int64_t)sizeof(double);
}

// Shared by lnz_lanczos_ritz (lanczos_ritz.hip) and lnz_lanczos_ritz_ws.
// flags: bit 0 = the basis goes to the workspace even if it would fit in LDS; bit 1 = the QL sweep
// instead of the parallel tridiagonal eigensolver; bit 2 = the eight-wave Lanczos phase where the
// wave-level form would run (basis in LDS); bits 3-4 = parts per row group of the wave-level form
// (1: one, 2: two, 3: four; 0: by size) — all for testing.
int lnz_launch_ritz_wg(const float* A, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                       const int32_t* n_nodes, int B, int N, int K, float* D, float* V,
                       int32_t* info, void* workspace, int64_t workspace_bytes, int flags,
                       hipStream_t s) {
  LNZ_REQUIRE(N <= kNMax, LNZ_ENOTSUP,
              "lnz_lanczos_ritz: N=%d > %d: use lnz_lanczos_ritz_large / _sym (streamed kernels)",
              N, kNMax);
  const bool qg = (flags & 1) || wg_lds_bytes(N, false) > (size_t)kLdsMax;
  size_t lds = wg_lds_bytes(N, qg);
  // The parallel tridiagonal eigensolver keeps 3 N + min(K, N) N doubles in the (then dead) A
  // region; a small graph's A is smaller than that (N = 40, K = 20: 7.4 KB against 6.6 KB — the QL
  // sweep ran instead, 0.49 ms against 0.2).  The region grows to hold it where LDS allows.
  size_t a_bytes = wg_a_bytes(N, qg);
  {
    const size_t want = ((size_t)3 * N + (size_t)(K < N ? K : N) * N) * sizeof(double);
    if (want > a_bytes && lds + (want - a_bytes) <= (size_t)kLdsMax) {
      lds += want - a_bytes;
      a_bytes = want;
    }
  }
  LNZ_REQUIRE(lds <= (size_t)kLdsMax, LNZ_ENOTSUP, "lnz_lanczos_ritz: N=%d needs %zu B of LDS", N,
              lds);
  if (qg) {
    const int64_t need = (int64_t)B * N * (N | 1) * (int64_t)sizeof(double);
    void* owned = nullptr;
    if (!workspace) {
      // no caller workspace: a stream-ordered allocation that lives for this launch only — not
      // while the stream is being captured into a graph (an allocation made at capture time
      // would be freed before the first replay): there the caller owns the workspace
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(s, &cap);
      LNZ_REQUIRE(cap == hipStreamCaptureStatusNone, LNZ_ENOTSUP,
                  "lnz_lanczos_ritz: N=%d keeps its Krylov basis in a %lld-byte workspace; under "
                  "stream capture pass one (lnz_lanczos_ritz_ws, lnz_lanczos_ritz_workspace_bytes)",
                  N, (long long)need);
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
    LNZ_DYNAMIC_LDS(kfn, lds, "lanczos_ritz_wg.hip");
    hipLaunchKernelGGL(kfn, dim3(B), dim3(kNT), lds, s, A, stride_b, stride_r, stride_c, n_nodes, N,
                       K, D, V, info, (double*)workspace, (flags >> 1) & 15, (int)a_bytes);
    const int rc = lnz::check_launch("lnz_lanczos_ritz");
    if (owned) (void)hipFreeAsync(owned, s);
    return rc;
  }
  auto kfn = lanczos_ritz_wg_kernel<false>;
  LNZ_DYNAMIC_LDS(kfn, lds, "lanczos_ritz_wg.hip");
  hipLaunchKernelGGL(kfn, dim3(B), dim3(kNT), lds, s, A, stride_b, stride_r, stride_c, n_nodes, N, K,
                     D, V, info, (double*)nullptr, (flags >> 1) & 15, (int)a_bytes);
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
