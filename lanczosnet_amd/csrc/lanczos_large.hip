// R2 + R6 for LARGE graphs (BASELINE config 5: N = 2048, K = 64): M-step Lanczos with full
// re-orthogonalisation -> QL on the M x M tridiagonal -> Ritz vectors V = Q S.  HBM BOUND.
//
// One 512-thread workgroup (8 wavefronts) per graph — 256 graphs fill the 256 CUs, each CU
// streaming its own A at the per-CU HBM rate.  Per Lanczos step the dense A (4 N^2 bytes; 16.8 MB
// at N = 2048, far beyond LDS/L2) is streamed ONCE:
//   * lane l of every wave owns columns {256 s + 4 l .. +3 | s = 0..7}: its slice of q sits in 32
//     fp64 REGISTERS for the whole step, so the inner loop has no LDS traffic at all;
//   * wave v owns rows v, v+8, ...: 8 x global_load_dwordx4 per row (one fully coalesced 8 KiB
//     row), two rows software-pipelined, fp64 FMAs, one DPP wave reduction per row.
// The Krylov basis (M x N fp64 = 1 MB per graph) lives in a caller-provided HBM workspace: the
// Gram-Schmidt dot products are "one wave per basis vector", the update is "thread owns rows"
// (full stream) / one read of the basis per pass (symmetric kernel).
//
// Algorithmic bytes per graph: M * 4 N^2 (A) + basis traffic ~ 4 * 8 N * M(M+1)/2
// + 4 N K (V) : 1.074 GB + 0.133 GB + 0.5 MB at N = 2048, M = K = 64.
#include "common.hpp"

// A (16.8 MB per graph at N = 2048, re-streamed every Lanczos step) never survives in a cache
// until its next use: non-temporal loads leave L2 / Infinity Cache to the fp64 Krylov basis.
__device__ __forceinline__ float4 lnz_stream_f4(const float* p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

namespace {

constexpr int TPB = 512;
constexpr int NWAVE = TPB / 64;
constexpr int NCH = 8;          // 256-column chunks -> N <= 2048
constexpr int MMAX = 64;        // Lanczos steps
constexpr int ZLD = MMAX;      // QL accumulator rows (aliases the q/w vectors, dead by then)
constexpr double kTol = 1e-8;
constexpr double kReorth = 1e-6;  // second Gram-Schmidt pass when |w1|^2 < kReorth |w0|^2
constexpr double kEpsD = 2.220446049250313e-16;

struct LargeSmem {
  union {
    struct {
      double qs[NCH * 256];
      double ws[NCH * 256];
    };
    double Zt[MMAX * ZLD];  // QL eigenvector accumulator, transposed: Zt[i][r] = S[r][i]
  };
  union {
    double ub[NWAVE / 2][NCH * 256];  // partial CGS updates on their way down the wave tree
    // symmetric SpMV: the contributions to chunk c from the blocks it shares with chunk s != c,
    // cslot[s - (s > c)][256 c ..] (the diagonal block's go straight to ws)
    double cslot[NCH - 1][NCH * 256];
  };
  double dd[MMAX];
  double ee[MMAX];
  double cs[MMAX];
  double red[NWAVE];
  double zero16[16];  // the symmetric SpMV's diagonal jobs read their row vector here
  int perm[MMAX];
  float sgn[MMAX];
};

__device__ inline double dpp_xadd_f64(double v, int sel) {
  int lo = __double2loint(v), hi = __double2hiint(v), l2, h2;
  switch (sel) {
    case 0:
      l2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
      h2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
      break;
    case 1:
      l2 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false);
      h2 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false);
      break;
    case 2:
      l2 = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false);
      h2 = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false);
      break;
    default:
      l2 = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false);
      h2 = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false);
      break;
  }
  return v + __hiloint2double(h2, l2);
}

// wave-wide fp64 sum, identical in every lane, fixed tree
__device__ inline double wave_sum_f64(double v) {
  v = dpp_xadd_f64(v, 0);
  v = dpp_xadd_f64(v, 1);
  v = dpp_xadd_f64(v, 2);
  v = dpp_xadd_f64(v, 3);
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 16 * k);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 16 * k);
    r[k] = __hiloint2double(hi, lo);
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}

__device__ inline double block_sum(LargeSmem& sm, double part, int tid) {
  double w = wave_sum_f64(part);
  if ((tid & 63) == 0) sm.red[tid >> 6] = w;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < NWAVE; ++k) t += sm.red[k];
  __syncthreads();
  return t;
}

// ---- symmetric SpMV (SYM): only the 256 x 256 chunk blocks (I, J >= I) of A are streamed -------
// An off-diagonal block serves both  w_I += A_IJ q_J  (row dots)  and  w_J += A_IJ^T q_I  (column
// sums, accumulated in the registers of the lane that owns the column), a diagonal block its row
// dots over the full chunk: 36 of 64 blocks = 56 % of the bytes at N = 2048.  A block (or a
// 64-row quarter of a diagonal block) is one job of one wave (claimed from a list, see SymSched).
// Every (contribution slot s, chunk I) pair is written by exactly one job — slot s > I: row dots
// of block (I, s); s == I: the diagonal block; s < I: column sums of block (s, I) — into LDS
// (LargeSmem::cslot, 112 KB; the diagonal block's straight into ws), and w is their sum in slot
// order: deterministic, no atomics, no memory traffic besides A itself.  (Until late in round 2
// the slots were an HBM workspace: 4 GB of extra traffic per launch that thrashed L2 against the A
// stream, and the launch took 26.3 or 29.2 ms depending on where the two allocations had landed.)
constexpr int SYM_MAXJOBS = 64;  // 28 off-diagonal blocks + 32 diagonal quarters at N = 2048
constexpr unsigned kOob = 0xffffffffu;  // buffer-load offset past every graph: reads as zeros
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int SYM_RG = 16;  // rows per group: one float4 per lane and row in flight

// A job = rows [row0, row0 + nrow) of chunk block (I, J), packed {I | J << 16, row0 | nrow << 16};
// J == I: row dots only.  The jobs of a Lanczos step sit in one list, large ones first, and the
// waves claim them one by one (an LDS counter): the eight waves of a workgroup do not stream at
// the same rate — the older wave of a SIMD wins the arbitration — and a static split left the
// slow ones streaming alone for the last quarter of every step.  Who runs a job does not matter to
// the result: its contribution slots are its own.
struct SymSched {
  uint2 all[SYM_MAXJOBS];
  uint2 mine[NWAVE][SYM_MAXJOBS];  // the jobs a wave has claimed in this step, in order
  int total;
  int next;
};

__device__ inline void sym_list(SymSched& sc, int N) {  // one thread
  const int nch = (N + 255) >> 8;
  int n = 0;
  auto add = [&](int I, int J, int row0, int nrow) {
    sc.all[n++] = make_uint2((unsigned)I | ((unsigned)J << 16), (unsigned)row0 | ((unsigned)nrow << 16));
  };
  for (int I = 0; I < nch; ++I)
    for (int J = I + 1; J < nch; ++J) add(I, J, 256 * I, min(256, N - 256 * I));
  for (int I = 0; I < nch; ++I)
    for (int q = 0; q < 4; ++q) {
      const int r0 = 256 * I + 64 * q;
      if (r0 < N) add(I, I, r0, min(64, N - r0));
    }
  sc.total = n;
  sc.next = 0;
}

// value of lane i (wave-uniform i) of a per-lane double
__device__ __forceinline__ double lane_f64(double v, int i) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), i);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), i);
  return __hiloint2double(hi, lo);
}

// Implicit-shift QL (tql2 recurrences) on the n x n tridiagonal (sm.dd, sm.ee), n <= 64, by ONE
// wavefront; sm.Zt (identity on entry) receives the eigenvectors transposed, sm.dd the eigenvalues.
__device__ inline void tql2_wave(LargeSmem& sm, const int n, const int lane) {
  double d = sm.dd[lane], e = sm.ee[lane];  // (MMAX == 64 entries, zero past n)
  double f = 0.0, tst1 = 0.0;
  for (int l = 0; l < n; ++l) {
    tst1 = fmax(tst1, fabs(lane_f64(d, l)) + fabs(lane_f64(e, l)));
    // first m >= l with m == n - 1 or a negligible coupling e_m
    unsigned long long stop = __ballot(lane >= n - 1 || !(fabs(e) > kEpsD * tst1));
    stop &= ~0ull << l;
    const int m = __builtin_ctzll(stop);
    if (m > l) {
      int iter = 0;
      double el;
      do {
        ++iter;
        double g = lane_f64(d, l);
        el = lane_f64(e, l);
        double p = (lane_f64(d, l + 1) - g) / (2.0 * el);
        double rr = sqrt(p * p + 1.0);
        if (p < 0) rr = -rr;
        const double dl = el / (p + rr);
        const double dl1 = el * (p + rr);
        const double hh = g - dl;
        if (lane == l) d = dl;
        if (lane == l + 1) d = dl1;
        if (lane >= l + 2 && lane < n) d -= hh;
        f += hh;
        p = lane_f64(d, m);
        double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
        const double el1 = lane_f64(e, l + 1);
        double carry = sm.Zt[m * ZLD + lane];
        double z0 = sm.Zt[(m - 1) * ZLD + lane];  // (m > l >= 0)
        double ei = lane_f64(e, m - 1), di = lane_f64(d, m - 1);
        for (int i = m - 1; i >= l; --i) {
          // the next rotation's inputs ahead of time: none of them is written by this rotation
          const int ip = i > l ? i - 1 : l;
          const double znext = sm.Zt[ip * ZLD + lane];
          const double ei_n = lane_f64(e, ip), di_n = lane_f64(d, ip);
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * ei;
          const double hp = c * p;
          const double tt = fma(p, p, ei * ei);
          const double num = fma(p, di, -(ei * g));  // (p d_i - e_i g): off the rsqrt chain
          // 1 / sqrt(tt): hardware seed (~2^-26) + two Newton steps (tt is a normal double here:
          // |e_i| > eps * tst1 for l <= i < m) — 7 dependent instructions on the rotation-to-rotation
          // chain instead of the library rsqrt's scaling and special cases
          double y = __builtin_amdgcn_rsq(tt);
#pragma unroll
          for (int it = 0; it < 2; ++it) {
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
          if (lane == i + 1) {
            e = e_next;
            d = d_next;
          }
          sm.Zt[(i + 1) * ZLD + lane] = s * z0 + c * carry;
          carry = c * z0 - s * carry;
          z0 = znext;
          ei = ei_n;
          di = di_n;
        }
        sm.Zt[l * ZLD + lane] = carry;
        p = -s * s2 * c3 * el1 * lane_f64(e, l) / dl1;
        el = s * p;
        if (lane == l) {
          e = el;
          d = c * p;
        }
      } while (fabs(el) > kEpsD * tst1 && iter < 60);
    }
    if (lane == l) {
      d += f;
      e = 0.0;
    }
  }
  sm.dd[lane] = d;
}

// 16 per-lane partial sums -> the 16 wave totals: after the two half / row swaps each lane of
// 16-lane row rho holds the partials of rows 4 rho .. 4 rho + 3, reduced over its row by DPP.
// Returns v[k] = total of row 4 * (lane >> 4) + k (all 16 lanes of the row).
__device__ inline void reduce16_f64(const double (&p)[16], double (&v)[4]) {
  double h8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // lanes l, l ^ 32: lower half keeps rows 0..7, upper rows 8..15
    unsigned xl = (unsigned)__double2loint(p[i]), xh = (unsigned)__double2hiint(p[i]);
    unsigned yl = (unsigned)__double2loint(p[i + 8]), yh = (unsigned)__double2hiint(p[i + 8]);
    auto rl = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
    auto rh = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
    h8[i] = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // 16-lane rows 2a, 2a + 1: even rows keep i, odd rows i + 4
    unsigned xl = (unsigned)__double2loint(h8[i]), xh = (unsigned)__double2hiint(h8[i]);
    unsigned yl = (unsigned)__double2loint(h8[i + 4]), yh = (unsigned)__double2hiint(h8[i + 4]);
    auto rl = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
    double t = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    t = dpp_xadd_f64(t, 0);
    t = dpp_xadd_f64(t, 1);
    t = dpp_xadd_f64(t, 2);
    v[i] = dpp_xadd_f64(t, 3);
  }
}

// One classical Gram-Schmidt pass of w (sm.ws) against q_0..q_j, reading the basis ONCE: the wave
// that forms c_i = <q_i, w> keeps q_i in registers and adds c_i q_i to its own partial update u
// (the lane's 32 columns); the eight partial updates are summed in a fixed tree through LDS
// (4 + 2 + 1 buffers) and wave 0 subtracts.  Leaves c_i in sm.cs[i].  (Symmetric kernel only.)
__device__ __forceinline__ void cgs_pass(LargeSmem& sm, const double* __restrict__ Qg, const int N,
                                         const int j, const int wave, const int lane) {
  double wreg[NCH][4], u[NCH][4];
#pragma unroll
  for (int s = 0; s < NCH; ++s) {
    const double2* w = reinterpret_cast<const double2*>(&sm.ws[256 * s + 4 * lane]);
    const double2 w0 = w[0], w1 = w[1];
    wreg[s][0] = w0.x; wreg[s][1] = w0.y; wreg[s][2] = w1.x; wreg[s][3] = w1.y;
    u[s][0] = u[s][1] = u[s][2] = u[s][3] = 0.0;
  }
  for (int i = wave; i <= j; i += NWAVE) {
    const double* qi = Qg + (int64_t)i * N;
    double qv[NCH][4];
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
      const int c0 = 256 * s + 4 * lane;
      double2 g0 = make_double2(0.0, 0.0), g1 = g0;
      if (c0 < N) {
        const double2* g = reinterpret_cast<const double2*>(qi + c0);
        g0 = g[0];
        g1 = g[1];
      }
      qv[s][0] = g0.x; qv[s][1] = g0.y; qv[s][2] = g1.x; qv[s][3] = g1.y;
      s0 = fma(g0.x, wreg[s][0], s0);
      s1 = fma(g0.y, wreg[s][1], s1);
      s0 = fma(g1.x, wreg[s][2], s0);
      s1 = fma(g1.y, wreg[s][3], s1);
    }
    const double c = wave_sum_f64(s0 + s1);
    if (lane == 0) sm.cs[i] = c;
#pragma unroll
    for (int s = 0; s < NCH; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k) u[s][k] = fma(c, qv[s][k], u[s][k]);
  }
  // tree sum of the partial updates: waves 4..7 -> 0..3, 2..3 -> 0..1, 1 -> 0
#pragma unroll
  for (int half = NWAVE / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      double* dst = &sm.ub[wave - half][0];
#pragma unroll
      for (int s = 0; s < NCH; ++s) {
        double2* d = reinterpret_cast<double2*>(dst + 256 * s + 4 * lane);
        d[0] = make_double2(u[s][0], u[s][1]);
        d[1] = make_double2(u[s][2], u[s][3]);
      }
    }
    __syncthreads();
    if (wave < half) {
      const double* src = &sm.ub[wave][0];
#pragma unroll
      for (int s = 0; s < NCH; ++s) {
        const double2* d = reinterpret_cast<const double2*>(src + 256 * s + 4 * lane);
        const double2 a = d[0], b2 = d[1];
        u[s][0] += a.x; u[s][1] += a.y; u[s][2] += b2.x; u[s][3] += b2.y;
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
      double2* w = reinterpret_cast<double2*>(&sm.ws[256 * s + 4 * lane]);
      w[0] = make_double2(wreg[s][0] - u[s][0], wreg[s][1] - u[s][1]);
      w[1] = make_double2(wreg[s][2] - u[s][2], wreg[s][3] - u[s][3]);
    }
  }
  __syncthreads();
}

// ---- sliced-ELL image of a sparse dense-stored A (K-step entry, LNZ_KSTEP_COMPACT) ---------------
// The normalised Laplacian of a G(n, p = 0.01) graph (BASELINE config 5) is 99 % zeros, and the
// K-step recurrence multiplies by it K times.  The dense matrix is therefore read from HBM ONCE, by
// ell_compact_rows_kernel, which gathers the nonzeros of every 64-row slab g into
//   vals[b][g][k][i], cols[b][g][k][i] : entry k of row 64 g + i  (k < cap; zero padded up to
//   widths[b][g] = the slab's longest row rounded up to ELL_UNROLL)
// — 0.4 MB per graph at n = 2048, p = 0.01 instead of 16.8 MB — and the Lanczos steps run on that
// image (MODE 2 below).  A graph with a row of more than `cap` nonzeros raises over[b]; it is left
// to the dense symmetric stream, launched behind (gate).  Skipping an exact zero changes no sum, so
// the result is the dense kernels' up to the order of the fp64 additions.
constexpr int ELL_UNROLL = 8;
struct EllImage {
  const float* vals;
  const uint16_t* cols;
  const int32_t* widths;
  int cap;
};

// One wave per ROW (four rows per workgroup), the whole row requested before the first ballot: lane
// l holds float4 64 u + l of the row, u < 8 — four columns each (PAIR = false), or two columns of
// the two channels of a channels-last [N][N][2] block whose channel 0 is A (PAIR = true: the
// product's collate layout is read in place, its .x / .z are the entries).  Entry k of row 64 g + i
// goes to vals / cols [(g * cap + k) * 64 + i]; the rows of a slab are written by 64 different
// waves, so the slab's width is an atomic max (widths zeroed by the caller) and the zero entries
// up to it are written by ell_pad_kernel behind this launch.  4.3 GB in 0.67 ms (6.4 TB/s); the
// r06 form with one wave per SLAB (a row's ballots behind the previous row's) ran at 4.2 TB/s.
// The same pass can leave the image the large-graph conv gathers from (csrc/conv_sparse.hip,
// lnz_large_sparse_image's format: entries [B][N][ccap] = bf16(value) << 16 | column in the SAME
// entry order, counts, flag bits 0 = the two channels differ somewhere (PAIR only: both are in the
// float4 anyway), 1 = a row beyond ccap) — the collated L is then read from HBM once per batch for
// the Ritz pairs AND the seven conv layers.
struct ConvImageOut {
  unsigned* ent;      // NULL: not wanted
  float* vals;        // NULL: not wanted (the exact-fp32 form's unrounded values)
  int32_t* counts;
  int32_t* flags;
  int cap;
};
typedef __bf16 lnz_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lnz_f32x2 __attribute__((ext_vector_type(2)));
__device__ inline unsigned conv_entry(float v, int col) {   // (= conv_sparse.hip pack_entry: round to nearest even)
  const lnz_bf16x2 p = __builtin_convertvector(lnz_f32x2{v, 0.0f}, lnz_bf16x2);
  return ((unsigned)__builtin_bit_cast(unsigned short, p[0]) << 16) | (unsigned)col;
}

template <bool PAIR>
__global__ __launch_bounds__(256) void ell_compact_rows_kernel(
    const float* __restrict__ A, int64_t sb, int64_t sr, int B, int N, int cap,
    float* __restrict__ vals, uint16_t* __restrict__ cols, int32_t* __restrict__ widths,
    int32_t* __restrict__ rowcnt, int32_t* __restrict__ over, ConvImageOut cv) {
  const int lane = threadIdx.x & 63;
  const int64_t rid = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (rid >= (int64_t)B * N) return;
  const int b = (int)(rid / N), r = (int)(rid - (int64_t)b * N);
  const int nslab = (N + 63) >> 6, g = r >> 6, i = r & 63;
  const float4* src = reinterpret_cast<const float4*>(A + (int64_t)b * sb + (int64_t)r * sr);
  const int64_t base = (((int64_t)b * nslab + g) * cap) * 64 + i;
  float* vs = vals + base;
  uint16_t* cs = cols + base;
  int k = 0;   // entries of this row so far (wave-uniform)
  unsigned* ce = cv.ent ? cv.ent + rid * cv.cap : nullptr;
  float* cvv = (cv.ent && cv.vals) ? cv.vals + rid * cv.cap : nullptr;
  bool differ = false;
  auto place = [&](const float v, const int col) {
    const bool nz = v != 0.f;
    const unsigned long long m = __ballot(nz);
    if (m == 0ull) return;
    const int pos = k + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (nz && pos < cap) {
      vs[(int64_t)pos * 64] = v;
      cs[(int64_t)pos * 64] = (uint16_t)col;
    }
    if (ce && nz && pos < cv.cap) {
      ce[pos] = conv_entry(v, col);
      if (cvv) cvv[pos] = v;
    }
    k += __popcll(m);
  };
  const int nq = PAIR ? N >> 1 : N >> 2;   // float4s per row
  for (int q0 = 0; q0 < nq; q0 += 64 * 8) {
    float4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + 64 * u + lane;
      x[u] = q < nq ? lnz_stream_f4(reinterpret_cast<const float*>(src + q)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + 64 * u + lane;
      if (PAIR) {
        differ |= (x[u].x != x[u].y) | (x[u].z != x[u].w);
        if (__ballot(x[u].x != 0.f || x[u].z != 0.f) == 0ull) continue;
        place(x[u].x, 2 * q);
        place(x[u].z, 2 * q + 1);
      } else {
        if (__ballot(x[u].x != 0.f || x[u].y != 0.f || x[u].z != 0.f || x[u].w != 0.f) == 0ull) continue;
        place(x[u].x, 4 * q);
        place(x[u].y, 4 * q + 1);
        place(x[u].z, 4 * q + 2);
        place(x[u].w, 4 * q + 3);
      }
    }
  }
  if (ce) {
    const int c = k < cv.cap ? k : cv.cap;
    if (c + lane < ((c + 7) & ~7)) {   // (the conv walks whole groups of eight)
      ce[c + lane] = 0u;
      if (cvv) cvv[c + lane] = 0.f;
    }
    const bool any_differ = __ballot(differ) != 0ull;
    if (lane == 0) {
      cv.counts[rid] = c;
      const int f = (any_differ ? 1 : 0) | (k > cv.cap ? 2 : 0);
      if (f) atomicOr(cv.flags, f);
    }
  }
  if (lane == 0) {
    if (k > cap) over[b] = 1;   // (every writer stores the same value)
    const int c = k < cap ? k : cap;
    rowcnt[rid] = c;
    atomicMax(widths + (int64_t)b * nslab + g, (c + ELL_UNROLL - 1) / ELL_UNROLL * ELL_UNROLL);
  }
}

// zero entries from a row's count up to its slab's width (one wave per slab, lane = row)
__global__ __launch_bounds__(256) void ell_pad_kernel(int B, int N, int cap, float* __restrict__ vals,
                                                      uint16_t* __restrict__ cols,
                                                      const int32_t* __restrict__ widths,
                                                      const int32_t* __restrict__ rowcnt) {
  const int lane = threadIdx.x & 63;
  const int nslab = (N + 63) >> 6;
  const int64_t sid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sid >= (int64_t)B * nslab) return;
  const int b = (int)(sid / nslab), g = (int)(sid - (int64_t)b * nslab);
  const int row = 64 * g + lane;
  const int w = widths[sid];
  const int c = row < N ? rowcnt[(int64_t)b * N + row] : 0;
  const int64_t base = sid * cap * 64 + lane;
  for (int k = c; k < w; ++k) {
    vals[base + (int64_t)k * 64] = 0.f;
    cols[base + (int64_t)k * 64] = 0;
  }
}

#ifdef LNZ_LARGE_PROBE
__device__ unsigned long long g_large_probe[16];
#define LNZ_PROBE(k)                                                  \
  do {                                                                \
    if (tid == 0) {                                                   \
      const unsigned long long t_ = wall_clock64();                   \
      atomicAdd(&g_large_probe[k], t_ - t_last);                      \
      t_last = t_;                                                    \
    }                                                                 \
  } while (0)
#else
#define LNZ_PROBE(k)
#endif

// MODE 0: full stream; 1: symmetric stream (upper chunk blocks); 2: the sliced-ELL image of A that
// ell_compact_rows_kernel gathered (A itself is not read at all).  `gate` (optional): the per-graph
// "row capacity exceeded" flags of the compaction — the workgroup of graph b runs only when
// (gate[b] != 0) == gate_want, so that the ELL launch and the dense fallback launch behind it
// split a batch between them without a host round trip.
template <int MODE>
__global__ __launch_bounds__(TPB) void lanczos_ritz_large_kernel(
    const float* __restrict__ A, int64_t sb, int64_t sr, int N, int M, int K,
    double* __restrict__ work, float* __restrict__ D,
    float* __restrict__ V, int32_t* __restrict__ info, const int32_t* __restrict__ n_nodes,
    const int32_t* __restrict__ gate, int gate_want, EllImage ell) {
  constexpr bool SYM = MODE == 1;
  constexpr bool ELL = MODE == 2;
  __shared__ __attribute__((aligned(16))) LargeSmem sm;
  __shared__ SymSched sched;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (gate && (gate[b] != 0) != (gate_want != 0)) return;
  // ragged batch: graph b has n_b <= N nodes, rows / columns >= n_b of its matrix are zero padding
  // (dataset/graph_data.py:222-260).  Only the start vector has to know: a Krylov vector that is
  // zero on the padding stays zero there, and the recurrence stops by itself after n_b steps.
  const int n_b = n_nodes ? min(max(n_nodes[b], 0), N) : N;
  const int wave = tid >> 6, lane = tid & 63;
  const float* Ab = A + (int64_t)b * sb;
  double* Qg = work + (int64_t)b * MMAX * N;
  if (SYM && tid == 0) sym_list(sched, N);
  // the graph's matrix as a buffer: 32-bit byte offsets, out-of-range offsets read as zeros
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(Ab), 0, (unsigned)(((int64_t)(N - 1) * sr + N) * 4), 0x00020000);
#ifdef LNZ_LARGE_PROBE
  unsigned long long t_last = wall_clock64();
#endif

  // start vector (same hash as the small-graph kernel)
  double part = 0.0;
  for (int r = tid; r < NCH * 256; r += TPB) {
    double w = 0.0;
    if (r < n_b) {
      unsigned hsh = (unsigned)(r + 1) * 2654435761u;
      w = 1.0 + (double)((hsh >> 8) & 0xffff) * (1.0 / 65536.0);
    }
    sm.ws[r] = w;
    part += w * w;
  }
  if (tid < MMAX) {
    sm.dd[tid] = 0.0;
    sm.ee[tid] = 0.0;
  }
  if (tid < 16) sm.zero16[tid] = 0.0;
  double nrm2 = block_sum(sm, part, tid);

  int steps = 0;
  for (int j = 0; j < M; ++j) {
    const double nrm = sqrt(nrm2);
    // invariant subspace reached: stop (slots stay zero); an empty graph takes no step at all
    if ((j > 0 || n_b == 0) && nrm <= kTol) break;
    if (j > 0 && tid == 0) sm.ee[j - 1] = nrm;
    const double ninv = 1.0 / nrm;
    for (int r = tid; r < NCH * 256; r += TPB) {
      double q = sm.ws[r] * ninv;
      sm.qs[r] = q;
      if (r < N) Qg[(int64_t)j * N + r] = q;
    }
    if (SYM && tid == 0) sched.next = 0;  // this step's job list is unclaimed again
    __syncthreads();
    LNZ_PROBE(0);

    if constexpr (SYM) {
      // ---- SpMV, symmetric: this wave's jobs in groups of SYM_RG rows.  A ring of SYM_RG row
      //      slots (one float4 per lane): the slot of a consumed row is refilled at once with the
      //      same row of the NEXT group, so SYM_RG rows (16 KiB per wave) stay in flight.  The
      //      loads are buffer loads issued unconditionally — a row or column that does not exist
      //      gets an out-of-range offset and comes back as zeros without touching memory — so
      //      the loop has no branch around a load and every row waits for exactly its own
      //      (predicated flat loads made hipcc drain the whole ring once per group).
      const int total = __builtin_amdgcn_readfirstlane(sched.total);
      uint2* mine = sched.mine[wave];
      int ngrab = 0;     // jobs claimed so far
      bool more = true;  // the list may have unclaimed jobs
      auto grab = [&]() {
        int id = 0;
        if (lane == 0) id = atomicAdd(&sched.next, 1);
        id = __builtin_amdgcn_readfirstlane(id);
        if (id < total) {
          if (lane == 0) mine[ngrab] = sched.all[id];
          ++ngrab;
        } else {
          more = false;
        }
      };
      auto job_at = [&](int k, int& I, int& J, int& row0, int& nrow) {  // wave-uniform, in SGPRs
        const uint2 raw = mine[k];
        const unsigned lo = __builtin_amdgcn_readfirstlane(raw.x);
        const unsigned hi = __builtin_amdgcn_readfirstlane(raw.y);
        I = (int)(lo & 0xffffu); J = (int)(lo >> 16); row0 = (int)(hi & 0xffffu); nrow = (int)(hi >> 16);
      };
      const unsigned srb = (unsigned)sr * 4u;  // row stride in bytes (the launcher checked the range)
      int pj = 0, pg = 0;        // prefetch cursor: job, first row of the group inside the job
      unsigned poff = kOob;      // this lane's byte offset of row 0 of the prefetch group
      int pvalid = 0;            // rows of the prefetch group (0: nothing left)
      auto next_prefetch = [&]() {
        pvalid = 0;
        poff = kOob;
        if (pj == ngrab && more) grab();
        if (pj < ngrab) {
          int I, J, row0, nrow;
          job_at(pj, I, J, row0, nrow);
          const int c0 = 256 * J + 4 * lane;
          if (c0 < N) poff = (unsigned)(row0 + pg) * srb + 4u * (unsigned)c0;
          pvalid = min(SYM_RG, nrow - pg);
          pg += SYM_RG;
          if (pg >= nrow) {
            pg = 0;
            ++pj;
          }
        }
      };
      float4 buf[SYM_RG];
      auto fetch = [&](int i) {
        const unsigned off = (i < pvalid && poff != kOob) ? poff + (unsigned)i * srb : kOob;
        const u4v raw = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, /*nt*/ 2);
        buf[i] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z),
                             __uint_as_float(raw.w));
      };
      next_prefetch();
#pragma unroll
      for (int i = 0; i < SYM_RG; ++i) {
        fetch(i);
        // the ring is filled in the order the loop consumes it: hipcc's vmcnt bookkeeping takes
        // the worst case over the loop's entries, and a shuffled fill would cost a drain per group
        __builtin_amdgcn_sched_barrier(0);
      }
      for (int cj = 0; cj < ngrab; ++cj) {  // (ngrab grows as the prefetch cursor claims jobs)
        int I, J, row0, nrow;
        job_at(cj, I, J, row0, nrow);
        // one code path for both kinds of job: a diagonal block reads its "q of the rows" from a
        // line of zeros, so its column sums stay zero and are not stored
        const bool offd = J != I;
        double qJ[4], colacc[4];
        {
          const double2* qp = reinterpret_cast<const double2*>(&sm.qs[256 * J + 4 * lane]);
          const double2 x = qp[0], y = qp[1];
          qJ[0] = x.x; qJ[1] = x.y; qJ[2] = y.x; qJ[3] = y.y;
          colacc[0] = colacc[1] = colacc[2] = colacc[3] = 0.0;
        }
        for (int cg = 0; cg < nrow; cg += SYM_RG) {
          const int r0 = row0 + cg;
          next_prefetch();
          // q of the group's rows, fetched ahead of the row loop (broadcast reads; rows past N
          // hold zeros)
          double qr[SYM_RG];
          {
            const double2* qp = reinterpret_cast<const double2*>(offd ? &sm.qs[r0] : sm.zero16);
#pragma unroll
            for (int i = 0; i < SYM_RG; i += 2) {
              const double2 t = qp[i >> 1];
              qr[i] = t.x;
              qr[i + 1] = t.y;
            }
          }
          double p[SYM_RG];
#pragma unroll
          for (int i = 0; i < SYM_RG; ++i) {
            const double ax = (double)buf[i].x, ay = (double)buf[i].y, az = (double)buf[i].z,
                         aw = (double)buf[i].w;
            fetch(i);
            p[i] = fma(ax, qJ[0], ay * qJ[1]) + fma(az, qJ[2], aw * qJ[3]);
            colacc[0] = fma(ax, qr[i], colacc[0]);
            colacc[1] = fma(ay, qr[i], colacc[1]);
            colacc[2] = fma(az, qr[i], colacc[2]);
            colacc[3] = fma(aw, qr[i], colacc[3]);
            // row by row, the column sums pinned to their row: left alone hipcc sinks all 64 of
            // their FMAs behind the reduction, keeps every row's fp64 copy alive until then and
            // spills — and a scratch reload in this loop costs a full vmcnt(0) drain
            asm volatile("" : "+v"(colacc[0]), "+v"(colacc[1]), "+v"(colacc[2]), "+v"(colacc[3]));
            __builtin_amdgcn_sched_barrier(0);
          }
          double v[4];
          reduce16_f64(p, v);
          if ((lane & 15) == 0) {  // row dots: slot J of chunk I (J > I), the diagonal block's to ws
            double* dst = (offd ? sm.cslot[J - 1] : sm.ws) + r0 + 4 * (lane >> 4);
            *reinterpret_cast<double2*>(dst) = make_double2(v[0], v[1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(v[2], v[3]);
          }
        }
        if (offd) {  // column sums: slot I of chunk J (I < J)
          double* dst = &sm.cslot[I][256 * J + 4 * lane];
          *reinterpret_cast<double2*>(dst) = make_double2(colacc[0], colacc[1]);
          *reinterpret_cast<double2*>(dst + 2) = make_double2(colacc[2], colacc[3]);
        }
      }
      LNZ_PROBE(1);
      __syncthreads();
      LNZ_PROBE(2);
      // w = the contributions of the eight blocks that touch a chunk, in block order (the
      // diagonal block's is already in ws)
      const int nch = (N + 255) >> 8;
      part = 0.0;
      for (int r = tid; r < NCH * 256; r += TPB) {
        const int c = r >> 8;
        const double diag = sm.ws[r];
        double acc = 0.0;
#pragma unroll
        for (int sl = 0; sl < NCH; ++sl) {
          const double x = sl == c ? diag : sm.cslot[sl - (sl > c ? 1 : 0)][r];
          if (sl < nch && c < nch) acc += x;
        }
        sm.ws[r] = acc;
        part = fma(acc, acc, part);  // |w|^2 on the way (rows past N are zeros)
      }
    } else if constexpr (ELL) {
      // ---- SpMV on the sliced-ELL image: lane i of the wave that owns slab g forms row 64 g + i,
      //      entry k of the slab's rows is one coalesced 256-byte (values) + 128-byte (columns)
      //      read; q is gathered from LDS.  Entries in the order the compaction met them
      //      (fixed), padded with zeros to the slab's width (a multiple of ELL_UNROLL).
      const int nslab = (N + 63) >> 6;
      part = 0.0;
      for (int g = wave; g < nslab; g += NWAVE) {
        const int w = __builtin_amdgcn_readfirstlane(ell.widths[(int64_t)b * nslab + g]);
        const int64_t base = (((int64_t)b * nslab + g) * ell.cap) * 64 + lane;
        const float* vp = ell.vals + base;
        const uint16_t* cp = ell.cols + base;
        double acc0 = 0.0, acc1 = 0.0;
        float vb[ELL_UNROLL];
        uint16_t cb[ELL_UNROLL];
        if (w > 0) {
#pragma unroll
          for (int i = 0; i < ELL_UNROLL; ++i) {
            vb[i] = vp[i * 64];
            cb[i] = cp[i * 64];
          }
        }
        for (int k0 = 0; k0 < w; k0 += ELL_UNROLL) {
          float vn[ELL_UNROLL];
          uint16_t cn[ELL_UNROLL];
          const bool more_k = k0 + ELL_UNROLL < w;  // (wave-uniform)
          if (more_k) {
#pragma unroll
            for (int i = 0; i < ELL_UNROLL; ++i) {
              vn[i] = vp[(k0 + ELL_UNROLL + i) * 64];
              cn[i] = cp[(k0 + ELL_UNROLL + i) * 64];
            }
          }
#pragma unroll
          for (int i = 0; i < ELL_UNROLL; i += 2) {
            acc0 = fma((double)vb[i], sm.qs[cb[i]], acc0);
            acc1 = fma((double)vb[i + 1], sm.qs[cb[i + 1]], acc1);
          }
          if (more_k) {
#pragma unroll
            for (int i = 0; i < ELL_UNROLL; ++i) {
              vb[i] = vn[i];
              cb[i] = cn[i];
            }
          }
        }
        const double acc = acc0 + acc1;
        sm.ws[64 * g + lane] = acc;   // (rows in [N, 64 nslab) have no entries: zeros)
        part = fma(acc, acc, part);
      }
    } else {
    // ---- SpMV: w = A q; q slice in registers, A streamed once ------------------------------
      double qreg[NCH][4];
#pragma unroll
      for (int s = 0; s < NCH; ++s) {
        const double2* p = reinterpret_cast<const double2*>(&sm.qs[256 * s + 4 * lane]);
        double2 x = p[0], y = p[1];
        qreg[s][0] = x.x;
        qreg[s][1] = x.y;
        qreg[s][2] = y.x;
        qreg[s][3] = y.y;
      }
      auto load_row = [&](int r, float4 (&a)[NCH]) {
        const float* row = Ab + (int64_t)r * sr;
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
          int c0 = 256 * s + 4 * lane;
          a[s] = (c0 < N) ? lnz_stream_f4(row + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto dot_row = [&](const float4 (&a)[NCH]) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
          s0 = fma((double)a[s].x, qreg[s][0], s0);
          s1 = fma((double)a[s].y, qreg[s][1], s1);
          s2 = fma((double)a[s].z, qreg[s][2], s2);
          s3 = fma((double)a[s].w, qreg[s][3], s3);
        }
        return (s0 + s1) + (s2 + s3);
      };
      {
        float4 a0[NCH], a1[NCH];
        int r = wave;
        if (r < N) load_row(r, a0);
        for (; r < N; r += 2 * NWAVE) {
          const int r1 = r + NWAVE, r2 = r + 2 * NWAVE;
          if (r1 < N) load_row(r1, a1);
          double t0 = wave_sum_f64(dot_row(a0));
          if (lane == 0) sm.ws[r] = t0;
          if (r2 < N) load_row(r2, a0);
          if (r1 < N) {
            double t1 = wave_sum_f64(dot_row(a1));
            if (lane == 0) sm.ws[r1] = t1;
          }
        }
      }
    }
    __syncthreads();
    LNZ_PROBE(3);

    // ---- Gram-Schmidt against q_0..q_j: one classical pass; a second one only when the first
    //      cancelled more than kReorth of w's norm (the projection's rounding error relative to
    //      what is left is eps * |w0| / |w1|: below 1e-3 that is still < 3e-13, far inside the
    //      fp32 outputs; near an invariant subspace |w1| collapses and the second pass runs).
    //      oracle/lanczos_kstep.py states the same rule.
    double coef = 0.0;
    if constexpr (MODE == 0) {
      part = 0.0;
      for (int r = tid; r < N; r += TPB) part = fma(sm.ws[r], sm.ws[r], part);
    }
    const double nrm2_before = block_sum(sm, part, tid);
    LNZ_PROBE(4);
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {
        part = 0.0;
        for (int r = tid; r < N; r += TPB) part = fma(sm.ws[r], sm.ws[r], part);
        nrm2 = block_sum(sm, part, tid);
        if (nrm2 >= kReorth * nrm2_before) break;
      }
      if constexpr (MODE != 0) {
        cgs_pass(sm, Qg, N, j, wave, lane);
      } else {
        // full-stream kernel (its SpMV holds q in 64 registers; the single-read pass spills
        // there): dots "one wave per basis vector", update "thread owns rows" — two reads of Q
        for (int i = wave; i <= j; i += NWAVE) {
          const double* qi = Qg + (int64_t)i * N;
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int s = 0; s < NCH; ++s) {
            int c0 = 256 * s + 4 * lane;
            if (c0 < N) {
              const double2* g = reinterpret_cast<const double2*>(qi + c0);
              const double2* w = reinterpret_cast<const double2*>(&sm.ws[c0]);
              double2 g0 = g[0], g1 = g[1], w0 = w[0], w1 = w[1];
              s0 = fma(g0.x, w0.x, s0);
              s1 = fma(g0.y, w0.y, s1);
              s0 = fma(g1.x, w1.x, s0);
              s1 = fma(g1.y, w1.y, s1);
            }
          }
          double c = wave_sum_f64(s0 + s1);
          if (lane == 0) sm.cs[i] = c;
        }
        __syncthreads();
        for (int r = tid; r < N; r += TPB) {
          double acc0 = 0.0, acc1 = 0.0;
          int i = 0;
          for (; i + 1 <= j; i += 2) {
            acc0 = fma(sm.cs[i], Qg[(int64_t)i * N + r], acc0);
            acc1 = fma(sm.cs[i + 1], Qg[(int64_t)(i + 1) * N + r], acc1);
          }
          if (i <= j) acc0 = fma(sm.cs[i], Qg[(int64_t)i * N + r], acc0);
          sm.ws[r] -= (acc0 + acc1);
        }
        __syncthreads();
      }
      coef += sm.cs[j];
      LNZ_PROBE(5 + pass);
    }
    if (tid == 0) sm.dd[j] = coef;
    part = 0.0;
    for (int r = tid; r < N; r += TPB) part = fma(sm.ws[r], sm.ws[r], part);
    nrm2 = block_sum(sm, part, tid);  // (after a skipped second pass: the same sum again)
    steps = j + 1;
    LNZ_PROBE(7);
  }
  __syncthreads();

  // ---- QL (tql2 recurrences) on the steps x steps tridiagonal; Zt starts as identity --------
  const int n = steps;
  for (int idx = tid; idx < MMAX * ZLD; idx += TPB) {
    int i = idx / ZLD, r = idx - i * ZLD;
    sm.Zt[idx] = (i == r) ? 1.0 : 0.0;
  }
  __syncthreads();
  // One wavefront runs the whole sweep: lane r holds d_r and e_r in registers (a wave-uniform
  // element is a v_readlane away: no LDS on the rotation chain) and owns column r of the
  // accumulator rows in LDS (nobody else touches it: no barrier, no fence); the rotation scalars
  // are recomputed by every lane.  Same recurrences as before, rotation for rotation.
  if (wave == 0) tql2_wave(sm, n, lane);
  __syncthreads();

  LNZ_PROBE(8);
  // ---- order by descending |theta| (ties: ascending theta, then index) ----------------------
  if (tid < n) {
    double di = sm.dd[tid], ai = fabs(di);
    int rank = 0;
    for (int jj = 0; jj < n; ++jj) {
      double dj = sm.dd[jj], aj = fabs(dj);
      bool before = (aj > ai) || (aj == ai && (dj < di || (dj == di && jj < tid)));
      rank += before ? 1 : 0;
    }
    sm.perm[rank] = tid;
  }
  __syncthreads();
  const int kk = K < n ? K : n;
  if (tid < kk) {  // sign: largest-magnitude coefficient of the Ritz vector in the Krylov basis > 0
    const double* s = &sm.Zt[sm.perm[tid] * ZLD];
    double best = 0.0;
    float sg = 1.0f;
    for (int x = 0; x < n; ++x) {
      double av = fabs(s[x]);
      if (av > best) {
        best = av;
        sg = s[x] < 0 ? -1.0f : 1.0f;
      }
    }
    sm.sgn[tid] = sg;
  }
  __syncthreads();

  // ---- D [K], V [N, K] = Q S -------------------------------------------------------------------
  for (int k = tid; k < K; k += TPB) D[(int64_t)b * K + k] = k < kk ? (float)sm.dd[sm.perm[k]] : 0.0f;
  // V = Q S on the fp64 matrix cores (v_mfma_f64_16x16x4_f64): wave w forms the 16-row tiles w, w + 8,
  // ... of V for all (<= 64) Ritz vectors — A = 16 basis entries of 4 Krylov vectors straight from
  // the workspace (the basis is read ONCE, 128-byte runs), B = the selected, signed Ritz coefficients
  // St[i][k] staged in the slot region (dead by now; pitch 80: the four k rows of a fragment fall on
  // disjoint banks).  The r05 form read one LDS coefficient per FMA: 67 MB of LDS reads per graph,
  // 0.31 ms of the compacted path's 4.5 ms.
  constexpr int SP = 80;
  double* St = &sm.cslot[0][0];   // [MMAX][SP]
  for (int idx = tid; idx < MMAX * MMAX; idx += TPB) {
    const int i = idx / MMAX, k = idx - i * MMAX;
    St[i * SP + k] = (k < kk && i < n) ? (double)sm.sgn[k] * sm.Zt[sm.perm[k] * ZLD + i] : 0.0;
  }
  __syncthreads();
  typedef double d4v __attribute__((ext_vector_type(4)));
  float* Vb = V + (int64_t)b * N * K;
  const int j16 = lane & 15, k4 = lane >> 4;
  const int ntile = (N + 15) >> 4, nblk = (K + 15) >> 4;
  for (int t = wave; t < ntile; t += NWAVE) {
    const int row = 16 * t + j16;
    double a[MMAX / 4];
#pragma unroll
    for (int s4 = 0; s4 < MMAX / 4; ++s4) {
      const int i = 4 * s4 + k4;
      a[s4] = (i < n && row < N) ? Qg[(int64_t)i * N + row] : 0.0;
    }
    d4v acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < MMAX / 4; ++s4) {
      const double* sp = &St[(4 * s4 + k4) * SP + j16];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nblk) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], sp[16 * c], acc[c], 0, 0, 0);
    }
    // C/D of the f64 form: register g holds row k4 + 4 g, column j16
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * t + k4 + 4 * g, k = 16 * c + j16;
        if (r < N && k < K) Vb[(int64_t)r * K + k] = (float)acc[c][g];
      }
  }
  LNZ_PROBE(9);
  if (info && tid == 0) info[b] = steps;
}

}  // namespace

// Krylov basis [B][MMAX][N] fp64
extern "C" int64_t lnz_lanczos_ritz_large_workspace_bytes(int B, int N) {
  return (int64_t)B * MMAX * N * (int64_t)sizeof(double);
}

static int launch_large(const float* A, int64_t stride_b, int64_t stride_r, int B, int N, int M,
                        int K, void* workspace, float* D, float* V, int32_t* info,
                        lnz_stream_t stream, bool sym, const char* who,
                        const int32_t* n_nodes = nullptr, const int32_t* gate = nullptr) {
  LNZ_REQUIRE(A && workspace && D && V && B > 0 && N > 0 && M > 0 && K > 0, LNZ_EINVAL,
              "%s: bad arguments (B=%d N=%d M=%d K=%d)", who, B, N, M, K);
  LNZ_REQUIRE(N <= NCH * 256 && M <= MMAX && K <= M, LNZ_ENOTSUP,
              "%s: N=%d <= 2048, K=%d <= M=%d <= 64 required", who, N, K, M);
  LNZ_REQUIRE(N % 4 == 0 && stride_r % 4 == 0 && stride_b % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(A) & 15) == 0,
              LNZ_ENOTSUP, "%s: rows must be contiguous, 16-byte aligned, N %% 4 == 0", who);
  LNZ_REQUIRE(M <= N, LNZ_EINVAL, "%s: M=%d > N=%d", who, M, N);
  LNZ_REQUIRE(!sym || (stride_r > 0 && ((int64_t)(N - 1) * stride_r + N) * 4 < (int64_t)0xffffffff),
              LNZ_ENOTSUP, "%s: one graph must span less than 4 GiB (row stride %lld)", who,
              (long long)stride_r);
  double* basis = (double*)workspace;
  const EllImage none = {nullptr, nullptr, nullptr, 0};
  if (sym)
    hipLaunchKernelGGL(lanczos_ritz_large_kernel<1>, dim3(B), dim3(TPB), 0, (hipStream_t)stream, A,
                       stride_b, stride_r, N, M, K, basis, D, V, info, n_nodes, gate, 1, none);
  else
    hipLaunchKernelGGL(lanczos_ritz_large_kernel<0>, dim3(B), dim3(TPB), 0, (hipStream_t)stream, A,
                       stride_b, stride_r, N, M, K, basis, D, V, info, n_nodes, gate, 1, none);
  return lnz::check_launch(who);
}

// ---- the K-step entry: the reference's `eigsh` branch for graphs beyond one workgroup's reach ------
static inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }
struct KstepLayout {
  int64_t basis, over, widths, vals, cols, rowcnt, total;
};
static KstepLayout kstep_layout(int B, int N, int flags, int row_cap) {
  KstepLayout L;
  const int64_t nslab = (N + 63) / 64;
  L.basis = 0;
  int64_t at = align256((int64_t)B * MMAX * N * (int64_t)sizeof(double));
  L.over = L.widths = L.vals = L.cols = L.rowcnt = at;
  if (flags & LNZ_KSTEP_COMPACT) {
    L.over = at;
    at = align256(at + (int64_t)B * 4);
    L.widths = at;
    at = align256(at + (int64_t)B * nslab * 4);
    L.vals = at;
    at = align256(at + (int64_t)B * nslab * row_cap * 64 * 4);
    L.cols = at;
    at = align256(at + (int64_t)B * nslab * row_cap * 64 * 2);
    L.rowcnt = at;
    at = align256(at + (int64_t)B * N * 4);
  }
  L.total = at;
  return L;
}

extern "C" int64_t lnz_lanczos_ritz_kstep_workspace_bytes(int B, int N, int flags, int row_cap) {
  if (B <= 0 || N <= 0 || row_cap < 0) return 0;
  return kstep_layout(B, N, flags, row_cap).total;
}

static int kstep_launch(const char* who, const float* A, int64_t stride_b, int64_t stride_r,
                        int64_t stride_c, const int32_t* n_nodes, int B, int N, int M, int K, int flags,
                        int row_cap, void* workspace, int64_t workspace_bytes, float* D, float* V,
                        int32_t* info, int32_t* dense_fallback, ConvImageOut cv, lnz_stream_t stream) {
  const bool sym = (flags & LNZ_KSTEP_SYMMETRIC) != 0;
  LNZ_REQUIRE(stride_c == 1 || (stride_c == 2 && (flags & LNZ_KSTEP_COMPACT) && dense_fallback),
              LNZ_ENOTSUP,
              "%s: stride_c=%lld: columns are contiguous, or (LNZ_KSTEP_COMPACT with dense_fallback) A is "
              "channel 0 of a channels-last [N][N][2] block", who, (long long)stride_c);
  if (!(flags & LNZ_KSTEP_COMPACT)) {
    LNZ_REQUIRE(workspace_bytes >= kstep_layout(B, N, flags, 0).total, LNZ_EINVAL,
                "%s: workspace of %lld bytes, %lld needed", who, (long long)workspace_bytes,
                (long long)kstep_layout(B, N, flags, 0).total);
    LNZ_REQUIRE(!dense_fallback, LNZ_EINVAL, "%s: dense_fallback is an output of LNZ_KSTEP_COMPACT", who);
    return launch_large(A, stride_b, stride_r, B, N, M, K, workspace, D, V, info, stream, sym, who, n_nodes);
  }
  LNZ_REQUIRE(row_cap >= ELL_UNROLL && row_cap % ELL_UNROLL == 0 && row_cap <= 1024, LNZ_EINVAL,
              "%s: row_cap=%d must be a multiple of %d in [%d, 1024]", who, row_cap, ELL_UNROLL, ELL_UNROLL);
  LNZ_REQUIRE(A && workspace && B > 0 && N > 0, LNZ_EINVAL, "%s: bad arguments", who);
  const KstepLayout L = kstep_layout(B, N, flags, row_cap);
  LNZ_REQUIRE(workspace_bytes >= L.total, LNZ_EINVAL, "%s: workspace of %lld bytes, %lld needed", who,
              (long long)workspace_bytes, (long long)L.total);
  LNZ_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, LNZ_EINVAL, "%s: workspace alignment", who);
  char* ws = (char*)workspace;
  int32_t* over = dense_fallback ? dense_fallback : (int32_t*)(ws + L.over);
  int32_t* widths = (int32_t*)(ws + L.widths);
  float* vals = (float*)(ws + L.vals);
  uint16_t* cols = (uint16_t*)(ws + L.cols);
  // the dense launcher's argument checks first (nothing is launched when they fail)
  LNZ_REQUIRE(D && V && M > 0 && K > 0, LNZ_EINVAL, "%s: bad arguments (B=%d N=%d M=%d K=%d)", who, B, N, M, K);
  LNZ_REQUIRE(N <= NCH * 256 && M <= MMAX && K <= M, LNZ_ENOTSUP,
              "%s: N=%d <= 2048, K=%d <= M=%d <= 64 required", who, N, K, M);
  LNZ_REQUIRE(N % 4 == 0 && stride_r % 4 == 0 && stride_b % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(A) & 15) == 0,
              LNZ_ENOTSUP, "%s: rows must be contiguous, 16-byte aligned, N %% 4 == 0", who);
  LNZ_REQUIRE(M <= N, LNZ_EINVAL, "%s: M=%d > N=%d", who, M, N);
  int32_t* rowcnt = (int32_t*)(ws + L.rowcnt);
  const int nslab = (N + 63) / 64;
  // (over and widths are neighbours in the workspace unless the caller keeps `over`: two memsets)
  if (hipMemsetAsync(over, 0, (size_t)B * 4, (hipStream_t)stream) != hipSuccess ||
      hipMemsetAsync(widths, 0, (size_t)B * nslab * 4, (hipStream_t)stream) != hipSuccess ||
      (cv.ent && hipMemsetAsync(cv.flags, 0, 4, (hipStream_t)stream) != hipSuccess)) {
    lnz::set_error("%s: hipMemsetAsync failed", who);
    return LNZ_ELAUNCH;
  }
  const int64_t rows = (int64_t)B * N;
  LNZ_REQUIRE(stride_c == 1 || N % 2 == 0, LNZ_ENOTSUP, "%s: stride_c = 2 needs an even N", who);
  if (stride_c == 2)
    hipLaunchKernelGGL(ell_compact_rows_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, A, stride_b, stride_r, B, N, row_cap, vals, cols, widths, rowcnt, over, cv);
  else
    hipLaunchKernelGGL(ell_compact_rows_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, A, stride_b, stride_r, B, N, row_cap, vals, cols, widths, rowcnt, over, cv);
  int rc = lnz::check_launch(who);
  if (rc != LNZ_OK) return rc;
  hipLaunchKernelGGL(ell_pad_kernel, dim3((unsigned)(((int64_t)B * nslab + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, B, N, row_cap, vals, cols, widths, rowcnt);
  rc = lnz::check_launch(who);
  if (rc != LNZ_OK) return rc;
  const EllImage img = {vals, cols, widths, row_cap};
  hipLaunchKernelGGL(lanczos_ritz_large_kernel<2>, dim3(B), dim3(TPB), 0, (hipStream_t)stream, A,
                     stride_b, stride_r, N, M, K, (double*)workspace, D, V, info, n_nodes,
                     (const int32_t*)over, 0, img);
  rc = lnz::check_launch(who);
  if (rc != LNZ_OK) return rc;
  // graphs the image could not hold: the dense stream (its workgroups leave at once otherwise).
  // The streams read contiguous rows: with stride_c = 2 the caller looks at dense_fallback instead.
  if (stride_c != 1) return LNZ_OK;
  return launch_large(A, stride_b, stride_r, B, N, M, K, workspace, D, V, info, stream, sym, who, n_nodes,
                      (const int32_t*)over);
}

extern "C" int lnz_lanczos_ritz_kstep(const float* A, int64_t stride_b, int64_t stride_r,
                                      int64_t stride_c, const int32_t* n_nodes, int B, int N, int M,
                                      int K, int flags, int row_cap, void* workspace,
                                      int64_t workspace_bytes, float* D, float* V, int32_t* info,
                                      int32_t* dense_fallback, lnz_stream_t stream) {
  return kstep_launch("lnz_lanczos_ritz_kstep", A, stride_b, stride_r, stride_c, n_nodes, B, N, M, K,
                      flags, row_cap, workspace, workspace_bytes, D, V, info, dense_fallback,
                      ConvImageOut{nullptr, nullptr, nullptr, nullptr, 0}, stream);
}

extern "C" int lnz_lanczos_ritz_kstep_image(const float* A, int64_t stride_b, int64_t stride_r,
                                            int64_t stride_c, const int32_t* n_nodes, int B, int N,
                                            int M, int K, int flags, int row_cap, void* workspace,
                                            int64_t workspace_bytes, float* D, float* V,
                                            int32_t* info, int32_t* dense_fallback,
                                            uint32_t* conv_entries, float* conv_values,
                                            int32_t* conv_counts, int conv_row_cap,
                                            int32_t* conv_flags, lnz_stream_t stream) {
  const char* who = "lnz_lanczos_ritz_kstep_image";
  LNZ_REQUIRE(flags & LNZ_KSTEP_COMPACT, LNZ_EINVAL, "%s: needs LNZ_KSTEP_COMPACT", who);
  LNZ_REQUIRE(conv_entries && conv_counts && conv_flags && conv_row_cap >= 32 && conv_row_cap % 8 == 0,
              LNZ_EINVAL, "%s: conv image outputs (row capacity a multiple of 8, at least 32)", who);
  return kstep_launch(who, A, stride_b, stride_r, stride_c, n_nodes, B, N, M, K, flags, row_cap,
                      workspace, workspace_bytes, D, V, info, dense_fallback,
                      ConvImageOut{conv_entries, conv_values, conv_counts, conv_flags, conv_row_cap}, stream);
}

extern "C" int lnz_lanczos_ritz_large(const float* A, int64_t stride_b, int64_t stride_r, int B,
                                      int N, int M, int K, void* workspace, float* D, float* V,
                                      int32_t* info, lnz_stream_t stream) {
  return launch_large(A, stride_b, stride_r, B, N, M, K, workspace, D, V, info, stream, false,
                      "lnz_lanczos_ritz_large");
}

extern "C" int lnz_lanczos_ritz_large_sym(const float* A, int64_t stride_b, int64_t stride_r,
                                          int B, int N, int M, int K, void* workspace, float* D,
                                          float* V, int32_t* info, lnz_stream_t stream) {
  return launch_large(A, stride_b, stride_r, B, N, M, K, workspace, D, V, info, stream, true,
                      "lnz_lanczos_ritz_large_sym");
}

#ifdef LNZ_LARGE_PROBE
extern "C" int lnz_debug_large_probe(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_large_probe), sizeof(unsigned long long) * 16) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_large_probe), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
