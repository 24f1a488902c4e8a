// R8: the filter MLPs of AdaLanczosNet (model/ada_lanczos_net.py:271-272; per conv layer
// Linear(K K S -> 4096) -> ReLU -> Linear(4096 -> 4096) -> ReLU -> Linear(4096 -> 4096) -> ReLU ->
// Linear(4096 -> K K S), M = batch rows) in the opt-in split-precision mode, hand-written.
//
// One launch = one Linear:   out = [relu]( alpha * (X W^T) + bias )
// with BOTH operands as two fp16 pieces (x = x_hi + x_lo, w = w_hi + w_lo; weights scaled by 2^10
// before the split so their low pieces stay out of fp16's subnormal range, alpha = 2^-10 undoes
// it) and the product as  x_hi w_hi + x_hi w_lo + x_lo w_hi  on v_mfma_f32_32x32x16_f16, fp32
// accumulate: exact to ~2^-22 relative, better than an fp32 GEMM's own rounding.
//
// Why not a library GEMM of three times the depth ([hi|hi|lo] x [w_hi|w_lo|w_hi], the r02 path):
//   * that GEMM stages SIX operand tiles per three MFMAs; here the FOUR pieces (x_hi, x_lo, w_hi,
//     w_lo) of a k-slice are staged once and feed all three products — 1.5x the flops per staged
//     byte.  At M = 1024 the grid is 256 output tiles of 128 x 128 (one per CU; nothing larger fills
//     the chip), which makes the chain L2-bandwidth bound, so bytes per flop is the lever;
//   * the epilogue applies bias + ReLU + the 2^-10 and writes the NEXT layer's operand directly as
//     (hi, lo) fp16 planes: no fp32 activation round trip through HBM and no separate split kernel
//     (41 MB per Linear at width 4096).
//
// Kernel: 256 threads = 2 x 2 wavefronts, each a 64 x 64 block of the 128 x 128 output tile as 2 x 2
// MFMA tiles (64 accumulator registers).  K runs in slices of 64: the four operand slices
// (128 rows x 128 B each) go global -> LDS with global_load_lds (16 B per lane, no staging
// registers) into one of two LDS buffers (2 x 64 KB), the next slice in flight while the current one
// is multiplied: one __syncthreads per slice.  LDS rows are 128 B = 8 chunks of 16 B; chunk c of
// row r is stored in slot c ^ ((r >> 1) & 7) (the swizzle is applied to the SOURCE address — the LDS
// destination of a global_load_lds is lane-linear).  A ds_read_b128 is served in four groups of 16
// lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...), each of which must hit 16 distinct 16-B slots
// of the 256-B bank row = (row parity, slot): with (r >> 1) in the swizzle every group does (with
// r & 7, rows 0 and 24 of a group collide: measured 2-way, 8.4 M conflict cycles per launch).
#include <type_traits>

#include "common.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef LNZ_F16X3_WN
#define LNZ_F16X3_WN 4
#endif
// slices per workgroup from which the slice-pipelined kernel is used (0: never)
#ifndef LNZ_F16X3_PIPE_MIN_SLICES
#define LNZ_F16X3_PIPE_MIN_SLICES 16
#endif
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kSlice = 128 * BK * 2;        // one operand slice: 128 rows x 64 fp16 = 16 KB
constexpr int kStage = 4 * kSlice;          // x_hi, x_lo, w_hi, w_lo
constexpr size_t kLds = 2 * (size_t)kStage; // two stages: 128 KB

__device__ __forceinline__ f32x16 mfma16(const f16x8 a, const f16x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Issue global -> LDS copies of one K slice: pieces q0 .. q0 + NQ - 1 of this wave's operand.  Wave w
// stages operand w (0 = x_hi, 1 = x_lo, 2 = w_hi, 3 = w_lo): 16 pieces of 1 KB (8 rows x 128 B) each.
template <int NQ>
__device__ __forceinline__ void stage_pieces(const uint16_t* __restrict__ src, const int ld,
                                             const int k0, unsigned char* lds_op, const int lane,
                                             const int q0) {
  const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const int q = q0 + qq;
    const int r = 8 * q + rsub;                       // row of the 128-row slice
    const int chunk = slot ^ ((r >> 1) & 7);          // which 16-B chunk of the row lands in `slot`
    const uint16_t* g = src + (int64_t)r * ld + k0 + 8 * chunk;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)g,
        (__attribute__((address_space(3))) void*)(lds_op + q * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ f16x8 frag(const unsigned char* lds_op, const int row, const int chunk) {
  return *reinterpret_cast<const f16x8*>(lds_op + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// SPLIT_OUT = 1: write (hi, lo) fp16 planes of relu(alpha * acc + bias) — the next layer's operand;
// SPLIT_OUT = 0: write fp32 alpha * acc + bias (the last Linear).
// WN = wavefronts along N (2: four waves of 64 x 64; 4: EIGHT waves of 64 x 32 — two per SIMD, so
// one wave's barrier wait / fragment reads / copy issue overlap the other's MFMAs; with one wave
// per SIMD the matrix pipe measured 40 % busy: 1400 of a slice's 3300 cycles)
// PIPE = 1 (with WN = 2: four waves of 64 x 64, one per SIMD, up to 512 registers each): the
// slice-pipelined main loop — see the comment there.  Used for workgroups with >= 16 slices.
template <int SPLIT_OUT, int WN, int PIPE>
__global__ __launch_bounds__(128 * WN) void f16x3_linear_kernel(
    const uint16_t* __restrict__ Xh, const uint16_t* __restrict__ Xl, const int ldx,
    const uint16_t* __restrict__ Wh, const uint16_t* __restrict__ Wl, const int ldw,
    const float* __restrict__ bias, const float alpha, const int relu, const int M, const int N,
    const int K, const int tiles_n, const int bh, uint16_t* __restrict__ Oh,
    uint16_t* __restrict__ Ol, float* __restrict__ Of, const int ldo, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int NW = 2 * WN;          // wavefronts
  constexpr int CT = 4 / WN;          // 32-column MFMA tiles per wave
  constexpr int PW = 64 / NW;         // copy pieces per wave and slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  // XCD-aware tile order.  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its
  // own 4 MB L2; at M = 1024 a workgroup streams 4 MB of operand pieces per Linear, so what an XCD's
  // 32 workgroups share decides the traffic beyond L2: a block of bh x bw tiles reads (bh + bw)
  // operand tile-rows instead of 2 per tile.  First version (one tile ROW per XCD: 1 + 32): 528 MB
  // per 4096 x 4096 Linear from Infinity Cache / HBM, the launch ran at that stream's rate
  // (0.135 ms); 4 x 8 blocks: 192 MB.
  int tm, tn;
  {
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (bh > 0) {
      const int bw = (nwg >> 3) / bh, bpr = tiles_n / bw;   // block width, blocks per tile row
      tm = (xcd / bpr) * bh + slot % bh;
      tn = (xcd % bpr) * bw + slot / bh;
    } else {
      tm = blockIdx.x / tiles_n;
      tn = blockIdx.x - tm * tiles_n;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // this wave's share of the copies: operand `op` (0 = x_hi, 1 = x_lo, 2 = w_hi, 3 = w_lo), pieces
  // [pq0, pq0 + PW)
  const int op = wave * PW / 16, pq0 = (wave * PW) & 15;
  const uint16_t* src;
  int ld;
  if (op == 0) src = Xh + (int64_t)m0 * ldx, ld = ldx;
  else if (op == 1) src = Xl + (int64_t)m0 * ldx, ld = ldx;
  else if (op == 2) src = Wh + (int64_t)n0 * ldw, ld = ldw;
  else src = Wl + (int64_t)n0 * ldw, ld = ldw;

  f32x16 acc[2][CT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b) acc[a][b] = lnz::splat16(0.0f);

  // split-K (gridDim.y > 1, small N: too few output tiles to fill the chip): this workgroup
  // multiplies slices [kt0, kt0 + T) and writes its raw fp32 partial tile; f16x3_reduce_kernel adds
  // the partials in a fixed order and applies alpha / bias / ReLU
  const int Tall = K / BK, nsplit = gridDim.y;
  const int kt0 = (int)((int64_t)blockIdx.y * Tall / nsplit);
  const int T = (int)((int64_t)(blockIdx.y + 1) * Tall / nsplit) - kt0;
  if constexpr (!PIPE) stage_pieces<PW>(src, ld, kt0 * BK, smem + op * kSlice, lane, pq0);
  const int arow = wr * 64 + (lane & 31), brow = wc * (32 * CT) + (lane & 31), g = lane >> 5;
  if constexpr (PIPE) {
    // ---- software pipeline at slice granularity (four waves of 64 x 64, one per SIMD) -----------
    // ALL 32 fragments of slice kt + 1 are read into a second register set and slice kt + 2 is
    // copied while the 48 MFMAs of slice kt run from registers loaded a slice earlier: 64 x 64 wave
    // tiles read 128 KB of fragments per slice instead of 192 KB, and no MFMA waits for a fragment
    // read issued in its own k-step.
    constexpr int NS = BK / 16;   // k-steps per slice
    f16x8 fa[2][NS][2][2], fb[2][NS][CT][2];   // [register set][k-step][tile][hi, lo]
    auto load_slice = [&](const unsigned char* st, auto setc) {
      constexpr int set = decltype(setc)::value;
#pragma unroll
      for (int sk = 0; sk < NS; ++sk) {
        const int chunk = 2 * sk + g;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          fa[set][sk][u][0] = frag(st, arow + 32 * u, chunk);
          fa[set][sk][u][1] = frag(st + kSlice, arow + 32 * u, chunk);
        }
#pragma unroll
        for (int u = 0; u < CT; ++u) {
          fb[set][sk][u][0] = frag(st + 2 * kSlice, brow + 32 * u, chunk);
          fb[set][sk][u][1] = frag(st + 3 * kSlice, brow + 32 * u, chunk);
        }
      }
    };
    auto clamp_k = [&](const int kt) { return (kt0 + (kt < T ? kt : T - 1)) * BK; };
    stage_pieces<PW>(src, ld, clamp_k(0), smem + op * kSlice, lane, pq0);
    stage_pieces<PW>(src, ld, clamp_k(1), smem + kStage + op * kSlice, lane, pq0);
    lnz::wait_vmcnt0();
    __syncthreads();
    load_slice(smem, std::integral_constant<int, 0>{});
    auto slice = [&](const int kt, auto setc) {
      constexpr int set = decltype(setc)::value;
      // slice kt + 1 has landed in stage set ^ 1, every wave holds slice kt in registers: stage
      // `set` is free for slice kt + 2 (past the end: the last slice again, into the idle stage)
      lnz::wait_vmcnt0();
      __syncthreads();
      stage_pieces<PW>(src, ld, clamp_k(kt + 2), smem + set * kStage + op * kSlice, lane, pq0);
      load_slice(smem + (set ^ 1) * kStage, std::integral_constant<int, set ^ 1>{});
#pragma unroll
      for (int sk = 0; sk < NS; ++sk)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < CT; ++b)
              acc[a][b] = mfma16(fa[set][sk][a][term == 0 ? 1 : 0], fb[set][sk][b][term == 1 ? 1 : 0],
                                 acc[a][b]);
      // issue order of a slice: one load per MFMA
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // global_load_lds
      }
#pragma unroll
      for (int i = 0; i < NS * (4 + 2 * CT); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // fragment read
      }
    };
    for (int kt = 0; kt < T; kt += 2) {
      slice(kt, std::integral_constant<int, 0>{});
      if (kt + 1 < T) slice(kt + 1, std::integral_constant<int, 1>{});
    }
    lnz::wait_vmcnt0();   // (the copies past the end are still landing in the idle stage)
  } else
  for (int kt = 0; kt < T; ++kt) {
    // slice kt has landed (the barrier's fence waits for this wave's copies) and every wave is done
    // with the other buffer, which slice kt + 1 overwrites while slice kt is multiplied
    lnz::wait_vmcnt0();   // explicit: see common.hpp
    __syncthreads();
    const unsigned char* cur = smem + (kt & 1) * kStage;
    unsigned char* nxt = smem + ((kt + 1) & 1) * kStage + op * kSlice;
    const bool more = kt + 1 < T;
    const unsigned char* xh = cur, *xl = cur + kSlice, *wh = cur + 2 * kSlice, *wl = cur + 3 * kSlice;
    // fragments of k-step s + 1 are read while the 12 MFMAs of k-step s run, and the 16 copy
    // instructions of the next slice are issued four per k-step BETWEEN the MFMAs: issued in one
    // block in front of them (first version) they cost ~1600 issue cycles per slice during which
    // this wave's matrix pipe idles — as long as the 1536 cycles of the slice's MFMAs themselves
    f16x8 ah[2][2], al[2][2], bh[2][CT], bl[2][CT];
    auto load_frags = [&](const int s, const int buf) {
      const int chunk = 2 * s + g;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ah[buf][u] = frag(xh, arow + 32 * u, chunk);
        al[buf][u] = frag(xl, arow + 32 * u, chunk);
      }
#pragma unroll
      for (int u = 0; u < CT; ++u) {
        bh[buf][u] = frag(wh, brow + 32 * u, chunk);
        bl[buf][u] = frag(wl, brow + 32 * u, chunk);
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int fb = s & 1;
      if (s + 1 < BK / 16) load_frags(s + 1, fb ^ 1);
      if (more) stage_pieces<PW / 4>(src, ld, (kt0 + kt + 1) * BK, nxt, lane, pq0 + (PW / 4) * s);
      // the products of one accumulator are issued apart from each other (a dependent MFMA waits
      // for its predecessor's passes); small terms first
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < CT; ++b) {
            const f16x8 xa = term == 0 ? al[fb][a] : ah[fb][a];
            const f16x8 wb = term == 1 ? bl[fb][b] : bh[fb][b];
            acc[a][b] = mfma16(xa, wb, acc[a][b]);
          }
      // issue order of a k-step: one copy instruction and two fragment reads per three MFMAs
#pragma unroll
      for (int i = 0; i < PW / 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // 3 MFMA
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // 1 VMEM (global_load_lds: load AND store)
        __builtin_amdgcn_sched_group_barrier(0x100, (4 + 2 * CT) / (PW / 4), 0);   // DS reads
      }
    }
  }

  // ---- epilogue: register r of lane (j, hh) holds C[cd_row(r, hh)][j] of its 32 x 32 tile
  const int j = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int b = 0; b < CT; ++b) {
    const int col = n0 + wc * (32 * CT) + 32 * b + j;
    const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + 32 * a + lnz::cd_row(r, hh);
        if (nsplit > 1) {
          if (row < M && col < N) part[((int64_t)blockIdx.y * M + row) * N + col] = acc[a][b][r];
          continue;
        }
        float v = fmaf(alpha, acc[a][b][r], bv);
        if (relu) v = fmaxf(v, 0.0f);
        if (row < M && col < N) {
          if (SPLIT_OUT) {
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            Oh[(int64_t)row * ldo + col] = __builtin_bit_cast(uint16_t, h);
            Ol[(int64_t)row * ldo + col] = __builtin_bit_cast(uint16_t, l);
          } else {
            Of[(int64_t)row * ldo + col] = v;
          }
        }
      }
    }
  }
}

// out = [relu](alpha * sum_s part[s] + bias), partials added in split order (deterministic)
template <int SPLIT_OUT>
__global__ __launch_bounds__(256) void f16x3_reduce_kernel(const float* __restrict__ part, int nsplit,
                                                           int M, int N, const float* __restrict__ bias,
                                                           float alpha, int relu,
                                                           uint16_t* __restrict__ Oh,
                                                           uint16_t* __restrict__ Ol,
                                                           float* __restrict__ Of, int ldo) {
  const int64_t n = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
    float a = part[i];
    for (int s = 1; s < nsplit; ++s) a += part[(int64_t)s * n + i];
    float v = fmaf(alpha, a, bias ? bias[col] : 0.0f);
    if (relu) v = fmaxf(v, 0.0f);
    if (SPLIT_OUT) {
      const _Float16 h = (_Float16)v;
      const _Float16 l = (_Float16)(v - (float)h);
      Oh[(int64_t)row * ldo + col] = __builtin_bit_cast(uint16_t, h);
      Ol[(int64_t)row * ldo + col] = __builtin_bit_cast(uint16_t, l);
    } else {
      Of[(int64_t)row * ldo + col] = v;
    }
  }
}

// fp32 [M, K] -> (hi, lo) fp16 planes [Mp, Kp] (zero beyond K; rows >= M zero)
__global__ __launch_bounds__(256) void f16x3_split_kernel(const float* __restrict__ X, int M, int K,
                                                          int64_t ldx, float scale, int Mp, int Kp,
                                                          uint16_t* __restrict__ H,
                                                          uint16_t* __restrict__ Lo) {
  const int64_t n = (int64_t)Mp * Kp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / Kp), c = (int)(i - (int64_t)r * Kp);
    const float v = (r < M && c < K) ? X[r * ldx + c] * scale : 0.0f;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    H[i] = __builtin_bit_cast(uint16_t, h);
    Lo[i] = __builtin_bit_cast(uint16_t, l);
  }
}

}  // namespace

extern "C" int lnz_f16x3_split(const float* X, int M, int K, int64_t ldx, float scale, int Mp, int Kp,
                               uint16_t* hi, uint16_t* lo, lnz_stream_t stream) {
  LNZ_REQUIRE(X && hi && lo && M > 0 && K > 0 && Mp >= M && Kp >= K, LNZ_EINVAL,
              "lnz_f16x3_split: bad arguments (M=%d K=%d Mp=%d Kp=%d)", M, K, Mp, Kp);
  const int64_t n = (int64_t)Mp * Kp;
  const int grid = (int)((n + 256 * 8 - 1) / (256 * 8) < 4096 ? (n + 256 * 8 - 1) / (256 * 8) : 4096);
  hipLaunchKernelGGL(f16x3_split_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, (hipStream_t)stream,
                     X, M, K, ldx, scale, Mp, Kp, hi, lo);
  return lnz::check_launch("lnz_f16x3_split");
}

extern "C" int lnz_f16x3_linear_splits(int M, int N, int K) {
  // output tiles of 128 x 128; below half a chip's worth of them the K range is split so that
  // about one workgroup per CU runs (N = 1056: 72 tiles -> 3 splits), each with >= 8 slices
  if (M <= 0 || N <= 0 || K < BK) return 1;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int ns = 1;
  if (tiles < 128) ns = 256 / tiles;
  const int max_by_k = (K / BK) / 8;
  ns = ns > max_by_k ? max_by_k : ns;
  ns = ns > 8 ? 8 : ns;
  return ns < 1 ? 1 : ns;
}

extern "C" int lnz_f16x3_linear(const uint16_t* x_hi, const uint16_t* x_lo, int ldx,
                                const uint16_t* w_hi, const uint16_t* w_lo, int ldw,
                                const float* bias, float alpha, int relu, int M, int N, int K,
                                uint16_t* out_hi, uint16_t* out_lo, float* out_f32, int ldo,
                                float* partials, lnz_stream_t stream) {
  LNZ_REQUIRE(x_hi && x_lo && w_hi && w_lo && M > 0 && N > 0 && K > 0, LNZ_EINVAL,
              "lnz_f16x3_linear: bad arguments (M=%d N=%d K=%d)", M, N, K);
  LNZ_REQUIRE(K % BK == 0 && ldx >= K && ldw >= K && ldx % 8 == 0 && ldw % 8 == 0, LNZ_ENOTSUP,
              "lnz_f16x3_linear: K=%d must be a multiple of %d and the rows 16-byte aligned", K, BK);
  LNZ_REQUIRE((out_hi && out_lo && !out_f32) || (out_f32 && !out_hi && !out_lo), LNZ_EINVAL,
              "lnz_f16x3_linear: give either (out_hi, out_lo) or out_f32");
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int grid = tiles_m * tiles_n;
  // block height of an XCD's share of the tiles (0: plain row-major order): the divisor of tiles_m
  // closest to the square root of the share that also tiles the grid
  int bh = 0;
  if (grid % 8 == 0) {
    const int per = grid / 8;
    for (int h = 1; h <= tiles_m && h * h <= per; ++h)
      if (tiles_m % h == 0 && per % h == 0 && tiles_n % (per / h) == 0 &&
          (tiles_m / h) * (tiles_n / (per / h)) == 8)
        bh = h;
  }
  hipStream_t s = (hipStream_t)stream;
  // operand rows are read in whole 128-row tiles: x needs tiles_m * 128 rows, w tiles_n * 128
  // rows allocated (the pack / the previous layer's output planes provide them)
  const int nsplit = partials ? lnz_f16x3_linear_splits(M, N, K) : 1;
  const dim3 g2(grid, nsplit);
  const bool pipe = LNZ_F16X3_PIPE_MIN_SLICES > 0 && (K / BK) / nsplit >= LNZ_F16X3_PIPE_MIN_SLICES;
#define LNZ_F16X3_LAUNCH(SO_, WN_, PIPE_, OH_, OL_, OF_)                                             \
  do {                                                                                               \
    auto kfn = f16x3_linear_kernel<SO_, WN_, PIPE_>;                                                 \
    LNZ_DYNAMIC_LDS(kfn, \
      kLds, "f16x3_linear.hip");                                                            \
    hipLaunchKernelGGL(kfn, g2, dim3(128 * WN_), kLds, s, x_hi, x_lo, ldx, w_hi, w_lo, ldw, bias,    \
                       alpha, relu, M, N, K, tiles_n, bh, OH_, OL_, OF_, ldo, partials);             \
  } while (0)
  if (out_f32) {
    if (pipe) LNZ_F16X3_LAUNCH(0, 2, 1, (uint16_t*)nullptr, (uint16_t*)nullptr, out_f32);
    else LNZ_F16X3_LAUNCH(0, LNZ_F16X3_WN, 0, (uint16_t*)nullptr, (uint16_t*)nullptr, out_f32);
    if (nsplit > 1)
      hipLaunchKernelGGL(f16x3_reduce_kernel<0>, dim3(1024), dim3(256), 0, s, partials, nsplit, M, N,
                         bias, alpha, relu, (uint16_t*)nullptr, (uint16_t*)nullptr, out_f32, ldo);
  } else {
    if (pipe) LNZ_F16X3_LAUNCH(1, 2, 1, out_hi, out_lo, (float*)nullptr);
    else LNZ_F16X3_LAUNCH(1, LNZ_F16X3_WN, 0, out_hi, out_lo, (float*)nullptr);
    if (nsplit > 1)
      hipLaunchKernelGGL(f16x3_reduce_kernel<1>, dim3(1024), dim3(256), 0, s, partials, nsplit, M, N,
                         bias, alpha, relu, out_hi, out_lo, (float*)nullptr, ldo);
  }
#undef LNZ_F16X3_LAUNCH
  return lnz::check_launch("lnz_f16x3_linear");
}
