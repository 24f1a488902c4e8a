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
// row r is stored in slot c ^ (r & 7) (the swizzle is applied to the SOURCE address — the LDS
// destination of a global_load_lds is lane-linear), which keeps the ds_read_b128 fragment reads
// (lane = row, 16 B = 8 k-values) at most 2-way conflicted.
#include "common.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kSlice = 128 * BK * 2;        // one operand slice: 128 rows x 64 fp16 = 16 KB
constexpr int kStage = 4 * kSlice;          // x_hi, x_lo, w_hi, w_lo
constexpr size_t kLds = 2 * (size_t)kStage; // two stages: 128 KB

__device__ __forceinline__ f32x16 mfma16(const f16x8 a, const f16x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Issue the global -> LDS copies of one K slice.  Wave w stages operand w (0 = x_hi, 1 = x_lo,
// 2 = w_hi, 3 = w_lo): 16 instructions of 1 KB (8 rows x 128 B) each.
__device__ __forceinline__ void stage_slice(const uint16_t* __restrict__ src, const int ld,
                                            const int k0, unsigned char* lds_op, const int lane) {
  const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = 8 * q + rsub;                       // row of the 128-row slice
    const int chunk = slot ^ (r & 7);                 // which 16-B chunk of the row lands in `slot`
    const uint16_t* g = src + (int64_t)r * ld + k0 + 8 * chunk;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)g,
        (__attribute__((address_space(3))) void*)(lds_op + q * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ f16x8 frag(const unsigned char* lds_op, const int row, const int chunk) {
  return *reinterpret_cast<const f16x8*>(lds_op + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// SPLIT_OUT = 1: write (hi, lo) fp16 planes of relu(alpha * acc + bias) — the next layer's operand;
// SPLIT_OUT = 0: write fp32 alpha * acc + bias (the last Linear).
template <int SPLIT_OUT>
__global__ __launch_bounds__(256) void f16x3_linear_kernel(
    const uint16_t* __restrict__ Xh, const uint16_t* __restrict__ Xl, const int ldx,
    const uint16_t* __restrict__ Wh, const uint16_t* __restrict__ Wl, const int ldw,
    const float* __restrict__ bias, const float alpha, const int relu, const int M, const int N,
    const int K, const int tiles_n, uint16_t* __restrict__ Oh, uint16_t* __restrict__ Ol,
    float* __restrict__ Of, const int ldo) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs, so workgroup
  // id -> (xcd, slot); an XCD's 1/8 of the tiles is a block of whole tile ROWS (they share the
  // x slices in that XCD's L2) walked column-major inside the block
  const int nwg = gridDim.x;
  const int per = nwg / 8;
  int t = blockIdx.x;
  if ((nwg & 7) == 0) t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const uint16_t* src;
  int ld;
  if (wave == 0) src = Xh + (int64_t)m0 * ldx, ld = ldx;
  else if (wave == 1) src = Xl + (int64_t)m0 * ldx, ld = ldx;
  else if (wave == 2) src = Wh + (int64_t)n0 * ldw, ld = ldw;
  else src = Wl + (int64_t)n0 * ldw, ld = ldw;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = lnz::splat16(0.0f);

  const int T = K / BK;
  stage_slice(src, ld, 0, smem + wave * kSlice, lane);
  const int arow = wr * 64 + (lane & 31), brow = wc * 64 + (lane & 31), g = lane >> 5;
  for (int kt = 0; kt < T; ++kt) {
    // slice kt has landed (the barrier's fence waits for this wave's copies) and every wave is done
    // with the other buffer, which slice kt + 1 now overwrites while slice kt is multiplied
    __syncthreads();
    unsigned char* cur = smem + (kt & 1) * kStage;
    if (kt + 1 < T) stage_slice(src, ld, (kt + 1) * BK, smem + ((kt + 1) & 1) * kStage + wave * kSlice, lane);
    const unsigned char* xh = cur, *xl = cur + kSlice, *wh = cur + 2 * kSlice, *wl = cur + 3 * kSlice;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int chunk = 2 * s + g;
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ah[u] = frag(xh, arow + 32 * u, chunk);
        al[u] = frag(xl, arow + 32 * u, chunk);
        bh[u] = frag(wh, brow + 32 * u, chunk);
        bl[u] = frag(wl, brow + 32 * u, chunk);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = mfma16(al[a], bh[b], acc[a][b]);   // small terms first
          acc[a][b] = mfma16(ah[a], bl[b], acc[a][b]);
          acc[a][b] = mfma16(ah[a], bh[b], acc[a][b]);
        }
    }
  }

  // ---- epilogue: register r of lane (j, hh) holds C[cd_row(r, hh)][j] of its 32 x 32 tile
  const int j = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + wc * 64 + 32 * b + j;
    const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + 32 * a + lnz::cd_row(r, hh);
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

extern "C" int lnz_f16x3_linear(const uint16_t* x_hi, const uint16_t* x_lo, int ldx,
                                const uint16_t* w_hi, const uint16_t* w_lo, int ldw,
                                const float* bias, float alpha, int relu, int M, int N, int K,
                                uint16_t* out_hi, uint16_t* out_lo, float* out_f32, int ldo,
                                lnz_stream_t stream) {
  LNZ_REQUIRE(x_hi && x_lo && w_hi && w_lo && M > 0 && N > 0 && K > 0, LNZ_EINVAL,
              "lnz_f16x3_linear: bad arguments (M=%d N=%d K=%d)", M, N, K);
  LNZ_REQUIRE(K % BK == 0 && ldx >= K && ldw >= K && ldx % 8 == 0 && ldw % 8 == 0, LNZ_ENOTSUP,
              "lnz_f16x3_linear: K=%d must be a multiple of %d and the rows 16-byte aligned", K, BK);
  LNZ_REQUIRE((out_hi && out_lo && !out_f32) || (out_f32 && !out_hi && !out_lo), LNZ_EINVAL,
              "lnz_f16x3_linear: give either (out_hi, out_lo) or out_f32");
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int grid = tiles_m * tiles_n;
  hipStream_t s = (hipStream_t)stream;
  // operand rows are read in whole 128-row tiles: x needs tiles_m * 128 rows, w tiles_n * 128
  // rows allocated (the pack / the previous layer's output planes provide them)
  if (out_f32) {
    auto kfn = f16x3_linear_kernel<0>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), kLds, s, x_hi, x_lo, ldx, w_hi, w_lo, ldw, bias,
                       alpha, relu, M, N, K, tiles_n, (uint16_t*)nullptr, (uint16_t*)nullptr, out_f32,
                       ldo);
  } else {
    auto kfn = f16x3_linear_kernel<1>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), kLds, s, x_hi, x_lo, ldx, w_hi, w_lo, ldw, bias,
                       alpha, relu, M, N, K, tiles_n, out_hi, out_lo, (float*)nullptr, ldo);
  }
  return lnz::check_launch("lnz_f16x3_linear");
}
