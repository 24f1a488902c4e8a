// Operand packing into v_mfma_f32_32x32x2_f32 fragment order, error state, ABI version.
#include "common.hpp"
#include "prep.hpp"
#include "split_pack.hpp"

namespace lnz {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
static thread_local char g_kernel[160] = "";
void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}
}  // namespace lnz

extern "C" int lnz_abi_version(void) { return LNZ_ABI_VERSION; }
extern "C" const char* lnz_last_error(void) { return lnz::g_err; }
extern "C" const char* lnz_last_kernel(void) { return lnz::g_kernel; }

// ---------------------------------------------------------------------------------------
// Wp[rt][q][lane][u] = W[32 rt + (lane & 31)][8 q + 4 (lane >> 5) + u]
// One thread per output float4; the write side is perfectly coalesced (16 B per lane,
// 1 KiB per wave), the read side walks 4 consecutive floats of one row.
// ---------------------------------------------------------------------------------------
__global__ void pack_rows_k8_kernel(const float* __restrict__ W, int rows, int cols, int64_t ld,
                                    int RT, int Q, float4* __restrict__ Wp) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (rt, q, lane)
  int64_t total = (int64_t)RT * Q * 64;
  if (idx >= total) return;
  int lane = (int)(idx & 63);
  int q = (int)((idx >> 6) % Q);
  int rt = (int)((idx >> 6) / Q);
  int row = 32 * rt + (lane & 31);
  int col0 = 8 * q + 4 * (lane >> 5);
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int col = col0 + u;
    v[u] = (row < rows && col < cols) ? W[(int64_t)row * ld + col] : 0.0f;
  }
  Wp[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

extern "C" int64_t lnz_packed_rows_k8_size(int rows, int cols) {
  int64_t RT = (rows + 31) / 32, Q = (cols + 7) / 8;
  return RT * Q * 64 * 4;
}

extern "C" int lnz_pack_rows_k8(const float* W, int rows, int cols, int64_t ld, float* Wp,
                                lnz_stream_t stream) {
  LNZ_REQUIRE(W && Wp && rows > 0 && cols > 0 && ld >= cols, LNZ_EINVAL,
              "lnz_pack_rows_k8: bad arguments (rows=%d cols=%d ld=%lld)", rows, cols,
              (long long)ld);
  int RT = (rows + 31) / 32, Q = (cols + 7) / 8;
  int64_t total = (int64_t)RT * Q * 64;
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_rows_k8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, rows,
                     cols, ld, RT, Q, (float4*)Wp);
  return lnz::check_launch("lnz_pack_rows_k8");
}

// The same weight stream for the split-precision GEMM1 of the strip kernel (gemm_mode 1,
// conv_strip.hip): same size and the same (rt, 16-k step, lane slot) indexing as lnz_pack_rows_k8 —
// the kernel's four-slot ring walks it unchanged —, but the two slots of a 32-k block hold that
// block's fp16 hi pieces, then its lo pieces: slot (rt, 2 b + piece), lane slot t = 64 (kq >> 1) +
// 32 (kq & 1) + wj  ->  eight halves of W[32 rt + wj][32 b + 8 kq + e], e = 0..7: the B operand of
// v_mfma_f32_16x16x32_f16 for lane (j, kq).
__global__ void pack_rows_k8_split_kernel(const float* __restrict__ W, int rows, int cols, int64_t ld,
                                          int RT, int NB, uint4* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (rt, b, piece, slot)
  int64_t total = (int64_t)RT * NB * 256;
  if (idx >= total) return;
  const int t = (int)(idx & 127), piece = (int)((idx >> 7) & 1);
  const int b = (int)((idx >> 8) % NB), rt = (int)((idx >> 8) / NB);
  const int kq = 2 * (t >> 6) + ((t >> 5) & 1);
  const int row = 32 * rt + (t & 31);
  unsigned w[4];
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2) {
    unsigned short h[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = 2 * e2 + u;
      const int col = 32 * b + 8 * kq + e;
      const float x = (row < rows && col < cols) ? W[(int64_t)row * ld + col] : 0.0f;
      const _Float16 xh = (_Float16)x;
      const _Float16 v = piece ? (_Float16)(x - (float)xh) : xh;
      h[u] = __builtin_bit_cast(unsigned short, v);
    }
    w[e2] = (unsigned)h[0] | ((unsigned)h[1] << 16);
  }
  out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

extern "C" int lnz_pack_rows_k8_split(const float* W, int rows, int cols, int64_t ld, float* Wp,
                                      lnz_stream_t stream) {
  LNZ_REQUIRE(W && Wp && rows > 0 && cols > 0 && cols % 32 == 0 && ld >= cols, LNZ_EINVAL,
              "lnz_pack_rows_k8_split: bad arguments (rows=%d cols=%d ld=%lld; cols must be a "
              "multiple of 32)", rows, cols, (long long)ld);
  int RT = (rows + 31) / 32, NB = cols / 32;
  int64_t total = (int64_t)RT * NB * 256;
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_rows_k8_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W,
                     rows, cols, ld, RT, NB, (uint4*)Wp);
  return lnz::check_launch("lnz_pack_rows_k8_split");
}

__global__ __launch_bounds__(256) void split_laplacian_pack_kernel(const float4* p, float4* dst, int64_t n4) {
  lnz::split_pack_chunk<256>(p, dst, n4, blockIdx.x, threadIdx.x);
}

extern "C" int lnz_split_laplacian_pack_to(const float* Lp, int64_t n_floats, uint16_t* dst,
                                           lnz_stream_t stream) {
  LNZ_REQUIRE(Lp && dst && n_floats > 0 && n_floats % 4 == 0, LNZ_EINVAL,
              "lnz_split_laplacian_pack: bad arguments (n_floats=%lld)", (long long)n_floats);
  LNZ_REQUIRE(((reinterpret_cast<uintptr_t>(Lp) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, LNZ_EINVAL,
              "lnz_split_laplacian_pack: 16-byte aligned buffers required");
  const int64_t n4 = n_floats / 4, blocks = (n4 + lnz::kSplitChunk - 1) / lnz::kSplitChunk;
  LNZ_REQUIRE(blocks < (1ll << 31), LNZ_ENOTSUP, "lnz_split_laplacian_pack: pack too large");
  hipLaunchKernelGGL(split_laplacian_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)Lp, (float4*)dst, n4);
  return lnz::check_launch("lnz_split_laplacian_pack");
}

extern "C" int lnz_split_laplacian_pack(float* Lp, int64_t n_floats, lnz_stream_t stream) {
  return lnz_split_laplacian_pack_to(Lp, n_floats, reinterpret_cast<uint16_t*>(Lp), stream);
}

// bp[rt][lane][r] = bias[32 rt + cd_row(r, lane >> 5)]
__global__ void pack_bias_rows_kernel(const float* __restrict__ bias, int rows, int RT,
                                      float* __restrict__ bp) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (rt, lane, r)
  if (idx >= RT * 64 * 16) return;
  int r = idx & 15;
  int lane = (idx >> 4) & 63;
  int rt = idx >> 10;
  int row = 32 * rt + lnz::cd_row(r, lane >> 5);
  bp[idx] = row < rows ? bias[row] : 0.0f;
}

extern "C" int lnz_pack_bias_rows(const float* bias, int rows, float* bp, lnz_stream_t stream) {
  LNZ_REQUIRE(bias && bp && rows > 0, LNZ_EINVAL, "lnz_pack_bias_rows: bad arguments");
  int RT = (rows + 31) / 32;
  int total = RT * 64 * 16;
  hipLaunchKernelGGL(pack_bias_rows_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, bias, rows, RT, bp);
  return lnz::check_launch("lnz_pack_bias_rows");
}

__global__ __launch_bounds__(256) void pack_laplacian_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int N, int C,
    float4* __restrict__ Lp, uint32_t* __restrict__ ident) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  pack_laplacian_body(L, sb, sr, sc, sch, N, C, Lp, tile, blockIdx.x, ident);
}

extern "C" int lnz_pack_laplacian_ident(const float* L, int64_t stride_b, int64_t stride_r,
                                        int64_t stride_c, int64_t stride_ch, int B, int N, int C,
                                        float* Lp, uint32_t* ident, lnz_stream_t stream) {
  LNZ_REQUIRE(L && Lp && B > 0 && C > 0 && C <= LNZ_MAX_CHANNELS, LNZ_EINVAL,
              "lnz_pack_laplacian: bad arguments (B=%d C=%d)", B, C);
  LNZ_REQUIRE(N > 0 && N <= LNZ_TILE, LNZ_ENOTSUP,
              "lnz_pack_laplacian: N=%d exceeds the %d-node tile built in this version", N,
              LNZ_TILE);
  size_t lds = (size_t)N * N * C * sizeof(float);
  LNZ_REQUIRE(lds <= 64 * 1024, LNZ_ENOTSUP,
              "lnz_pack_laplacian: N*N*C*4 = %zu B exceeds the 64 KiB staging tile", lds);
  hipLaunchKernelGGL(pack_laplacian_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, L,
                     stride_b, stride_r, stride_c, stride_ch, N, C, (float4*)Lp, ident);
  return lnz::check_launch("lnz_pack_laplacian");
}

extern "C" int lnz_pack_laplacian(const float* L, int64_t stride_b, int64_t stride_r,
                                  int64_t stride_c, int64_t stride_ch, int B, int N, int C,
                                  float* Lp, lnz_stream_t stream) {
  return lnz_pack_laplacian_ident(L, stride_b, stride_r, stride_c, stride_ch, B, N, C, Lp, nullptr,
                                  stream);
}

__global__ __launch_bounds__(1024) void plan_tiles_kernel(const uint8_t* __restrict__ mask, int B,
                                                          int N, int n_cu, int allow_pairs,
                                                          int wg_cap, int32_t* __restrict__ plan,
                                                          int32_t* __restrict__ n_wg, int K,
                                                          int32_t* __restrict__ gain_rows,
                                                          int32_t* __restrict__ n_gain_rows,
                                                          int32_t* __restrict__ strips,
                                                          int32_t* __restrict__ n_strips) {
  // two workgroups: the tile plan (+ live eigen slots) and the strip plan are independent chains
  // of scans over the mask — side by side instead of behind each other
  __shared__ __attribute__((aligned(16))) unsigned char strip_scratch[kStripScratch];
  if (blockIdx.x == 0) {
    if (plan)
      plan_tiles_body(mask, B, N, n_cu, allow_pairs, wg_cap, plan, n_wg, K, gain_rows, n_gain_rows);
  } else if (strips) {
    plan_strips_body(mask, B, N, n_cu, strips, n_strips, strip_scratch);
  }
}

// Both byte movers that precede the Lanczos kernel in ONE launch: workgroups 0 and 1 plan the batch
// (tile plan, strip plan), workgroups 2..B+1 pack a molecule's Laplacian tile each — the single-workgroup planner (~15 us of
// latency-bound work) runs under the packing instead of behind it.
__global__ __launch_bounds__(1024) void pack_plan_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int N, int C,
    float4* __restrict__ Lp, const uint8_t* __restrict__ mask, int B, int n_cu, int allow_pairs,
    int wg_cap, int32_t* __restrict__ plan, int32_t* __restrict__ n_wg, int K,
    int32_t* __restrict__ gain_rows, int32_t* __restrict__ n_gain_rows,
    uint32_t* __restrict__ ident, int32_t* __restrict__ strips, int32_t* __restrict__ n_strips) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ __attribute__((aligned(16))) unsigned char strip_scratch[kStripScratch];
  if (blockIdx.x < 2) {  // dispatched first: the planners' latency chains start immediately
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0)
      plan_tiles_body(mask, B, N, n_cu, allow_pairs, wg_cap, plan, n_wg, K, gain_rows, n_gain_rows);
    else if (strips)
      plan_strips_body(mask, B, N, n_cu, strips, n_strips, strip_scratch);
  } else {
    pack_laplacian_body(L, sb, sr, sc, sch, N, C, Lp, tile, (int)blockIdx.x - 2, ident);
  }
}

extern "C" int lnz_pack_laplacian_plan(const float* L, int64_t stride_b, int64_t stride_r,
                                       int64_t stride_c, int64_t stride_ch, int B, int N, int C,
                                       float* Lp, const uint8_t* mask, int n_cu, int allow_pairs,
                                       int32_t* plan, int32_t* n_wg, int K, int32_t* gain_rows,
                                       int32_t* n_gain_rows, uint32_t* ident, int32_t* strips,
                                       int32_t* n_strips, lnz_stream_t stream) {
  LNZ_REQUIRE(L && Lp && mask && plan && n_wg && B > 0 && C > 0 && C <= LNZ_MAX_CHANNELS &&
                  n_cu > 0,
              LNZ_EINVAL, "lnz_pack_laplacian_plan: bad arguments (B=%d C=%d n_cu=%d)", B, C, n_cu);
  LNZ_REQUIRE(N > 0 && N <= LNZ_TILE, LNZ_ENOTSUP, "lnz_pack_laplacian_plan: N=%d > %d", N,
              LNZ_TILE);
  LNZ_REQUIRE(!gain_rows || (n_gain_rows && K > 0), LNZ_EINVAL,
              "lnz_pack_laplacian_plan: gain_rows needs n_gain_rows and K > 0");
  size_t lds = (size_t)N * N * C * sizeof(float);
  LNZ_REQUIRE(lds <= 40 * 1024, LNZ_ENOTSUP,
              "lnz_pack_laplacian_plan: N*N*C*4 = %zu B exceeds the 40 KiB staging tile", lds);
  LNZ_REQUIRE(!strips || n_strips, LNZ_EINVAL, "lnz_pack_laplacian_plan: strips need n_strips");
  hipLaunchKernelGGL(pack_plan_kernel, dim3(B + 2), dim3(1024), lds, (hipStream_t)stream, L,
                     stride_b, stride_r, stride_c, stride_ch, N, C, (float4*)Lp, mask, B, n_cu,
                     allow_pairs, lnz_plan_wg_cap(B, n_cu), plan, n_wg, K, gain_rows, n_gain_rows,
                     ident, strips, n_strips);
  return lnz::check_launch("lnz_pack_laplacian_plan");
}

extern "C" int lnz_plan_wg_cap(int B, int n_cu) {
  if (B <= 0 || n_cu <= 0) return 0;
  return B < n_cu ? B : ((B + 4 * n_cu - 1) / (4 * n_cu)) * n_cu;
}

extern "C" int lnz_plan_batch(const uint8_t* mask, int B, int N, int n_cu, int allow_pairs,
                              int32_t* plan, int32_t* n_wg, int K, int32_t* gain_rows,
                              int32_t* n_gain_rows, int32_t* strips, int32_t* n_strips,
                              lnz_stream_t stream) {
  LNZ_REQUIRE(mask && plan && n_wg && B > 0 && N > 0 && N <= LNZ_TILE && n_cu > 0, LNZ_EINVAL,
              "lnz_plan_batch: bad arguments (B=%d N=%d n_cu=%d)", B, N, n_cu);
  LNZ_REQUIRE(!gain_rows || (n_gain_rows && K > 0), LNZ_EINVAL,
              "lnz_plan_batch: gain_rows needs n_gain_rows and K > 0");
  LNZ_REQUIRE(!strips || n_strips, LNZ_EINVAL, "lnz_plan_batch: strips need n_strips");
  hipLaunchKernelGGL(plan_tiles_kernel, dim3(2), dim3(1024), 0, (hipStream_t)stream, mask, B, N,
                     n_cu, allow_pairs, lnz_plan_wg_cap(B, n_cu), plan, n_wg, K, gain_rows,
                     n_gain_rows, strips, n_strips);
  return lnz::check_launch("lnz_plan_batch");
}

extern "C" int lnz_strip_cap(int B) {  // every chunk of LNZ_STRIP_MAX_B molecules: <= kStripBins strips
  if (B <= 0) return 0;
  const int full = B / LNZ_STRIP_MAX_B, rest = B - full * LNZ_STRIP_MAX_B;
  return full * kStripBins + (rest < kStripBins ? rest : kStripBins);
}

extern "C" int lnz_plan_strips(const uint8_t* mask, int B, int N, int n_cu, int32_t* strips,
                               int32_t* n_strips, lnz_stream_t stream) {
  LNZ_REQUIRE(mask && strips && n_strips && B > 0 && N > 0 && n_cu > 0, LNZ_EINVAL,
              "lnz_plan_strips: bad arguments (B=%d N=%d n_cu=%d)", B, N, n_cu);
  LNZ_REQUIRE(N <= LNZ_TILE, LNZ_ENOTSUP, "lnz_plan_strips: built for N <= %d (N=%d)", LNZ_TILE, N);
  hipLaunchKernelGGL(plan_tiles_kernel, dim3(2), dim3(1024), 0, (hipStream_t)stream, mask, B, N,
                     n_cu, 0, 0, (int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr,
                     (int32_t*)nullptr, strips, n_strips);
  return lnz::check_launch("lnz_plan_strips");
}

extern "C" int lnz_plan_tiles(const uint8_t* mask, int B, int N, int n_cu, int allow_pairs,
                              int32_t* plan, int32_t* n_wg, lnz_stream_t stream) {
  return lnz_plan_batch(mask, B, N, n_cu, allow_pairs, plan, n_wg, 0, nullptr, nullptr, nullptr,
                        nullptr, stream);
}

// A HIP stream confined to compute units [first_cu, end_cu) of the current device (see the header).
extern "C" int lnz_stream_create_cu_masked(int first_cu, int end_cu, lnz_stream_t* stream) {
  LNZ_REQUIRE(stream, LNZ_EINVAL, "lnz_stream_create_cu_masked: stream is NULL");
  int dev = 0, n_cu = 0;
  LNZ_REQUIRE(hipGetDevice(&dev) == hipSuccess &&
                  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess,
              LNZ_ELAUNCH, "lnz_stream_create_cu_masked: no device");
  LNZ_REQUIRE(0 <= first_cu && first_cu < end_cu && end_cu <= n_cu, LNZ_EINVAL,
              "lnz_stream_create_cu_masked: compute units [%d, %d) of %d", first_cu, end_cu, n_cu);
  uint32_t words[16] = {0};
  const int n_words = (n_cu + 31) / 32;
  LNZ_REQUIRE(n_words <= 16, LNZ_ENOTSUP, "lnz_stream_create_cu_masked: %d compute units", n_cu);
  for (int cu = first_cu; cu < end_cu; ++cu) words[cu / 32] |= 1u << (cu % 32);
  hipStream_t s = nullptr;
  LNZ_REQUIRE(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, words) == hipSuccess, LNZ_ELAUNCH,
              "lnz_stream_create_cu_masked: hipExtStreamCreateWithCUMask failed");
  *stream = (lnz_stream_t)s;
  return LNZ_OK;
}
