// Large-graph spectral convolution on the NONZEROS of the Laplacian (BASELINE.json configs[4]:
// N = 2048 nodes, batch 256, bf16 operands / fp32 accumulate — the normalised Laplacian of a
// G(n, p = 0.01) graph is 99 % zeros) — reference model/lanczos_net_general.py:157-182, the
// node-space term  sum_e L_e (X W_e^T)  of
//
//     X' = relu( sum_e L_e (X W_e^T)  +  V [ sum_s diag(g_s) (V^T X) W_s^T ]  +  b )
//
// The streamed form (csrc/conv_large.hip) reads every entry of the packed operators once per layer:
// 2.1 GB of bf16 per layer and folded channel, 0.47 ms — for products that are zero 99 times in
// 100.  Here the dense fp32 operator is read from HBM ONCE per batch,
//
//   lnz_large_sparse_image   L [B,N,N,C] (any strides) -> the nonzeros of channel 0 row by row:
//                            ent [B][N][cap] u32 = bf16(value) << 16 | column (entry k of a row, any
//                            order but a fixed one; the value rounded exactly as the streamed
//                            form's pack rounds it), counts [B][N]; every other channel is compared
//                            with channel 0 on the way (flags bit 0: differs somewhere — the caller
//                            claimed one operator class, the reference's single-edge-type collate,
//                            dataset/graph_data.py:225-262); flags bit 1: a row holds more than cap
//                            nonzeros.  Either bit sends the batch to the streamed kernels: the
//                            caller checks.
//
// and a layer is  lnz_large_gemm1_rows (Z = X W^T, bf16, row major)  +  lnz_large_spectral  +
// lnz_large_conv with C = 0 (the lift V T + bias on the matrix pipe, no activation)  +
//
//   lnz_large_sparse_conv    X[r][:] = act( X[r][:] + sum_k value[r][k] Z[column[r][k]][:] )
//
// — the streamed form's products (bf16 x bf16, fp32 accumulate) without the zeros, in entry order.
// (The same image falls out of the K-step Lanczos entry's own pass over L: lnz_lanczos_ritz_kstep_image,
// csrc/lanczos_large.hip — the collated Laplacian is then read from HBM once per batch.)
#include "common.hpp"

#include <type_traits>

namespace {

typedef unsigned short u16;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int DH = 128;
#ifndef LNZ_SPARSE_ROWS_PER_WAVE   // (tools/experiments/build_variant.sh sweeps)
#define LNZ_SPARSE_ROWS_PER_WAVE 8
#endif
#ifndef LNZ_SPARSE_WAVES
#define LNZ_SPARSE_WAVES 4
#endif
#ifndef LNZ_SPARSE_GS
#define LNZ_SPARSE_GS 8
#endif
// A row's count is rounded up to GS (the image pads rows with zero entries up to a multiple of 8) and
// a turn of the gather is ONE uniform decision followed by a straight line of GS .. 32 loads and their
// FMAs.  With a uniform test in front of every group of GS instead (B = 256, N = 2048, 21.5 entries
// per row): GS = 8 0.275 ms; 4 0.311, 2 0.449, 1 0.700 — every test is a point the loads behind it
// wait at —; 16 0.288, 32 0.347 (padding gathers).  Straight-line turns: GS = 8 0.264, GS = 4 0.265.
constexpr int GS = LNZ_SPARSE_GS;
static_assert(GS == 4 || GS == 8, "the image pads rows to multiples of 8; a turn is at most 8 groups");
constexpr int ROWS_PER_WAVE = LNZ_SPARSE_ROWS_PER_WAVE, WAVES = LNZ_SPARSE_WAVES, TILE_ROWS = ROWS_PER_WAVE * WAVES;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// bf16(value) << 16 | column; round to nearest even (v_cvt_pk_bf16_f32), as conv_large.hip's pack
__device__ inline unsigned pack_entry(float v, int col) {
  const bf16x2 p = __builtin_convertvector(f32x2{v, 0.0f}, bf16x2);
  return ((unsigned)__builtin_bit_cast(u16, p[0]) << 16) | (unsigned)col;
}

__device__ inline int lane_rank(unsigned long long m) {  // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// ---- image: one wavefront per row ---------------------------------------------------------------
// FORM 2: channels-last pair of channels (sc == 2, sch == 1, rows 16-byte aligned): a float4 is two
// columns x two channels, 1 KiB contiguous per wave load, eight loads in flight.  FORM 1: one
// operator in contiguous rows (sc == 1 and a single channel or a zero channel stride — an expanded
// view): a float4 is four columns.  FORM 0: 4-byte loads at the given strides.
template <int FORM>
__global__ __launch_bounds__(256) void sparse_image_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int B, int N, int C,
    int cap, unsigned* __restrict__ ent, float* __restrict__ vals, int32_t* __restrict__ counts,
    int32_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t rid = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (rid >= (int64_t)B * N) return;
  const int b = (int)(rid / N), r = (int)(rid - (int64_t)b * N);
  const float* Lr = L + (int64_t)b * sb + (int64_t)r * sr;
  unsigned* oe = ent + rid * cap;
  float* ov = vals ? vals + rid * cap : nullptr;   // (the exact-fp32 form's values, optional)
  int k = 0;            // entries of this row so far (wave-uniform)
  bool differ = false;  // a channel differs from channel 0 in this lane's columns
  auto place = [&](const float v, const int col) {
    const bool nz = v != 0.0f;   // (a NaN is kept)
    const unsigned long long m = __ballot(nz);
    if (m == 0ull) return;
    const int pos = k + lane_rank(m);
    if (nz && pos < cap) {
      oe[pos] = pack_entry(v, col);
      if (ov) ov[pos] = v;
    }
    k += __popcll(m);
  };
  if constexpr (FORM == 1) {
    const f32x4* src = reinterpret_cast<const f32x4*>(Lr);
    const int nq = N >> 2;   // (N is a multiple of 4 in this form)
    for (int q0 = 0; q0 < nq; q0 += 64 * 8) {
      f32x4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + 64 * u + lane;
        x[u] = q < nq ? __builtin_nontemporal_load(src + q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + 64 * u + lane;
        if (__ballot((x[u][0] != 0.0f) | (x[u][1] != 0.0f) | (x[u][2] != 0.0f) | (x[u][3] != 0.0f)) == 0ull)
          continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) place(x[u][c], 4 * q + c);
      }
    }
  } else if constexpr (FORM == 2) {
    const f32x4* src = reinterpret_cast<const f32x4*>(Lr);
    const int nq = N >> 1;   // float4 = columns 2 q, 2 q + 1 (N is even in this form)
    for (int q0 = 0; q0 < nq; q0 += 64 * 8) {
      f32x4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + 64 * u + lane;
        x[u] = q < nq ? __builtin_nontemporal_load(src + q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + 64 * u + lane;
        differ |= (x[u][0] != x[u][1]) | (x[u][2] != x[u][3]);
        if (__ballot((x[u][0] != 0.0f) | (x[u][2] != 0.0f)) == 0ull) continue;
        place(x[u][0], 2 * q);
        place(x[u][2], 2 * q + 1);
      }
    }
  } else {
    const int nchk = sch == 0 ? 1 : C;   // (a zero channel stride: one operator by construction)
    for (int c0 = 0; c0 < N; c0 += 64 * 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int col = c0 + 64 * u + lane;
        x[u] = col < N ? Lr[(int64_t)col * sc] : 0.0f;
      }
      for (int c = 1; c < nchk; ++c) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int col = c0 + 64 * u + lane;
          const float y = col < N ? Lr[(int64_t)col * sc + (int64_t)c * sch] : 0.0f;
          differ |= x[u] != y;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) place(x[u], c0 + 64 * u + lane);
    }
  }
  const int cnt = k < cap ? k : cap;
  const int cnt8 = (cnt + 7) & ~7;   // (cap is a multiple of 8) the conv walks whole groups of eight
  if (cnt + lane < cnt8) {
    oe[cnt + lane] = 0u;
    if (ov) ov[cnt + lane] = 0.0f;
  }
  const bool any_differ = __ballot(differ) != 0ull;
  if (lane == 0) {
    counts[rid] = cnt;
    const int f = (any_differ ? 1 : 0) | (k > cap ? 2 : 0);
    if (f) atomicOr(flags, f);
  }
}

// ---- conv: X[r][:] = act( X[r][:] + sum_k value[r][k] Z[column[r][k]][:] ) ---------------------
// Workgroup = 32 rows of one graph (4 waves x 8 rows: 0.276 ms; x 16: 0.288, x 32: 0.340 — fewer
// graphs share an L2 at a time); blockIdx -> (graph, tile) deals the tiles
// of a graph to ONE XCD (workgroup i runs on XCD i % 8), whose L2 then holds the graph's Z (512
// KiB).  One row at a time per wave, lanes along the 128 features (one dword = two bf16 features per
// lane and entry: every gather is the 256 contiguous bytes of one node, its offset in an SGPR of a
// buffer load — no vector address arithmetic), the row's entries one coalesced load (lane k <-
// entry k, requested one row ahead) and broadcast by v_readlane; up to 32 gathers go out before the
// first is used.  The launch moves 256 B through the L2 -> L1 path per nonzero: 3.5 GB per layer at
// config 5 = 0.28 ms at ~12 TB/s, and that path is what bounds it (measured: 7 -> 4 vector
// instructions per entry and 8 -> 32 gathers in flight changed nothing; a form with a feature
// quarter of Z staged in LDS and ds_bpermute broadcasts was issue bound at 0.32 ms; the lift V T in
// vector FMAs inside this kernel, T in LDS, cost 0.25 ms against the 0.08 ms of the MFMA launch).
__global__ __launch_bounds__(64 * WAVES) void sparse_conv_kernel(
    const unsigned* __restrict__ ent, const int32_t* __restrict__ counts, int cap,
    const u16* __restrict__ Z, int B, int N, int tiles, int relu, float* __restrict__ X) {
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int b = xcd + 8 * (seq / tiles), tile = seq % tiles;
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r0 = tile * TILE_ROWS + wave * ROWS_PER_WAVE;
  if (r0 >= N) return;
  const int nr = min(ROWS_PER_WAVE, N - r0);
  const int64_t row0 = (int64_t)b * N + r0;
  const int cv = lane < nr ? counts[row0 + lane] : 0;
  // the graph's Z as a buffer: dword `lane` of a node's 256 B = features 2 lane, 2 lane + 1
  const __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(Z + (int64_t)b * N * DH), 0, (unsigned)N * DH * 2, 0x00020000);
  const unsigned zoff = 4u * lane;
  auto entries = [&](const int rr, const int k0, const int cnt8) -> unsigned {
    return k0 + lane < cnt8 ? ent[(row0 + rr) * cap + k0 + lane] : 0u;
  };
  int cnt8n = (__builtin_amdgcn_readlane(cv, 0) + GS - 1) & ~(GS - 1);
  unsigned en = entries(0, 0, cnt8n);
  for (int rr = 0; rr < nr; ++rr) {
    const int cnt8 = cnt8n;
    unsigned e = en;
    float* xr = X + (row0 + rr) * DH + 2 * lane;
    f32x2 acc = *reinterpret_cast<const f32x2*>(xr);
    if (rr + 1 < nr) {   // (uniform) the next row's entries
      cnt8n = (__builtin_amdgcn_readlane(cv, rr + 1) + GS - 1) & ~(GS - 1);
      en = entries(rr + 1, 0, cnt8n);
    }
    for (int k0 = 0; k0 < cnt8; k0 += 64) {
      if (k0 > 0) e = entries(rr, k0, cnt8);   // (rows of more than 64 entries)
      const int m = min(64, cnt8 - k0);
      for (int k = 0; k < m; k += 32) {
        // ONE uniform decision per turn, then a straight line of 8, 16, 24 or 32 gathers and their
        // FMAs (a test per group of eight was a point the loads behind it waited at)
        auto turn = [&](auto nc) {
          constexpr int n = decltype(nc)::value;
          unsigned z[n];
          float s[n];
#pragma unroll
          for (int u = 0; u < n; ++u) {
            const unsigned se = (unsigned)__builtin_amdgcn_readlane((int)e, k + u);
            s[u] = __uint_as_float(se & 0xffff0000u);
            z[u] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(z_rsrc, zoff, (se & 0xffffu) * (DH * 2), 0);
          }
#pragma unroll
          for (int u = 0; u < n; ++u) {
            acc[0] = fmaf(s[u], __uint_as_float(z[u] << 16), acc[0]);
            acc[1] = fmaf(s[u], __uint_as_float(z[u] & 0xffff0000u), acc[1]);
          }
        };
        switch (min(32, m - k) / GS) {   // (m - k is a multiple of GS)
#define LNZ_TURN(q) case q: if constexpr (q * GS <= 32) turn(std::integral_constant<int, (q * GS <= 32 ? q * GS : 32)>{}); break;
          LNZ_TURN(1) LNZ_TURN(2) LNZ_TURN(3) LNZ_TURN(4) LNZ_TURN(5) LNZ_TURN(6) LNZ_TURN(7) LNZ_TURN(8)
#undef LNZ_TURN
          default: break;
        }
      }
    }
    if (relu) {
      acc[0] = acc[0] > 0.0f ? acc[0] : 0.0f;
      acc[1] = acc[1] > 0.0f ? acc[1] : 0.0f;
    }
    *reinterpret_cast<f32x2*>(xr) = acc;
  }
}

// ---- the same in exact fp32 (the split-precision modes' node-space term): fp32 values x fp32
// features, Zf [B][N][128] fp32 (= lnz_f32_linear's X W^T), a lane's dwordx2 = features 2 lane,
// 2 lane + 1: 512 B through the L2 -> L1 path per nonzero (twice the bf16 form's), no unpacking.
__global__ __launch_bounds__(64 * WAVES) void sparse_conv_f32_kernel(
    const unsigned* __restrict__ ent, const float* __restrict__ vals, const int32_t* __restrict__ counts,
    int cap, const float* __restrict__ Zf, int B, int N, int tiles, int relu, float* __restrict__ X) {
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int b = xcd + 8 * (seq / tiles), tile = seq % tiles;
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r0 = tile * TILE_ROWS + wave * ROWS_PER_WAVE;
  if (r0 >= N) return;
  const int nr = min(ROWS_PER_WAVE, N - r0);
  const int64_t row0 = (int64_t)b * N + r0;
  const int cv = lane < nr ? counts[row0 + lane] : 0;
  const __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(Zf + (int64_t)b * N * DH), 0, (unsigned)N * DH * 4, 0x00020000);
  const unsigned zoff = 8u * lane;
  auto entries = [&](const int rr, const int k0, const int cnt8, unsigned& e, float& v) {
    const bool in = k0 + lane < cnt8;
    const int64_t o = (row0 + rr) * cap + k0 + lane;
    e = in ? ent[o] : 0u;
    v = in ? vals[o] : 0.0f;
  };
  int cnt8n = (__builtin_amdgcn_readlane(cv, 0) + 7) & ~7;
  unsigned en;
  float vn;
  entries(0, 0, cnt8n, en, vn);
  for (int rr = 0; rr < nr; ++rr) {
    const int cnt8 = cnt8n;
    unsigned e = en;
    float v = vn;
    float* xr = X + (row0 + rr) * DH + 2 * lane;
    f32x2 acc = *reinterpret_cast<const f32x2*>(xr);
    if (rr + 1 < nr) {   // (uniform) the next row's entries
      cnt8n = (__builtin_amdgcn_readlane(cv, rr + 1) + 7) & ~7;
      entries(rr + 1, 0, cnt8n, en, vn);
    }
    for (int k0 = 0; k0 < cnt8; k0 += 64) {
      if (k0 > 0) entries(rr, k0, cnt8, e, v);   // (rows of more than 64 entries)
      const int m = min(64, cnt8 - k0);
      for (int k = 0; k < m; k += 16) {
        auto turn = [&](auto nc) {   // (one uniform decision, then a straight line: see the bf16 form)
          constexpr int n = decltype(nc)::value;
          f32x2 z[n];
          float s[n];
#pragma unroll
          for (int u = 0; u < n; ++u) {
            const unsigned col = (unsigned)__builtin_amdgcn_readlane((int)e, k + u) & 0xffffu;
            s[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k + u));
            z[u] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(z_rsrc, zoff, col * (DH * 4), 0));
          }
#pragma unroll
          for (int u = 0; u < n; ++u) {
            acc[0] = fmaf(s[u], z[u][0], acc[0]);
            acc[1] = fmaf(s[u], z[u][1], acc[1]);
          }
        };
        if (m - k > 8) turn(std::integral_constant<int, 16>{});
        else turn(std::integral_constant<int, 8>{});
      }
    }
    if (relu) {
      acc[0] = acc[0] > 0.0f ? acc[0] : 0.0f;
      acc[1] = acc[1] > 0.0f ? acc[1] : 0.0f;
    }
    *reinterpret_cast<f32x2*>(xr) = acc;
  }
}

}  // namespace

extern "C" int lnz_large_sparse_image(const float* L, int64_t stride_b, int64_t stride_r,
                                      int64_t stride_c, int64_t stride_ch, int B, int N, int C,
                                      int row_cap, uint32_t* entries, float* values,
                                      int32_t* counts, int32_t* flags, lnz_stream_t stream) {
  LNZ_REQUIRE(L && entries && counts && flags && B > 0 && N > 0 && C > 0, LNZ_EINVAL,
              "lnz_large_sparse_image: bad arguments (B=%d N=%d C=%d)", B, N, C);
  LNZ_REQUIRE(N <= 65536, LNZ_ENOTSUP, "lnz_large_sparse_image: N=%d > 65536 (16-bit columns)", N);
  LNZ_REQUIRE(row_cap >= 32 && row_cap % 8 == 0, LNZ_EINVAL,
              "lnz_large_sparse_image: row_cap=%d must be a multiple of 8, at least 32", row_cap);
  const int64_t rows = (int64_t)B * N;
  LNZ_REQUIRE((rows + 3) / 4 <= 0x7fffffffll, LNZ_ENOTSUP, "lnz_large_sparse_image: B x N too large");
  hipStream_t s = (hipStream_t)stream;
  LNZ_REQUIRE(hipMemsetAsync(flags, 0, sizeof(int32_t), s) == hipSuccess, LNZ_ELAUNCH,
              "lnz_large_sparse_image: hipMemsetAsync failed");
  const bool aligned = (((uintptr_t)L) & 15) == 0 && stride_b % 4 == 0 && stride_r % 4 == 0;
  const bool pair = aligned && C == 2 && stride_c == 2 && stride_ch == 1 && N % 2 == 0;
  const bool rows1 = aligned && stride_c == 1 && (C == 1 || stride_ch == 0) && N % 4 == 0;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (pair)
    hipLaunchKernelGGL(sparse_image_kernel<2>, grid, dim3(256), 0, s, L, stride_b, stride_r,
                       stride_c, stride_ch, B, N, C, row_cap, entries, values, counts, flags);
  else if (rows1)
    hipLaunchKernelGGL(sparse_image_kernel<1>, grid, dim3(256), 0, s, L, stride_b, stride_r,
                       stride_c, stride_ch, B, N, C, row_cap, entries, values, counts, flags);
  else
    hipLaunchKernelGGL(sparse_image_kernel<0>, grid, dim3(256), 0, s, L, stride_b, stride_r,
                       stride_c, stride_ch, B, N, C, row_cap, entries, values, counts, flags);
  lnz::note_kernel("sparse_image_kernel<%s>", pair ? "pair" : rows1 ? "rows" : "strided");
  return lnz::check_launch("lnz_large_sparse_image");
}

extern "C" int lnz_large_sparse_conv(const uint32_t* entries, const int32_t* counts, int row_cap,
                                     const uint16_t* Z, int B, int N, int relu, float* X,
                                     lnz_stream_t stream) {
  LNZ_REQUIRE(entries && counts && Z && X && B > 0 && N > 0, LNZ_EINVAL,
              "lnz_large_sparse_conv: bad arguments");
  LNZ_REQUIRE(row_cap >= 32 && row_cap % 8 == 0, LNZ_EINVAL,
              "lnz_large_sparse_conv: row_cap=%d must be a multiple of 8, at least 32", row_cap);
  LNZ_REQUIRE((((uintptr_t)X) & 7) == 0, LNZ_EINVAL, "lnz_large_sparse_conv: X must be 8-byte aligned");
  const int tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
  const int64_t grid = (int64_t)8 * tiles * ((B + 7) / 8);
  LNZ_REQUIRE(grid <= 0x7fffffffll, LNZ_ENOTSUP, "lnz_large_sparse_conv: B x N too large");
  hipLaunchKernelGGL(sparse_conv_kernel, dim3((unsigned)grid), dim3(64 * WAVES), 0,
                     (hipStream_t)stream, entries, counts, row_cap, Z, B, N, tiles, relu, X);
  lnz::note_kernel("sparse_conv_kernel");
  return lnz::check_launch("lnz_large_sparse_conv");
}

extern "C" int lnz_large_sparse_conv_f32(const uint32_t* entries, const float* values,
                                         const int32_t* counts, int row_cap, const float* Zf, int B,
                                         int N, int relu, float* X, lnz_stream_t stream) {
  LNZ_REQUIRE(entries && values && counts && Zf && X && B > 0 && N > 0, LNZ_EINVAL,
              "lnz_large_sparse_conv_f32: bad arguments");
  LNZ_REQUIRE(row_cap >= 32 && row_cap % 8 == 0, LNZ_EINVAL,
              "lnz_large_sparse_conv_f32: row_cap=%d must be a multiple of 8, at least 32", row_cap);
  LNZ_REQUIRE((((uintptr_t)X) & 7) == 0 && (((uintptr_t)Zf) & 7) == 0, LNZ_EINVAL,
              "lnz_large_sparse_conv_f32: X / Zf must be 8-byte aligned");
  LNZ_REQUIRE((int64_t)N * DH * 4 <= 0x7fffffffll, LNZ_ENOTSUP, "lnz_large_sparse_conv_f32: N too large");
  const int tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
  const int64_t grid = (int64_t)8 * tiles * ((B + 7) / 8);
  LNZ_REQUIRE(grid <= 0x7fffffffll, LNZ_ENOTSUP, "lnz_large_sparse_conv_f32: B x N too large");
  hipLaunchKernelGGL(sparse_conv_f32_kernel, dim3((unsigned)grid), dim3(64 * WAVES), 0,
                     (hipStream_t)stream, entries, values, counts, row_cap, Zf, B, N, tiles, relu, X);
  lnz::note_kernel("sparse_conv_f32_kernel");
  return lnz::check_launch("lnz_large_sparse_conv_f32");
}
