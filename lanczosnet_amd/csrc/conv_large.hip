// Large-graph spectral convolution (BASELINE.json configs[4]: LanczosNetGeneral, N = 2048 nodes,
// K = 64 Ritz pairs, batch 256, bf16 operands / fp32 accumulate) — reference
// model/lanczos_net_general.py:157-182 at sizes where one graph's operators (E+1 dense N x N
// Laplacians, 16.8 MB each in fp32) are far beyond LDS and L2: the layer
//
//     X' = relu( sum_e L_e (X W_e^T)  +  V [ sum_s diag(g_s) (V^T X) W_s^T ]  +  b )
//
// is HBM bound on the L_e stream (2 N^2 * 128 flop per N^2 operator entries: 64 flop/B in fp32,
// 128 flop/B on packed bf16 — the bf16 matrix pipe needs 320 flop/B to saturate), so the design
// is a streaming one:
//
//   lnz_large_pack_operators   once per batch: channels-last fp32 L [B,N,N,C] -> channel-major
//                              bf16 planes Lb [P][B][C][N][Nk] (k padded to 64), V -> Vb.  The
//                              channels-last layout would make every per-channel read strided;
//                              packed bf16 halves the bytes all num_layer passes stream.
//   lnz_large_gemm1            per layer: Z_e = X W_e^T on the matrix pipe, written TRANSPOSED
//                              (Zt [P][B][C][128][Nk] bf16) — exactly the B-operand image the
//                              conv kernel stages.
//   lnz_large_spectral         per layer, one workgroup per graph, exact fp32 MFMA: Y = V^T X
//                              (K x d_in, the projection to eigen space), T = sum_s diag(g_s)
//                              (Y W_s^T) — the S long-scale channels mixed on K = 64 rows instead
//                              of N = 2048 — written as Tt [P][B][128][64] bf16.
//   lnz_large_conv             per layer: 256-row output tiles, 8 wavefronts x (32 rows x 128
//                              columns) of v_mfma_f32_16x16x32_bf16 accumulators; every wave
//                              streams ITS rows of L_e from HBM straight into A fragments
//                              (global_load_dwordx4 = one fragment, 64 B contiguous per row and
//                              instruction, double buffered), the Zt k-block (16 KB) is staged
//                              through LDS once per workgroup and feeds all 8 waves; the lift
//                              V T is one more k-block of the same loop; bias + ReLU in the
//                              epilogue.  The 8 row tiles of a graph are dealt to ONE XCD
//                              (blockIdx -> (graph, tile) map below) so that graph's Zt (1 MB)
//                              is served by that XCD's L2.
//
// P (planes) = 1: plain bf16 operands (config 5's mode; ~1e-2 after 7 layers).  P = 3: every
// fp32 operand is split into three bf16 pieces x = x0 + x1 + x2 (24 mantissa bits) and every
// product is the six piece products of order <= 2, accumulated in fp32: fp32-grade results
// (1e-6) from the same kernels at 6x the matrix work and 3x the operand bytes — the parity mode.
#include "common.hpp"

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr int DH = 128;        // hidden width (output columns of every layer)
constexpr int KB = 64;         // k-block of the conv loop (bf16 elements = one 128 B line per row)
constexpr int BP = 72;         // LDS pitch of a staged B row in bf16 (144 B: conflict-free b128)

__device__ inline __bf16 to_bf16(float x) {  // round to nearest even (v_cvt_pk_bf16_f32)
  bf16x2 p = __builtin_convertvector(f32x2{x, 0.0f}, bf16x2);
  return p[0];
}
__device__ inline float bf16_float(__bf16 b) {
  return __uint_as_float((unsigned)__builtin_bit_cast(u16, b) << 16);
}
__device__ inline u16 bits(__bf16 b) { return __builtin_bit_cast(u16, b); }

// x -> pieces p[0..P-1] with x ~= sum p[i] (each piece the bf16 rounding of the remainder)
template <int P>
__device__ inline void split_bf16(float x, __bf16* p) {
  float r = x;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    p[i] = to_bf16(r);
    r -= bf16_float(p[i]);
  }
}

// ------------------------------------------------------------------------------------------
// pack: L [B,N,N,C] fp32 (any strides) -> Lb [P][B][C][N][Nk];  V [B,N,K] -> Vb [P][B][N][64]
// One workgroup per (graph, row): reads the row's N*C floats (channels-last: contiguous),
// writes C bf16 rows.  Columns k in [N, Nk) and eigen slots in [K, 64) are zero.
// ------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void large_pack_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch,
    const float* __restrict__ V, int B, int N, int Nk, int C, int K, u16* __restrict__ Lb,
    u16* __restrict__ Vb) {
  const int r = blockIdx.x, b = blockIdx.y;
  const float* Lr = L + (int64_t)b * sb + (int64_t)r * sr;
  const int64_t plane_l = (int64_t)B * C * N * Nk;
  // thread = (8 consecutive k, channel c): channels-last source -> the C threads of a k-group read
  // one contiguous 32*C-byte piece; one 16 B store per plane into the channel-major bf16 row
  for (int idx = threadIdx.x; idx < C * (Nk / 8); idx += 256) {
    const int j = idx / C, c = idx - j * C;
    bf16x8 out[P];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = 8 * j + u;
      const float x = k < N ? Lr[(int64_t)k * sc + (int64_t)c * sch] : 0.0f;
      __bf16 p[P];
      split_bf16<P>(x, p);
#pragma unroll
      for (int i = 0; i < P; ++i) out[i][u] = p[i];
    }
    const int64_t o = (((int64_t)b * C + c) * N + r) * Nk + 8 * j;
#pragma unroll
    for (int i = 0; i < P; ++i) *reinterpret_cast<bf16x8*>(Lb + i * plane_l + o) = out[i];
  }
  const int64_t plane_v = (int64_t)B * N * 64;
  for (int k = threadIdx.x; k < 64; k += 256) {
    const float x = k < K ? V[((int64_t)b * N + r) * K + k] : 0.0f;
    __bf16 p[P];
    split_bf16<P>(x, p);
#pragma unroll
    for (int i = 0; i < P; ++i) Vb[i * plane_v + ((int64_t)b * N + r) * 64 + k] = bits(p[i]);
  }
}

// ------------------------------------------------------------------------------------------
// GEMM1: Zt[c][o][n] = sum_i W_c[o][i] X[n][i]   (= (X W_c^T)^T, the conv kernel's B image)
// Workgroup = 4 waves x 32 node rows; per channel the weight block Wb[c] (128 x dinp bf16 per
// plane) is staged in LDS and is the A operand of v_mfma_f32_32x32x16_bf16, the node rows are
// the B operand (fp32 X converted / split in registers).
// ------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void large_gemm1_kernel(
    const float* __restrict__ X, int ldx, int din, int dinp, const u16* __restrict__ Wb,
    int B, int N, int Nk, int C, u16* __restrict__ Zt) {
  extern __shared__ __attribute__((aligned(16))) u16 Ws[];  // [P][128][dinp + 8]
  const int wp = dinp + 8;
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 128 + 32 * w;
  const int n = n0 + (lane & 31), h = lane >> 5;
  const int nc = n < N ? n : N - 1;
  const float* xr = X + ((int64_t)b * N + nc) * ldx;
  const int nks = dinp / 16;
  const int64_t plane_w = (int64_t)C * DH * dinp;
  const int64_t plane_z = (int64_t)B * C * DH * Nk;
  for (int c = 0; c < C; ++c) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < P * DH * (dinp / 8); idx += 256) {
      const int q = idx % (dinp / 8), o = (idx / (dinp / 8)) % DH, p = idx / ((dinp / 8) * DH);
      const f32x4 v = *reinterpret_cast<const f32x4*>(Wb + p * plane_w + ((int64_t)c * DH + o) * dinp + 8 * q);
      *reinterpret_cast<f32x4*>(Ws + ((int64_t)p * DH + o) * wp + 8 * q) = v;
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = lnz::splat16(0.0f);
    for (int ks = 0; ks < nks; ++ks) {
      // B fragment: X[n][16 ks + 8 h .. + 7]
      bf16x8 bf[P];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = 16 * ks + 8 * h + u;
        const float x = i < din ? xr[i] : 0.0f;
        __bf16 p[P];
        split_bf16<P>(x, p);
#pragma unroll
        for (int q = 0; q < P; ++q) bf[q][u] = p[q];
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bf16x8 af[P];
#pragma unroll
        for (int q = 0; q < P; ++q)
          af[q] = *reinterpret_cast<const bf16x8*>(Ws + ((int64_t)q * DH + 32 * mt + (lane & 31)) * wp +
                                                   16 * ks + 8 * h);
        // small terms first: (order 2), (order 1), (order 0)
#pragma unroll
        for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
          for (int i = 0; i <= ord; ++i)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[ord - i], acc[mt], 0, 0, 0);
      }
    }
    if (n < N) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * mt + lnz::cd_row(r, h);
          __bf16 p[P];
          split_bf16<P>(acc[mt][r], p);
          const int64_t off = (((int64_t)b * C + c) * DH + o) * Nk + n;
#pragma unroll
          for (int q = 0; q < P; ++q) Zt[q * plane_z + off] = bits(p[q]);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Spectral block, one workgroup (8 waves) per graph, exact fp32 (v_mfma_f32_32x32x2_f32):
//   phase 1  Y[k][i] = sum_n V[n][k] X[n][i]            wave (kt, it): 32 slots x 32 columns
//   phase 2  T[k][o] = sum_s g_s[k] * sum_i Y[k][i] W_s[o][i]   wave (kt, ot); per scale the
//            unscaled product accumulates in its own C tile and its rows are scaled into T
//            (16 FMAs) — a VALU multiply in front of every MFMA costs ~40 cycles each
// Wt: [S * dinp][128] fp32 = the long-scale column blocks of the mix weight, transposed.
// Output Tt [P][B][128][64] bf16 (the B image of the conv kernel's lift block).
// ------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(512) void large_spectral_kernel(
    const float* __restrict__ X, int ldx, int din, int dinp, const float* __restrict__ V,
    const float* __restrict__ G, const float* __restrict__ Wt, int B, int N, int K, int S,
    u16* __restrict__ Tt) {
  __shared__ float Ys[64][DH + 4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int kt = w >> 2, qt = w & 3;
  // ---- phase 1
  {
    const int slot = 32 * kt + l31, col = 32 * qt + l31;
    const bool va = slot < K, vb = col < din;
    f32x16 acc = lnz::splat16(0.0f);
    if (32 * qt < dinp) {
      const float* vp = V + (int64_t)b * N * K + slot;
      const float* xp = X + (int64_t)b * N * ldx + col;
      for (int n = 0; n < N; n += 16) {
        float a[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int nn = n + 2 * u + h;
          const bool in = nn < N;
          a[u] = (va && in) ? vp[(int64_t)nn * K] : 0.0f;
          x[u] = (vb && in) ? xp[(int64_t)nn * ldx] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = lnz::mfma32(a[u], x[u], acc);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Ys[32 * kt + lnz::cd_row(r, h)][32 * qt + l31] = acc[r];
  }
  __syncthreads();
  // ---- phase 2
  {
    const int o = 32 * qt + l31;
    f32x16 T = lnz::splat16(0.0f);
    const float* yrow = &Ys[32 * kt + l31][0];
    for (int s = 0; s < S; ++s) {
      f32x16 U = lnz::splat16(0.0f);
      const float* wp = Wt + (int64_t)s * dinp * DH + o;
      for (int i = 0; i < dinp; i += 16) {
        float a[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a[u] = yrow[i + 2 * u + h];
          x[u] = wp[(int64_t)(i + 2 * u + h) * DH];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) U = lnz::mfma32(a[u], x[u], U);
      }
      const float* g = G + ((int64_t)b * S + s) * K;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int slot = 32 * kt + lnz::cd_row(r, h);
        T[r] = fmaf(slot < K ? g[slot] : 0.0f, U[r], T[r]);
      }
    }
    const int64_t plane = (int64_t)B * DH * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot = 32 * kt + lnz::cd_row(r, h);
      __bf16 p[P];
      split_bf16<P>(T[r], p);
#pragma unroll
      for (int q = 0; q < P; ++q) Tt[q * plane + ((int64_t)b * DH + o) * 64 + slot] = bits(p[q]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv: Xout[rows][0..127] = relu( sum_c Lb_c[rows][:] Zt_c^T + Vb[rows][:] Tt^T + bias )
// ------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(512) void large_conv_kernel(
    const u16* __restrict__ Lb, const u16* __restrict__ Vb, const u16* __restrict__ Zt,
    const u16* __restrict__ Tt, const float* __restrict__ bias, int B, int N, int Nk, int C,
    int tiles, int relu, float* __restrict__ Xout) {
  extern __shared__ __attribute__((aligned(16))) u16 Bs[];  // [2 buffers][P][128][BP]
  // blockIdx -> (graph, row tile): workgroup i runs on XCD i % 8; the `tiles` row tiles of a
  // graph are consecutive workgroups of ONE XCD, so they share that L2's copy of the graph's Zt
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int b = xcd + 8 * (seq / tiles), tile = seq % tiles;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int r0 = tile * 256 + 32 * w;
  const int64_t plane_l = (int64_t)B * C * N * Nk, plane_v = (int64_t)B * N * 64;
  const int64_t plane_z = (int64_t)B * C * DH * Nk, plane_t = (int64_t)B * DH * 64;
  const int nkb = Nk / KB;
  const int total = C * nkb + 1;  // + the lift block (V T)
  // this lane's two A rows (row tiles of 16), clamped: rows >= N are computed and dropped
  int ra[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int r = r0 + 16 * rt + l15;
    ra[rt] = r < N ? r : N - 1;
  }
  // staging role of this thread for the B image: row tid >> 2 (0..127), 32 B part tid & 3
  const int srow = tid >> 2, spart = tid & 3;

  auto a_ptr = [&](int g, int rt, int ks, int p) -> const u16* {
    if (g < C * nkb) {
      const int c = g / nkb, kb = g - c * nkb;
      return Lb + p * plane_l + (((int64_t)b * C + c) * N + ra[rt]) * Nk + kb * KB + 32 * ks + 8 * kq;
    }
    return Vb + p * plane_v + ((int64_t)b * N + ra[rt]) * 64 + 32 * ks + 8 * kq;
  };
  auto b_ptr = [&](int g, int p) -> const u16* {
    if (g < C * nkb) {
      const int c = g / nkb, kb = g - c * nkb;
      return Zt + p * plane_z + (((int64_t)b * C + c) * DH + srow) * Nk + kb * KB + 16 * spart;
    }
    return Tt + p * plane_t + ((int64_t)b * DH + srow) * 64 + 16 * spart;
  };

  f32x4 acc[2][8];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[rt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  bf16x8 A[2][P][2][2];  // [buffer][plane][row tile][k-step]; buffer index always a constant
  f32x4 st[P][2];        // B staging registers (32 B per thread and plane)
  auto load_a = [&](int g, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          A[buf][p][rt][ks] = *reinterpret_cast<const bf16x8*>(a_ptr(g, rt, ks, p));
  };
  auto load_b = [&](int g) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const f32x4* s = reinterpret_cast<const f32x4*>(b_ptr(g, p));
      st[p][0] = s[0];
      st[p][1] = s[1];
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      f32x4* d = reinterpret_cast<f32x4*>(Bs + ((int64_t)(buf * P + p) * DH + srow) * BP + 16 * spart);
      d[0] = st[p][0];
      d[1] = st[p][1];
    }
  };
  // one k-block: prefetch block g + 1 (A fragments from HBM, B image into registers), run the
  // MFMAs of block g from A[cur] and LDS buffer cur, publish the next B image, one barrier
  auto block = [&](int g, auto curc) {
    constexpr int cur = decltype(curc)::value;
    if (g + 1 < total) {
      load_a(g + 1, std::integral_constant<int, cur ^ 1>{});
      load_b(g + 1);
    }
    const u16* bt = Bs + (int64_t)cur * P * DH * BP;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        bf16x8 bf[P];
#pragma unroll
        for (int p = 0; p < P; ++p)
          bf[p] = *reinterpret_cast<const bf16x8*>(bt + ((int64_t)p * DH + 16 * nt + l15) * BP + 32 * ks + 8 * kq);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
            for (int i = 0; i <= ord; ++i)
              acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cur][i][rt][ks], bf[ord - i],
                                                                   acc[rt][nt], 0, 0, 0);
      }
    if (g + 1 < total) store_b(cur ^ 1);
    __syncthreads();
  };

  load_a(0, std::integral_constant<int, 0>{});
  load_b(0);
  store_b(0);
  __syncthreads();
  for (int g = 0; g < total; g += 2) {
    block(g, std::integral_constant<int, 0>{});
    if (g + 1 < total) block(g + 1, std::integral_constant<int, 1>{});
  }
  // epilogue: C/D layout of 16x16: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int col = 16 * nt + l15;
    const float bv = bias[col];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * rt + 4 * kq + r;
        if (row < N) {
          float v = acc[rt][nt][r] + bv;
          if (relu) v = v > 0.0f ? v : 0.0f;
          Xout[((int64_t)b * N + row) * DH + col] = v;
        }
      }
  }
}

}  // namespace

extern "C" int64_t lnz_large_nk(int N) { return ((int64_t)N + KB - 1) / KB * KB; }

extern "C" int lnz_large_pack_operators(const float* L, int64_t stride_b, int64_t stride_r,
                                        int64_t stride_c, int64_t stride_ch, const float* V, int B,
                                        int N, int C, int K, int planes, uint16_t* Lb,
                                        uint16_t* Vb, lnz_stream_t stream) {
  LNZ_REQUIRE(L && V && Lb && Vb && B > 0 && N > 0 && C > 0 && K > 0, LNZ_EINVAL,
              "lnz_large_pack_operators: bad arguments (B=%d N=%d C=%d K=%d)", B, N, C, K);
  LNZ_REQUIRE(K <= 64, LNZ_ENOTSUP, "lnz_large_pack_operators: K=%d > 64", K);
  LNZ_REQUIRE(planes == 1 || planes == 3, LNZ_EINVAL, "lnz_large_pack_operators: planes must be 1 or 3");
  const int Nk = (int)lnz_large_nk(N);
  dim3 grid(N, B);
  if (planes == 1)
    hipLaunchKernelGGL(large_pack_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, L, stride_b,
                       stride_r, stride_c, stride_ch, V, B, N, Nk, C, K, Lb, Vb);
  else
    hipLaunchKernelGGL(large_pack_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, L, stride_b,
                       stride_r, stride_c, stride_ch, V, B, N, Nk, C, K, Lb, Vb);
  return lnz::check_launch("lnz_large_pack_operators");
}

extern "C" int lnz_large_gemm1(const float* X, int ldx, int din, const uint16_t* Wb, int B, int N,
                               int C, int planes, uint16_t* Zt, lnz_stream_t stream) {
  LNZ_REQUIRE(X && Wb && Zt && B > 0 && N > 0 && C > 0 && din > 0 && ldx >= din, LNZ_EINVAL,
              "lnz_large_gemm1: bad arguments");
  LNZ_REQUIRE(planes == 1 || planes == 3, LNZ_EINVAL, "lnz_large_gemm1: planes must be 1 or 3");
  const int dinp = (din + 15) / 16 * 16;
  LNZ_REQUIRE(dinp <= 128, LNZ_ENOTSUP, "lnz_large_gemm1: input width %d > 128", din);
  const int Nk = (int)lnz_large_nk(N);
  const size_t lds = (size_t)planes * DH * (dinp + 8) * sizeof(uint16_t);
  dim3 grid((N + 127) / 128, B);
  if (planes == 1) {
    hipLaunchKernelGGL(large_gemm1_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, X, ldx, din,
                       dinp, Wb, B, N, Nk, C, Zt);
  } else {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)large_gemm1_kernel<3>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, 3 * DH * 136 * 2);
      attr = true;
    }
    hipLaunchKernelGGL(large_gemm1_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, X, ldx, din,
                       dinp, Wb, B, N, Nk, C, Zt);
  }
  return lnz::check_launch("lnz_large_gemm1");
}

extern "C" int lnz_large_spectral(const float* X, int ldx, int din, const float* V, const float* G,
                                  const float* Wt, int B, int N, int K, int S, int planes,
                                  uint16_t* Tt, lnz_stream_t stream) {
  LNZ_REQUIRE(X && V && G && Wt && Tt && B > 0 && N > 0 && K > 0 && S > 0 && din > 0 && ldx >= din,
              LNZ_EINVAL, "lnz_large_spectral: bad arguments");
  LNZ_REQUIRE(K <= 64, LNZ_ENOTSUP, "lnz_large_spectral: K=%d > 64", K);
  LNZ_REQUIRE(planes == 1 || planes == 3, LNZ_EINVAL, "lnz_large_spectral: planes must be 1 or 3");
  const int dinp = (din + 15) / 16 * 16;
  LNZ_REQUIRE(dinp <= 128, LNZ_ENOTSUP, "lnz_large_spectral: input width %d > 128", din);
  if (planes == 1)
    hipLaunchKernelGGL(large_spectral_kernel<1>, dim3(B), dim3(512), 0, (hipStream_t)stream, X, ldx,
                       din, dinp, V, G, Wt, B, N, K, S, Tt);
  else
    hipLaunchKernelGGL(large_spectral_kernel<3>, dim3(B), dim3(512), 0, (hipStream_t)stream, X, ldx,
                       din, dinp, V, G, Wt, B, N, K, S, Tt);
  return lnz::check_launch("lnz_large_spectral");
}

extern "C" int lnz_large_conv(const uint16_t* Lb, const uint16_t* Vb, const uint16_t* Zt,
                              const uint16_t* Tt, const float* bias, int B, int N, int C, int planes,
                              int relu, float* Xout, lnz_stream_t stream) {
  LNZ_REQUIRE(Lb && Vb && Zt && Tt && bias && Xout && B > 0 && N > 0 && C > 0, LNZ_EINVAL,
              "lnz_large_conv: bad arguments");
  LNZ_REQUIRE(planes == 1 || planes == 3, LNZ_EINVAL, "lnz_large_conv: planes must be 1 or 3");
  const int Nk = (int)lnz_large_nk(N);
  const int tiles = (N + 255) / 256;
  const int grid = 8 * tiles * ((B + 7) / 8);
  const size_t lds = (size_t)2 * planes * DH * BP * sizeof(uint16_t);
  if (planes == 1) {
    hipLaunchKernelGGL(large_conv_kernel<1>, dim3(grid), dim3(512), lds, (hipStream_t)stream, Lb, Vb,
                       Zt, Tt, bias, B, N, Nk, C, tiles, relu, Xout);
  } else {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)large_conv_kernel<3>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * DH * BP * 2);
      attr = true;
    }
    hipLaunchKernelGGL(large_conv_kernel<3>, dim3(grid), dim3(512), lds, (hipStream_t)stream, Lb, Vb,
                       Zt, Tt, bias, B, N, Nk, C, tiles, relu, Xout);
  }
  return lnz::check_launch("lnz_large_conv");
}
