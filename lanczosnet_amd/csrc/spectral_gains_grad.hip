// Training, R7a backward: parameter gradients of the spectral-filter MLPs
//   G[l][b][s][k] = MLP_l([D[b,k]^p_1 .. D[b,k]^p_S])_s,   MLP = Linear(S,128) ReLU Linear(128,128)
//   ReLU Linear(128,128) ReLU Linear(128,S)                   (model/lanczos_net.py:95-123,146-149)
// from dG[l][b][k][s] (lnz_lanczosnet_gain_grad), for every conv layer in ONE launch.  The reference
// gets them from autograd through nn.Sequential; as batched library GEMMs over the ~17 k live eigen
// rows that is a forward recomputation plus eleven thin products per step (1.0 ms of a 4.8 ms step:
// 128 x 128 outputs with K = rows fill a fraction of the chip).  Here a workgroup walks over 64-row
// tiles of one layer's rows and keeps the whole chain on chip:
//   x -> h1 -> h2 -> h3 (LDS), dh3 = (dout W6) [h3 > 0], dh2 = (dh3 W4) [h2 > 0], dh1 = (dh2 W2) [h1 > 0],
// the four 64 x 128 x 128 products (h1 W2^T, h2 W4^T, dh3 W4, dh2 W2) as the strip kernel's GEMM1
// (wave w owns output columns [16 w, 16 w + 16), A fragments by ds_read_b128 at pitch 136, one weight
// float4 per lane and 16-k step from L2 — four strided words for the two products with a transposed
// weight, read from the same row-major matrix), and
// the weight gradients dW4 += dh3^T h2, dW2 += dh2^T h1 as outer-product MFMAs over the tile's rows
// into accumulators that live in registers ACROSS the tiles (wave w: rows [16 w, 16 w + 16) of dW,
// all eight column blocks).  The thin ends (S <= 8 inputs / outputs) and the bias gradients run on
// the VALU.  Every workgroup writes ONE partial of every gradient; the caller adds the `parts`
// partials of a layer in a fixed order (deterministic, no atomics).  Exact fp32 throughout.
#include "common.hpp"
#include "gains_body.hpp"

namespace {

using lnz_gains::DistArr;
using lnz_gains::powi;

constexpr int P = 136;      // LDS row pitch (floats): conflict-free ds_read_b128 fragments
constexpr int TR = 64;      // rows per tile
constexpr int SUB = TR / 16;
constexpr int SX = 8;       // thin-end width (S <= 8)
typedef const __attribute__((address_space(3))) float* lds_cptr;
typedef const __attribute__((address_space(3))) f32x4* lds_c4ptr;

__device__ __forceinline__ f32x4 lds4(lds_cptr p) { return *(lds_c4ptr)p; }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// out[r][c] = sum_k A[r][k] M[c][k] (TRANS: M[k][c]), c = 16 w + j   (A: LDS [TR][P], M: row-major
// [128][128] in L2).  C/D register q of acc[I] = out[16 I + 4 kq + q][16 w + j]
template <bool TRANS>
__device__ __forceinline__ void gemm_rows(const float* A, const float* __restrict__ M, const int wave,
                                          const int j, const int kq, f32x4 (&acc)[SUB]) {
  // raw buffer loads: one lane offset, the step's displacement in the scalar offset (plain global
  // loads made the compiler keep 32 hoisted 64-bit addresses per matrix alive across the tile loop)
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(M), 0, 128 * 128 * 4, 0x00020000);
  f32x4 bw[8];
  if constexpr (TRANS) {
    const unsigned voff = (unsigned)(((4 * kq) * 128 + 16 * wave + j) * 4);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        bw[s][t] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (16 * s + t) * 128 * 4, 0));
  } else {
    const unsigned voff = (unsigned)(((16 * wave + j) * 128 + 4 * kq) * 4);
#pragma unroll
    for (int s = 0; s < 8; ++s)
      bw[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 64 * s, 0));
  }
  const lds_cptr xa = (lds_cptr)(A + j * P + 4 * kq);
#pragma unroll
  for (int I = 0; I < SUB; ++I) acc[I] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    f32x4 a[SUB];
#pragma unroll
    for (int I = 0; I < SUB; ++I) a[I] = lds4(xa + 16 * I * P + 16 * s);
#pragma unroll
    for (int I = 0; I < SUB; ++I) acc[I] = mfma16(a[I][0], bw[s][0], acc[I]);
#pragma unroll
    for (int I = 0; I < SUB; ++I) acc[I] = mfma16(a[I][1], bw[s][1], acc[I]);
#pragma unroll
    for (int I = 0; I < SUB; ++I) acc[I] = mfma16(a[I][2], bw[s][2], acc[I]);
#pragma unroll
    for (int I = 0; I < SUB; ++I) acc[I] = mfma16(a[I][3], bw[s][3], acc[I]);
  }
}

// dW[16 w + m][i] += sum_r Gd[r][16 w + m] H[r][i]: register q of acc[ib] = dW[16 w + 4 kq + q][16 ib + j]
__device__ __forceinline__ void outer_acc(const float* Gd, const float* H, const int wave, const int j,
                                          const int kq, f32x4 (&acc)[8]) {
#pragma unroll 4
  for (int kk = 0; kk < TR / 4; ++kk) {
    const int r = 4 * kk + kq;
    const float av = Gd[r * P + 16 * wave + j];
    float bv[8];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) bv[ib] = H[r * P + 16 * ib + j];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) acc[ib] = mfma16(av, bv[ib], acc[ib]);
  }
}

struct GradArgs {
  const float* D;
  const int32_t* rows;
  const int32_t* n_rows;
  const float* dG;      // [L][B K][S]
  const float* p[16][8];  // [layer][W0 [128][S], b0, W2 [128][128], b2, W4, b4, W6 [S][128], b6 (unused)]
  float* out;           // [L][parts][T]: dW0 [128][S] | dW2 [128][128] | dW4 | dW6 [S][128] | db0, db2, db4
                        // [3][128] | db6 [S] | pad to a multiple of four
  int BK, S, parts, T;
  DistArr dist;
};

__global__ __launch_bounds__(512) void spectral_mlp_grad_kernel(const GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H1 = smem;               // [TR][P]
  float* H2 = H1 + TR * P;
  float* H3 = H2 + TR * P;        // h3, later dh2
  float* Gd = H3 + TR * P;        // dh3, later dh1
  float* Xs = Gd + TR * P;        // [TR][SX] features
  float* Ds = Xs + TR * SX;       // [TR][SX] dout
  float* W0s = Ds + TR * SX;      // [128][SX]
  float* W6s = W0s + 128 * SX;    // [SX][128]
  float* bs = W6s + SX * 128;     // [3][128]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, j = lane & 15, kq = lane >> 4;
  const int l = blockIdx.y, part = blockIdx.x, S = a.S;
  const int R = a.rows ? *a.n_rows : a.BK;
  const float* __restrict__ W2 = a.p[l][2];
  const float* __restrict__ W4 = a.p[l][4];

  for (int i = tid; i < 128 * SX; i += 512) {
    const int o = i / SX, s = i - o * SX;
    W0s[i] = s < S ? a.p[l][0][o * S + s] : 0.0f;
    const int s6 = i >> 7, c6 = i & 127;
    W6s[i] = s6 < S ? a.p[l][6][s6 * 128 + c6] : 0.0f;
  }
  for (int i = tid; i < 3 * 128; i += 512) bs[i] = a.p[l][1 + 2 * (i >> 7)][i & 127];

  f32x4 acc2[8], acc4[8];
#pragma unroll
  for (int ib = 0; ib < 8; ++ib) acc2[ib] = acc4[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  // thin ends: thread t owns dW6 entries t and t + 512 (< 8 * 128: (s, column)), dW0 entries likewise
  // ((row, s)); threads < 384 one bias-gradient column each, threads 384 .. 391 one of db6
  float g6[2] = {0.f, 0.f}, g0[2] = {0.f, 0.f}, gb = 0.f;

  for (int tile = part; tile * TR < R; tile += a.parts) {
    __syncthreads();  // (the previous tile's last readers of Gd / Xs; the staging above)
    // ---- features x = D^p and the incoming dG rows
    if (tid < TR * SX) {
      const int r = tid / SX, s = tid - r * SX;
      const int idx = tile * TR + r;
      float x = 0.f, g = 0.f;
      if (idx < R && s < S) {
        const int row = a.rows ? a.rows[idx] : idx;
        x = powi(a.D[row], a.dist.v[s]);
        g = a.dG[((int64_t)l * a.BK + row) * S + s];
      }
      Xs[tid] = x;
      Ds[tid] = g;
    }
    __syncthreads();
    // ---- h1 = relu(x W0^T + b0)
#pragma unroll 2
    for (int i = tid; i < TR * 128; i += 512) {
      const int r = i >> 7, o = i & 127;
      float v = bs[o];
#pragma unroll
      for (int s = 0; s < SX; ++s) v = fmaf(Xs[r * SX + s], W0s[o * SX + s], v);
      H1[r * P + o] = fmaxf(v, 0.f);
    }
    __syncthreads();
    f32x4 acc[SUB];
    const int col = 16 * wave + j;
    // ---- h2 = relu(h1 W2^T + b2), h3 = relu(h2 W4^T + b4)
    gemm_rows<false>(H1, W2, wave, j, kq, acc);
#pragma unroll
    for (int I = 0; I < SUB; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) H2[(16 * I + 4 * kq + q) * P + col] = fmaxf(acc[I][q] + bs[128 + col], 0.f);
    __syncthreads();
    gemm_rows<false>(H2, W4, wave, j, kq, acc);
#pragma unroll
    for (int I = 0; I < SUB; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) H3[(16 * I + 4 * kq + q) * P + col] = fmaxf(acc[I][q] + bs[256 + col], 0.f);
    __syncthreads();
    // ---- dh3 = (dout W6) [h3 > 0]  -> Gd;  dW6 += dout^T h3, db6 += column sums of dout
#pragma unroll 2
    for (int i = tid; i < TR * 128; i += 512) {
      const int r = i >> 7, o = i & 127;
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < SX; ++s) v = fmaf(Ds[r * SX + s], W6s[s * 128 + o], v);
      Gd[r * P + o] = H3[r * P + o] > 0.f ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u, s = e >> 7, c = e & 127;
      float v = 0.f;
#pragma unroll 4
      for (int r = 0; r < TR; ++r) v = fmaf(Ds[r * SX + s], H3[r * P + c], v);
      g6[u] += v;
    }
    if (tid >= 384 && tid < 384 + SX) {
      float v = 0.f;
#pragma unroll 4
      for (int r = 0; r < TR; ++r) v += Ds[r * SX + (tid - 384)];
      gb += v;
    }
    __syncthreads();
    // ---- dW4 += dh3^T h2, db4; dh2 = (dh3 W4) [h2 > 0] -> H3's buffer
    outer_acc(Gd, H2, wave, j, kq, acc4);
    if (tid >= 256 && tid < 384) {
      float v = 0.f;
#pragma unroll 4
      for (int r = 0; r < TR; ++r) v += Gd[r * P + (tid - 256)];
      gb += v;
    }
    gemm_rows<true>(Gd, W4, wave, j, kq, acc);
    __syncthreads();  // (every wave has read h3's successor dh3 and is through with H3 as h3)
#pragma unroll
    for (int I = 0; I < SUB; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int at = (16 * I + 4 * kq + q) * P + col;
        H3[at] = H2[at] > 0.f ? acc[I][q] : 0.f;
      }
    __syncthreads();
    // ---- dW2 += dh2^T h1, db2; dh1 = (dh2 W2) [h1 > 0] -> Gd
    outer_acc(H3, H1, wave, j, kq, acc2);
    if (tid >= 128 && tid < 256) {
      float v = 0.f;
#pragma unroll 4
      for (int r = 0; r < TR; ++r) v += H3[r * P + (tid - 128)];
      gb += v;
    }
    gemm_rows<true>(H3, W2, wave, j, kq, acc);
#pragma unroll
    for (int I = 0; I < SUB; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int at = (16 * I + 4 * kq + q) * P + col;
        Gd[at] = H1[at] > 0.f ? acc[I][q] : 0.f;
      }
    __syncthreads();
    // ---- dW0 += dh1^T x, db0
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u, o = e / SX, s = e - o * SX;
      float v = 0.f;
#pragma unroll 4
      for (int r = 0; r < TR; ++r) v = fmaf(Gd[r * P + o], Xs[r * SX + s], v);
      g0[u] += v;
    }
    if (tid < 128) {
      float v = 0.f;
#pragma unroll 4
      for (int r = 0; r < TR; ++r) v += Gd[r * P + tid];
      gb += v;
    }
  }

  // ---- this workgroup's partials
  float* __restrict__ o0 = a.out + ((int64_t)l * a.parts + part) * a.T;   // dW0
  float* __restrict__ o2 = o0 + 128 * S;                                   // dW2
  float* __restrict__ o4 = o2 + 16384;                                     // dW4
  float* __restrict__ o6 = o4 + 16384;                                     // dW6
  float* __restrict__ ob = o6 + S * 128;                                   // db0 | db2 | db4 | db6
#pragma unroll
  for (int ib = 0; ib < 8; ++ib)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int at = (16 * wave + 4 * kq + q) * 128 + 16 * ib + j;
      o2[at] = acc2[ib][q];
      o4[at] = acc4[ib][q];
    }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 512 * u;
    const int s6 = e >> 7, c6 = e & 127;
    if (s6 < S) o6[s6 * 128 + c6] = g6[u];
    const int o = e / SX, s = e - o * SX;
    if (s < S) o0[o * S + s] = g0[u];
  }
  if (tid < 384) ob[tid] = gb;
  else if (tid < 384 + SX && tid - 384 < S) ob[tid] = gb;
}

constexpr size_t kGradLds = (size_t)(4 * TR * P + 2 * TR * SX + 2 * 128 * SX + 3 * 128) * sizeof(float);

}  // namespace

extern "C" int lnz_spectral_mlp_grad_floats(int S) {   // floats of one partial (layout: header)
  if (S < 1 || S > SX) return 0;
  return (128 * S + 2 * 16384 + S * 128 + 384 + S + 3) & ~3;
}

extern "C" int lnz_spectral_mlp_grad_parts(int n_rows_max, int num_layer, int n_cu) {
  if (n_rows_max <= 0 || num_layer <= 0 || n_cu <= 0) return 0;
  const int tiles = (n_rows_max + TR - 1) / TR;
  int parts = n_cu / num_layer;
  if (parts < 1) parts = 1;
  return parts < tiles ? parts : tiles;
}

extern "C" int lnz_spectral_mlp_grad(const float* D, int B, int K, const int32_t* dist_host, int S,
                                     int num_layer, const int32_t* rows, const int32_t* n_rows,
                                     const float* dG, const float* const* ptrs, int parts,
                                     float* partials, lnz_stream_t stream) {
  LNZ_REQUIRE(D && dist_host && dG && ptrs && partials, LNZ_EINVAL, "lnz_spectral_mlp_grad: null pointer");
  LNZ_REQUIRE(B > 0 && K > 0 && num_layer > 0 && num_layer <= 16 && parts > 0 && parts <= 65535,
              LNZ_EINVAL, "lnz_spectral_mlp_grad: bad sizes (B=%d K=%d L=%d parts=%d)", B, K, num_layer, parts);
  LNZ_REQUIRE(S >= 1 && S <= SX, LNZ_ENOTSUP, "lnz_spectral_mlp_grad: S=%d not in 1..%d", S, SX);
  LNZ_REQUIRE(!rows == !n_rows, LNZ_EINVAL, "lnz_spectral_mlp_grad: rows and n_rows come together");
  GradArgs a;
  a.D = D, a.rows = rows, a.n_rows = n_rows, a.dG = dG;
  a.out = partials;
  for (int l = 0; l < num_layer; ++l)
    for (int i = 0; i < 8; ++i) {
      LNZ_REQUIRE(ptrs[l * 8 + i] || i == 7, LNZ_EINVAL, "lnz_spectral_mlp_grad: null parameter pointer");
      a.p[l][i] = ptrs[l * 8 + i];
    }
  a.BK = B * K, a.S = S, a.parts = parts, a.T = lnz_spectral_mlp_grad_floats(S);
  for (int s = 0; s < lnz_gains::SMAX; ++s) a.dist.v[s] = s < S ? dist_host[s] : 0;
  // per launch: the attribute is per device, and a process may drive several
  LNZ_DYNAMIC_LDS(spectral_mlp_grad_kernel, kGradLds, "spectral_gains_grad.hip");
  hipLaunchKernelGGL(spectral_mlp_grad_kernel, dim3(parts, num_layer), dim3(512), kGradLds,
                     (hipStream_t)stream, a);
  return lnz::check_launch("lnz_spectral_mlp_grad");
}
