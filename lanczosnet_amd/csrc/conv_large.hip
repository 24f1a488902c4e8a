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
//   lnz_large_pack_operators   once per batch: channels-last fp32 L [B,N,N,C] -> bf16 planes in
//                              FRAGMENT-TILE order Lb [P][B][C][RT][nkb][4][64][8]: the 32 rows x
//                              64 k block a wavefront consumes per k-block is one contiguous
//                              4 KiB chunk laid out as its four A fragments (row tile, k-step) x
//                              lane x 8 bf16, consecutive k-blocks consecutive — every wave
//                              streams ONE sequential 4 KiB-granular region (128 KiB per channel
//                              at N = 2048), every global_load_dwordx4 is 1 KiB contiguous.
//                              (Row-major bf16 rows made the 256 rows of a workgroup a 4 KiB-
//                              stride gather that leaned on a few HBM channels: 3.9 TB/s.)
//                              V -> Vb [P][B][RT][4][64][8] likewise.  Packed bf16 also halves
//                              the bytes all num_layer passes stream.
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
// P = 2: the cheaper parity mode — two fp16 pieces per operand (22 mantissa bits), three piece
// products, 2x the operand bytes (see ElemTraits).
#include "common.hpp"

#include <type_traits>

// The packed operators are read exactly once per launch: non-temporal loads keep them from
// evicting the (re-read) Zt / Tt blocks from L2.  (Measured and NOT kept: non-temporal loads of
// the fp32 source in the pack kernel, 2.55 -> 3.0 ms; non-temporal stores of Lb, no change.)
#define LNZ_STREAM_LOAD(p) __builtin_nontemporal_load(p)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr int DH = 128;        // hidden width (output columns of every layer)
constexpr int KB = 64;         // k-block of the conv loop (bf16 elements = one 128 B line per row)
constexpr int BP = 64;         // LDS pitch of a staged B row in bf16: 128 B, 16 B chunks XOR-swizzled

__device__ inline __bf16 to_bf16(float x) {  // round to nearest even (v_cvt_pk_bf16_f32)
  bf16x2 p = __builtin_convertvector(f32x2{x, 0.0f}, bf16x2);
  return p[0];
}
__device__ inline float bf16_float(__bf16 b) {
  return __uint_as_float((unsigned)__builtin_bit_cast(u16, b) << 16);
}
__device__ inline u16 bits(__bf16 b) { return __builtin_bit_cast(u16, b); }

// ---- element type by plane count: P = 1, 3 -> bf16 pieces; P = 2 -> fp16 pieces -----------------
// P = 2 is the cheaper parity mode: x = hi + lo in fp16 (22 mantissa bits), products hi hi + hi lo
// + lo hi, 2/3 of the operand bytes and half the matrix work of P = 3.  fp16 has 5 exponent bits:
// the A operands (Laplacian entries, Ritz vectors, mix weights — all <= 1 in magnitude and often
// ~1e-2) are scaled by 2^10 before the split so that their low pieces stay out of the subnormal
// range, and the consumer multiplies its accumulator by 2^-10 (exact); B operands (activations)
// are split as they are.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int P>
struct ElemTraits {
  static constexpr bool kHalf = P == 2;
  static constexpr float kAScale = P == 2 ? 1024.0f : 1.0f;   // applied to A operands
  static constexpr float kAInv = P == 2 ? 1.0f / 1024.0f : 1.0f;
};
template <int P>
__device__ inline u16 to_piece(float x) {
  if (ElemTraits<P>::kHalf) return __builtin_bit_cast(u16, (_Float16)x);
  return bits(to_bf16(x));
}
template <int P>
__device__ inline float piece_float(u16 b) {
  if (ElemTraits<P>::kHalf) return (float)__builtin_bit_cast(_Float16, b);
  return __uint_as_float((unsigned)b << 16);
}
// x -> pieces p[0..P-1] with x ~= sum p[i] (each piece the rounding of the remainder)
template <int P>
__device__ inline void split_pieces(float x, u16* p) {
  float r = x;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    p[i] = to_piece<P>(r);
    r -= piece_float<P>(p[i]);
  }
}
template <int P>
__device__ inline f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  if (ElemTraits<P>::kHalf)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int P>
__device__ inline f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  if (ElemTraits<P>::kHalf)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// pack: L [B,N,N,C] fp32 (any strides) -> Lb [P][B][C][RT][nkb][4][64][8];  V [B,N,K] -> Vb
// [P][B][RT][4][64][8].  Chunk (row group rg of 32 rows, k-block kb of 64): fragment f = 2 rt +
// ks (rt: rows 16 rt .. + 15 of the group, ks: k 32 ks .. + 31 of the block), lane = 16 kq + r15
// holds A[row = 32 rg + 16 rt + r15][k = 64 kb + 32 ks + 8 kq .. + 7] — the A operand of
// v_mfma_f32_16x16x32_bf16 as it is loaded.  Rows >= N, columns >= N and eigen slots >= K: zero.
// One workgroup per (graph, row group): thread (row = t / 8, j = t % 8) owns 8 consecutive k of
// its row for ALL channels per k-block — 32 C contiguous bytes of a channels-last row — and
// writes one 16 B fragment piece per channel and plane; the workgroup emits whole 4 KiB chunks.
// ------------------------------------------------------------------------------------------
__device__ inline int frag_slot(int row32, int j) {  // 16 B slot of (row in group, k-octet j) in a chunk
  return (((row32 >> 4) * 2 + (j >> 2)) * 64 + (j & 3) * 16 + (row32 & 15));
}

// Channel folding (config 5 / the graph configuration have num_edge_type = 1, reference
// dataset/graph_data.py:225-262: the bond-type channel IS the simple-graph channel, and
// sum_c L_c X W_c^T = L (X (sum_c W_c)^T) over a class of equal operators): the caller names the
// Cd DISTINCT source channels to pack (`src`) and, per source channel c, the packed slot it is
// claimed to equal (`rep`); the kernel has every channel of its 32 B x C piece of the row in hand
// anyway and verifies each claim, and for every two packed channels reports whether they differ
// anywhere: bit 8 c + c' (c' < c) of *neq is set iff channel c differs from channel c' somewhere
// among the compared pairs.  A claim that fails makes the caller repack unfolded; a pair of
// packed channels that never differed is folded from the next batch on.
struct LargeChanMap {
  signed char nsrc;      // Cd
  signed char src[8];    // packed slot d  -> source channel
  signed char rep[8];    // source channel -> packed slot of its class (src[rep[c]] == c: packed itself)
  signed char check[8];  // source channel -> compare against the channels packed before it
};

template <int P>
__global__ __launch_bounds__(256) void large_pack_kernel(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch,
    const float* __restrict__ V, int B, int N, int nkb, int C, int K, LargeChanMap cm,
    unsigned long long* __restrict__ neq, u16* __restrict__ Lb, u16* __restrict__ Vb) {
  const int rg = blockIdx.x, b = blockIdx.y, RT = gridDim.x;
  const int row32 = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int r = 32 * rg + row32;
  const bool rv = r < N;
  const float* Lr = L + (int64_t)b * sb + (int64_t)(rv ? r : 0) * sr;
  const int Cd = cm.nsrc;
  const int64_t plane_l = (int64_t)B * Cd * RT * nkb * 2048;
  const int slot = frag_slot(row32, j);
  // channels-last pair of channels (the collate layout with one edge type): one 64 B run per thread
  const bool fast2 = sch == 1 && sc == 2 && C == 2 && (((uintptr_t)Lr) & 15) == 0;
  // rows contiguous per channel (channel-major tensors, expanded single-channel views)
  const bool fast1 = sc == 1 && (((uintptr_t)Lr) & 15) == 0 && ((sch & 3) == 0);
  unsigned long long differ = 0;
  auto emit = [&](const float* x, int d, int kb) {
    bf16x8 out[P];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      u16 p[P];
      split_pieces<P>(x[u] * ElemTraits<P>::kAScale, p);
#pragma unroll
      for (int i = 0; i < P; ++i) out[i][u] = __builtin_bit_cast(__bf16, p[i]);
    }
    const int64_t o = ((((int64_t)b * Cd + d) * RT + rg) * nkb + kb) * 2048 + (int64_t)slot * 8;
#pragma unroll
    for (int i = 0; i < P; ++i) *reinterpret_cast<bf16x8*>(Lb + i * plane_l + o) = out[i];
  };
  auto fetch = [&](float* x, int c, int k0) {  // channel c, k0 .. k0 + 7 of this thread's row
    if (fast1 && rv && k0 + 8 <= N) {
      const f32x4* src = reinterpret_cast<const f32x4*>(Lr + (int64_t)c * sch + k0);
      const f32x4 a0 = src[0], a1 = src[1];
#pragma unroll
      for (int u = 0; u < 4; ++u) { x[u] = a0[u]; x[4 + u] = a1[u]; }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u;
        x[u] = (rv && k < N) ? Lr[(int64_t)k * sc + (int64_t)c * sch] : 0.0f;
      }
    }
  };
  for (int kb = 0; kb < nkb; ++kb) {
    const int k0 = 64 * kb + 8 * j;
    if (fast2) {
      float x[2][8];
      if (rv && k0 + 8 <= N) {
        const f32x4* src = reinterpret_cast<const f32x4*>(Lr + (int64_t)k0 * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 a = src[q];
          x[0][2 * q] = a[0]; x[1][2 * q] = a[1]; x[0][2 * q + 1] = a[2]; x[1][2 * q + 1] = a[3];
        }
      } else {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + u;
            x[cc][u] = (rv && k < N) ? Lr[(int64_t)k * 2 + cc] : 0.0f;
          }
      }
      emit(x[0], 0, kb);
      if (Cd == 2) emit(x[1], 1, kb);
      if (cm.check[1]) {
        bool ne = false;
#pragma unroll
        for (int u = 0; u < 8; ++u) ne |= x[0][u] != x[1][u];
        differ |= ne ? (1ull << 8) : 0ull;
      }
    } else {
      for (int c = 0; c < C; ++c) {
        const bool own = cm.src[cm.rep[c]] == c;  // packed itself
        if (!own && !cm.check[c]) continue;        // equal by construction (zero channel stride)
        float x[8];
        fetch(x, c, k0);
        if (own) emit(x, cm.rep[c], kb);
        if (cm.check[c]) {
          // a folded channel against its representative; a packed one against all packed before it
          for (int d = 0; d < Cd; ++d) {
            const int c2 = cm.src[d];
            if (c2 >= c || (!own && d != cm.rep[c])) continue;
            float y[8];
            fetch(y, c2, k0);
            bool ne = false;
#pragma unroll
            for (int u = 0; u < 8; ++u) ne |= x[u] != y[u];
            differ |= ne ? (1ull << (8 * c + c2)) : 0ull;
          }
        }
      }
    }
  }
  if (neq) {
    // wave-level OR (the bit pattern is the same for nearly all lanes: at most 28 rounds of ballot)
    for (int c = 1; c < C; ++c)
      for (int c2 = 0; c2 < c; ++c2) {
        const unsigned long long bit = 1ull << (8 * c + c2);
        if (__builtin_amdgcn_ballot_w64((differ & bit) != 0) != 0 && (threadIdx.x & 63) == 0)
          atomicOr(neq, bit);
      }
  }
  // the Ritz vectors: one chunk per row group
  {
    const int64_t plane_v = (int64_t)B * RT * 2048;
    bf16x8 out[P];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = 8 * j + u;
      const float x = (rv && k < K) ? V[((int64_t)b * N + r) * K + k] : 0.0f;
      u16 p[P];
      split_pieces<P>(x * ElemTraits<P>::kAScale, p);
#pragma unroll
      for (int i = 0; i < P; ++i) out[i][u] = __builtin_bit_cast(__bf16, p[i]);
    }
    const int64_t o = ((int64_t)b * RT + rg) * 2048 + (int64_t)slot * 8;
#pragma unroll
    for (int i = 0; i < P; ++i) *reinterpret_cast<bf16x8*>(Vb + i * plane_v + o) = out[i];
  }
}

// The Ritz vectors alone (the sparse conv has no packed Laplacian): Vb exactly as large_pack_kernel
// writes it.
template <int P>
__global__ __launch_bounds__(256) void large_pack_vectors_kernel(const float* __restrict__ V, int B,
                                                                 int N, int K, u16* __restrict__ Vb) {
  const int rg = blockIdx.x, b = blockIdx.y, RT = gridDim.x;
  const int row32 = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int r = 32 * rg + row32;
  const int64_t plane_v = (int64_t)B * RT * 2048;
  bf16x8 out[P];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int k = 8 * j + u;
    const float x = (r < N && k < K) ? V[((int64_t)b * N + r) * K + k] : 0.0f;
    u16 p[P];
    split_pieces<P>(x * ElemTraits<P>::kAScale, p);
#pragma unroll
    for (int i = 0; i < P; ++i) out[i][u] = __builtin_bit_cast(__bf16, p[i]);
  }
  const int64_t o = ((int64_t)b * RT + rg) * 2048 + (int64_t)frag_slot(row32, j) * 8;
#pragma unroll
  for (int i = 0; i < P; ++i) *reinterpret_cast<bf16x8*>(Vb + i * plane_v + o) = out[i];
}

// ------------------------------------------------------------------------------------------
// GEMM1: Zt[c][o][n] = sum_i W_c[o][i] X[n][i]   (= (X W_c^T)^T, the conv kernel's B image)
// Workgroup = 128 node rows; the X tile is read ONCE, coalesced, converted (split) to bf16 and
// staged in LDS (B operand of v_mfma_f32_32x32x16_bf16 for all channels and all four waves);
// wave w owns the 32 outputs o = 32 w .. + 31 and keeps their weight fragments in registers per
// channel (Wf is stored in fragment order: one 1 KiB wave load per fragment); the 128 x 128
// output tile of a channel goes through LDS so that Zt rows are written as 256 B runs.
// ------------------------------------------------------------------------------------------
// ROWS (P = 1, one channel; the sparse conv of csrc/conv_sparse.hip): the product is written ROW
// major instead, Z [B][N][128] bf16 — a node's 128 features are the 256 contiguous bytes the
// gather of lnz_large_sparse_conv reads per nonzero.
template <int P, bool ROWS = false>
__global__ __launch_bounds__(256) void large_gemm1_kernel(
    const float* __restrict__ X, int ldx, int din, int dinp, const u16* __restrict__ Wf,
    int B, int N, int Nk, int C, u16* __restrict__ Zt) {
  extern __shared__ __attribute__((aligned(16))) u16 smem_g1[];
  const int xp = dinp + 8;                      // LDS pitch of an X row (bf16): conflict-free b128
  u16* Xs = smem_g1;                            // [P][128][xp]
  u16* Zs = smem_g1 + (size_t)P * 128 * xp;     // [128 o][136]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * 128;
  const int l31 = lane & 31, h = lane >> 5;
  const int nks = dinp / 16;
  // ---- stage the X tile: 4 consecutive columns per thread and step
  {
    const int qn = dinp / 4;
    const bool vec = (din == dinp) && (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
    // eight pieces per thread and round, all requested before the first is converted: one piece
    // per iteration was one global-memory round trip per iteration — 16 in series for the 128 x
    // 128 tile, the whole 21 us of this launch on a batch of 100-node graphs
    const int total = 128 * qn;
    for (int base = tid; base < total; base += 256 * 8) {
      f32x4 xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + 256 * u;
        const int n = idx / qn, q = idx - n * qn;
        const int row = n0 + n;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (idx < total && row < N) {
          const float* src = X + ((int64_t)b * N + row) * ldx + 4 * q;
          if (vec) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = (4 * q + t < din) ? src[t] : 0.0f;
          }
        }
        xv[u] = v;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + 256 * u;
        if (idx >= total) continue;
        const int n = idx / qn, q = idx - n * qn;
        u16 o[P][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          u16 p[P];
          split_pieces<P>(xv[u][t], p);
#pragma unroll
          for (int i = 0; i < P; ++i) o[i][t] = p[i];
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {
          uint2 v;
          v.x = (unsigned)o[i][0] | ((unsigned)o[i][1] << 16);
          v.y = (unsigned)o[i][2] | ((unsigned)o[i][3] << 16);
          *reinterpret_cast<uint2*>(Xs + ((size_t)i * 128 + n) * xp + 4 * q) = v;
        }
      }
    }
  }
  __syncthreads();
  const int64_t plane_z = (int64_t)B * C * DH * Nk;
  const int64_t plane_w = (int64_t)C * 4 * nks * 512;     // fragment-ordered weights per plane
  for (int c = 0; c < C; ++c) {
    // this wave's weight fragments (A operand): [P][nks] x 8 bf16 per lane
    bf16x8 af[P][8];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        if (ks < nks)
          af[p][ks] = *reinterpret_cast<const bf16x8*>(
              Wf + p * plane_w + (((int64_t)c * 4 + w) * nks + ks) * 512 + lane * 8);
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[nt] = lnz::splat16(0.0f);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < nks) {
          bf16x8 bf[P];
#pragma unroll
          for (int p = 0; p < P; ++p)
            bf[p] = *reinterpret_cast<const bf16x8*>(Xs + ((size_t)p * 128 + 32 * nt + l31) * xp +
                                                     16 * ks + 8 * h);
          // small terms first: (order 2), (order 1), (order 0)
#pragma unroll
          for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
            for (int i = 0; i <= ord; ++i)
              acc[nt] = mfma_32x32x16<P>(af[i][ks], bf[ord - i], acc[nt]);
        }
      }
    }
    if (ElemTraits<P>::kHalf) {  // the weight fragments carry the 2^10 scale of the A operands
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] *= ElemTraits<P>::kAInv;
    }
    if constexpr (ROWS) {
      // acc[nt][4 q .. 4 q + 3] = features 32 w + 8 q + 4 h .. + 3 of node n0 + 32 nt + l31: 8 B stores
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + 32 * nt + l31;
        if (n < N) {
          u16* dst = Zt + ((int64_t)b * N + n) * DH + 32 * w + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint2 v;
            v.x = (unsigned)to_piece<P>(acc[nt][4 * q]) | ((unsigned)to_piece<P>(acc[nt][4 * q + 1]) << 16);
            v.y = (unsigned)to_piece<P>(acc[nt][4 * q + 2]) | ((unsigned)to_piece<P>(acc[nt][4 * q + 3]) << 16);
            *reinterpret_cast<uint2*>(dst + 8 * q) = v;
          }
        }
      }
      continue;
    }
    // ---- output, plane by plane through LDS: Zs[o][n] <- piece p of the tile, then 256 B runs
#pragma unroll
    for (int p = 0; p < P; ++p) {
      __syncthreads();  // Zs free (previous plane / channel copied out)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const u16 pc = to_piece<P>(acc[nt][r]);
          acc[nt][r] -= piece_float<P>(pc);
          Zs[(32 * w + lnz::cd_row(r, h)) * 136 + 32 * nt + l31] = pc;
        }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, o = q >> 4, part = q & 15;
        const int n = n0 + 8 * part;
        u16* dst = Zt + p * plane_z + (((int64_t)b * C + c) * DH + o) * Nk + n;
        const u16* src = Zs + o * 136 + 8 * part;
        if (n + 8 <= N) {
          *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
        } else {
          for (int u = 0; u < 8; ++u)
            if (n + u < N) dst[u] = src[u];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Spectral block, exact fp32 (v_mfma_f32_32x32x2_f32), two launches:
//   large_project_kernel   Y[k][i] += sum_{n in this workgroup's row chunk} V[n][k] X[n][i]
//                          grid (B, row chunks): wave (kt, it) = 32 slots x 32 columns; partial
//                          tiles are added into Ybuf [B][64][128] with fp32 atomics (Ybuf is zero
//                          on entry: large_spectral_kernel leaves it zeroed for the next layer)
//   large_spectral_kernel  one workgroup per graph: T[k][o] = sum_s g_s[k] * sum_i Y[k][i]
//                          W_s[o][i]; wave (kt, ot); per scale the unscaled product accumulates
//                          in its own C tile and its rows are scaled into T (16 FMAs) — a VALU
//                          multiply in front of every MFMA costs ~40 cycles each
// Wt: pack_rows_k8 image of W_long [128][S * dinp] (the long-scale column blocks of the mix weight,
// each zero padded to dinp columns): [4][S * dinp / 8][64][4] fp32.
// Output Tt [P][B][128][64] bf16 (the B image of the conv kernel's lift block).
// ------------------------------------------------------------------------------------------
// WITHZ (the sparse conv's layer, planes = 1): the X tile in LDS also gives Z = bf16(X) bf16(W)^T
// for its 64 rows — large_gemm1_kernel<1, rows>'s products bit for bit (same fragments, same k
// order) — so the layer reads X once instead of twice: wave (kt, qt) = rows 32 kt .. + 31 of the
// step x features 32 qt .. + 31, its weight fragments in registers for the whole launch.
template <bool WITHZ>
__global__ __launch_bounds__(512) void large_project_kernel(
    const float* __restrict__ X, int ldx, int din, int dinp, const float* __restrict__ V, int N,
    int K, int rows_per_wg, float* __restrict__ Ybuf, const u16* __restrict__ Wf,
    u16* __restrict__ Z) {
  // 64-row steps: the V [64 x 64] and X [64 x 128] pieces are read once per workgroup with
  // coalesced 16 B loads into LDS (double buffered) and feed all eight wave tiles from there —
  // straight from global, every wave re-read its 128 B column slice of every row (2 x 4 B loads
  // per MFMA): the load path, not the matrix pipe, bounded it.
  __shared__ __attribute__((aligned(16))) float Vs[2][64][64 + 4];
  __shared__ __attribute__((aligned(16))) float Xs[2][64][DH + 4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int kt = w >> 2, qt = w & 3;
  const int nbeg = blockIdx.x * rows_per_wg;
  const int nend = min(N, nbeg + rows_per_wg);
  const bool work = 32 * qt < dinp;  // wave tile inside the (padded) input width
  const float* Vg = V + (int64_t)b * N * K;
  const float* Xg = X + (int64_t)b * N * ldx;
  const bool vecv = (K % 4 == 0) && ((((uintptr_t)V) & 15) == 0);
  const bool vecx = (ldx % 4 == 0) && ((((uintptr_t)X) & 15) == 0);
  f32x4 rv[2], rx[4];
  auto fetch = [&](int n0) {  // rows n0 .. n0 + 63 -> registers (zeros beyond nend / K / din)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 512 * i, r = q >> 4, c4 = (q & 15) * 4;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (n0 + r < nend) {
        const float* src = Vg + (int64_t)(n0 + r) * K + c4;
        if (vecv && c4 + 4 <= K) v = *reinterpret_cast<const f32x4*>(src);
        else
          for (int u = 0; u < 4; ++u) v[u] = c4 + u < K ? src[u] : 0.0f;
      }
      rv[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 512 * i, r = q >> 5, c4 = (q & 31) * 4;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (n0 + r < nend && c4 < dinp) {
        const float* src = Xg + (int64_t)(n0 + r) * ldx + c4;
        if (vecx && c4 + 4 <= din) v = *reinterpret_cast<const f32x4*>(src);
        else
          for (int u = 0; u < 4; ++u) v[u] = c4 + u < din ? src[u] : 0.0f;
      }
      rx[i] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 512 * i;
      *reinterpret_cast<f32x4*>(&Vs[buf][q >> 4][(q & 15) * 4]) = rv[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 512 * i;
      *reinterpret_cast<f32x4*>(&Xs[buf][q >> 5][(q & 31) * 4]) = rx[i];
    }
  };
  f32x16 acc = lnz::splat16(0.0f);
  const int nks = dinp / 16;
  bf16x8 af[WITHZ ? 8 : 1];
  if constexpr (WITHZ) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      af[ks] = ks < nks ? *reinterpret_cast<const bf16x8*>(Wf + ((int64_t)qt * nks + ks) * 512 + lane * 8)
                        : bf16x8{};
  }
  fetch(nbeg);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int n0 = nbeg; n0 < nend; n0 += 64) {
    const bool more = n0 + 64 < nend;
    if (more) fetch(n0 + 64);
    if constexpr (WITHZ) {
      f32x16 z = lnz::splat16(0.0f);
      const float* xr = &Xs[buf][32 * kt + l31][8 * h];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < nks) {   // (uniform)
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr + 16 * ks);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(xr + 16 * ks + 4);
          bf16x8 bx;
#pragma unroll
          for (int u = 0; u < 4; ++u) bx[u] = to_bf16(x0[u]), bx[4 + u] = to_bf16(x1[u]);
          z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bx, z, 0, 0, 0);
        }
      }
      const int n = n0 + 32 * kt + l31;
      if (n < nend) {
        u16* dst = Z + ((int64_t)b * N + n) * DH + 32 * qt + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint2 v;
          v.x = (unsigned)bits(to_bf16(z[4 * q])) | ((unsigned)bits(to_bf16(z[4 * q + 1])) << 16);
          v.y = (unsigned)bits(to_bf16(z[4 * q + 2])) | ((unsigned)bits(to_bf16(z[4 * q + 3])) << 16);
          *reinterpret_cast<uint2*>(dst + 8 * q) = v;
        }
      }
    }
    if (work) {
#pragma unroll 4
      for (int u = 0; u < 32; u += 8) {
        float a[8], x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          a[t] = Vs[buf][2 * (u + t) + h][32 * kt + l31];
          x[t] = Xs[buf][2 * (u + t) + h][32 * qt + l31];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = lnz::mfma32(a[t], x[t], acc);
      }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (!work) return;
  float* y = Ybuf + (int64_t)b * 64 * DH;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    atomicAdd(y + (32 * kt + lnz::cd_row(r, h)) * DH + 32 * qt + l31, acc[r]);
}

template <int P>
__global__ __launch_bounds__(512) void large_spectral_kernel(
    int dinp, float* __restrict__ Ybuf, const float* __restrict__ G, const float* __restrict__ Wt,
    int B, int K, int S, u16* __restrict__ Tt) {
  __shared__ float Ys[64][DH + 4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int kt = w >> 2, qt = w & 3;
  __shared__ float Gs[16][64];   // the graph's gains by scale and slot (zero beyond K)
  {
    f32x4* y = reinterpret_cast<f32x4*>(Ybuf + (int64_t)b * 64 * DH);
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = y[threadIdx.x + 512 * i];   // (all four in flight)
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 512) {
      const int sc = idx >> 6, slot = idx & 63;
      Gs[sc][slot] = (sc < S && slot < K) ? G[((int64_t)b * S + sc) * K + slot] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = 4 * (threadIdx.x + 512 * i);
      *reinterpret_cast<f32x4*>(&Ys[idx / DH][idx % DH]) = v[i];
      y[threadIdx.x + 512 * i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // ready for the next layer's projection
    }
  }
  __syncthreads();
  {
    const int o = 32 * qt + l31;
    f32x16 T = lnz::splat16(0.0f);
    // k-steps visit the input columns in the order i = 8 q + 4 h + t (t = 0..3 per quad of MFMAs):
    // this lane's A values are 4 consecutive floats of its Y row (one ds_read_b128), its B values
    // one dwordx4 of the pack_rows_k8 image of the long-scale weight block (1 KiB per wave load)
    // K <= 32 (the reference's graph configuration: K = 20): the upper 32 slots are padding — the
    // two waves of a column tile share the scales instead of the slot tiles (the launch is the
    // matrix time of ONE compute unit per graph: 46 -> 27 us for 64 graphs of 100 nodes).
    const bool small_k = K <= 32;
    const int ktile = small_k ? 0 : kt;
    const int s_mid = (S + 1) >> 1;
    const int s_beg = (small_k && kt == 1) ? s_mid : 0, s_end = (small_k && kt == 0) ? s_mid : S;
    const float* yrow = &Ys[32 * ktile + l31][4 * h];
    const int Q = S * dinp / 8;
    const f32x4* wq = reinterpret_cast<const f32x4*>(Wt) + ((int64_t)qt * Q) * 64 + lane;
    // The weight fragments of a whole scale (dinp / 8 <= 16 loads of 1 KiB per wave) are requested
    // one scale AHEAD of the MFMAs that use them, branch free (steps beyond dinp re-read step 0,
    // scales beyond S re-read the last one: with a conditional load the wait counters fall back to
    // "everything").  Loaded where they were used — two loads in front of every eight MFMAs — the
    // launch was one L2 round trip per 8 MFMAs: 53 us for 64 graphs of 100 nodes (7 launches of it
    // were 0.37 of the graph configuration's 0.80 ms forward), [see DESIGN 4.5b].
    const int NQ = dinp / 8;
    auto load_scale = [&](f32x4 (&wv)[16], const int sc) {
      const f32x4* ws = wq + (int64_t)(sc < S ? sc : S - 1) * NQ * 64;
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) wv[qq] = ws[(int64_t)(qq < NQ ? qq : 0) * 64];
    };
    auto run_scale = [&](const f32x4 (&wv)[16], const int sc) {
      f32x16 U = lnz::splat16(0.0f);
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        if (qq < NQ) {   // (uniform)
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(yrow + 8 * qq);
#pragma unroll
          for (int t = 0; t < 4; ++t) U = lnz::mfma32(a0[t], wv[qq][t], U);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) T[r] = fmaf(Gs[sc][32 * ktile + lnz::cd_row(r, h)], U[r], T[r]);
    };
    f32x4 w0[16], w1[16];
    load_scale(w0, s_beg);
    for (int sc = s_beg; sc < s_end; sc += 2) {
      load_scale(w1, sc + 1);
      run_scale(w0, sc);
      load_scale(w0, sc + 2);
      if (sc + 1 < s_end) run_scale(w1, sc + 1);
    }
    if (small_k) {   // (uniform) the second wave's scales join the first's; slots 32..63 are zero
      float* tx = &Ys[0][0];   // [column tile][register][lane]
      __syncthreads();         // every wave is through with Y
      if (kt == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tx[(qt * 16 + r) * 64 + lane] = T[r];
      }
      __syncthreads();
      if (kt == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r] += tx[(qt * 16 + r) * 64 + lane];
      } else {
        T = lnz::splat16(0.0f);
      }
    }
    const int64_t plane = (int64_t)B * DH * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot = 32 * kt + lnz::cd_row(r, h);
      u16 p[P];
      split_pieces<P>(T[r], p);
#pragma unroll
      for (int q = 0; q < P; ++q) Tt[q * plane + ((int64_t)b * DH + o) * 64 + slot] = p[q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv: Xout[rows][0..127] = relu( sum_c Lb_c[rows][:] Zt_c^T + Vb[rows][:] Tt^T + bias )
// ------------------------------------------------------------------------------------------
// B image in LDS: row n (output column) = 8 chunks of 16 B (64 k); chunk j of row n is stored at
// chunk position j ^ (n & 7).  ds_read_b128 is serviced in the four lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): with lane = 16 kq + column, a group holds
// 8 columns at chunk j and the other 8 at chunk j + 1, and the XOR key makes their sixteen 16 B
// slots cover the 256 B bank row exactly once; the 8-lane groups of the staging ds_write_b128
// (one row's 8 chunks) are conflict free as well.  (A 144 B row pitch looked conflict free for
// contiguous 16-lane groups and measured 44 % conflict cycles.)
template <int P, int NW>
__global__ __launch_bounds__(64 * NW) void large_conv_kernel(
    const u16* __restrict__ Lb, const u16* __restrict__ Vb, const u16* __restrict__ Zt,
    const u16* __restrict__ Tt, const float* __restrict__ bias, int B, int N, int Nk, int C,
    int tiles, int relu, float* __restrict__ Xout) {
  extern __shared__ __attribute__((aligned(16))) u16 Bs[];  // [2 buffers][P][128][BP]
  constexpr int NT = 64 * NW;            // threads
  constexpr int NPC = 1024 / NT;         // 16 B pieces of the B image per thread and plane
  // blockIdx -> (graph, row tile): workgroup i runs on XCD i % 8; the `tiles` row tiles of a
  // graph are consecutive workgroups of ONE XCD, so they share that L2's copy of the graph's Zt
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int b = xcd + 8 * (seq / tiles), tile = seq % tiles;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  // rows per wave: NRT row tiles of 16
  constexpr int NRT = 2, NG = NRT / 2;   // (NRT = 4 for P = 1 was measured: spills, 7 % slower)
  const int r0 = tile * (16 * NRT * NW) + 16 * NRT * w;
  const int RT = (N + 31) / 32;
  const int nkb = Nk / KB;
  const int64_t plane_l = (int64_t)B * C * RT * nkb * 2048, plane_v = (int64_t)B * RT * 2048;
  const int64_t plane_z = (int64_t)B * C * DH * Nk, plane_t = (int64_t)B * DH * 64;
  const int total = C * nkb + 1;  // + the lift block (V T)
  // this wave's first 32-row group (clamped: groups past the last compute a copy that is dropped)
  const int rg = (tile * NW + w) * NG < RT ? (tile * NW + w) * NG : RT - 1;
  const int g1 = (NG > 1 && rg + 1 < RT) ? 1 : 0;   // second group of a 64-row wave (or a copy)
  // staging role of this thread for the B image: pieces q = tid + NT i, row q >> 3, chunk q & 7
  // (the 8 chunks of a row = 128 contiguous bytes of Zt are 8 consecutive lanes)
  const int srow = tid >> 3, schunk = tid & 7;

  // Load streams.  Every lane walks two pointers block by block (uniform increments): its 16 B
  // piece of the wave's A chunks and its 32 B piece of the B image.  Node-space blocks: channel c,
  // k-block kb of Lb / Zt; the last block is the lift (Vb / Tt).  `la` / `lb` count the blocks
  // the A / B stream has issued (they run DEPTH - 1 resp. BDIST blocks ahead of the MFMAs).
  // The k-blocks of a channel are visited in a ROTATED order (start kb0, wrap around) that differs
  // from graph to graph, so the chip is spread over the k range at any moment, but is the SAME for
  // the row tiles of one graph: they run side by side on one XCD and walk the graph's Zt image in
  // step, so each of its k-blocks is fetched into that L2 once and hit by the other tiles (with a
  // per-workgroup start the counters showed 1.45 GB of Zt re-fetch per launch next to 4.3 GB of
  // operator stream).
  const int kb0 = (int)(((unsigned)b * 7u + ((unsigned)b >> 3)) % (unsigned)nkb);
  const u16* pa = Lb + (((int64_t)b * C * RT + rg) * nkb + kb0) * 2048 + lane * 8;
  const u16* pb = Zt + ((int64_t)b * C * DH + srow) * Nk + (int64_t)kb0 * KB + 8 * schunk;
  const int64_t wrap_step = -(int64_t)(nkb - 1) * KB;       // physical last k-block -> first
  const int64_t a_chan_step = (int64_t)RT * nkb * 2048;     // same row group, next channel
  const int64_t b_chan_step = (int64_t)DH * Nk;
  const int nnode = C * nkb;
  int la = 0, la_kb = 0, la_pos = kb0, lb = 0, lb_kb = 0, lb_pos = kb0;

  f32x4 acc[NRT][8];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[rt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // Software pipeline over the k-blocks, DEPTH - 1 blocks ahead: a k-block is only ~0.45 us of
  // matrix work per SIMD (P = 1) while an HBM load under load returns after ~2 us, and a CU must
  // keep >= 25 KB in flight for its share of 8 TB/s.  A fragments: register ring of DEPTH slots
  // (loaded straight from HBM); B image: register ring of DEPTH slots, written to the double-
  // buffered LDS tile one block before its use.  All ring indices are compile-time constants
  // (the loop is unrolled DEPTH times).
  constexpr int DEPTH = P == 1 ? 4 : 2;  // A ring: prefetch distance DEPTH - 1 blocks (P = 2: 4 measured, no gain)
  constexpr int DB = 2;                  // B ring: 2 register slots; distance 2 (P = 1) / 1 (P = 3)
  constexpr int BDIST = P == 1 ? 2 : 1;
  bf16x8 A[DEPTH][P][NRT][2];  // [slot][plane][row tile][k-step]
  f32x4 st[DB][P][NPC];      // B staging registers (16 B pieces per thread, plane and slot)
  // The streams are branch free: past the last block they keep re-reading it (valid memory, the
  // data is never used) — conditional loads would turn every ring register into a phi of two
  // definitions and cost the kernel its register allocation.
  auto load_a = [&](auto slotc) {  // next block of the A stream -> ring slot
    constexpr int slot = decltype(slotc)::value;
    const bool lift = la == nnode;
    pa = lift ? Vb + ((int64_t)b * RT + rg) * 2048 + lane * 8 : pa;
    const int64_t pl = la >= nnode ? plane_v : plane_l;
    // the wave's second row group: its chunks follow the first group's nkb chunks (one chunk for Vb)
    const int64_t goff = (int64_t)g1 * (la >= nnode ? 2048 : (int64_t)nkb * 2048);
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          A[slot][p][rt][ks] = LNZ_STREAM_LOAD(reinterpret_cast<const bf16x8*>(
              pa + p * pl + (rt >> 1) * goff + ((rt & 1) * 2 + ks) * 512));
    ++la_kb;
    int64_t step = la_pos == nkb - 1 ? -(int64_t)(nkb - 1) * 2048 : (int64_t)2048;
    la_pos = la_pos == nkb - 1 ? 0 : la_pos + 1;
    step += la_kb == nkb ? a_chan_step : (int64_t)0;
    step = la >= nnode ? (int64_t)0 : step;
    la_kb = la_kb == nkb ? 0 : la_kb;
    ++la;
    pa += step;
  };
  auto load_b = [&](auto slotc) {  // next block of the B stream -> register slot
    constexpr int slot = decltype(slotc)::value;
    pb = lb == nnode ? Tt + ((int64_t)b * DH + srow) * 64 + 8 * schunk : pb;
    const int64_t pl = lb >= nnode ? plane_t : plane_z;
    const int64_t rstep = (int64_t)(NT / 8) * (lb >= nnode ? 64 : Nk);  // NT / 8 rows further
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int i = 0; i < NPC; ++i)
        st[slot][p][i] = *reinterpret_cast<const f32x4*>(pb + p * pl + i * rstep);
    ++lb_kb;
    int64_t step = lb_pos == nkb - 1 ? wrap_step : (int64_t)KB;
    lb_pos = lb_pos == nkb - 1 ? 0 : lb_pos + 1;
    step += lb_kb == nkb ? b_chan_step : (int64_t)0;
    step = lb >= nnode ? (int64_t)0 : step;
    lb_kb = lb_kb == nkb ? 0 : lb_kb;
    ++lb;
    pb += step;
  };
  auto store_b = [&](int g, auto slotc) {  // (past the last block: an image nobody reads)
    constexpr int slot = decltype(slotc)::value;
    const int buf = g & 1;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        const int row = srow + (NT / 8) * i;  // (NT / 8) % 8 == 0: same swizzle key as srow
        *reinterpret_cast<f32x4*>(Bs + ((int64_t)(buf * P + p) * DH + row) * BP +
                                  8 * (schunk ^ (srow & 7))) = st[slot][p][i];
      }
  };
  // one k-block (gs = g mod 4, compile time): issue the loads of the blocks ahead, run the MFMAs
  // of block g from A[slot] and LDS buffer g & 1, publish the B image of block g + 1, one barrier.
  // The B registers of block g were written to LDS during block g - 1, so slot g % 2 is free for
  // block g + 2 from the start of block g.
  auto block = [&](int g, auto gsc) {
    constexpr int gs = decltype(gsc)::value;
    load_a(std::integral_constant<int, (gs + DEPTH - 1) % DEPTH>{});
    if constexpr (BDIST == 2) load_b(std::integral_constant<int, gs % 2>{});
    // the block's loads go out BEFORE its MFMAs: left free, hipcc sinks them to the end of the
    // first block of the unrolled body and waits vmcnt(0) there — the A ring drained once per
    // four blocks
    __builtin_amdgcn_sched_barrier(0);
    const u16* bt = Bs + (int64_t)(g & 1) * P * DH * BP;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        bf16x8 bf[P];
#pragma unroll
        for (int p = 0; p < P; ++p)
          bf[p] = *reinterpret_cast<const bf16x8*>(bt + ((int64_t)p * DH + 16 * nt + l15) * BP +
                                                   8 * ((4 * ks + kq) ^ (l15 & 7)));
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
          for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
            for (int i = 0; i <= ord; ++i)
              // (operands swapped: D[feature][node] — see the epilogue)
              acc[rt][nt] = mfma_16x16x32<P>(bf[ord - i], A[gs % DEPTH][i][rt][ks], acc[rt][nt]);
      }
    if constexpr (BDIST == 1) load_b(std::integral_constant<int, (gs + 1) % 2>{});
    store_b(g + 1, std::integral_constant<int, (gs + 1) % 2>{});
    __syncthreads();
  };

  // (the prologue's loads in program order: hipcc otherwise reorders them, the s_waitcnt of the
  // loop head has to cover the reordered entry state as well, and block 0 of every 4-block
  // iteration drained the whole A ring with vmcnt(0))
  load_a(std::integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  load_b(std::integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DEPTH > 2) {
    load_a(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(std::integral_constant<int, 2 % DEPTH>{});
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (BDIST == 2) load_b(std::integral_constant<int, 1>{});
  __builtin_amdgcn_sched_barrier(0);
  store_b(0, std::integral_constant<int, 0>{});
  __syncthreads();
  int g = 0;
  for (; g + 4 <= total; g += 4) {  // steady state: no conditionals
    block(g, std::integral_constant<int, 0>{});
    block(g + 1, std::integral_constant<int, 1>{});
    block(g + 2, std::integral_constant<int, 2>{});
    block(g + 3, std::integral_constant<int, 3>{});
  }
  if (g < total) {  // 1..3 remaining blocks
    block(g, std::integral_constant<int, 0>{});
    if (g + 1 < total) {
      block(g + 1, std::integral_constant<int, 1>{});
      if (g + 2 < total) block(g + 2, std::integral_constant<int, 2>{});
    }
  }
  // epilogue.  The MFMAs ran with the operands SWAPPED (first = the B image's feature rows, second
  // = the operator's node rows; both fragment layouts are "lane = 16 kq + index, 8 consecutive k",
  // and the products are the same numbers in the same k order): D[feature][node], so the C/D
  // layout col = lane & 15, row = 4 * (lane >> 4) + reg puts FOUR CONSECUTIVE FEATURES of one node
  // into a lane's four registers — one 16-byte store per tile instead of four 4-byte ones.  With
  // the row-per-register layout the 64 dword stores per lane were a ~20 us store tail per
  // workgroup (issue bound), a third of a one-channel tile's time.
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int col = 16 * nt + 4 * kq;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const int row = r0 + 16 * rt + l15;
      if (row < N) {
        f32x4 v = acc[rt][nt] * ElemTraits<P>::kAInv + bv;   // (2^-10: fp16 mode's A scale)
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(Xout + ((int64_t)b * N + row) * DH + col) = v;
      }
    }
  }
}

}  // namespace

extern "C" int64_t lnz_large_nk(int N) { return ((int64_t)N + KB - 1) / KB * KB; }

extern "C" int lnz_large_pack_operators_fold(const float* L, int64_t stride_b, int64_t stride_r,
                                             int64_t stride_c, int64_t stride_ch, const float* V,
                                             int B, int N, int C, int K, int planes,
                                             const int32_t* chan_src_host, int n_src,
                                             const int32_t* chan_rep_host, const int32_t* chan_check_host,
                                             unsigned long long* neq, uint16_t* Lb, uint16_t* Vb,
                                             lnz_stream_t stream) {
  LNZ_REQUIRE(L && V && Lb && Vb && B > 0 && N > 0 && C > 0 && K > 0, LNZ_EINVAL,
              "lnz_large_pack_operators: bad arguments (B=%d N=%d C=%d K=%d)", B, N, C, K);
  LNZ_REQUIRE(K <= 64, LNZ_ENOTSUP, "lnz_large_pack_operators: K=%d > 64", K);
  LNZ_REQUIRE(planes >= 1 && planes <= 3, LNZ_EINVAL, "lnz_large_pack_operators: planes must be 1, 2 or 3");
  LNZ_REQUIRE(C <= 8, LNZ_ENOTSUP, "lnz_large_pack_operators: C=%d > 8 channels", C);
  LargeChanMap cm = {};
  if (!chan_src_host) n_src = C;  // identity: every channel packed
  const bool mapped = chan_src_host != nullptr;
  LNZ_REQUIRE(!mapped || (chan_rep_host && n_src >= 1 && n_src <= C), LNZ_EINVAL,
              "lnz_large_pack_operators_fold: channel map needs chan_rep_host and 1 <= n_src <= C "
              "(C=%d n_src=%d)", C, n_src);
  if (mapped) {
    cm.nsrc = (signed char)n_src;
    for (int d = 0; d < n_src; ++d) {
      LNZ_REQUIRE(chan_src_host[d] >= 0 && chan_src_host[d] < C && (d == 0 || chan_src_host[d] > chan_src_host[d - 1]),
                  LNZ_EINVAL, "lnz_large_pack_operators_fold: chan_src_host must be ascending channels");
      cm.src[d] = (signed char)chan_src_host[d];
    }
    for (int c = 0; c < C; ++c) {
      LNZ_REQUIRE(chan_rep_host[c] >= 0 && chan_rep_host[c] < n_src && chan_src_host[chan_rep_host[c]] <= c, LNZ_EINVAL,
                  "lnz_large_pack_operators_fold: chan_rep_host[%d] must name a packed slot of an "
                  "earlier (or the same) channel", c);
      cm.rep[c] = (signed char)chan_rep_host[c];
      cm.check[c] = (signed char)((neq && (!chan_check_host || chan_check_host[c])) ? 1 : 0);
    }
    for (int d = 0; d < n_src; ++d)
      LNZ_REQUIRE(chan_rep_host[chan_src_host[d]] == d, LNZ_EINVAL,
                  "lnz_large_pack_operators_fold: a packed channel must represent itself");
  } else {
    cm.nsrc = (signed char)C;
    for (int c = 0; c < C; ++c) {
      cm.src[c] = cm.rep[c] = (signed char)c;
      cm.check[c] = (signed char)(neq ? 1 : 0);
    }
  }
  const int nkb = (int)(lnz_large_nk(N) / KB);
  dim3 grid((N + 31) / 32, B);
  if (planes == 1)
    hipLaunchKernelGGL(large_pack_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, L, stride_b,
                       stride_r, stride_c, stride_ch, V, B, N, nkb, C, K, cm, neq, Lb, Vb);
  else if (planes == 2)
    hipLaunchKernelGGL(large_pack_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, L, stride_b,
                       stride_r, stride_c, stride_ch, V, B, N, nkb, C, K, cm, neq, Lb, Vb);
  else
    hipLaunchKernelGGL(large_pack_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, L, stride_b,
                       stride_r, stride_c, stride_ch, V, B, N, nkb, C, K, cm, neq, Lb, Vb);
  return lnz::check_launch("lnz_large_pack_operators");
}

extern "C" int lnz_large_pack_operators(const float* L, int64_t stride_b, int64_t stride_r,
                                        int64_t stride_c, int64_t stride_ch, const float* V, int B,
                                        int N, int C, int K, int planes, uint16_t* Lb,
                                        uint16_t* Vb, lnz_stream_t stream) {
  return lnz_large_pack_operators_fold(L, stride_b, stride_r, stride_c, stride_ch, V, B, N, C, K,
                                       planes, nullptr, 0, nullptr, nullptr, nullptr, Lb, Vb, stream);
}

extern "C" int lnz_large_gemm1(const float* X, int ldx, int din, const uint16_t* Wf, int B, int N,
                               int C, int planes, uint16_t* Zt, lnz_stream_t stream) {
  LNZ_REQUIRE(X && Wf && Zt && B > 0 && N > 0 && C > 0 && din > 0 && ldx >= din, LNZ_EINVAL,
              "lnz_large_gemm1: bad arguments");
  LNZ_REQUIRE(planes >= 1 && planes <= 3, LNZ_EINVAL, "lnz_large_gemm1: planes must be 1, 2 or 3");
  const int dinp = (din + 15) / 16 * 16;
  LNZ_REQUIRE(dinp <= 128, LNZ_ENOTSUP, "lnz_large_gemm1: input width %d > 128", din);
  const int Nk = (int)lnz_large_nk(N);
  const size_t lds = ((size_t)planes * 128 * (dinp + 8) + 128 * 136) * sizeof(uint16_t);
  dim3 grid((N + 127) / 128, B);
  // the dynamic-LDS limit is a PER-DEVICE attribute (one process may drive several devices:
  // nn.DataParallel, runner/qm8_runner.py:62): set on the current device at every launch
#define LNZ_LAUNCH_GEMM1(PP)                                                                        \
  do {                                                                                              \
    LNZ_DYNAMIC_LDS(large_gemm1_kernel<PP>, \
      lds, "conv_large.hip");                \
    hipLaunchKernelGGL(large_gemm1_kernel<PP>, grid, dim3(256), lds, (hipStream_t)stream, X, ldx,   \
                       din, dinp, Wf, B, N, Nk, C, Zt);                                             \
  } while (0)
  if (planes == 1) LNZ_LAUNCH_GEMM1(1);
  else if (planes == 2) LNZ_LAUNCH_GEMM1(2);
  else LNZ_LAUNCH_GEMM1(3);
#undef LNZ_LAUNCH_GEMM1
  return lnz::check_launch("lnz_large_gemm1");
}

extern "C" int lnz_large_gemm1_rows(const float* X, int ldx, int din, const uint16_t* Wf, int B,
                                    int N, uint16_t* Z, lnz_stream_t stream) {
  LNZ_REQUIRE(X && Wf && Z && B > 0 && N > 0 && din > 0 && ldx >= din, LNZ_EINVAL,
              "lnz_large_gemm1_rows: bad arguments");
  const int dinp = (din + 15) / 16 * 16;
  LNZ_REQUIRE(dinp <= 128, LNZ_ENOTSUP, "lnz_large_gemm1_rows: input width %d > 128", din);
  const size_t lds = ((size_t)128 * (dinp + 8) + 128 * 136) * sizeof(uint16_t);
  LNZ_DYNAMIC_LDS((large_gemm1_kernel<1, true>), lds, "conv_large.hip");
  hipLaunchKernelGGL((large_gemm1_kernel<1, true>), dim3((N + 127) / 128, B), dim3(256), lds,
                     (hipStream_t)stream, X, ldx, din, dinp, Wf, B, N, 0, 1, Z);
  return lnz::check_launch("lnz_large_gemm1_rows");
}

static int large_spectral_launch(const float* X, int ldx, int din, const float* V, const float* G,
                                 const float* Wt, int B, int N, int K, int S, int planes,
                                 float* Ybuf, uint16_t* Tt, const uint16_t* Wf_rows, uint16_t* Z_rows,
                                 lnz_stream_t stream) {
  LNZ_REQUIRE(X && V && G && Wt && Ybuf && Tt && B > 0 && N > 0 && K > 0 && S > 0 && din > 0 &&
                  ldx >= din,
              LNZ_EINVAL, "lnz_large_spectral: bad arguments");
  LNZ_REQUIRE(K <= 64, LNZ_ENOTSUP, "lnz_large_spectral: K=%d > 64", K);
  LNZ_REQUIRE(S <= 16, LNZ_ENOTSUP, "lnz_large_spectral: %d long scales > 16", S);
  LNZ_REQUIRE(planes >= 1 && planes <= 3, LNZ_EINVAL, "lnz_large_spectral: planes must be 1, 2 or 3");
  const int dinp = (din + 15) / 16 * 16;
  LNZ_REQUIRE(dinp <= 128, LNZ_ENOTSUP, "lnz_large_spectral: input width %d > 128", din);
  // row chunks: one workgroup per compute unit, at least 128 rows each (multiple of 16).  (A
  // workgroup's first tile is an exposed HBM round trip and its partial tile ends in 8 K atomics:
  // B = 256, N = 2048 by the number of workgroups: 4096 0.178 ms, 2048 0.155, 512 0.131, 256 0.128.)
  static const int target_wgs = [] {  // read once (thread-safe static initialisation)
    const char* e = getenv("LNZ_LARGE_PROJECT_WGS");
    const int v = e ? atoi(e) : 256;
    return v < 1 ? 256 : v;
  }();
  int chunks = (target_wgs + B - 1) / B;
  int rows = ((N + chunks - 1) / chunks + 63) / 64 * 64;
  if (rows < 128) rows = 128;
  chunks = (N + rows - 1) / rows;
  if (Wf_rows)
    hipLaunchKernelGGL(large_project_kernel<true>, dim3(chunks, B), dim3(512), 0, (hipStream_t)stream, X,
                       ldx, din, dinp, V, N, K, rows, Ybuf, Wf_rows, Z_rows);
  else
    hipLaunchKernelGGL(large_project_kernel<false>, dim3(chunks, B), dim3(512), 0, (hipStream_t)stream, X,
                       ldx, din, dinp, V, N, K, rows, Ybuf, (const u16*)nullptr, (u16*)nullptr);
  if (planes == 1)
    hipLaunchKernelGGL(large_spectral_kernel<1>, dim3(B), dim3(512), 0, (hipStream_t)stream, dinp,
                       Ybuf, G, Wt, B, K, S, Tt);
  else if (planes == 2)
    hipLaunchKernelGGL(large_spectral_kernel<2>, dim3(B), dim3(512), 0, (hipStream_t)stream, dinp,
                       Ybuf, G, Wt, B, K, S, Tt);
  else
    hipLaunchKernelGGL(large_spectral_kernel<3>, dim3(B), dim3(512), 0, (hipStream_t)stream, dinp,
                       Ybuf, G, Wt, B, K, S, Tt);
  return lnz::check_launch("lnz_large_spectral");
}

extern "C" int lnz_large_pack_vectors(const float* V, int B, int N, int K, int planes, uint16_t* Vb,
                                      lnz_stream_t stream) {
  LNZ_REQUIRE(V && Vb && B > 0 && N > 0 && K > 0, LNZ_EINVAL, "lnz_large_pack_vectors: bad arguments");
  LNZ_REQUIRE(K <= 64, LNZ_ENOTSUP, "lnz_large_pack_vectors: K=%d > 64", K);
  LNZ_REQUIRE(planes >= 1 && planes <= 3, LNZ_EINVAL, "lnz_large_pack_vectors: planes must be 1, 2 or 3");
  dim3 grid((N + 31) / 32, B);
  if (planes == 1)
    hipLaunchKernelGGL(large_pack_vectors_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, V, B, N, K, Vb);
  else if (planes == 2)
    hipLaunchKernelGGL(large_pack_vectors_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, V, B, N, K, Vb);
  else
    hipLaunchKernelGGL(large_pack_vectors_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, V, B, N, K, Vb);
  return lnz::check_launch("lnz_large_pack_vectors");
}

extern "C" int lnz_large_spectral(const float* X, int ldx, int din, const float* V, const float* G,
                                  const float* Wt, int B, int N, int K, int S, int planes,
                                  float* Ybuf, uint16_t* Tt, lnz_stream_t stream) {
  return large_spectral_launch(X, ldx, din, V, G, Wt, B, N, K, S, planes, Ybuf, Tt, nullptr, nullptr, stream);
}

extern "C" int lnz_large_spectral_gemm1_rows(const float* X, int ldx, int din, const float* V,
                                             const float* G, const float* Wt, const uint16_t* Wf,
                                             int B, int N, int K, int S, float* Ybuf, uint16_t* Tt,
                                             uint16_t* Z, lnz_stream_t stream) {
  LNZ_REQUIRE(Wf && Z, LNZ_EINVAL, "lnz_large_spectral_gemm1_rows: bad arguments");
  return large_spectral_launch(X, ldx, din, V, G, Wt, B, N, K, S, 1, Ybuf, Tt, Wf, Z, stream);
}

extern "C" int lnz_large_conv(const uint16_t* Lb, const uint16_t* Vb, const uint16_t* Zt,
                              const uint16_t* Tt, const float* bias, int B, int N, int C, int planes,
                              int relu, float* Xout, lnz_stream_t stream) {
  // C == 0: no node-space channel — the launch is the lift V T (+ bias, activation) alone, the
  // form the sparse conv (csrc/conv_sparse.hip) adds its gathered products to; Lb / Zt unused
  LNZ_REQUIRE(Vb && Tt && bias && Xout && B > 0 && N > 0 && C >= 0 && (C == 0 || (Lb && Zt)), LNZ_EINVAL,
              "lnz_large_conv: bad arguments");
  if (C == 0) Lb = Vb, Zt = Tt;   // (the streams start on their lift block; never dereferenced as operators)
  LNZ_REQUIRE(planes >= 1 && planes <= 3, LNZ_EINVAL, "lnz_large_conv: planes must be 1, 2 or 3");
  const int Nk = (int)lnz_large_nk(N);
  const size_t lds = (size_t)2 * planes * DH * BP * sizeof(uint16_t);
  // 8-wave workgroups = 256-row tiles.  (LNZ_LARGE_CONV_WAVES=4: 128-row tiles, two workgroups per
  // CU that drift apart — measured 12 % slower: twice the B-image staging per operator byte.)
  static const int nw = [] {  // read once (thread-safe static initialisation)
    const char* e = getenv("LNZ_LARGE_CONV_WAVES");
    return (e && atoi(e) == 4) ? 4 : 8;
  }();
  const int nwp = planes >= 2 ? 8 : nw;  // the multi-plane B images (64 / 96 KB): one workgroup per CU
  const int tile_rows = 32 * nwp;
  const int tiles = (N + tile_rows - 1) / tile_rows;
  const int grid = 8 * tiles * ((B + 7) / 8);
  // (the dynamic-LDS limit is a per-device attribute: set on the current device at every launch)
#define LNZ_LAUNCH_CONV(PP, WW)                                                                     \
  do {                                                                                              \
    LNZ_DYNAMIC_LDS((large_conv_kernel<PP, WW>), lds, "conv_large.hip");                            \
    hipLaunchKernelGGL((large_conv_kernel<PP, WW>), dim3(grid), dim3(64 * WW), lds,                 \
                       (hipStream_t)stream, Lb, Vb, Zt, Tt, bias, B, N, Nk, C, tiles, relu, Xout);  \
  } while (0)
  if (planes == 1 && nwp == 4) LNZ_LAUNCH_CONV(1, 4);
  else if (planes == 1) LNZ_LAUNCH_CONV(1, 8);
  else if (planes == 2) LNZ_LAUNCH_CONV(2, 8);
  else LNZ_LAUNCH_CONV(3, 8);
#undef LNZ_LAUNCH_CONV
  return lnz::check_launch("lnz_large_conv");
}
