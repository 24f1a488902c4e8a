// R7 (second half) + R9 + R10: the fused LanczosNet forward.
//
// One workgroup (dhid/32 wavefronts) per PAIR of molecules runs the WHOLE network on chip:
// embedding -> num_layer x [ X' = relu( sum_c M_c X W_c^T + b ) ] -> gated head -> masked mean.
//
// Per layer and message channel c two chained matrix-core GEMMs (v_mfma_f32_32x32x2_f32, exact
// fp32 fma chains), evaluated as  M_c (X W_c^T)  instead of the reference's  (M_c X) W_c^T :
//
//   GEMM1  Z_c [32 nodes x dhid] = X [32 x din] * W_c^T
//          A = X from LDS (row-major, pitch 132 floats, one ds_read_b128 = 4 k-steps),
//          B = W_c pre-packed in fragment order (one global_load_dwordx4 = 4 k-steps, L2 hits),
//          wave w owns output-feature tile w for both molecules: each weight fragment feeds
//          2 x 4 MFMAs (the per-CU vector-memory path, not L2, limits a 1-molecule tiling).
//   GEMM2  out += M_c [32 x 32] * Z_c
//          B = the C/D registers of GEMM1 *as they are*: register r of lane (j, hh) holds
//              Z_c[cd_row(r,hh)][j], which is exactly what k-step r needs when the contraction
//              index is visited in the order m(r,hh) = cd_row(r,hh);
//          A = M_c in the matching order: edge/short channels from the packed Laplacian
//              (4 coalesced dwordx4 per channel), spectral channels built in registers:
//              L_s = V diag(g_s) V^T by 10 MFMAs, whose C/D registers are — L_s being
//              symmetric — already the A fragments.
//   short-diffusion channels apply M = L_0 p times to Z_c in registers (same chaining).
//
// So `cat(msg)` (model/lanczos_net.py:180), the [B,N,N,S] filter stack (:123) and the strided
// `L[:,:,:,ii]` clones (:172-178) never exist; per layer the only LDS traffic is X (written once
// by the epilogue, read by both waves) and the only barrier is one __syncthreads per layer.
// HBM bytes per molecule: Lp 28,672 + V 2,560 + G 4,480 + ids 256 + mask 32 in, 64 out; the
// 7.4 MB of packed weights are shared by all workgroups and stay L2 / Infinity-Cache resident.
#include "common.hpp"

namespace {

constexpr int MOLS = 2;      // molecules per workgroup; every wave works on both
constexpr int PITCH = 132;   // LDS row pitch (floats): conflict-free ds_read_b128 A fragments
constexpr int KHMAX = 16;    // eigen slots per lane half (K <= 32)

__device__ inline f32x16 frag_from4(const float4 (&v)[4]) {
  f32x16 f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f[4 * g + 0] = v[g].x;
    f[4 * g + 1] = v[g].y;
    f[4 * g + 2] = v[g].z;
    f[4 * g + 3] = v[g].w;
  }
  return f;
}

// NWV wavefronts per workgroup = dhid / 32: wave w owns output-feature tile w for BOTH molecules
// of the workgroup, so every packed-weight fragment it loads feeds 2 x 4 MFMAs.
// FK = filter kind: 0 = diagonal gains on Ritz vectors (LanczosNet), 1 = dense K x K filters on
// the Lanczos basis (AdaLanczosNet: M = Q DD Q^T, model/ada_lanczos_net.py:280-281).
template <int NWV, int KHT, int FK>
__global__ __launch_bounds__(64 * NWV) void lanczosnet_forward_kernel(const lnz_forward_args a) {
  __shared__ __attribute__((aligned(16))) float Xs[2][MOLS][32][PITCH];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 31, hh = lane >> 5;
  const int N = a.N, K = a.K, B = a.B;
  const int dhid = a.dhid;
  const int C = a.n_short + a.n_long + a.n_edge;
  // molecule ids (an odd batch repeats the last one; its result is dropped).  a.order: optional
  // permutation that pairs a small with a large molecule so that all workgroups skip the same
  // amount of padded GEMM2 work (the launch is one round: its time is the slowest workgroup's).
  int mb[MOLS];
#pragma unroll
  for (int m = 0; m < MOLS; ++m) {
    int x = blockIdx.x * MOLS + m;
    x = x < B ? x : B - 1;
    mb[m] = a.order ? a.order[x] : x;
  }

  // ---- embedding gather (model/lanczos_net.py:154) / float features (lanczos_net_general.py:156)
  {
    const int d4 = a.din0 >> 2;
    for (int idx = tid; idx < MOLS * 32 * d4; idx += 64 * NWV) {
      int m = idx / (32 * d4);
      int rem = idx - m * 32 * d4;
      int row = rem / d4, c4 = rem - row * d4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N) {
        if (a.node_feat) {
          int64_t id = a.node_feat[(int64_t)mb[m] * N + row];
          id = id < 0 ? 0 : (id >= a.num_atom ? a.num_atom - 1 : id);
          v = reinterpret_cast<const float4*>(a.embedding + id * a.din0)[c4];
        } else {
          v = reinterpret_cast<const float4*>(a.node_feat_f + ((int64_t)mb[m] * N + row) * a.din0)[c4];
        }
      }
      *reinterpret_cast<float4*>(&Xs[0][m][row][4 * c4]) = v;
    }
  }

  // ---- rows of M_c beyond the last real node are zero (zero-padded Laplacian rows/cols, zero
  //      rows of V): GEMM2 k-steps 4g..4g+3 only touch node rows 8g..8g+7, so a molecule with
  //      n real nodes needs 4*ceil(n/8) of the 16 steps (wave-uniform per molecule).
  int g2steps[MOLS];
#pragma unroll
  for (int m = 0; m < MOLS; ++m) {
    int last = 0;
    for (int i = lane; i < N; i += 64) last = a.mask[(int64_t)mb[m] * N + i] ? i + 1 : last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
    g2steps[m] = __builtin_amdgcn_readfirstlane(((last + 7) >> 3) * 4);
  }

  // ---- basis fragments.  FK = 0: vreg[m][t] = V[mol m][j][KH*hh + t] (Ritz vectors, two k-halves)
  //      FK = 1: vreg[m][r] = Q[mol m][j][cd_row(r,hh)] — the k-order that lets the same registers
  //      serve as B operand of R = DD Q^T and as A operand of L_s = Q R.
  const int KH = (K + 1) >> 1;
  float vreg[MOLS][KHT];
#pragma unroll
  for (int m = 0; m < MOLS; ++m) {
#pragma unroll
    for (int t = 0; t < KHT; ++t) {
      int k = FK ? lnz::cd_row(t, hh) : KH * hh + t;
      bool ok = FK ? (k < K) : (t < KH && k < K);
      vreg[m][t] = (ok && j < N) ? a.V[((int64_t)mb[m] * N + j) * K + k] : 0.0f;
    }
  }
  __syncthreads();

#ifdef LNZ_PROFILE_PHASES
  long long t_g1 = 0, t_g2 = 0, t_ep = 0, t_all = clock64();
#define LNZ_T0 long long _t0 = clock64();
#define LNZ_ACC(x) { long long _t1 = clock64(); x += _t1 - _t0; _t0 = _t1; }
#else
#define LNZ_T0
#define LNZ_ACC(x)
#endif
  int cur = 0;
  for (int l = 0; l < a.num_layer; ++l) {
    const int din = l == 0 ? a.din0 : dhid;
    const int Q = din >> 3;
    const float4* __restrict__ Wl = reinterpret_cast<const float4*>(a.Wp + a.w_off[l]);
    const float* __restrict__ bl = a.bias + a.b_off[l];

    f32x16 out[MOLS];
    {
      const float bv = bl[32 * wave + j];
#pragma unroll
      for (int m = 0; m < MOLS; ++m) out[m] = lnz::splat16(bv);
    }

    // B-operand stream of this wave's feature tile: contiguous over g = c*Q + q for the whole
    // layer, read through a 4-slot register ring with prefetch distance 3 steps.  One running
    // pointer + immediate offsets; over-reads 3 steps past the layer (host keeps 3 KiB slack).
    const int Gtot = C * Q;
    const float4* __restrict__ wp = Wl + (int64_t)wave * Gtot * 64 + lane;
    float4 ring[4];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) ring[sl] = wp[sl * 64];
    const float* xrow[MOLS];
#pragma unroll
    for (int m = 0; m < MOLS; ++m) xrow[m] = &Xs[cur][m][j][4 * hh];

    // Operands of GEMM2 (per molecule, 16 registers: the 10 spectral gains g_s[k] of a long
    // channel OR the four float4 Laplacian fragments of an edge/short channel) are fetched one
    // channel ahead, right after the previous fragments are consumed and before that
    // molecule's GEMM2 — so they have >= 16 MFMAs to land and are waited for together with the
    // oldest ring slot at the next GEMM1 loop header.
    float mop[MOLS][16];
    auto fetch_m_operands = [&](int c, int m) {
      const bool lng = (c >= a.n_short) && (c < a.n_short + a.n_long);
      if (lng) {
        if (FK == 0) {
          const float* gp = a.G + (((int64_t)l * B + mb[m]) * a.n_long + (c - a.n_short)) * K;
#pragma unroll
          for (int t = 0; t < KHT; ++t) {
            int k = KH * hh + t;
            mop[m][t] = (t < KH && k < K) ? gp[k] : 0.0f;
          }
        } else {
          // row j of the symmetric K x K filter DD_s, columns in cd_row order
          const float* dp = a.G + ((((int64_t)l * B + mb[m]) * a.n_long + (c - a.n_short)) * K + j) * K;
#pragma unroll
          for (int t = 0; t < KHT; ++t) {
            int k2 = lnz::cd_row(t, hh);
            mop[m][t] = (j < K && k2 < K) ? dp[k2] : 0.0f;
          }
        }
      } else {
        const int e = c < a.n_short ? 0 : c - a.n_short - a.n_long;
        const float4* lp =
            reinterpret_cast<const float4*>(a.Lp) + ((int64_t)mb[m] * a.n_edge + e) * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = lp[g * 64 + lane];
          mop[m][4 * g + 0] = v.x;
          mop[m][4 * g + 1] = v.y;
          mop[m][4 * g + 2] = v.z;
          mop[m][4 * g + 3] = v.w;
        }
      }
    };
#pragma unroll
    for (int m = 0; m < MOLS; ++m) fetch_m_operands(0, m);

    for (int c = 0; c < C; ++c) {
      const bool is_long = (c >= a.n_short) && (c < a.n_short + a.n_long);

      LNZ_T0
      // ---------------- GEMM1: Z_m = X_m W_c^T ----------------
      f32x16 Z[MOLS];
#pragma unroll
      for (int m = 0; m < MOLS; ++m) Z[m] = lnz::splat16(0.0f);
      const float* xq[MOLS];
      float4 acur[MOLS];
#pragma unroll
      for (int m = 0; m < MOLS; ++m) {
        xq[m] = xrow[m];
        acur[m] = *reinterpret_cast<const float4*>(xq[m]);
      }
#pragma unroll 1
      for (int q0 = 0; q0 < Q; q0 += 4) {
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
          ring[(u4 + 3) & 3] = wp[(u4 + 3) * 64];
          // next A fragments (the read one step past the channel's last is in-bounds, unused)
          float4 anext[MOLS];
#pragma unroll
          for (int m = 0; m < MOLS; ++m)
            anext[m] = *reinterpret_cast<const float4*>(xq[m] + 8 * (u4 + 1));
          // keep the prefetches ahead of this step's MFMAs (hipcc otherwise sinks all loads of
          // the unrolled body to its end and waits vmcnt(0) at the top of the next iteration)
          __builtin_amdgcn_sched_barrier(0);
          const float4 bv = ring[u4];
#pragma unroll
          for (int m = 0; m < MOLS; ++m) Z[m] = lnz::mfma32(acur[m].x, bv.x, Z[m]);
#pragma unroll
          for (int m = 0; m < MOLS; ++m) Z[m] = lnz::mfma32(acur[m].y, bv.y, Z[m]);
#pragma unroll
          for (int m = 0; m < MOLS; ++m) Z[m] = lnz::mfma32(acur[m].z, bv.z, Z[m]);
#pragma unroll
          for (int m = 0; m < MOLS; ++m) Z[m] = lnz::mfma32(acur[m].w, bv.w, Z[m]);
#pragma unroll
          for (int m = 0; m < MOLS; ++m) acur[m] = anext[m];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MOLS; ++m) xq[m] += 32;
        wp += 4 * 64;
      }

      LNZ_ACC(t_g1)
      // ---------------- per molecule: M_c fragments, next operands, GEMM2 ----------------
#pragma unroll
      for (int m = 0; m < MOLS; ++m) {
        f32x16 Mf;
        if (is_long) {
          f32x16 acc = lnz::splat16(0.0f);
          if (FK == 0) {
#pragma unroll
            for (int t = 0; t < KHT; ++t) {
              if (t < KH) acc = lnz::mfma32(vreg[m][t] * mop[m][t], vreg[m][t], acc);
            }
          } else {
            // R[k1][n] = sum_k2 DD[k1][k2] Q[n][k2]  then  L_s[i][n] = sum_k1 Q[i][k1] R[k1][n]
            f32x16 R = lnz::splat16(0.0f);
#pragma unroll
            for (int t = 0; t < KHT; ++t) R = lnz::mfma32(mop[m][t], vreg[m][t], R);
#pragma unroll
            for (int t = 0; t < KHT; ++t) acc = lnz::mfma32(vreg[m][t], R[t], acc);
          }
          Mf = acc;  // L_s[cd_row(r,hh)][j] == L_s[j][cd_row(r,hh)]  (symmetric)
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) Mf[r] = mop[m][r];
        }
        if (c + 1 < C) fetch_m_operands(c + 1, m);

        // short diffusion: Z <- L_0^(p-1) Z
        if (c < a.n_short) {
          const int p = a.short_dist[c];
          for (int rep = 1; rep < p; ++rep) {
            f32x16 T = lnz::splat16(0.0f);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (r < g2steps[m]) T = lnz::mfma32(Mf[r], Z[m][r], T);
            }
            Z[m] = T;
          }
        }
        // GEMM2: out_m += M_c,m Z_m
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          if (r < g2steps[m]) {
            out[m] = lnz::mfma32(Mf[r + 0], Z[m][r + 0], out[m]);
            out[m] = lnz::mfma32(Mf[r + 1], Z[m][r + 1], out[m]);
            out[m] = lnz::mfma32(Mf[r + 2], Z[m][r + 2], out[m]);
            out[m] = lnz::mfma32(Mf[r + 3], Z[m][r + 3], out[m]);
          }
        }
      }
      LNZ_ACC(t_g2)
    }

    // ---------------- epilogue: ReLU, X' -> LDS (other buffer), one barrier per layer -------
    LNZ_T0
    const int nxt = cur ^ 1;
#pragma unroll
    for (int m = 0; m < MOLS; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Xs[nxt][m][lnz::cd_row(r, hh)][32 * wave + j] = fmaxf(out[m][r], 0.0f);
    }
    __syncthreads();
    cur = nxt;
    LNZ_ACC(t_ep)
  }
#ifdef LNZ_PROFILE_PHASES
  if (a.state_out && lane == 0 && blockIdx.x < 8) {
    float* d = a.state_out + ((int64_t)B * 32 * dhid) + (blockIdx.x * 4 + wave) * 4;
    d[0] = (float)t_g1; d[1] = (float)t_g2; d[2] = (float)t_ep; d[3] = (float)(clock64() - t_all);
  }
#endif

  // ---- optional debug/test output of the final node state
  if (a.state_out) {
    for (int idx = tid; idx < MOLS * 32 * dhid; idx += 64 * NWV) {
      int m = idx / (32 * dhid);
      int rem = idx - m * 32 * dhid;
      int row = rem / dhid, col = rem - row * dhid;
      if (blockIdx.x * MOLS + m < B)
        a.state_out[((int64_t)mb[m] * 32 + row) * dhid + col] = Xs[cur][m][row][col];
    }
  }

  // ---- head (model/lanczos_net.py:185-194): one 32-column tile = [W_o ; w_a ; 0];
  //      wave m handles molecule m
  if (wave < MOLS && blockIdx.x * MOLS + wave < B) {
    const int m = wave;
    const int P = a.dout;
    f32x16 acc = lnz::splat16(a.bias_head[j]);
    const float4* wh = reinterpret_cast<const float4*>(a.Wp_head) + lane;
    const float* xr = &Xs[cur][m][j][4 * hh];
    const int Q = dhid >> 3;
#pragma unroll 2
    for (int q = 0; q < Q; ++q) {
      float4 av = *reinterpret_cast<const float4*>(xr + 8 * q);
      float4 bv = wh[q * 64];
      acc = lnz::mfma32(av.x, bv.x, acc);
      acc = lnz::mfma32(av.y, bv.y, acc);
      acc = lnz::mfma32(av.z, bv.z, acc);
      acc = lnz::mfma32(av.w, bv.w, acc);
    }
    // acc[r] of lane (j,hh) = Y[cd_row(r,hh)][j]; the gate logit is column P of the same row
    float sum = 0.0f, cnt = 0.0f;
    const int src = 32 * hh + P;
    const int64_t mol = mb[m];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float logit = __shfl(acc[r], src, 64);
      float gate = 1.0f / (1.0f + __expf(-logit));
      int row = lnz::cd_row(r, hh);
      bool msk = row < N && a.mask[mol * N + row] != 0;
      sum += msk ? gate * acc[r] : 0.0f;
      cnt += msk ? 1.0f : 0.0f;
    }
    sum += __shfl_xor(sum, 32, 64);
    cnt += __shfl_xor(cnt, 32, 64);
    if (hh == 0 && j < P) a.score[mol * P + j] = sum / cnt;
  }
}

}  // namespace

namespace lnz {
int launch_forward_f16x3(const lnz_forward_args& a, hipStream_t s);  // conv_forward_f16.hip
}

extern "C" int64_t lnz_forward_args_size(void) { return (int64_t)sizeof(lnz_forward_args); }

extern "C" int lnz_lanczosnet_forward(const lnz_forward_args* args, lnz_stream_t stream) {
  LNZ_REQUIRE(args, LNZ_EINVAL, "lnz_lanczosnet_forward: null args");
  const lnz_forward_args& a = *args;
  LNZ_REQUIRE(a.B > 0 && a.N > 0 && a.K > 0 && a.num_layer > 0, LNZ_EINVAL,
              "lnz_lanczosnet_forward: bad sizes (B=%d N=%d K=%d L=%d)", a.B, a.N, a.K,
              a.num_layer);
  LNZ_REQUIRE(a.N <= LNZ_TILE, LNZ_ENOTSUP,
              "lnz_lanczosnet_forward: N=%d > %d-node tile (multi-tile molecules not built yet)",
              a.N, LNZ_TILE);
  LNZ_REQUIRE(a.K <= 2 * KHMAX, LNZ_ENOTSUP, "lnz_lanczosnet_forward: K=%d > %d", a.K, 2 * KHMAX);
  LNZ_REQUIRE(a.num_layer <= 16, LNZ_ENOTSUP, "lnz_lanczosnet_forward: num_layer=%d > 16",
              a.num_layer);
  LNZ_REQUIRE(a.dhid == 64 || a.dhid == 128, LNZ_ENOTSUP,
              "lnz_lanczosnet_forward: hidden width %d not in {64,128}", a.dhid);
  LNZ_REQUIRE(a.din0 > 0 && a.din0 % 32 == 0 && a.din0 <= 128, LNZ_ENOTSUP,
              "lnz_lanczosnet_forward: input width %d must be a multiple of 32, <= 128 "
              "(zero-pad features and weight columns on the host)", a.din0);
  LNZ_REQUIRE(a.dout >= 1 && a.dout <= 31, LNZ_ENOTSUP,
              "lnz_lanczosnet_forward: output width %d not in 1..31", a.dout);
  LNZ_REQUIRE(a.n_short >= 0 && a.n_short <= 8 && a.n_long >= 0 && a.n_edge >= 1 &&
                  a.n_short + a.n_long + a.n_edge <= LNZ_MAX_CHANNELS,
              LNZ_EINVAL, "lnz_lanczosnet_forward: bad channel counts");
  LNZ_REQUIRE((a.node_feat && a.embedding && a.num_atom > 0) || a.node_feat_f, LNZ_EINVAL,
              "lnz_lanczosnet_forward: need node_feat+embedding or node_feat_f");
  LNZ_REQUIRE(a.mask && (a.Lp || (a.gemm_mode == 1 && a.Lp16)) && a.V && a.Wp && a.bias &&
                  a.Wp_head && a.bias_head && a.score,
              LNZ_EINVAL, "lnz_lanczosnet_forward: null tensor pointer");
  LNZ_REQUIRE(a.n_long == 0 || a.G, LNZ_EINVAL, "lnz_lanczosnet_forward: G missing");
  hipStream_t s = (hipStream_t)stream;
  LNZ_REQUIRE(a.gemm_mode == 0 || a.gemm_mode == 1, LNZ_EINVAL,
              "lnz_lanczosnet_forward: gemm_mode %d", a.gemm_mode);
  if (a.gemm_mode == 1) return lnz::launch_forward_f16x3(a, s);
  const int grid = (a.B + MOLS - 1) / MOLS;
  LNZ_REQUIRE(a.filter_kind == 0 || a.filter_kind == 1, LNZ_EINVAL,
              "lnz_lanczosnet_forward: filter_kind %d", a.filter_kind);
#define LNZ_LAUNCH(NWV_, KHT_, FK_)                                                             \
  hipLaunchKernelGGL((lanczosnet_forward_kernel<NWV_, KHT_, FK_>), dim3(grid), dim3(64 * NWV_), \
                     0, s, a)
  if (a.filter_kind == 0) {
    const bool k20 = a.K <= 20;  // QM8 config: 10 eigen slots per lane half
    if (a.dhid == 128 && k20) LNZ_LAUNCH(4, 10, 0);
    else if (a.dhid == 128) LNZ_LAUNCH(4, KHMAX, 0);
    else if (k20) LNZ_LAUNCH(2, 10, 0);
    else LNZ_LAUNCH(2, KHMAX, 0);
  } else {
    const bool k24 = a.K <= 24;  // cd_row order: 12 steps cover k < 24
    if (a.dhid == 128 && k24) LNZ_LAUNCH(4, 12, 1);
    else if (a.dhid == 128) LNZ_LAUNCH(4, KHMAX, 1);
    else if (k24) LNZ_LAUNCH(2, 12, 1);
    else LNZ_LAUNCH(2, KHMAX, 1);
  }
#undef LNZ_LAUNCH
  return lnz::check_launch("lnz_lanczosnet_forward");
}
