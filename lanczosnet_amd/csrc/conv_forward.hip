// R7 (second half) + R9 + R10: the fused LanczosNet forward.
//
// One 128-thread workgroup (2 wavefronts) per molecule runs the WHOLE network on chip:
// embedding -> num_layer x [ X' = relu( sum_c M_c X W_c^T + b ) ] -> gated head -> masked mean.
//
// Per layer and message channel c two chained matrix-core GEMMs (v_mfma_f32_32x32x2_f32, exact
// fp32 fma chains), evaluated as  M_c (X W_c^T)  instead of the reference's  (M_c X) W_c^T :
//
//   GEMM1  Z_c [32 nodes x dhid] = X [32 x din] * W_c^T
//          A = X from LDS (row-major, pitch 132 floats, one ds_read_b128 = 4 k-steps),
//          B = W_c pre-packed in fragment order (one global_load_dwordx4 = 4 k-steps, L2 hits),
//          wave w owns output-feature tiles [w*OTW, (w+1)*OTW).
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

constexpr int NW = 2;        // wavefronts per molecule
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

template <int OTW>
__global__ __launch_bounds__(64 * NW) void lanczosnet_forward_kernel(const lnz_forward_args a) {
  __shared__ __attribute__((aligned(16))) float Xs[2][32][PITCH];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 31, hh = lane >> 5;
  const int N = a.N, K = a.K, B = a.B;
  const int dhid = a.dhid;
  const int C = a.n_short + a.n_long + a.n_edge;
  const int OT = OTW * NW;  // dhid / 32

  // ---- embedding gather (model/lanczos_net.py:154) / float features (lanczos_net_general.py:156)
  {
    const int d4 = a.din0 >> 2;
    for (int idx = tid; idx < 32 * d4; idx += 64 * NW) {
      int row = idx / d4, c4 = idx - row * d4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N) {
        if (a.node_feat) {
          int64_t id = a.node_feat[(int64_t)b * N + row];
          id = id < 0 ? 0 : (id >= a.num_atom ? a.num_atom - 1 : id);
          v = reinterpret_cast<const float4*>(a.embedding + id * a.din0)[c4];
        } else {
          v = reinterpret_cast<const float4*>(a.node_feat_f + ((int64_t)b * N + row) * a.din0)[c4];
        }
      }
      *reinterpret_cast<float4*>(&Xs[0][row][4 * c4]) = v;
    }
  }

  // ---- Ritz-vector fragments: vreg[t] = V[b][j][KH*hh + t]
  const int KH = (K + 1) >> 1;
  float vreg[KHMAX];
#pragma unroll
  for (int t = 0; t < KHMAX; ++t) {
    int k = KH * hh + t;
    vreg[t] = (t < KH && k < K && j < N) ? a.V[((int64_t)b * N + j) * K + k] : 0.0f;
  }
  __syncthreads();

  int cur = 0;
  for (int l = 0; l < a.num_layer; ++l) {
    const int din = l == 0 ? a.din0 : dhid;
    const int Q = din >> 3;
    const float4* __restrict__ Wl = reinterpret_cast<const float4*>(a.Wp + a.w_off[l]);
    const float* __restrict__ bl = a.bias + a.b_off[l];

    f32x16 out[OTW];
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) out[ot] = lnz::splat16(bl[32 * (wave * OTW + ot) + j]);

    for (int c = 0; c < C; ++c) {
      // ---------------- A operand of GEMM2: M_c fragments ----------------
      f32x16 Mf;
      const bool is_long = (c >= a.n_short) && (c < a.n_short + a.n_long);
      if (is_long) {
        const int s = c - a.n_short;
        const float* gp = a.G + (((int64_t)l * B + b) * a.n_long + s) * K;
        f32x16 acc = lnz::splat16(0.0f);
#pragma unroll
        for (int t = 0; t < KHMAX; ++t) {
          if (t < KH) {
            int k = KH * hh + t;
            float g = k < K ? gp[k] : 0.0f;
            acc = lnz::mfma32(vreg[t] * g, vreg[t], acc);
          }
        }
        Mf = acc;  // L_s[cd_row(r,hh)][j] == L_s[j][cd_row(r,hh)]
      } else {
        const int e = c < a.n_short ? 0 : c - a.n_short - a.n_long;
        const float4* lp = reinterpret_cast<const float4*>(a.Lp) + ((int64_t)b * a.n_edge + e) * 256;
        float4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = lp[g * 64 + lane];
        Mf = frag_from4(v);
      }

      // ---------------- GEMM1: Z = X W_c^T ----------------
      f32x16 Z[OTW];
#pragma unroll
      for (int ot = 0; ot < OTW; ++ot) Z[ot] = lnz::splat16(0.0f);
      const float4* __restrict__ wb[OTW];
#pragma unroll
      for (int ot = 0; ot < OTW; ++ot)
        wb[ot] = Wl + ((int64_t)(wave * OTW + ot) * (C * Q) + (int64_t)c * Q) * 64 + lane;
      const float* xrow = &Xs[cur][j][4 * hh];
#pragma unroll 2
      for (int q = 0; q < Q; ++q) {
        float4 av = *reinterpret_cast<const float4*>(xrow + 8 * q);
        float4 bv[OTW];
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) bv[ot] = wb[ot][q * 64];
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) Z[ot] = lnz::mfma32(av.x, bv[ot].x, Z[ot]);
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) Z[ot] = lnz::mfma32(av.y, bv[ot].y, Z[ot]);
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) Z[ot] = lnz::mfma32(av.z, bv[ot].z, Z[ot]);
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) Z[ot] = lnz::mfma32(av.w, bv[ot].w, Z[ot]);
      }

      // ---------------- short diffusion: Z <- L_0^(p-1) Z ----------------
      if (c < a.n_short) {
        const int p = a.short_dist[c];
        for (int rep = 1; rep < p; ++rep) {
#pragma unroll
          for (int ot = 0; ot < OTW; ++ot) {
            f32x16 T = lnz::splat16(0.0f);
#pragma unroll
            for (int r = 0; r < 16; ++r) T = lnz::mfma32(Mf[r], Z[ot][r], T);
            Z[ot] = T;
          }
        }
      }

      // ---------------- GEMM2: out += M_c Z ----------------
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) out[ot] = lnz::mfma32(Mf[r], Z[ot][r], out[ot]);
      }
    }

    // ---------------- epilogue: ReLU, X' -> LDS (other buffer), one barrier per layer -------
    const int nxt = cur ^ 1;
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Xs[nxt][lnz::cd_row(r, hh)][32 * (wave * OTW + ot) + j] = fmaxf(out[ot][r], 0.0f);
    }
    __syncthreads();
    cur = nxt;
  }

  // ---- optional debug/test output of the final node state
  if (a.state_out) {
    float* so = a.state_out + (int64_t)b * 32 * dhid;
    for (int idx = tid; idx < 32 * dhid; idx += 64 * NW) {
      int row = idx / dhid, col = idx - row * dhid;
      so[idx] = Xs[cur][row][col];
    }
  }

  // ---- head (model/lanczos_net.py:185-194): one 32-column tile = [W_o ; w_a ; 0], wave 0 only
  if (wave == 0) {
    const int P = a.dout;
    f32x16 acc = lnz::splat16(a.bias_head[j]);
    const float4* wh = reinterpret_cast<const float4*>(a.Wp_head) + lane;
    const float* xrow = &Xs[cur][j][4 * hh];
    const int Q = dhid >> 3;
#pragma unroll 2
    for (int q = 0; q < Q; ++q) {
      float4 av = *reinterpret_cast<const float4*>(xrow + 8 * q);
      float4 bv = wh[q * 64];
      acc = lnz::mfma32(av.x, bv.x, acc);
      acc = lnz::mfma32(av.y, bv.y, acc);
      acc = lnz::mfma32(av.z, bv.z, acc);
      acc = lnz::mfma32(av.w, bv.w, acc);
    }
    // acc[r] of lane (j,hh) = Y[cd_row(r,hh)][j]; the gate logit is column P of the same row
    float sum = 0.0f, cnt = 0.0f;
    const int src = 32 * hh + P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float logit = __shfl(acc[r], src, 64);
      float gate = 1.0f / (1.0f + __expf(-logit));
      int row = lnz::cd_row(r, hh);
      bool m = row < N && a.mask[(int64_t)b * N + row] != 0;
      sum += m ? gate * acc[r] : 0.0f;
      cnt += m ? 1.0f : 0.0f;
    }
    sum += __shfl_xor(sum, 32, 64);
    cnt += __shfl_xor(cnt, 32, 64);
    if (hh == 0 && j < P) a.score[(int64_t)b * P + j] = sum / cnt;
  }
}

}  // namespace

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
  LNZ_REQUIRE(a.din0 > 0 && a.din0 % 8 == 0 && a.din0 <= 128, LNZ_ENOTSUP,
              "lnz_lanczosnet_forward: input width %d must be a multiple of 8, <= 128", a.din0);
  LNZ_REQUIRE(a.dout >= 1 && a.dout <= 31, LNZ_ENOTSUP,
              "lnz_lanczosnet_forward: output width %d not in 1..31", a.dout);
  LNZ_REQUIRE(a.n_short >= 0 && a.n_short <= 8 && a.n_long >= 0 && a.n_edge >= 1 &&
                  a.n_short + a.n_long + a.n_edge <= LNZ_MAX_CHANNELS,
              LNZ_EINVAL, "lnz_lanczosnet_forward: bad channel counts");
  LNZ_REQUIRE((a.node_feat && a.embedding && a.num_atom > 0) || a.node_feat_f, LNZ_EINVAL,
              "lnz_lanczosnet_forward: need node_feat+embedding or node_feat_f");
  LNZ_REQUIRE(a.mask && a.Lp && a.V && a.Wp && a.bias && a.Wp_head && a.bias_head && a.score,
              LNZ_EINVAL, "lnz_lanczosnet_forward: null tensor pointer");
  LNZ_REQUIRE(a.n_long == 0 || a.G, LNZ_EINVAL, "lnz_lanczosnet_forward: G missing");
  hipStream_t s = (hipStream_t)stream;
  if (a.dhid == 128) {
    hipLaunchKernelGGL(lanczosnet_forward_kernel<2>, dim3(a.B), dim3(64 * NW), 0, s, a);
  } else {
    hipLaunchKernelGGL(lanczosnet_forward_kernel<1>, dim3(a.B), dim3(64 * NW), 0, s, a);
  }
  return lnz::check_launch("lnz_lanczosnet_forward");
}
