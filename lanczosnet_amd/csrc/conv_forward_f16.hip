// Opt-in split-precision variant of the fused LanczosNet forward (lnz_forward_args.gemm_mode = 1).
//
// GEMM1  Z_c = X W_c^T  is 77 % of the fp32 kernel's matrix work.  Here both operands are split
// into fp16 pieces x = x_hi + x_lo (x_hi = half(x), x_lo = half(x - x_hi): 22 mantissa bits) and
//   X W^T  ~=  X_hi W_hi^T + X_hi W_lo^T + X_lo W_hi^T          (fp32 accumulate)
// runs on v_mfma_f32_32x32x16_f16 (32 cycles per 32x32x16) instead of v_mfma_f32_32x32x2_f32
// (64 cycles per 32x32x2): 3/16 of the issue cycles.  The dropped x_lo w_lo term is 2^-22 relative;
// end-to-end deviation from fp64 is 6e-7 (exact-fp32 path: 2e-7; parity bar 1e-5).
// GEMM2 (M_c Z_c) of the edge / short channels uses the same split (M_c pieces pre-split by
// lnz_pack_laplacian_f16x2, Z_c split from the GEMM1 accumulators).  The long-scale spectral
// channels run in eigen space like the fp32 kernel (conv_forward.hip): Y = V^T X once per layer
// (exact fp32 MFMA, Y stored as fp16 hi/lo in the other X buffer), GEMM1 of every long channel on
// Y, T += diag(g_s) (Y W_s^T) on the VALU in fp32, one lift out += V T (split) — no L_s build, no
// per-channel GEMM2.  The head accumulation and everything else stay exact fp32.
//
// With GEMM1 this cheap the per-CU vector-memory path becomes the limiter unless every packed
// weight fragment is reused more: ONE workgroup per CU-sized group of FOUR molecules, four
// wavefronts (one per SIMD, up to 512 registers each), wave w owns output-feature tile w for all
// four molecules, so one 2 KiB weight fragment (hi + lo) feeds 4 x 3 MFMAs.  X lives in LDS as
// fp16 hi/lo row-major tiles (pitch 136 halves: conflict-free ds_read_b128 A fragments), written
// by the epilogue of the previous layer.  All layers use an input width of 128 (layer-0 features
// and weight columns are zero padded), so a channel is always 8 k-blocks and ring slots are static.
#include "common.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int M4 = 4;          // molecules per workgroup
constexpr int P16 = 136;       // LDS row pitch in halves
constexpr int KB = 8;          // k-blocks (of 16) per channel: input width 128
constexpr int RING_D = 6;      // weight prefetch distance in k-blocks
constexpr int KHT = 10;        // eigen slots per lane half (K <= 20)
constexpr int GK = 24;         // gains per (molecule, channel) kept in LDS (slots k < 24, K <= 20 live)
constexpr int VP = 33;         // Ritz-vector LDS row pitch (floats): conflict-free column reads

union H8 {
  uint4 u;
  f16x8 h;
};

__device__ inline f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ inline void split_store(_Float16* xh, _Float16* xl, float x) {
  _Float16 h = (_Float16)x;
  *xh = h;
  *xl = (_Float16)(x - (float)h);
}

// registers [8 blk, 8 blk + 8) of a C/D tile -> fp16 hi / lo operand vectors
__device__ inline void split8(const f32x16& v, int blk, f16x8& h, f16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x = v[8 * blk + e];
    _Float16 xh = (_Float16)x;
    h[e] = xh;
    l[e] = (_Float16)(x - (float)xh);
  }
}

__global__ __launch_bounds__(256) void lanczosnet_forward_f16x3_kernel(const lnz_forward_args a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  // [buf][piece][mol][32][P16]
  auto Xp = [&](int buf, int piece, int m) { return smem + (((buf * 2 + piece) * M4 + m) * 32) * P16; };
  // Ritz vectors of the four molecules, fp32 [mol][node][slot] (pitch VP), behind the X tiles:
  // column reads feed the projection, row reads the lift
  float* Vs = reinterpret_cast<float*>(smem + 2 * 2 * M4 * 32 * P16);
  float* Gl = Vs + M4 * 32 * VP;  // this layer's gains [mol][s][slot k < GK]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 31, hh = lane >> 5;
  const int N = a.N, K = a.K, B = a.B;
  const int C = a.n_short + a.n_long + a.n_edge;
  // molecule ids of this workgroup's four slots (a.plan: balanced deal of lnz_plan_tiles, single
  // tiles only).  An unused slot repeats slot 0's molecule; its results are dropped.
  if (a.plan && (int)blockIdx.x >= *a.n_wg) return;
  int mb[M4];
  bool used[M4];
#pragma unroll
  for (int m = 0; m < M4; ++m) {
    const int x = blockIdx.x * M4 + m;
    int id = a.plan ? a.plan[3 * x] : (x < B ? x : -1);
    used[m] = id >= 0;
    mb[m] = id;
  }
#pragma unroll
  for (int m = 1; m < M4; ++m) mb[m] = used[m] ? mb[m] : mb[0];

  // ---- embedding gather / float features, zero padded to 128 columns, split into hi/lo ----------
  for (int idx = tid; idx < M4 * 32 * 128; idx += 256) {
    int m = idx >> 12, row = (idx >> 7) & 31, col = idx & 127;
    float v = 0.0f;
    if (row < N && col < a.din0) {
      if (a.node_feat) {
        int64_t id = a.node_feat[(int64_t)mb[m] * N + row];
        id = id < 0 ? 0 : (id >= a.num_atom ? a.num_atom - 1 : id);
        v = a.embedding[id * a.din0 + col];
      } else {
        v = a.node_feat_f[((int64_t)mb[m] * N + row) * a.din0 + col];
      }
    }
    split_store(Xp(0, 0, m) + row * P16 + col, Xp(0, 1, m) + row * P16 + col, v);
  }

  int g2steps[M4];
#pragma unroll
  for (int m = 0; m < M4; ++m) {
    int last = 0;
    for (int i = lane; i < N; i += 64) last = a.mask[(int64_t)mb[m] * N + i] ? i + 1 : last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
    g2steps[m] = __builtin_amdgcn_readfirstlane(((last + 7) >> 3) * 4);
  }
  int nblkm[M4];  // k-blocks of GEMM2 a molecule needs: block 1 covers node rows 16..31
#pragma unroll
  for (int m = 0; m < M4; ++m) nblkm[m] = g2steps[m] > 8 ? 2 : 1;

  for (int idx = tid; idx < M4 * 32 * 32; idx += 256) {
    const int m = idx >> 10, node = (idx >> 5) & 31, slot = idx & 31;
    const int mol = m == 0 ? mb[0] : m == 1 ? mb[1] : m == 2 ? mb[2] : mb[3];
    Vs[(m * 32 + node) * VP + slot] =
        (node < N && slot < K) ? a.V[((int64_t)mol * N + node) * K + slot] : 0.0f;
  }
  __syncthreads();

#ifdef LNZ_PROFILE_PHASES
  long long t_g1 = 0, t_g2 = 0, t_ep = 0, t_pr = 0, t_all = clock64();
#define LNZ_T0 long long _t0 = clock64();
#define LNZ_ACC(x) { long long _t1 = clock64(); x += _t1 - _t0; _t0 = _t1; }
#else
#define LNZ_T0
#define LNZ_ACC(x)
#endif
  int cur = 0;
  for (int l = 0; l < a.num_layer; ++l) {
    const float* __restrict__ bl = a.bias + a.b_off[l];
    f32x16 out[M4];
    {
      const float bv = bl[32 * wave + j];
#pragma unroll
      for (int m = 0; m < M4; ++m) out[m] = lnz::splat16(bv);
    }
    // weight stream of this wave's feature tile: [g = c*8 + kb][piece][lane] uint4, contiguous
    const uint4* __restrict__ wp =
        reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.Wp16) + a.w16_off[l]) +
        (int64_t)wave * (C * KB) * 128 + lane;
    uint4 ringh[KB], ringl[KB];
#pragma unroll
    for (int sl = 0; sl < RING_D; ++sl) {
      ringh[sl] = wp[sl * 128];
      ringl[sl] = wp[sl * 128 + 64];
    }
    const int nxt = cur ^ 1;
    const _Float16* xa[M4][2];
    const _Float16* ya[M4][2];
#pragma unroll
    for (int m = 0; m < M4; ++m) {
      xa[m][0] = Xp(cur, 0, m) + j * P16 + 8 * hh;
      xa[m][1] = Xp(cur, 1, m) + j * P16 + 8 * hh;
      ya[m][0] = Xp(nxt, 0, m) + j * P16 + 8 * hh;
      ya[m][1] = Xp(nxt, 1, m) + j * P16 + 8 * hh;
    }

    // ---------------- eigen-space projection Y_m = V_m^T X_m (exact fp32 MFMA) ----------------
    //   A: V^T fragments (lane = slot row j, step r = node cd_row(r,hh)), L2 resident;
    //   B: X rows rebuilt from their fp16 pieces; Y (rows = eigen slots) goes, split, into the
    //   other X buffer — free until this layer's epilogue
    const float* vsl = Vs;
    // gains of conv layer `ll` for the four molecules -> LDS
    auto stage_gains = [&](int ll) {
      for (int idx = tid; idx < M4 * a.n_long * GK; idx += 256) {
        const int m = idx / (a.n_long * GK);
        const int rem = idx - m * a.n_long * GK;
        const int sc = rem / GK, k = rem - sc * GK;
        const int mol = m == 0 ? mb[0] : m == 1 ? mb[1] : m == 2 ? mb[2] : mb[3];
        Gl[idx] = k < K ? a.G[(((int64_t)ll * B + mol) * a.n_long + sc) * K + k] : 0.0f;
      }
    };
    // Layer 0: X_0 sits in LDS only (embedding gather) -> project from its fp16 pieces.  Later
    // layers are projected by the previous layer's epilogue from the C/D registers.
    if (a.n_long > 0 && l == 0) {
      LNZ_T0
#pragma unroll
      for (int m = 0; m < M4; ++m) {
        f32x16 Y = lnz::splat16(0.0f);
        const float* vp = vsl + m * 32 * VP + j;
        const _Float16* x0 = Xp(cur, 0, m) + 32 * wave + j;
        const _Float16* x1 = Xp(cur, 1, m) + 32 * wave + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (4 * g < g2steps[m]) {  // node rows beyond the molecule are zero
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) {
              const int node = lnz::cd_row(r, hh);
              const float vt = vp[node * VP];
              const float xb = (float)x0[node * P16] + (float)x1[node * P16];
              Y = lnz::mfma32(vt, xb, Y);
            }
          }
        }
        _Float16* yh = Xp(nxt, 0, m) + 32 * wave + j;
        _Float16* yl = Xp(nxt, 1, m) + 32 * wave + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = lnz::cd_row(r, hh);
          split_store(yh + row * P16, yl + row * P16, Y[r]);
        }
        // one molecule at a time: the scheduler otherwise hoists all four molecules' 192 LDS
        // reads to the top and spills their addresses
        __builtin_amdgcn_sched_barrier(0);
      }
      stage_gains(0);
      __syncthreads();
      LNZ_ACC(t_pr)
    }
    f32x16 Tsum[M4];
#pragma unroll
    for (int m = 0; m < M4; ++m) Tsum[m] = lnz::splat16(0.0f);

    // GEMM2 operands of the CURRENT channel, fetched at its start (they have the whole GEMM1 to
    // land): 16 floats per molecule as four dwordx4 — the Laplacian fragments of an edge/short
    // channel, or (long channel) the gains of the 16 slot rows this lane's C/D registers hold.
    float mop[M4][16];
    auto fetch_m_operands = [&](int c, int m) {
      const bool lng = (c >= a.n_short) && (c < a.n_short + a.n_long);
      const float* src;
      int stride4;
      if (lng) {
        return;  // gains come from LDS after GEMM1
      } else {
        // [blk][piece][lane] uint4: g = 2 blk + piece
        const int e = c < a.n_short ? 0 : c - a.n_short - a.n_long;
        src = reinterpret_cast<const float*>(a.Lp16) +
              (((int64_t)mb[m] * a.n_edge + e) * 256 + lane) * 4;
        stride4 = 64;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = *reinterpret_cast<const float4*>(src + (int64_t)g * stride4 * 4);
        mop[m][4 * g + 0] = v.x;
        mop[m][4 * g + 1] = v.y;
        mop[m][4 * g + 2] = v.z;
        mop[m][4 * g + 3] = v.w;
      }
    };

    for (int c = 0; c < C; ++c) {
      const bool is_long = (c >= a.n_short) && (c < a.n_short + a.n_long);
#pragma unroll
      for (int m = 0; m < M4; ++m) fetch_m_operands(c, m);

      LNZ_T0
      // ---------------- GEMM1 (split fp16): Z_m = X_m W_c^T ----------------
      f32x16 Z[M4];
#pragma unroll
      for (int m = 0; m < M4; ++m) Z[m] = lnz::splat16(0.0f);
      H8 ah[M4], al[M4];
#pragma unroll
      for (int m = 0; m < M4; ++m) {
        ah[m].u = *reinterpret_cast<const uint4*>(is_long ? ya[m][0] : xa[m][0]);
        al[m].u = *reinterpret_cast<const uint4*>(is_long ? ya[m][1] : xa[m][1]);
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        // prefetch weights RING_D k-blocks ahead (over-reads past the layer into the slack)
        ringh[(kb + RING_D) % KB] = wp[(kb + RING_D) * 128];
        ringl[(kb + RING_D) % KB] = wp[(kb + RING_D) * 128 + 64];
        H8 nh[M4], nl[M4];
        if (kb + 1 < KB) {
#pragma unroll
          for (int m = 0; m < M4; ++m) {
            nh[m].u = *reinterpret_cast<const uint4*>((is_long ? ya[m][0] : xa[m][0]) + 16 * (kb + 1));
            nl[m].u = *reinterpret_cast<const uint4*>((is_long ? ya[m][1] : xa[m][1]) + 16 * (kb + 1));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        H8 wh, wl;
        wh.u = ringh[kb];
        wl.u = ringl[kb];
#pragma unroll
        for (int m = 0; m < M4; ++m) Z[m] = mfma16(ah[m].h, wh.h, Z[m]);
#pragma unroll
        for (int m = 0; m < M4; ++m) Z[m] = mfma16(ah[m].h, wl.h, Z[m]);
#pragma unroll
        for (int m = 0; m < M4; ++m) Z[m] = mfma16(al[m].h, wh.h, Z[m]);
        if (kb + 1 < KB) {
#pragma unroll
          for (int m = 0; m < M4; ++m) {
            ah[m] = nh[m];
            al[m] = nl[m];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      wp += KB * 128;

      LNZ_ACC(t_g1)
      if (is_long) {
        // ---------------- eigen space: T_m += diag(g_c) Z_m; after the last long channel the
        //                  lift out_m += V_m T_m (split fp16), then the Y buffer is released --------
#pragma unroll
        for (int m = 0; m < M4; ++m) {
          // gains of the slot rows of this lane's C/D registers: rows 8g + 4hh + u < GK
          const float* gp = Gl + (m * a.n_long + (c - a.n_short)) * GK + 4 * hh;
#pragma unroll
          for (int g = 0; g < GK / 8; ++g) {
            const float4 gv = *reinterpret_cast<const float4*>(gp + 8 * g);
            Tsum[m][4 * g + 0] = fmaf(gv.x, Z[m][4 * g + 0], Tsum[m][4 * g + 0]);
            Tsum[m][4 * g + 1] = fmaf(gv.y, Z[m][4 * g + 1], Tsum[m][4 * g + 1]);
            Tsum[m][4 * g + 2] = fmaf(gv.z, Z[m][4 * g + 2], Tsum[m][4 * g + 2]);
            Tsum[m][4 * g + 3] = fmaf(gv.w, Z[m][4 * g + 3], Tsum[m][4 * g + 3]);
          }
        }
        if (c + 1 == a.n_short + a.n_long) {
#pragma unroll
          for (int m = 0; m < M4; ++m) {
            const float* vrow = vsl + (m * 32 + j) * VP;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
              if (16 * blk < K) {  // k-block 1 holds slots 16..31
                H8 vh, vl, th, tl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const int slot = lnz::cd_row(8 * blk + e, hh);
                  const float x = vrow[slot];
                  const _Float16 xh = (_Float16)x;
                  vh.h[e] = xh;
                  vl.h[e] = (_Float16)(x - (float)xh);
                }
                split8(Tsum[m], blk, th.h, tl.h);
                out[m] = mfma16(vh.h, th.h, out[m]);
                out[m] = mfma16(vh.h, tl.h, out[m]);
                out[m] = mfma16(vl.h, th.h, out[m]);
              }
            }
          }
        }
        LNZ_ACC(t_g2)
        continue;
      }
      // ---------------- M_c operand pieces (A of GEMM2), per molecule and k-block --------------
      // edge/short: pre-split fragments from lnz_pack_laplacian_f16x2.
      H8 mh[M4][2], ml[M4][2];
      {
#pragma unroll
        for (int m = 0; m < M4; ++m) {
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            mh[m][blk].u = make_uint4(__float_as_uint(mop[m][8 * blk + 0]), __float_as_uint(mop[m][8 * blk + 1]),
                                      __float_as_uint(mop[m][8 * blk + 2]), __float_as_uint(mop[m][8 * blk + 3]));
            ml[m][blk].u = make_uint4(__float_as_uint(mop[m][8 * blk + 4]), __float_as_uint(mop[m][8 * blk + 5]),
                                      __float_as_uint(mop[m][8 * blk + 6]), __float_as_uint(mop[m][8 * blk + 7]));
          }
        }
      }
      // ---------------- GEMM2 (split fp16): out_m += M_c,m Z_m ---------------------------------
      // k-block 0 covers node rows 0..15, block 1 rows 16..31 (skipped per molecule when it has
      // <= 16 nodes).  Short-diffusion channels first apply M = L_0 (p-1) more times to Z.
      if (c < a.n_short) {  // Z <- L_0^(p-1) Z
        const int p = a.short_dist[c];
        for (int rep = 1; rep < p; ++rep) {
          f32x16 T[M4];
#pragma unroll
          for (int m = 0; m < M4; ++m) T[m] = lnz::splat16(0.0f);
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int m = 0; m < M4; ++m) {
              if (blk < nblkm[m]) {
                H8 zh, zl;
                split8(Z[m], blk, zh.h, zl.h);
                T[m] = mfma16(mh[m][blk].h, zh.h, T[m]);
                T[m] = mfma16(mh[m][blk].h, zl.h, T[m]);
                T[m] = mfma16(ml[m][blk].h, zh.h, T[m]);
              }
            }
          }
#pragma unroll
          for (int m = 0; m < M4; ++m) Z[m] = T[m];
        }
      }
      {  // k-block 0 (node rows 0..15): all molecules, four chains interleaved
        H8 zh[M4], zl[M4];
#pragma unroll
        for (int m = 0; m < M4; ++m) split8(Z[m], 0, zh[m].h, zl[m].h);
#pragma unroll
        for (int m = 0; m < M4; ++m) out[m] = mfma16(mh[m][0].h, zh[m].h, out[m]);
#pragma unroll
        for (int m = 0; m < M4; ++m) out[m] = mfma16(mh[m][0].h, zl[m].h, out[m]);
#pragma unroll
        for (int m = 0; m < M4; ++m) out[m] = mfma16(ml[m][0].h, zh[m].h, out[m]);
      }
#pragma unroll
      for (int m = 0; m < M4; ++m) {  // k-block 1 (rows 16..31): only molecules with > 16 nodes
        if (nblkm[m] > 1) {
          H8 zh, zl;
          split8(Z[m], 1, zh.h, zl.h);
          out[m] = mfma16(mh[m][1].h, zh.h, out[m]);
          out[m] = mfma16(mh[m][1].h, zl.h, out[m]);
          out[m] = mfma16(ml[m][1].h, zh.h, out[m]);
        }
      }
      LNZ_ACC(t_g2)
    }

    // ---------------- epilogue ----------------------------------------------------------------
    //   barrier: every wave is done with X (cur) and Y (nxt).  Then X' = relu(out) -> nxt (split)
    //   and, from the same C/D registers (exact fp32, they are the B operand as they stand), the
    //   next layer's projection Y' = V^T X' -> cur (split); its gains -> LDS; barrier.
    LNZ_T0
    __syncthreads();
    const bool more = l + 1 < a.num_layer && a.n_long > 0;
#pragma unroll
    for (int m = 0; m < M4; ++m) {
      _Float16* xh = Xp(nxt, 0, m) + 32 * wave + j;
      _Float16* xl = Xp(nxt, 1, m) + 32 * wave + j;
      f32x16 Y = lnz::splat16(0.0f);
      const float* vp = vsl + m * 32 * VP + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[m][r] = fmaxf(out[m][r], 0.0f);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (more && 4 * g < g2steps[m]) {  // node rows beyond the molecule have zero Ritz rows
#pragma unroll
          for (int r = 4 * g; r < 4 * g + 4; ++r)
            Y = lnz::mfma32(vp[lnz::cd_row(r, hh) * VP], out[m][r], Y);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = lnz::cd_row(r, hh);
        split_store(xh + row * P16, xl + row * P16, out[m][r]);
      }
      if (more) {
        _Float16* yh = Xp(cur, 0, m) + 32 * wave + j;
        _Float16* yl = Xp(cur, 1, m) + 32 * wave + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = lnz::cd_row(r, hh);
          split_store(yh + row * P16, yl + row * P16, Y[r]);
        }
      }
    }
    if (more) stage_gains(l + 1);
    __syncthreads();
    cur = nxt;
    LNZ_ACC(t_ep)
  }
#ifdef LNZ_PROFILE_PHASES
  if (a.state_out && lane == 0 && blockIdx.x < 8) {
    float* d = a.state_out + ((int64_t)B * 32 * 128) + (blockIdx.x * 8 + wave) * 8;  // tools/phase_probe.py layout
    d[0] = (float)t_g1; d[1] = (float)t_g2; d[2] = (float)t_ep; d[3] = (float)(clock64() - t_all);
    d[4] = (float)t_pr;
  }
#endif

  if (a.state_out) {
    for (int idx = tid; idx < M4 * 32 * 128; idx += 256) {
      int m = idx >> 12, row = (idx >> 7) & 31, col = idx & 127;
      if (m == 0 ? used[0] : m == 1 ? used[1] : m == 2 ? used[2] : used[3])
        a.state_out[((int64_t)mb[m] * 32 + row) * 128 + col] =
            (float)Xp(cur, 0, m)[row * P16 + col] + (float)Xp(cur, 1, m)[row * P16 + col];
    }
  }

  // ---- head: wave m handles molecule m (split fp16 GEMM against the packed [32,128] head) ------
  if (wave == 0 ? used[0] : wave == 1 ? used[1] : wave == 2 ? used[2] : used[3]) {
    const int m = wave;
    const int P = a.dout;
    f32x16 acc = lnz::splat16(a.bias_head[j]);
    const uint4* wh = reinterpret_cast<const uint4*>(a.Wp16_head) + lane;
    const _Float16* x0 = Xp(cur, 0, m) + j * P16 + 8 * hh;
    const _Float16* x1 = Xp(cur, 1, m) + j * P16 + 8 * hh;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      H8 xh, xl, bh, blo;
      xh.u = *reinterpret_cast<const uint4*>(x0 + 16 * kb);
      xl.u = *reinterpret_cast<const uint4*>(x1 + 16 * kb);
      bh.u = wh[kb * 128];
      blo.u = wh[kb * 128 + 64];
      acc = mfma16(xh.h, bh.h, acc);
      acc = mfma16(xh.h, blo.h, acc);
      acc = mfma16(xl.h, bh.h, acc);
    }
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

constexpr size_t kSmemBytes = (size_t)2 * 2 * M4 * 32 * P16 * sizeof(_Float16)  // X tiles 139,264 B
                              + (size_t)M4 * 32 * VP * sizeof(float)             // + Ritz vectors 16,896 B
                              + (size_t)M4 * 16 * GK * sizeof(float);            // + gains of one layer (<= 16 channels) 6,144 B

}  // namespace

namespace lnz {
int launch_forward_f16x3(const lnz_forward_args& a, hipStream_t s) {
  LNZ_REQUIRE(a.dhid == 128 && a.filter_kind == 0 && a.K <= 2 * KHT && a.din0 <= 128, LNZ_ENOTSUP,
              "lnz_lanczosnet_forward(gemm_mode=1): needs dhid=128, filter_kind=0, K<=20, din0<=128");
  LNZ_REQUIRE(a.Wp16 && a.Wp16_head && a.Lp16, LNZ_EINVAL,
              "lnz_lanczosnet_forward(gemm_mode=1): Wp16 / Wp16_head / Lp16 missing");
  {
    // the attribute is PER DEVICE (one process may drive several: nn.DataParallel,
    // runner/qm8_runner.py:62) — set it on the current device at every launch, no cached flag
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lanczosnet_forward_f16x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmemBytes);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS %zu): %s", kSmemBytes, hipGetErrorString(e));
      return LNZ_ELAUNCH;
    }
  }
  const int grid = a.plan ? a.plan_wg_cap : (a.B + M4 - 1) / M4;
  hipLaunchKernelGGL(lanczosnet_forward_f16x3_kernel, dim3(grid), dim3(256), kSmemBytes, s, a);
  return check_launch("lnz_lanczosnet_forward(f16x3)");
}
}  // namespace lnz
