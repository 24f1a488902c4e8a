// R7 (second half) + R9 + R10: the fused LanczosNet forward.
//
// One workgroup = two halves of dhid/32 wavefronts; a half runs the WHOLE network on chip for up
// to two 32-row node tiles (lnz_plan_tiles deals the batch's tiles over one workgroup per CU):
// embedding -> num_layer x [ X' = relu( sum_c M_c X W_c^T + b ) ] -> gated head -> masked mean.
// A tile holds one molecule, or two small ones (8|24 or 16|16 rows) with block-diagonal operators
// — the per-lane molecule id is all that differs.
//
// Node-space channels (edge types, short diffusion) are two chained matrix-core GEMMs
// (v_mfma_f32_32x32x2_f32, exact fp32 fma chains), evaluated as  M_c (X W_c^T)  instead of the
// reference's  (M_c X) W_c^T :
//
//   GEMM1  Z_c [32 nodes x dhid] = X [32 x din] * W_c^T
//          A = X from LDS (row-major, pitch 132 floats, one ds_read_b128 = 4 k-steps),
//          B = W_c pre-packed in fragment order (one global_load_dwordx4 = 4 k-steps, L2 hits)
//              through a register ring; wave w owns output-feature tile w for all its tiles, so
//              each weight fragment feeds MT x 4 MFMAs.
//   GEMM2  out += M_c [32 x 32] * Z_c
//          B = the C/D registers of GEMM1 *as they are*: register r of lane (j, hh) holds
//              Z_c[cd_row(r,hh)][j], which is exactly what k-step r needs when the contraction
//              index is visited in the order m(r,hh) = cd_row(r,hh);
//          A = M_c in the matching order from the packed Laplacian (4 coalesced dwordx4).
//   short-diffusion channels apply M = L_0 p times to Z_c in registers (same chaining).
//
// The long-scale spectral channels  sum_s V diag(g_s) V^T X W_s^T  run in EIGEN SPACE (see the
// comment above forward_half): Y = V^T X once per layer (from the previous epilogue's C/D
// registers; from LDS for the first layer), GEMM1 of every channel on Y
// with the slot rows scaled by the gains, one lift back through V — the filters L_s never exist.
//
// So `cat(msg)` (model/lanczos_net.py:180), the [B,N,N,S] filter stack (:123) and the strided
// `L[:,:,:,ii]` clones (:172-178) never exist; per layer the only LDS traffic is X / Y and there
// are two __syncthreads per layer (one without spectral channels).
// HBM bytes per molecule: Lp 28,672 + V 2,560 + G 4,480 + ids 256 + mask 32 in, 64 out; the
// 7.4 MB of packed weights are shared by all workgroups and stay L2 / Infinity-Cache resident.
#include "common.hpp"
#include "conv_tiles.hpp"
#include <type_traits>

namespace {

constexpr int MOLS = 2;      // node tiles per workgroup half; every wave of the half works on both
constexpr int PITCH = 132;   // LDS row pitch (floats): conflict-free ds_read_b128 A fragments
// LDS-qualified pointer: keeps the A-fragment reads ds_read_b128 (a generic pointer that the
// compiler cannot trace back to __shared__ becomes flat_load, which also ties up vmcnt)
typedef const __attribute__((address_space(3))) float* lds_cptr;
__device__ __forceinline__ float4 lds_f4(lds_cptr p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) f4v* lds_c4ptr;
  const f4v v = *(lds_c4ptr)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
constexpr int VPITCH = 36;   // Ritz-vector row pitch: 16 B aligned, conflict-free float4 rows
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

// One HALF of a workgroup: NWV wavefronts (wave w owns output-feature tile w) running the whole
// network for MT node tiles, so every packed-weight fragment a wave loads feeds MT x 4 MFMAs (the
// per-CU vector-memory path, not L2, limits a 1-tile tiling).  The two halves of a workgroup share
// nothing but the CU and the per-layer barrier.
// FK = filter kind: 0 = diagonal gains on Ritz vectors (LanczosNet), 1 = dense K x K filters on
// the Lanczos basis built in NODE space (AdaLanczosNet: M = Q DD Q^T,
// model/ada_lanczos_net.py:280-281; single tiles; kept for A/B runs), 2 = the same dense filters in
// EIGEN space: sum_s Q DD_s Q^T X W_s^T = Q [ sum_s DD_s (Y W_s^T) ], Y = Q^T X — the diagonal
// variant's structure with the 16 row-scaling FMAs of a channel replaced by one 32 x 32 MFMA chain
// (A = DD_s fragments from global memory, B = the channel's GEMM1 result as it stands in its C/D
// registers); pair tiles, identity-channel shortcut and the weight-ring schedule come with it.
template <int MT>
__device__ __forceinline__ TileDesc pick(const TileDesc (&td)[MT], int m) {
  static_assert(MT <= 2, "select written for two tiles");
  TileDesc t = td[0];
  if (MT > 1 && m) t = td[MT - 1];
  return t;
}
template <int MT>
__device__ __forceinline__ int pick(const int (&v)[MT], int m) {
  return (MT > 1 && m) ? v[MT - 1] : v[0];
}

// Long-scale spectral channels run in EIGEN SPACE (FK = 0):
//   sum_s V diag(g_s) V^T X W_s^T  =  V [ sum_s diag(g_s) (Y W_s^T) ],   Y = V^T X
// so a layer projects once (Y, 16 MFMAs per tile and wave), runs GEMM1 of every long channel on Y,
// scales the result rows (= eigen slots) by the channel's gains on the VALU, and lifts the sum back
// through V once (<= 16 MFMAs) — instead of building L_s and running GEMM2 per channel.
// Slot rows of a tile follow its node rows: row rho < split is eigen slot k = rho of molecule A,
// row rho >= split is slot k = rho - split of molecule B (a molecule has <= min(n, K) live slots and
// n <= its row extent), so V^T and V are block-diagonal exactly like the Laplacians and the
// summation order of a molecule's terms never depends on its tile partner.
// MODE 0 = forward; MODE 3 = forward that also stores every layer's activations (training);
// MODE 1 = input-gradient pass (lnz_lanczosnet_input_grad): the same two chained GEMMs run on dY
//          with per-channel transposed weights, kernel layer t = conv layer num_layer-1-t, the
//          epilogue masks with the stored activation instead of bias + ReLU;
// MODE 2 = message pass (lnz_lanczosnet_messages): GEMM1 is skipped — Z is the stored X_l block in
//          C/D order — and every channel's M_c X_l is written out instead of accumulated.
// DEEPK: which weight-ring loop the GEMM1 uses — 1: the 8-slot ring in every layer (all input
//   widths multiples of 64: the QM8 model), 0: the 4-slot ring, -1: chosen per layer at run time.
//   With both loops present the register allocator gives their rings different registers and the
//   join behind EVERY channel copies one ring into the other behind an s_waitcnt vmcnt(0): the
//   whole prefetch is drained fourteen times a layer.
template <int NWV, int KHT, int FK, int MT, int MODE, int DEEPK = -1>
__device__ __forceinline__ void forward_half(KArgs& a, const TileDesc (&td)[MT],
                                             float (*Xs)[2][32][PITCH],  // [2 buffers][tile][..]
                                             float (*Vm)[32][VPITCH],      // [tile][node row][slot row]
                                             float* Gs,  // [2 buffers][tile][n_long][2 halves][16]
                                             const int htid, const int wave) {
  constexpr bool FWD = MODE == 0 || MODE == 3;
  constexpr bool ES = FK == 0 || FK == 2;  // long channels in eigen space
  constexpr bool DENSE = FK == 2;          // ... with dense K x K gains (block diagonal per tile)
  // Laplacian fragments of a node-space channel: fetched in front of the LAST ring-depth steps of
  // the channel's own GEMM1 (kernels with one ring loop), else one channel ahead
  constexpr bool PEEL = DEEPK >= 0 && ES && MODE != 2;
#ifdef LNZ_EXP_PRIO
  // the two-tile half is the critical path of a 3-tile workgroup: let it issue first, the
  // one-tile half fills the matrix-pipe slots it leaves
  __builtin_amdgcn_s_setprio(MT == 2 ? 3 : 0);
#endif
  const int lane = htid & 63;
  const int j = lane & 31, hh = lane >> 5;
  const int N = a.N, K = a.K, B = a.B;
  const int dhid = a.dhid;
  const int C = a.n_short + a.n_long + a.n_edge;
  const bool es = ES && a.n_long > 0;

  // per-lane view of each tile: molecule and local node of tile row j
  int molj[MT], jl[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const bool first = j < td[m].split;
    molj[m] = first ? td[m].ta : td[m].tb;
    jl[m] = first ? j : j - td[m].split;
  }

  // ---- embedding gather (model/lanczos_net.py:154) / float features (lanczos_net_general.py:156)
  //      MODE 1: the incoming gradient dY of the last conv layer
  if (MODE != 2) {
    const int d4 = a.din0 >> 2;
    for (int idx = htid; idx < MT * 32 * d4; idx += 64 * NWV) {
      const int m = idx / (32 * d4);
      const int rem = idx - m * 32 * d4;
      const int row = rem / d4, c4 = rem - row * d4;
      const TileDesc t = pick(td, m);
      const bool first = row < t.split;
      const int mol = first ? t.ta : t.tb;
      const int lrow = first ? row : row - t.split;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1) {
        if (mol >= 0 && lrow < 32)
          v = reinterpret_cast<const float4*>(
              a.dy + (((int64_t)(a.num_layer - 1) * B + mol) * 32 + lrow) * dhid)[c4];
      } else if (lrow < N) {
        if (a.node_feat) {
          int64_t id = a.node_feat[(int64_t)mol * N + lrow];
          id = id < 0 ? 0 : (id >= a.num_atom ? a.num_atom - 1 : id);
          v = reinterpret_cast<const float4*>(a.embedding + id * a.din0)[c4];
        } else {
          v = reinterpret_cast<const float4*>(a.node_feat_f + ((int64_t)mol * N + lrow) * a.din0)[c4];
        }
      }
      *reinterpret_cast<float4*>(&Xs[0][m][row][4 * c4]) = v;
    }
  }

  // ---- rows of M_c beyond the last real node are zero (zero-padded Laplacian rows/cols, zero
  //      rows of V): GEMM2 k-steps 4g..4g+3 only touch node rows 8g..8g+7, so bit g of g2mask says
  //      whether that group of steps is needed (wave-uniform per tile): molecule A needs groups
  //      g < ceil(nA/8), molecule B groups split/8 <= g < split/8 + ceil(nB/8).
  //      smask is the same for the eigen-slot rows (lift-back k-steps): A has min(nA, K) live
  //      slots from row 0, B min(nB, K) from row split (slots k >= n are zero, dataset/qm8.py:264-291).
  int g2mask[MT], smask[MT], nA[MT], nB[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    tile_extents(a, td[m], lane, nA[m], nB[m]);
    g2mask[m] = row_group_mask(nA[m], nB[m], td[m].split);
    smask[m] = row_group_mask(nA[m] < K ? nA[m] : K, nB[m] < K ? nB[m] : K, td[m].split);
  }

  // ---- edge-type channels that are identities on every molecule of a tile (a bond type the
  //      molecule does not contain): out += Z_c instead of M_c Z_c, no Laplacian fragments
  int idm[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    unsigned v = (FWD && a.ident) ? a.ident[td[m].ta] : 0u;
    if (FWD && a.ident && td[m].tb >= 0) v &= a.ident[td[m].tb];
    idm[m] = __builtin_amdgcn_readfirstlane((int)v);
  }

  // ---- basis fragments.
  //      FK = 0: Ritz vectors staged in LDS as the tile's [node row][slot row] matrix (zero off
  //        the diagonal blocks), Vm[m][j][rho].  The lift-back reads row j as A operand (float4s
  //        of slots cd_row(r..r+3, hh): the order in which the C/D registers of the slot-row sum
  //        chain as B operand), the projection reads column rho as A operand of V^T.
  //      FK = 1: vreg[m][r] = Q[mol m][j][cd_row(r,hh)] — the k-order that lets the same registers
  //        serve as B operand of R = DD Q^T and as A operand of L_s = Q R.
  float vreg[MT][FK == 1 ? KHT : 1];
  if (ES) {
    for (int idx = htid; idx < MT * 32 * 32; idx += 64 * NWV) {
      const int m = idx >> 10, jj = (idx >> 5) & 31, rho = idx & 31;
      Vm[m][jj][rho] = ritz_tile_elem(a, pick(td, m), jj, rho);
    }
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int t = 0; t < (FK == 1 ? KHT : 1); ++t) {
        int k = lnz::cd_row(t, hh);
        vreg[m][t] = (k < K && j < N) ? finite_or_zero(a.V[((int64_t)td[m].ta * N + j) * K + k]) : 0.0f;
      }
    }
  }
  // ---- spectral gains of one layer, staged in LDS by slot row: Gs[buf][m][s][rho] = g_s[k] of
  //      slot row rho of the molecule owning it (zero for unused slots).  Layer l
  //      uses buffer l & 1; layer l+1 is staged at the start of layer l (its buffer was last read
  //      in layer l-1, which every wave left through the barrier).
  auto stage_gains = [&](int l) {
    if (FK != 0 || a.n_long == 0) return;
    float* dst = Gs + (l & 1) * MT * a.n_long * 32;
    for (int idx = htid; idx < MT * a.n_long * 32; idx += 64 * NWV) {
      const int m = idx / (a.n_long * 32);
      const int rem = idx - m * a.n_long * 32;
      const int sc = rem >> 5, rho = rem & 31;
      const TileDesc t = pick(td, m);
      const bool isA = rho < t.split;
      const int k = isA ? rho : rho - t.split;
      // slots k >= n belong to zero-padded eigen columns: their gains are never computed by
      // lnz_spectral_gains_rows and never read here (G may be uninitialised there)
      const bool ok = k < K && k < (isA ? pick(nA, m) : pick(nB, m));
      const int mol = isA ? t.ta : t.tb;
      dst[idx] = ok ? a.G[(((int64_t)l * B + mol) * a.n_long + sc) * K + k] : 0.0f;
    }
  };
  stage_gains(FWD ? 0 : MODE == 1 ? a.num_layer - 1 : a.msg_layer);
  __syncthreads();

#ifdef LNZ_PROFILE_PHASES
  long long t_g1 = 0, t_g2 = 0, t_ep = 0, t_pr = 0, t_mb = 0, t_all = clock64();
#define LNZ_T0 long long _t0 = clock64();
#define LNZ_TR _t0 = clock64();
#define LNZ_ACC(x) { long long _t1 = clock64(); x += _t1 - _t0; _t0 = _t1; }
#else
#define LNZ_T0
#define LNZ_TR
#define LNZ_ACC(x)
#endif
  int cur = 0;
  const int n_iter = MODE == 2 ? 1 : a.num_layer;
  for (int l = 0; l < n_iter; ++l) {
    // la = conv layer handled by this iteration (MODE 1 walks the stack backwards)
    const int la = FWD ? l : MODE == 1 ? a.num_layer - 1 - l : a.msg_layer;
    const int din = MODE == 2 ? 8 : (l == 0 ? a.din0 : dhid);
    const int Q = din >> 3;
    const float4* __restrict__ Wl = reinterpret_cast<const float4*>(a.Wp + a.w_off[l]);
    if (FWD && l + 1 < a.num_layer) stage_gains(l + 1);
    if (MODE == 1 && la > 0) stage_gains(la - 1);
    const float* gsl = Gs + (la & 1) * MT * a.n_long * 32;
    // width this iteration produces: waves beyond it only keep the barrier
    const int wout = FWD ? dhid : MODE == 1 ? (la == 0 ? a.bwd_din0 : dhid)
                                                  : (la == 0 ? a.din0 : dhid);
    const bool active = FWD || 32 * wave < wout;

    f32x16 out[MT];
    {
      const float bv = FWD ? (a.bias + a.b_off[l])[32 * wave + j] : 0.0f;
#pragma unroll
      for (int m = 0; m < MT; ++m) out[m] = lnz::splat16(bv);
    }

    // B-operand stream of this wave's feature tile: contiguous over g = c*Q + q for the whole
    // layer, read through a 4-slot register ring with prefetch distance 3 steps.  One running
    // pointer + immediate offsets; over-reads up to 7 steps past the layer (host keeps 8 KiB slack).
    const int Gtot = C * Q;
    const float4* __restrict__ wp = Wl + (int64_t)wave * Gtot * 64 + lane;
    // Layers whose channels have a multiple of 8 k-steps use all 8 ring slots (prefetch distance
    // 7 steps: with one tile a step is only 4 MFMAs, and three steps do not cover an L2 miss);
    // the narrow first layer keeps the 4-slot rotation.
    float4 ring[8];
    const bool deep = DEEPK >= 0 ? DEEPK == 1  // (the backward modes have no registers to spare)
                                 : (MODE == 0 || MODE == 3) && (Q & 7) == 0;
    if (MODE != 2 && active) {
#pragma unroll
      for (int sl = 0; sl < 7; ++sl)
        if (sl < 3 || deep) ring[sl] = wp[sl * 64];
    }
    // MODE 2: this wave's 32 columns of X_l in C/D order — Z of every channel
    f32x16 Xblk[MODE == 2 ? MT : 1];
    if (MODE == 2 && active) {
      const float* src = la == 0 ? a.x0 : a.act + (int64_t)(la - 1) * B * 32 * dhid;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // rows 8g + 4hh + u: one owner per group (split % 8 == 0)
          const bool first = 8 * g < td[m].split;
          const int mol = first ? td[m].ta : td[m].tb;
          const int lrow0 = 8 * g + 4 * hh - (first ? 0 : td[m].split);
          const float* p = src + ((int64_t)(mol >= 0 ? mol : 0) * 32 + lrow0) * wout + 32 * wave + j;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            Xblk[MODE == 2 ? m : 0][4 * g + u] = mol >= 0 ? p[u * wout] : 0.0f;
        }
      }
    }

    lds_cptr xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xrow[m] = (lds_cptr)&Xs[cur][m][j][4 * hh];

    // ---------------- eigen-space projection Y = V^T X (long channels, FK = 0) ----------------
    //   A operand: V^T fragments, lane (slot row j, hh) step r = Vm[node row cd_row(r,hh)][slot j];
    //   B operand: X rows in the same order.  Y lives in the other X buffer (free until the
    //   epilogue), where the long channels' GEMM1 reads it as A operand.  Only the FIRST layer is
    //   projected here, from LDS: every later layer's Y is produced by the previous epilogue
    //   straight from its C/D registers.  MODE 2 keeps this wave's block in registers.
    const int nxt = cur ^ 1;
    LNZ_T0
    f32x16 Yblk[(ES && MODE == 2) ? MT : 1];
    if (es && (MODE == 2 || l == 0)) {
      if (32 * wave < din || MODE == 2) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (MODE == 2 && !active) continue;
          float vt[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) vt[r] = Vm[m][lnz::cd_row(r, hh)][j];
          f32x16 Y = lnz::splat16(0.0f);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if ((g2mask[m] >> g) & 1) {
#pragma unroll
              for (int r = 4 * g; r < 4 * g + 4; ++r) {
                const float xb = MODE == 2 ? Xblk[MODE == 2 ? m : 0][r]
                                           : Xs[cur][m][lnz::cd_row(r, hh)][32 * wave + j];
                Y = lnz::mfma32(vt[r], xb, Y);
              }
            }
          }
          if (MODE == 2) {
            Yblk[(ES && MODE == 2) ? m : 0] = Y;
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) Xs[nxt][m][lnz::cd_row(r, hh)][32 * wave + j] = Y[r];
          }
        }
      }
      if (MODE != 2) __syncthreads();
    }
    LNZ_ACC(t_pr)
    lds_cptr yrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) yrow[m] = (lds_cptr)&Xs[nxt][m][j][4 * hh];

    // Operands of GEMM2 (per tile, 16 registers: the spectral gains of this lane half's eigen
    // slots for a long channel OR the four float4 Laplacian fragments of an edge/short channel)
    // are fetched one channel ahead, right after the previous fragments are consumed and before
    // that tile's GEMM2 — so they have >= 16 MFMAs to land and are waited for together with the
    // oldest ring slot at the next GEMM1 loop header.
    float mop[MT][16];
    auto fetch_m_operands = [&](int c, int m) {
      const bool lng = (c >= a.n_short) && (c < a.n_short + a.n_long);
      if (lng) {
        if (FK == 1) {
          // row j of the symmetric K x K filter DD_s, columns in cd_row order
          const float* dp =
              a.G + ((((int64_t)la * B + td[m].ta) * a.n_long + (c - a.n_short)) * K + j) * K;
#pragma unroll
          for (int t = 0; t < (FK == 1 ? KHT : 1); ++t) {
            int k2 = lnz::cd_row(t, hh);
            mop[m][t] = (j < K && k2 < K) ? dp[k2] : 0.0f;
          }
        } else if (DENSE) {
          // fragment group g of lane (rho = j, hh) = DDtile[rho][8g + 4hh + 0..3]: slot row rho of
          // the tile belongs to molecule A (rho < split, its slot rho) or B (slot rho - split, its
          // columns behind A's) — block diagonal like the Laplacian fragments below; rows /
          // columns >= K are zero.  DD_s is [K][K] row major (symmetric): 16-byte loads when K % 4
          // == 0 (QM8: K = 20), element loads otherwise.
          const int g0 = td[m].split >> 3;
          const bool rowA = j < td[m].split;
          const int kr = jl[m];  // this lane's slot within its molecule
          const float* dp = a.G + ((((int64_t)la * B + (molj[m] >= 0 ? molj[m] : 0)) * a.n_long +
                                    (c - a.n_short)) * K + (kr < K ? kr : 0)) * K;
          const bool live = kr < K && molj[m] >= 0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int k0 = 8 * (g - (rowA ? 0 : g0)) + 4 * hh;  // first column, molecule-local
            const bool blk = live && (rowA ? g < g0 : g >= g0);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((K & 3) == 0) {
              if (blk && k0 < K) v = *reinterpret_cast<const float4*>(dp + k0);
            } else if (blk) {
              v.x = k0 + 0 < K ? dp[k0 + 0] : 0.0f;
              v.y = k0 + 1 < K ? dp[k0 + 1] : 0.0f;
              v.z = k0 + 2 < K ? dp[k0 + 2] : 0.0f;
              v.w = k0 + 3 < K ? dp[k0 + 3] : 0.0f;
            }
            mop[m][4 * g + 0] = v.x;
            mop[m][4 * g + 1] = v.y;
            mop[m][4 * g + 2] = v.z;
            mop[m][4 * g + 3] = v.w;
          }
        }  // FK = 0: the eigen-space block below reads its gains from LDS itself
      } else if (c >= a.n_short && ((idm[m] >> (c - a.n_short - a.n_long)) & 1)) {
        // identity channel: nothing to fetch
      } else {
        const int e = c < a.n_short ? 0 : c - a.n_short - a.n_long;
        // fragment group g of lane (j,hh) = M[j][8g + 4hh + 0..3].  Rows of molecule A take its
        // own groups 0..split/8-1; rows of molecule B take B's groups 0.. as tile groups
        // split/8..3 (its columns sit behind A's); the off-diagonal blocks are zero.
        const int g0 = td[m].split >> 3;
        const bool rowA = j < td[m].split;
        // per-lane base shifted by the row block's first group: group g sits at lpl[g * 64]
        // (constant offsets, nothing per-group to keep in registers)
        const float4* lpl = reinterpret_cast<const float4*>(a.Lp) +
                            ((int64_t)molj[m] * a.n_edge + e) * 256 + 32 * hh + jl[m] -
                            (rowA ? 0 : g0 * 64);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rowA ? g < g0 : g >= g0) v = lpl[g * 64];
          mop[m][4 * g + 0] = v.x;
          mop[m][4 * g + 1] = v.y;
          mop[m][4 * g + 2] = v.z;
          mop[m][4 * g + 3] = v.w;
        }
      }
    };
    if (active && !es && !PEEL) {  // (PEEL: fetched inside the channel's GEMM1)
#pragma unroll
      for (int m = 0; m < MT; ++m) fetch_m_operands(0, m);
    }

    auto store_message = [&](int c, int m, const f32x16& P) {
      const int64_t ld = (int64_t)C * wout;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if ((g2mask[m] >> g) & 1) {  // rows of groups no molecule owns are never written
          const bool first = 8 * g < td[m].split;
          const int mol = first ? td[m].ta : td[m].tb;
          const int lrow0 = 8 * g + 4 * hh - (first ? 0 : td[m].split);
          // row_off: compact row numbering (real nodes only); else 32 rows per molecule
          const int64_t r0 = a.row_off ? (int64_t)a.row_off[mol] + lrow0 : (int64_t)mol * 32 + lrow0;
          const int nmol = a.row_off ? (first ? nA[m] : nB[m]) : 32;
          float* p = a.msg + r0 * ld + (int64_t)c * wout + 32 * wave + j;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (lrow0 + u < nmol) p[u * ld] = P[4 * g + u];
        }
      }
    };

    // GEMM1 of one channel into Z: Z_m (+)= [diag(g)] A_m W_c^T, A rows from `rows` (X or Y).
    f32x16 Z[MT];
    auto gemm1_rd = [&](auto depth, const lds_cptr (&rows)[MT], auto&& before_last) {
      constexpr int RD = decltype(depth)::value;  // ring slots = steps per unrolled body
      lds_cptr xq[MT];
      float4 acur[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        xq[m] = rows[m];
        acur[m] = lds_f4(xq[m]);
      }
      auto rd_steps = [&]() {
#pragma unroll
        for (int u4 = 0; u4 < RD; ++u4) {
#ifndef LNZ_EXP_NO_WLOAD
          ring[(u4 + RD - 1) & (RD - 1)] = wp[(u4 + RD - 1) * 64];
#endif
          // next A fragments (the read one step past the channel's last is in-bounds, unused)
          float4 anext[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m)
            anext[m] = lds_f4(xq[m] + 8 * (u4 + 1));
          // keep the prefetches ahead of this step's MFMAs (hipcc otherwise sinks all loads of
          // the unrolled body to its end and waits vmcnt(0) at the top of the next iteration)
          const float4 bv = ring[u4];
#pragma unroll
          for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].x, bv.x, Z[m]);
#pragma unroll
          for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].y, bv.y, Z[m]);
#pragma unroll
          for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].z, bv.z, Z[m]);
#pragma unroll
          for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].w, bv.w, Z[m]);
#pragma unroll
          for (int m = 0; m < MT; ++m) acur[m] = anext[m];
          // Issue order of a step: ONE load between pairs of MFMAs instead of all of the step's
          // loads in one gap (the matrix pipe hides a few single-issue instructions per gap; a
          // bunch of five does not fit): next A fragments first (they are used one step later),
          // the weight-ring load last (used seven steps later).  Same-process A/B on the bench
          // batch: 0.735 -> 0.709 ms; bunched = the old sched_barrier(0) fences.
          if (MT == 2) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ds_read
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // global load
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) xq[m] += 8 * RD;
        wp += RD * 64;
      };
      // all but the last RD steps in the loop, the last RD straight-line behind `before_last`:
      // vector loads issued there (the channel's Laplacian fragments) are followed by a KNOWN
      // number of ring loads, so their s_waitcnt does not have to drain the weight ring — a load
      // carried across a loop of unknown trip count can only be waited for with vmcnt(0)
      // (kernels that carry both ring loops — DEEPK < 0, odd widths — keep one body per loop:
      // four bodies per call site spill; there `before_last` is empty)
      if constexpr (DEEPK >= 0) {
#pragma unroll 1
        for (int q0 = RD; q0 < Q; q0 += RD) rd_steps();
        before_last();
        rd_steps();
      } else {
        before_last();
#pragma unroll 1
        for (int q0 = 0; q0 < Q; q0 += RD) rd_steps();
      }
    };
    auto gemm1 = [&](const lds_cptr (&rows)[MT], auto&& before_last) {
      if constexpr (DEEPK == 1) gemm1_rd(std::integral_constant<int, 8>{}, rows, before_last);
      else if constexpr (DEEPK == 0) gemm1_rd(std::integral_constant<int, 4>{}, rows, before_last);
      else if (deep) gemm1_rd(std::integral_constant<int, 8>{}, rows, before_last);
      else gemm1_rd(std::integral_constant<int, 4>{}, rows, before_last);
    };

    // ---------------- eigen-space block: all long channels of the layer ----------------
    //   runs BEFORE the node-space channels (its own loop nest keeps the register allocation of
    //   both parts tight); with short channels present the weight stream is re-entered at the
    //   long block, rewound for the short channels and skipped over afterwards.
    const int c_end = a.n_short + a.n_long;           // first edge channel
    const int c_first = a.n_short > 0 ? 0 : c_end;    // first node-space channel
    auto prime_ring = [&](int step0) {
      wp = Wl + ((int64_t)wave * Gtot + step0) * 64 + lane;
#pragma unroll
      for (int sl = 0; sl < 7; ++sl)
        if (sl < 3 || deep) ring[sl] = wp[sl * 64];
    };
    if (es && active) {
      LNZ_T0
      if (MODE != 2 && a.n_short > 0) prime_ring(a.n_short * Q);
        if (MODE != 2) {
          // T_m = sum_s diag(g_s) (Y_m W_s^T): each channel's GEMM1 runs unscaled — a VALU
          // multiply in front of every MFMA costs ~40 cycles per MFMA (tools/mfma_issue_probe.hip)
          // — and its C/D rows (= eigen slots) are scaled into T by 16 FMAs per tile; then
          // out_m += V_m T_m
          f32x16 T[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) T[m] = lnz::splat16(0.0f);
          for (int s = 0; s < a.n_long; ++s) {
#pragma unroll
            for (int m = 0; m < MT; ++m) Z[m] = lnz::splat16(0.0f);
            if constexpr (DENSE) {
              // this channel's DD fragments land under its own GEMM1 (fetched in front of its
              // last ring-depth steps where the ring loop is peeled, else in front of the loop)
              // (tile 0's under the GEMM1, tile 1's under tile 0's MFMA chain: with both sets live
              // across the GEMM1 the kernel spills)
              gemm1(yrow, [&] { fetch_m_operands(a.n_short + s, 0); });
              // T_m += DD_s,m Z_m: GEMM2 with the DD fragments as M, on the live slot groups
#pragma unroll
              for (int m = 0; m < MT; ++m) {
                if (m + 1 < MT) fetch_m_operands(a.n_short + s, m + 1);
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                  if ((smask[m] >> (r >> 2)) & 1) {
                    T[m] = lnz::mfma32(mop[m][r + 0], Z[m][r + 0], T[m]);
                    T[m] = lnz::mfma32(mop[m][r + 1], Z[m][r + 1], T[m]);
                    T[m] = lnz::mfma32(mop[m][r + 2], Z[m][r + 2], T[m]);
                    T[m] = lnz::mfma32(mop[m][r + 3], Z[m][r + 3], T[m]);
                  }
                }
              }
              continue;
            }
            gemm1(yrow, [] {});
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float* g4 = gsl + (m * a.n_long + s) * 32 + 4 * hh;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float4 gv = *reinterpret_cast<const float4*>(g4 + 8 * g);
                T[m][4 * g + 0] = fmaf(gv.x, Z[m][4 * g + 0], T[m][4 * g + 0]);
                T[m][4 * g + 1] = fmaf(gv.y, Z[m][4 * g + 1], T[m][4 * g + 1]);
                T[m][4 * g + 2] = fmaf(gv.z, Z[m][4 * g + 2], T[m][4 * g + 2]);
                T[m][4 * g + 3] = fmaf(gv.w, Z[m][4 * g + 3], T[m][4 * g + 3]);
              }
            }
          }
          LNZ_ACC(t_g1)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            if (!PEEL && c_first < C) fetch_m_operands(c_first, m);
            const float* vs = &Vm[m][j][4 * hh];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
              if ((smask[m] >> t4) & 1) {
                const float4 v = *reinterpret_cast<const float4*>(vs + 8 * t4);
                out[m] = lnz::mfma32(v.x, T[m][4 * t4 + 0], out[m]);
                out[m] = lnz::mfma32(v.y, T[m][4 * t4 + 1], out[m]);
                out[m] = lnz::mfma32(v.z, T[m][4 * t4 + 2], out[m]);
                out[m] = lnz::mfma32(v.w, T[m][4 * t4 + 3], out[m]);
              }
            }
          }
          LNZ_ACC(t_mb)
        } else {
          // messages g_s * Y lifted back through V, one per channel
          for (int s = 0; s < a.n_long; ++s) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float* g4 = gsl + (m * a.n_long + s) * 32 + 4 * hh;
              const float* vs = &Vm[m][j][4 * hh];
              // scale first, then the MFMA chain: a VALU multiply in front of every MFMA stalls
              // the matrix pipe (tools/mfma_issue_probe.hip)
              const f32x16& Y = Yblk[(ES && MODE == 2) ? m : 0];
              f32x16 T;
              if constexpr (DENSE) {
                // T = DD_s Y: the dense filter's fragments as M operand of one MFMA chain
                fetch_m_operands(a.n_short + s, m);
                T = lnz::splat16(0.0f);
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                  if ((smask[m] >> (r >> 2)) & 1) {
                    T = lnz::mfma32(mop[m][r + 0], Y[r + 0], T);
                    T = lnz::mfma32(mop[m][r + 1], Y[r + 1], T);
                    T = lnz::mfma32(mop[m][r + 2], Y[r + 2], T);
                    T = lnz::mfma32(mop[m][r + 3], Y[r + 3], T);
                  }
                }
              } else {
#pragma unroll
              for (int t4 = 0; t4 < 4; ++t4) {
                const float4 g = *reinterpret_cast<const float4*>(g4 + 8 * t4);
                T[4 * t4 + 0] = g.x * Y[4 * t4 + 0];
                T[4 * t4 + 1] = g.y * Y[4 * t4 + 1];
                T[4 * t4 + 2] = g.z * Y[4 * t4 + 2];
                T[4 * t4 + 3] = g.w * Y[4 * t4 + 3];
              }
              }
              f32x16 P = lnz::splat16(0.0f);
#pragma unroll
              for (int t4 = 0; t4 < 4; ++t4) {
                if ((smask[m] >> t4) & 1) {
                  const float4 v = *reinterpret_cast<const float4*>(vs + 8 * t4);
                  P = lnz::mfma32(v.x, T[4 * t4 + 0], P);
                  P = lnz::mfma32(v.y, T[4 * t4 + 1], P);
                  P = lnz::mfma32(v.z, T[4 * t4 + 2], P);
                  P = lnz::mfma32(v.w, T[4 * t4 + 3], P);
                }
              }
              store_message(a.n_short + s, m, P);
            }
          }
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (c_first < C) fetch_m_operands(c_first, m);
        }
      if (MODE != 2 && a.n_short > 0) prime_ring(0);
    }
    // The node-space channels, in one or two contiguous ranges of the weight stream: without
    // eigen space all of [0, C); in eigen space (the long block is done) the short channels
    // [0, n_short) and then, behind the long block, the edge channels [c_end, C) — the ring is
    // re-primed BETWEEN the ranges, outside the channel loop: with the jump inside it the loop had
    // three ways in, and hipcc's s_waitcnt placement drained the weight ring (vmcnt(1)) at the top
    // of every 8-step iteration of the GEMM1 loop.
    for (int range = 0; range < 2; ++range) {
      int c_lo = es ? c_first : 0, c_hi = (es && a.n_short > 0) ? a.n_short : C;
      if (range == 1) {
        if (!(es && a.n_short > 0) || c_end >= C) break;
        c_lo = c_end;
        c_hi = C;
        if (MODE != 2 && active) prime_ring(c_end * Q);
      }
    for (int c = c_lo; active && c < c_hi; ++c) {
      const bool is_long = (c >= a.n_short) && (c < a.n_short + a.n_long);

      LNZ_T0
      // ---------------- GEMM1: Z_m = X_m W_c^T ----------------
#pragma unroll
      for (int m = 0; m < MT; ++m) Z[m] = MODE == 2 ? Xblk[MODE == 2 ? m : 0] : lnz::splat16(0.0f);
      if (MODE != 2) {
        // FK = 0: this channel's Laplacian fragments are fetched in front of the LAST ring-depth
        // steps of its GEMM1 (>= 32 MFMAs to land)
        gemm1(xrow, [&] {
          if (PEEL) {
#pragma unroll
            for (int m = 0; m < MT; ++m) fetch_m_operands(c, m);
          }
        });
      }

      LNZ_ACC(t_g1)
      // ---------------- per tile: M_c fragments, next operands, GEMM2 ----------------
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        // M_c is the identity on this tile's molecules: out += Z_c (bit-identical on their nodes)
        const bool idc = FWD && !is_long && c >= a.n_short &&
                         ((idm[m] >> (c - a.n_short - a.n_long)) & 1);
        // FK = 0: the fragments fetched one channel ahead are used in place and the next
        // channel's are fetched once this tile's MFMAs are issued (no register copy in front of
        // the GEMM2 chain; they still have a whole GEMM1 to land)
        constexpr bool INPLACE = ES;
        f32x16 Mf;
        if (INPLACE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) Mf[r] = 0.0f;  // unused (keeps the type uniform below)
        } else if (is_long) {
          // FK = 1: R[k1][n] = sum_k2 DD[k1][k2] Q[n][k2], L_s[i][n] = sum_k1 Q[i][k1] R[k1][n]
          f32x16 acc = lnz::splat16(0.0f);
          f32x16 R = lnz::splat16(0.0f);
#pragma unroll
          for (int t = 0; t < (FK == 1 ? KHT : 1); ++t) R = lnz::mfma32(mop[m][t], vreg[m][t], R);
#pragma unroll
          for (int t = 0; t < (FK == 1 ? KHT : 1); ++t) acc = lnz::mfma32(vreg[m][t], R[t], acc);
          Mf = acc;  // L_s[cd_row(r,hh)][j] == L_s[j][cd_row(r,hh)]  (symmetric)
        } else if (!idc) {
#pragma unroll
          for (int r = 0; r < 16; ++r) Mf[r] = mop[m][r];
        }
        if (!INPLACE) {
          const int cn = (es && c + 1 == a.n_short) ? c_end : c + 1;
          if (cn < C) fetch_m_operands(cn, m);
        }
#define LNZ_MF(r) (INPLACE ? mop[m][r] : Mf[r])

        // short diffusion: Z <- L_0^(p-1) Z
        if (c < a.n_short) {
          const int p = a.short_dist[c];
          for (int rep = 1; rep < p; ++rep) {
            f32x16 T = lnz::splat16(0.0f);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if ((g2mask[m] >> (r >> 2)) & 1) T = lnz::mfma32(LNZ_MF(r), Z[m][r], T);
            }
            Z[m] = T;
          }
        }
        // GEMM2: out_m += M_c,m Z_m   (MODE 2: the message M_c,m X_m itself, written out)
        f32x16 P = lnz::splat16(0.0f);
        if (idc) {
#pragma unroll
          for (int r = 0; r < 16; ++r) out[m][r] += Z[m][r];
        }
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          if (!idc && ((g2mask[m] >> (r >> 2)) & 1)) {
            if (MODE == 2) {
              P = lnz::mfma32(LNZ_MF(r + 0), Z[m][r + 0], P);
              P = lnz::mfma32(LNZ_MF(r + 1), Z[m][r + 1], P);
              P = lnz::mfma32(LNZ_MF(r + 2), Z[m][r + 2], P);
              P = lnz::mfma32(LNZ_MF(r + 3), Z[m][r + 3], P);
            } else {
              out[m] = lnz::mfma32(LNZ_MF(r + 0), Z[m][r + 0], out[m]);
              out[m] = lnz::mfma32(LNZ_MF(r + 1), Z[m][r + 1], out[m]);
              out[m] = lnz::mfma32(LNZ_MF(r + 2), Z[m][r + 2], out[m]);
              out[m] = lnz::mfma32(LNZ_MF(r + 3), Z[m][r + 3], out[m]);
            }
          }
        }
#undef LNZ_MF
        if (INPLACE && !PEEL) {  // one channel ahead
          const int cn = (es && c + 1 == a.n_short) ? c_end : c + 1;
          if (cn < C) fetch_m_operands(cn, m);
        }
        if (MODE == 2) store_message(c, m, P);
      }
      LNZ_ACC(t_g2)
    }
    }

    // ---------------- epilogue: X' -> LDS (other buffer), one barrier per layer -------------
    //   MODE 0: ReLU (+ the activation store training asks for)
    //   MODE 1: dY_{la-1} = dX_la * [X_la > 0] -> LDS and dy[la-1]; the last iteration writes dX_0
    //   eigen space: a barrier first (every wave is done with X and Y), then X' goes where Y was
    //   and the NEXT layer's projection Y' = V^T X' — computed from the same C/D registers, which
    //   are its B operand as they stand — goes where X was.
    LNZ_TR
    if (es && MODE != 2) __syncthreads();
    if (MODE != 2 && active) {
      const int col = 32 * wave + j;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // rows 8g + 4hh + u: one owner per group (split % 8 == 0)
          const bool first = 8 * g < td[m].split;
          const int mol = first ? td[m].ta : td[m].tb;
          const int lrow0 = 8 * g + 4 * hh - (first ? 0 : td[m].split);
          const int64_t rowbase = (int64_t)(mol >= 0 ? mol : 0) * 32 + lrow0;
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = out[m][4 * g + u];
          if (FWD) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.0f);
            if (MODE == 3 && mol >= 0) {
              float* p = a.act_out + ((int64_t)l * B * 32 + rowbase) * dhid + col;
#pragma unroll
              for (int u = 0; u < 4; ++u) p[u * dhid] = v[u];
            }
          } else if (la > 0) {
            const int64_t at = ((int64_t)(la - 1) * B * 32 + rowbase) * dhid + col;
            const float* xa = a.act + at;
            float* dyp = a.dy + at;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              v[u] = (mol >= 0 && xa[u * dhid] > 0.0f) ? v[u] : 0.0f;
              if (mol >= 0) dyp[u * dhid] = v[u];
            }
          } else if (mol >= 0) {
            float* p = a.dx0 + rowbase * a.bwd_din0 + col;
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u * a.bwd_din0] = v[u];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            Xs[nxt][m][8 * g + 4 * hh + u][col] = v[u];
            out[m][4 * g + u] = v[u];
          }
        }
        if (es && l + 1 < n_iter) {
          f32x16 Y = lnz::splat16(0.0f);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if ((g2mask[m] >> g) & 1) {
#pragma unroll
              for (int r = 4 * g; r < 4 * g + 4; ++r)
                Y = lnz::mfma32(Vm[m][lnz::cd_row(r, hh)][j], out[m][r], Y);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) Xs[cur][m][lnz::cd_row(r, hh)][col] = Y[r];
        }
      }
      if (MODE == 1 && la > 0 && (a.dy_compact || a.dbias_part)) {
        // What the weight / bias gradients of conv layer la - 1 need, as a second pass over the
        // values this lane just wrote to LDS (the accumulators are dead by now: no registers
        // taken from the loop above): dY_{la-1} in the COMPACT row numbering of the message
        // matrix (real nodes only), and this half's column sums (rows of padded nodes and of
        // unowned tile rows are zero here) — one writer per (workgroup half, layer, column).
        float colsum = 0.0f;
#pragma unroll 1
        for (int m = 0; m < MT; ++m) {
          const TileDesc t = pick(td, m);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const bool first = 8 * g < t.split;
            const int mol = first ? t.ta : t.tb;
            const int lrow0 = 8 * g + 4 * hh - (first ? 0 : t.split);
            const int nmol = first ? pick(nA, m) : pick(nB, m);
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = Xs[nxt][m][8 * g + 4 * hh + u][col];
            colsum += (v[0] + v[1]) + (v[2] + v[3]);
            if (a.dy_compact && mol >= 0) {
              float* dc = a.dy_compact +
                          ((int64_t)(la - 1) * a.dy_compact_rows + a.row_off[mol] + lrow0) * dhid + col;
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (lrow0 + u < nmol) dc[u * dhid] = v[u];
            }
          }
        }
        if (a.dbias_part) {
          colsum += __shfl_xor(colsum, 32, 64);
          const int half = threadIdx.x / (64 * NWV);
          if (hh == 0)
            a.dbias_part[(((int64_t)blockIdx.x * 2 + half) * a.num_layer + (la - 1)) * dhid + col] = colsum;
        }
      }
    }
    __syncthreads();
    cur = nxt;
    LNZ_ACC(t_ep)
  }
#ifdef LNZ_PROFILE_PHASES
  if (a.state_out && lane == 0 && blockIdx.x < 8) {
    float* d = a.state_out + ((int64_t)B * 32 * dhid) + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
    d[0] = (float)t_g1; d[1] = (float)t_g2; d[2] = (float)t_ep; d[3] = (float)(clock64() - t_all);
    d[4] = (float)t_pr; d[5] = (float)t_mb; d[6] = (float)MT;
  }
#endif

  if (!FWD) return;

  // ---- optional debug/test output of the final node state (rows of the tile each molecule owns)
  if (a.state_out) {
    for (int idx = htid; idx < MT * 32 * dhid; idx += 64 * NWV) {
      const int m = idx / (32 * dhid);
      const int rem = idx - m * 32 * dhid;
      const int row = rem / dhid, col = rem - row * dhid;
      const TileDesc t = pick(td, m);
      const bool first = row < t.split;
      const int mol = first ? t.ta : t.tb;
      const int lrow = first ? row : row - t.split;
      if (mol >= 0) a.state_out[((int64_t)mol * 32 + lrow) * dhid + col] = Xs[cur][m][row][col];
    }
  }

  // ---- head (model/lanczos_net.py:185-194): one 32-column tile = [W_o ; w_a ; 0];
  //      wave m handles tile m.  Register group r>>2 of a C/D fragment covers tile rows
  //      8(r>>2)..+7, so the two molecules of a pair tile reduce over separate registers.
  if (wave < MT) {
    const int m = wave;
    const TileDesc t = pick(td, m);
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
    const bool pr = t.tb >= 0;
    const int64_t mol0 = t.ta, mol1 = pr ? t.tb : t.ta;
    const int g0 = t.split >> 3;
    float sum0 = 0.0f, sum1 = 0.0f, cnt0 = 0.0f, cnt1 = 0.0f;
    const int src = 32 * hh + P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float logit = __shfl(acc[r], src, 64);
      float gate = 1.0f / (1.0f + __expf(-logit));
      const int row = lnz::cd_row(r, hh);
      const bool second = (r >> 2) >= g0;
      const int lrow = second ? row - t.split : row;
      const int64_t mol = second ? mol1 : mol0;
      const bool msk = lrow < N && a.mask[mol * N + lrow] != 0;
      const float val = msk ? gate * acc[r] : 0.0f, one = msk ? 1.0f : 0.0f;
      sum0 += second ? 0.0f : val;
      cnt0 += second ? 0.0f : one;
      sum1 += second ? val : 0.0f;
      cnt1 += second ? one : 0.0f;
    }
    sum0 += __shfl_xor(sum0, 32, 64);
    cnt0 += __shfl_xor(cnt0, 32, 64);
    sum1 += __shfl_xor(sum1, 32, 64);
    cnt1 += __shfl_xor(cnt1, 32, 64);
    if (hh == 0 && j < P) {
      a.score[mol0 * P + j] = sum0 / cnt0;
      if (pr) a.score[mol1 * P + j] = sum1 / cnt1;
    }
  }
}

// One workgroup = 2 halves x NWV wavefronts; half h works on slots 2h, 2h+1 of the workgroup's
// plan entry (0, 1 or 2 node tiles).  With the plan of lnz_plan_tiles there is one workgroup per
// CU while the batch fits in one round, holding floor/ceil of (tiles / CUs) tiles.
template <int NWV, int KHT, int FK, int MODE, int DEEPK = -1>
__global__ __launch_bounds__(128 * NWV) void lanczosnet_forward_kernel(const lnz_forward_args) {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) float Xs[2][2][MOLS][32][PITCH];  // [half][buffer][tile]
  const int tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float Vm[2][MOLS][32][VPITCH];  // FK = 0: Ritz vectors
  extern __shared__ __attribute__((aligned(16))) float Gs_all[];  // FK = 0: [half][2][MOLS][n_long][32]
  float* Gs = Gs_all + (tid / (64 * NWV)) * 2 * MOLS * a.n_long * 32;
  const int W = a.plan ? *a.n_wg : (a.B + 3) / 4;
  if ((int)blockIdx.x >= W) return;  // the grid is sized for the unpaired worst case
  const int half = __builtin_amdgcn_readfirstlane(tid / (64 * NWV));
  const int htid = tid - half * 64 * NWV;
  const int wave = __builtin_amdgcn_readfirstlane(htid >> 6);

  TileDesc td[MOLS];
  int nt = 0;
#pragma unroll
  for (int m = 0; m < MOLS; ++m) {
    td[m] = load_tile_desc(a, (int)blockIdx.x * 4 + 2 * half + m);
    nt += td[m].ta >= 0 ? 1 : 0;  // slots fill from 0: a used slot 1 implies a used slot 0
  }
  if (nt == 2) {
    forward_half<NWV, KHT, FK, 2, MODE, DEEPK>(a, td, Xs[half], Vm[half], Gs, htid, wave);
  } else if (nt == 1) {
    const TileDesc t1[1] = {td[0]};
    forward_half<NWV, KHT, FK, 1, MODE, DEEPK>(a, t1, Xs[half], Vm[half], Gs, htid, wave);
  } else {
    // keep the barrier count of the other half: setup, (eigen space: the first layer's
    // projection,) and per layer one barrier (eigen space: two)
    const bool es = (FK == 0 || FK == 2) && a.n_long > 0;
    const int nb = MODE == 2 ? 2 : (es ? 2 : 1) * a.num_layer + 1 + (es ? 1 : 0);
    for (int l = 0; l < nb; ++l) __syncthreads();
  }
}


// -----------------------------------------------------------------------------------------
// Gradient of the loss w.r.t. the spectral gains (training), in eigen space:
//   out_l += V [ sum_s diag(g_s) (V^T X_l) W_s^T ]   =>   dG[l][b][k][s] = sum_o P[k][o] Q_s[k][o],
//   P = V^T dY_l  (slot rows x dhid),   Q_s = (V^T X_l) W_s^T
// One half = NWV wavefronts x MT node tiles, every conv layer in turn: X_l and dY_l are staged in
// the two LDS tile buffers, projected (Y to LDS over X_l, P kept in C/D registers), then each long
// channel's GEMM1 runs on Y with the FORWARD weight pack and its result is multiplied with P and
// reduced over the wave's 32 output columns (lane shuffles) and over the NWV waves (LDS, fixed
// order: deterministic).  No gains, Laplacians or activations other than X_l / dY_l are read.
// sum over the 16 lanes of a DPP row, in every lane of the row (quad swaps, then the two mirrors)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
  return v;
}

// DEEP: the weight-ring depth as a template constant (1: 8 slots — every layer's channels have a
// multiple of 8 k-steps —, 0: 4 slots): with both ring loops in one kernel their rings get
// different registers and the join behind every channel drains the prefetch (see forward_half).
template <int NWV, int MT, int DEEP>
__device__ __forceinline__ void gain_grad_half(KArgs& a, const TileDesc (&td)[MT],
                                               float (*Xs)[2][32][PITCH], float (*Vm)[32][VPITCH],
                                               const int htid, const int wave) {
  const int lane = htid & 63;
  const int j = lane & 31, hh = lane >> 5;
  const int K = a.K, B = a.B, dhid = a.dhid, S = a.n_long;
  const int C = a.n_short + a.n_long + a.n_edge;

  int g2mask[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int nA, nB;
    tile_extents(a, td[m], lane, nA, nB);
    g2mask[m] = row_group_mask(nA, nB, td[m].split);
  }
  for (int idx = htid; idx < MT * 32 * 32; idx += 64 * NWV) {
    const int m = idx >> 10, jj = (idx >> 5) & 31, rho = idx & 31;
    Vm[m][jj][rho] = ritz_tile_elem(a, pick(td, m), jj, rho);
  }

  for (int la = 0; la < a.num_layer; ++la) {
    const int din = la == 0 ? a.din0 : dhid;
    const int Q = din >> 3;
    // ---- stage X_la (buffer 0) and dY_la (buffer 1): rows of the tile's molecules, zero elsewhere
    {
      const float* xsrc = la == 0 ? a.x0 : a.act + (int64_t)(la - 1) * B * 32 * dhid;
      const int d4 = din >> 2, e4 = dhid >> 2;
      const int total = MT * 32 * (d4 + e4);
      // four loads in flight per thread: with one load -> one LDS store per trip every trip waited
      // out a global round trip (8-16 trips per layer)
      constexpr int UN = 4;
      for (int base = htid; base < total; base += 64 * NWV * UN) {
        float4 v[UN];
        float* dst[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int idx = base + u * 64 * NWV;
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[u] = nullptr;
          if (idx < total) {
            const int m = idx / (32 * (d4 + e4));
            const int rem = idx - m * 32 * (d4 + e4);
            const int row = rem / (d4 + e4), c = rem - row * (d4 + e4);
            const TileDesc t = pick(td, m);
            const bool first = row < t.split;
            const int mol = first ? t.ta : t.tb;
            const int lrow = first ? row : row - t.split;
            if (c < d4) {
              if (mol >= 0)
                v[u] = reinterpret_cast<const float4*>(xsrc + ((int64_t)mol * 32 + lrow) * din)[c];
              dst[u] = &Xs[0][m][row][4 * c];
            } else {
              if (mol >= 0)
                v[u] = reinterpret_cast<const float4*>(
                    a.dy + (((int64_t)la * B + mol) * 32 + lrow) * dhid)[c - d4];
              dst[u] = &Xs[1][m][row][4 * (c - d4)];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
          if (dst[u]) *reinterpret_cast<float4*>(dst[u]) = v[u];
      }
    }
    __syncthreads();
    // ---- projections: P = V^T dY (registers), Y = V^T X (registers, then LDS over X)
    f32x16 Pb[MT], Yb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float vt[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vt[r] = Vm[m][lnz::cd_row(r, hh)][j];
      f32x16 Pm = lnz::splat16(0.0f), Ym = lnz::splat16(0.0f);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if ((g2mask[m] >> g) & 1) {
#pragma unroll
          for (int r = 4 * g; r < 4 * g + 4; ++r) {
            Pm = lnz::mfma32(vt[r], Xs[1][m][lnz::cd_row(r, hh)][32 * wave + j], Pm);
            if (32 * wave < din)
              Ym = lnz::mfma32(vt[r], Xs[0][m][lnz::cd_row(r, hh)][32 * wave + j], Ym);
          }
        }
      }
      Pb[m] = Pm;
      Yb[m] = Ym;
    }
    __syncthreads();  // every wave has read X and dY
    if (32 * wave < din) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[0][m][lnz::cd_row(r, hh)][32 * wave + j] = Yb[m][r];
      }
    }
    __syncthreads();
    // ---- long channels: Q_s = Y W_s^T, row-wise <P, Q_s> over this wave's 32 columns
    const float4* __restrict__ Wl = reinterpret_cast<const float4*>(a.Wp + a.w_off[la]);
    const float4* __restrict__ wp =
        Wl + ((int64_t)wave * (C * Q) + (int64_t)a.n_short * Q) * 64 + lane;
    // weight stream of the layer's S long channels (contiguous in the pack) through a register ring:
    // 8 slots / distance 7 where a channel has a multiple of 8 k-steps (every layer of the QM8
    // model), like the forward's GEMM1 — with the 4-slot ring (distance 3: 3 x 4 MFMAs of cover per
    // tile) the loop ran at the L2 latency: 0.73 ms for the launch against 0.13 ms of MFMA time
    float4 ring[DEEP ? 8 : 4];
#pragma unroll
    for (int sl = 0; sl < (DEEP ? 7 : 3); ++sl) ring[sl] = wp[sl * 64];
    lds_cptr yrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) yrow[m] = (lds_cptr)&Xs[0][m][j][4 * hh];
    for (int s = 0; s < S; ++s) {
      f32x16 Z[MT];
      lds_cptr xq[MT];
      float4 acur[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        Z[m] = lnz::splat16(0.0f);
        xq[m] = yrow[m];
        acur[m] = lds_f4(xq[m]);
      }
      auto steps = [&](auto depth) {
        constexpr int RD = decltype(depth)::value;
#pragma unroll 1
        for (int q0 = 0; q0 < Q; q0 += RD) {
#pragma unroll
          for (int u4 = 0; u4 < RD; ++u4) {
            ring[(u4 + RD - 1) & (RD - 1)] = wp[(u4 + RD - 1) * 64];
            float4 anext[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) anext[m] = lds_f4(xq[m] + 8 * (u4 + 1));
            const float4 bv = ring[u4];
#pragma unroll
            for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].x, bv.x, Z[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].y, bv.y, Z[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].z, bv.z, Z[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) Z[m] = lnz::mfma32(acur[m].w, bv.w, Z[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acur[m] = anext[m];
            // one load between pairs of MFMAs (see forward_half's GEMM1)
            if (MT == 2) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
          }
#pragma unroll
          for (int m = 0; m < MT; ++m) xq[m] += 8 * RD;
          wp += RD * 64;
        }
      };
      steps(std::integral_constant<int, DEEP ? 8 : 4>{});
      // partial[rho] over the 16 lanes of a DPP row (four row-local DPP adds — the __shfl_xor
      // butterfly over 32 lanes compiled to five ds_bpermute_b32 per value, 276 in the kernel, and
      // made the LDS crossbar the launch's bottleneck); the dY buffer (free since the projection)
      // collects [wave][lane row within the half][s][rho] per tile
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float* red = &Xs[1][m][0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = row16_sum(Pb[m][r] * Z[m][r]);
          if ((lane & 15) == 0)
            red[((wave * 2 + ((lane >> 4) & 1)) * S + s) * 32 + lnz::cd_row(r, hh)] = v;
        }
      }
    }
    __syncthreads();
    // ---- sum over the waves (fixed order) and store dG[la][mol][k][s]
    for (int idx = htid; idx < MT * S * 32; idx += 64 * NWV) {
      const int m = idx / (S * 32);
      const int rem = idx - m * S * 32;
      const int s = rem >> 5, rho = rem & 31;
      const TileDesc t = pick(td, m);
      const bool isA = rho < t.split;
      const int k = isA ? rho : rho - t.split;
      const int mol = isA ? t.ta : t.tb;
      if (mol >= 0 && k < K) {
        const float* red = &Xs[1][m][0][0];
        float v = 0.0f;
        for (int w = 0; w < 2 * NWV; ++w) v += red[(w * S + s) * 32 + rho];
        a.dgains[(((int64_t)la * B + mol) * K + k) * S + s] = v;
      }
    }
    __syncthreads();
  }
}

template <int NWV, int DEEP>
__global__ __launch_bounds__(128 * NWV) void lanczosnet_gain_grad_kernel(const lnz_forward_args) {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) float Xs[2][2][MOLS][32][PITCH];  // [half][buffer][tile]
  __shared__ __attribute__((aligned(16))) float Vm[2][MOLS][32][VPITCH];
  const int tid = threadIdx.x;
  const int W = a.plan ? *a.n_wg : (a.B + 3) / 4;
  if ((int)blockIdx.x >= W) return;
  const int half = __builtin_amdgcn_readfirstlane(tid / (64 * NWV));
  const int htid = tid - half * 64 * NWV;
  const int wave = __builtin_amdgcn_readfirstlane(htid >> 6);
  TileDesc td[MOLS];
  int nt = 0;
#pragma unroll
  for (int m = 0; m < MOLS; ++m) {
    td[m] = load_tile_desc(a, (int)blockIdx.x * 4 + 2 * half + m);
    nt += td[m].ta >= 0 ? 1 : 0;
  }
  if (nt == 2) {
    gain_grad_half<NWV, 2, DEEP>(a, td, Xs[half], Vm[half], htid, wave);
  } else if (nt == 1) {
    const TileDesc t1[1] = {td[0]};
    gain_grad_half<NWV, 1, DEEP>(a, t1, Xs[half], Vm[half], htid, wave);
  } else {
    for (int l = 0; l < 5 * a.num_layer; ++l) __syncthreads();  // five barriers per layer
  }
}

}  // namespace

namespace lnz {
bool forward16_eligible(const lnz_forward_args& a, int mode);         // conv_forward16.hip
int launch_forward16(const lnz_forward_args& a, int mode, hipStream_t s);
bool strip_forward_eligible(const lnz_forward_args& a, int mode);     // conv_strip.hip
int launch_strip_forward(const lnz_forward_args& a, int mode, hipStream_t s);
bool strip_messages_eligible(const lnz_forward_args& a);
int launch_strip_messages(const lnz_forward_args& a, hipStream_t s);
bool strip_gain_grad_eligible(const lnz_forward_args& a);
int launch_strip_gain_grad(const lnz_forward_args& a, hipStream_t s);
}

extern "C" int64_t lnz_forward_args_size(void) { return (int64_t)sizeof(lnz_forward_args); }

// The inference forward runs on 16 x 16 MFMA tiles with all eight waves on all of a workgroup's
// node tiles (conv_forward16.hip) where that kernel is built; LNZ_FORWARD16=0 keeps the 32 x 32
// kernel of this file for A/B runs (lanczosnet_amd/utils/flop_model.py reads the same variable).
static bool forward16_enabled() {  // (read per launch: tests and A/B runs switch it in-process)
  const char* e = getenv("LNZ_FORWARD16");
  return !e || atoi(e) != 0;
}

// LNZ_STRIPS=0: inference launches that carry a strip plan run on the 32-row tile plan all the
// same (A/B runs; lanczosnet_amd/utils/flop_model.py reads the same variable).
static bool strips_enabled() {
  const char* e = getenv("LNZ_STRIPS");
  return !e || atoi(e) != 0;
}

static int launch_conv(const lnz_forward_args& a, int mode, hipStream_t s, const char* who) {
  LNZ_REQUIRE(a.B > 0 && a.N > 0 && a.K > 0 && a.num_layer > 0, LNZ_EINVAL,
              "%s: bad sizes (B=%d N=%d K=%d L=%d)", who, a.B, a.N, a.K, a.num_layer);
  LNZ_REQUIRE(a.N <= LNZ_TILE, LNZ_ENOTSUP,
              "%s: N=%d > %d-node tile (multi-tile molecules not built yet)", who, a.N, LNZ_TILE);
  LNZ_REQUIRE(a.K <= 2 * KHMAX, LNZ_ENOTSUP, "%s: K=%d > %d", who, a.K, 2 * KHMAX);
  LNZ_REQUIRE(a.num_layer <= 16, LNZ_ENOTSUP, "%s: num_layer=%d > 16", who, a.num_layer);
  LNZ_REQUIRE(a.dhid == 64 || a.dhid == 128, LNZ_ENOTSUP, "%s: hidden width %d not in {64,128}",
              who, a.dhid);
  LNZ_REQUIRE(a.din0 > 0 && a.din0 % 32 == 0 && a.din0 <= 128, LNZ_ENOTSUP,
              "%s: input width %d must be a multiple of 32, <= 128 (zero-pad features and weight "
              "columns on the host)", who, a.din0);
  LNZ_REQUIRE(a.n_short >= 0 && a.n_short <= 8 && a.n_long >= 0 && a.n_edge >= 1 &&
                  a.n_short + a.n_long + a.n_edge <= LNZ_MAX_CHANNELS,
              LNZ_EINVAL, "%s: bad channel counts", who);
  LNZ_REQUIRE(a.mask && a.V && a.Lp, LNZ_EINVAL,
              "%s: null tensor pointer", who);
  LNZ_REQUIRE(a.n_long == 0 || a.G, LNZ_EINVAL, "%s: G missing", who);
  LNZ_REQUIRE(a.gemm_mode == 0 || a.gemm_mode == 1, LNZ_EINVAL, "%s: gemm_mode %d", who,
              a.gemm_mode);
  LNZ_REQUIRE(a.filter_kind == 0 || a.filter_kind == 1, LNZ_EINVAL, "%s: filter_kind %d", who,
              a.filter_kind);
  LNZ_REQUIRE(!a.plan || (a.n_wg && a.plan_wg_cap > 0), LNZ_EINVAL,
              "%s: plan without n_wg / plan_wg_cap", who);
  const bool dense_es = a.filter_kind == 1;  // dense K x K filters run in eigen space
  if (mode == 0) {
    LNZ_REQUIRE(a.dout >= 1 && a.dout <= 31, LNZ_ENOTSUP, "%s: output width %d not in 1..31", who,
                a.dout);
    LNZ_REQUIRE((a.node_feat && a.embedding && a.num_atom > 0) || a.node_feat_f, LNZ_EINVAL,
                "%s: need node_feat+embedding or node_feat_f", who);
    LNZ_REQUIRE(a.Wp && a.bias && a.Wp_head && a.bias_head && a.score, LNZ_EINVAL,
                "%s: null tensor pointer", who);
    LNZ_REQUIRE(!a.act_out || (a.gemm_mode == 0 && (a.filter_kind == 0 || dense_es)), LNZ_ENOTSUP,
                "%s: act_out needs gemm_mode 0 and diagonal gains or dense filters in eigen space",
                who);
    if (a.gemm_mode == 1) {
      LNZ_REQUIRE(lnz::strip_forward_eligible(a, 0), LNZ_ENOTSUP,
                  "%s: gemm_mode 1 (split-precision GEMM1) runs on the strip plan only: strips, hidden "
                  "width 128, input width 128, diagonal gains, no short-diffusion channels, <= 12 long and <= 32 "
                  "channels in all",
                  who);
      return lnz::launch_strip_forward(a, 0, s);
    }
    // (the training forward — act_out — has no 32 x 32-tile kernel any more: the switch that
    // selects those for A/B runs applies to inference launches only)
    const bool tiles16 = forward16_enabled() || a.act_out;
    if (tiles16 && strips_enabled() && lnz::strip_forward_eligible(a, 0))
      return lnz::launch_strip_forward(a, 0, s);
    if (tiles16 && lnz::forward16_eligible(a, 0)) return lnz::launch_forward16(a, 0, s);
    LNZ_REQUIRE(!a.act_out, LNZ_ENOTSUP,
                "%s: the activation store (training forward) is built on the 16 x 16-tile kernels: "
                "hidden width 128, input width 64 or 128, <= 12 long and <= 32 channels in all, "
                "K %% 4 == 0 for dense filters", who);
    if (getenv("LNZ_FORWARD16_VERBOSE"))
      fprintf(stderr, "lnz forward on 32x32 tiles: fk %d dense_es %d gemm %d dhid %d din0 %d short %d long %d "
              "edge %d K %d B %d L %d\n", a.filter_kind, (int)dense_es, a.gemm_mode, a.dhid, a.din0,
              a.n_short, a.n_long, a.n_edge, a.K, a.B, a.num_layer);
  } else {
    LNZ_REQUIRE(a.gemm_mode == 0 && (a.filter_kind == 0 || dense_es) && a.dhid == 128, LNZ_ENOTSUP,
                "%s: built for gemm_mode 0, hidden width 128, diagonal gains or dense filters in "
                "eigen space", who);
    LNZ_REQUIRE(a.act || a.num_layer == 1, LNZ_EINVAL, "%s: act missing", who);
    if (mode == 1) {
      LNZ_REQUIRE(a.Wp && a.dy && a.dx0 && a.din0 == a.dhid && a.bwd_din0 > 0 &&
                      a.bwd_din0 % 32 == 0 && a.bwd_din0 <= a.dhid,
                  LNZ_EINVAL, "%s: need Wp (transposed packs), dy, dx0, din0 == dhid, bwd_din0", who);
      LNZ_REQUIRE(!a.dy_compact || (a.row_off && a.dy_compact_rows > 0), LNZ_EINVAL,
                  "%s: dy_compact needs row_off and dy_compact_rows", who);
      if (strips_enabled() && lnz::strip_forward_eligible(a, 1))
        return lnz::launch_strip_forward(a, 1, s);
      if (lnz::forward16_eligible(a, 1)) return lnz::launch_forward16(a, 1, s);
      // (the 32 x 32-tile instantiations of this pass — up to 79 spilled registers — went in r05)
      LNZ_REQUIRE(false, LNZ_ENOTSUP,
                  "%s: built on the 16 x 16-tile kernels: <= 12 long and <= 32 channels in all, "
                  "bwd_din0 %% 16 == 0, K %% 4 == 0 for dense filters", who);
    } else {
      LNZ_REQUIRE(a.msg && a.msg_layer >= 0 && a.msg_layer < a.num_layer &&
                      (a.msg_layer > 0 || a.x0),
                  LNZ_EINVAL, "%s: need msg, msg_layer in range, x0 for layer 0", who);
      LNZ_REQUIRE(a.msg_layer == 0 || a.act, LNZ_EINVAL, "%s: act missing", who);
      // (the 32-row-tile instantiations of this pass — 204 and 598 spilled scalars, 21 spilled vector
      // registers — went in r05: every training batch carries a strip plan)
      LNZ_REQUIRE(lnz::strip_messages_eligible(a), LNZ_ENOTSUP,
                  "%s: built on the strip plan: strips, hidden width 128, input width %% 16 == 0 and <= 128, "
                  "<= 16 long scales, K %% 4 == 0 for dense filters", who);
      return lnz::launch_strip_messages(a, s);
    }
  }
  const int grid = a.plan ? a.plan_wg_cap : (a.B + 3) / 4;  // upper bound of the workgroup count
  // dynamic LDS: per-layer spectral gains of both halves, double buffered (filter_kind 0)
  const size_t gs_bytes = a.filter_kind == 0 ? (size_t)2 * 2 * MOLS * a.n_long * 32 * sizeof(float) : 0;
  LNZ_REQUIRE(gs_bytes <= 12288, LNZ_ENOTSUP,
              "%s: %d long-diffusion channels exceed the 12 whose gains fit in LDS next to the node "
              "tiles", who, a.n_long);
#define LNZ_LAUNCH_D(NWV_, KHT_, FK_, MODE_, DEEPK_)                                             \
  do {                                                                                           \
    auto kfn = lanczosnet_forward_kernel<NWV_, KHT_, FK_, MODE_, DEEPK_>;                        \
    lnz::note_kernel("lanczosnet_forward_kernel<%d,%d,%d,%d,%d>", NWV_, KHT_, FK_, MODE_, DEEPK_); \
    if (gs_bytes)                                                                                \
      LNZ_DYNAMIC_LDS(kfn, \
      gs_bytes, "conv_forward.hip");                                                  \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(128 * NWV_), gs_bytes, s, a);                       \
  } while (0)
#define LNZ_LAUNCH(NWV_, KHT_, FK_, MODE_) LNZ_LAUNCH_D(NWV_, KHT_, FK_, MODE_, -1)
  // diagonal gains (filter_kind 0) run in eigen space for any K <= 32: the KHT parameter only
  // sizes the dense-filter variant's register arrays.  The weight-ring depth is a template
  // constant wherever it is known on the host (forward_half DEEPK): every layer of a width-128
  // model with an input width that is a multiple of 64 takes the 8-slot ring in the forward
  // modes; the backward modes always take the 4-slot ring.
  const bool all_deep = a.dhid == 128 && a.din0 % 64 == 0;
  if (a.filter_kind == 0) {
    if (all_deep) LNZ_LAUNCH_D(4, 10, 0, 0, 1);
    else if (a.dhid == 128) LNZ_LAUNCH(4, 10, 0, 0);
    else LNZ_LAUNCH(2, 10, 0, 0);
  } else {
    // dense K x K filters (AdaLanczosNet) in eigen space: pair tiles, DD fragments as GEMM2 operand
    // 4-slot weight ring in every layer: the DD fragments are live across the channel's GEMM1
    // next to T, Z and out — the 8-slot ring does not fit in 256 registers beside them
    if (a.dhid == 128) LNZ_LAUNCH_D(4, 10, 2, 0, 0);
    else LNZ_LAUNCH_D(2, 10, 2, 0, 0);
  }
#undef LNZ_LAUNCH_D
#undef LNZ_LAUNCH
  return lnz::check_launch(who);
}

extern "C" int lnz_lanczosnet_forward(const lnz_forward_args* args, lnz_stream_t stream) {
  LNZ_REQUIRE(args, LNZ_EINVAL, "lnz_lanczosnet_forward: null args");
  return launch_conv(*args, 0, (hipStream_t)stream, "lnz_lanczosnet_forward");
}

extern "C" int lnz_lanczosnet_input_grad(const lnz_forward_args* args, lnz_stream_t stream) {
  LNZ_REQUIRE(args, LNZ_EINVAL, "lnz_lanczosnet_input_grad: null args");
  return launch_conv(*args, 1, (hipStream_t)stream, "lnz_lanczosnet_input_grad");
}

extern "C" int lnz_lanczosnet_gain_grad(const lnz_forward_args* args, lnz_stream_t stream) {
  LNZ_REQUIRE(args, LNZ_EINVAL, "lnz_lanczosnet_gain_grad: null args");
  const lnz_forward_args& a = *args;
  const char* who = "lnz_lanczosnet_gain_grad";
  LNZ_REQUIRE(a.B > 0 && a.N > 0 && a.N <= LNZ_TILE && a.K > 0 && a.K <= 2 * KHMAX &&
                  a.num_layer > 0 && a.num_layer <= 16,
              LNZ_EINVAL, "%s: bad sizes (B=%d N=%d K=%d L=%d)", who, a.B, a.N, a.K, a.num_layer);
  LNZ_REQUIRE(a.gemm_mode == 0 && a.filter_kind == 0 && a.dhid == 128 && a.n_long > 0 &&
                  a.n_long <= 16 && a.din0 > 0 && a.din0 % 32 == 0 && a.din0 <= 128,
              LNZ_ENOTSUP, "%s: built for gemm_mode 0, filter_kind 0, hidden width 128, "
              "1..16 long channels, input width a multiple of 32", who);
  LNZ_REQUIRE(a.mask && a.V && a.Wp && a.dy && a.x0 && a.dgains && (a.act || a.num_layer == 1),
              LNZ_EINVAL, "%s: null tensor pointer (mask, V, Wp, dy, x0, act, dgains)", who);
  LNZ_REQUIRE(!a.plan || (a.n_wg && a.plan_wg_cap > 0), LNZ_EINVAL,
              "%s: plan without n_wg / plan_wg_cap", who);
  // [wave][lane row][s][32] partial sums of a tile live in its 32 x PITCH dY buffer
  LNZ_REQUIRE(8 * a.n_long * 32 <= 32 * PITCH, LNZ_ENOTSUP, "%s: too many long channels", who);
  if (forward16_enabled() && strips_enabled() && lnz::strip_gain_grad_eligible(a))
    return lnz::launch_strip_gain_grad(a, (hipStream_t)stream);
  const int grid = a.plan ? a.plan_wg_cap : (a.B + 3) / 4;
  if (a.din0 % 64 == 0)
    hipLaunchKernelGGL((lanczosnet_gain_grad_kernel<4, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((lanczosnet_gain_grad_kernel<4, 0>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
  return lnz::check_launch(who);
}

extern "C" int lnz_lanczosnet_messages(const lnz_forward_args* args, lnz_stream_t stream) {
  LNZ_REQUIRE(args, LNZ_EINVAL, "lnz_lanczosnet_messages: null args");
  return launch_conv(*args, 2, (hipStream_t)stream, "lnz_lanczosnet_messages");
}
