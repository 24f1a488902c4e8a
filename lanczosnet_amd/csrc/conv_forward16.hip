// R7 + R9 + R10 on v_mfma_f32_16x16x4_f32: the inference forward of the LanczosNet model
// (diagonal spectral gains, no short-diffusion channels, hidden width 128) with ALL eight waves of
// a workgroup on ALL of its node tiles.
//
// conv_forward.hip splits a workgroup in two halves of four waves (32 output columns each) that
// own one or two tiles: a three-tile workgroup runs 2 | 1, the lone two-tile wave of a SIMD cannot
// fill the matrix pipe once its one-tile neighbour is through, and a weight fragment feeds 4 or 8
// MFMAs.  Here wave w owns output columns [16 w, 16 w + 16) of every tile (NT = 1..4 per
// workgroup, two 16-row subtiles each): a 16-k step of GEMM1 is ONE weight float4 per lane, 2 NT
// A fragments (ds_read_b128) and 8 NT MFMAs, every wave of the workgroup carries the same work,
// and the 32-cycle instruction keeps half the accumulator bytes per flop.  tools/mfma16_issue_probe.hip
// (whole chip, same loads): 32x32x2 halves 2|1 0.896 of the fp32 peak, 1|1 0.85; this step 0.93
// (NT = 3), 0.905 (NT = 2), 0.948 (NT = 4).
//
// Same algebra, packs and plan as conv_forward.hip (eigen-space long channels, node-space edge
// channels with the GEMM1 result chained into GEMM2 as B operand, epilogue that projects the next
// layer's Y from the C/D registers); only the fragment indexing differs:
//   lane = (j = lane & 15, kq = lane >> 4);  A operand: A[row j][k = kq];  B: B[k = kq][col j];
//   C/D register r: D[row 4 kq + r][col j].
//   weights   the 32x32x2 pack holds W_c[32 rt + jj][8 q + 4 hh + 0..3] at float4
//             (rt Gtot + c Q + q) 64 + 32 hh + jj; wave w (rt = w >> 1) reads, for its 16-k step q',
//             float4 (rt Gtot + c Q + 2 q' + (kq >> 1)) 64 + 32 (kq & 1) + 16 (w & 1) + j
//             = W_c[16 w + j][16 q' + 4 kq + 0..3]: MFMA x of the step contracts k = 16 q' + 4 kq + x,
//             so the A operand is X[row][16 q' + 4 kq + 0..3] — one ds_read_b128 (row pitch 136
//             floats: conflict free for the instruction's four 16-lane groups).
//   Laplacian the pack holds M[jl][8 g + 4 hh + 0..3] at float4 (mol n_edge + e) 256 + 64 g + 32 hh
//             + jl; the fragment of (row subtile I, node subtile J) is M[16 I + j][16 J + 4 kq + 0..3]:
//             g = 2 J + (kq >> 1), hh = kq & 1 — GEMM2 step r of (I, J) contracts node 16 J + 4 kq + r,
//             which is C/D register r of Z[J].
// Rows that belong to no molecule are computed like every other row (they stay finite: relu(bias));
// only CONTRACTIONS over a dead 16-row subtile are skipped (wave uniform).
#include "common.hpp"
#include "conv_tiles.hpp"
#include <type_traits>

#ifndef LNZ_F16_PRIO
#define LNZ_F16_PRIO 1
#endif

namespace {

constexpr int VP = 36;  // Ritz tile row pitch (floats)
typedef const __attribute__((address_space(3))) float* lds_cptr;
typedef const __attribute__((address_space(3))) f32x4* lds_c4ptr;

__device__ __forceinline__ f32x4 lds4(lds_cptr p) { return *(lds_c4ptr)p; }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }

// element m of a small register array (explicit selects: a loop over the array, even a fully
// unrolled one, makes hipcc keep the array in scratch memory and index it)
template <int NT>
__device__ __forceinline__ int pick_int(const int (&v)[NT], int m) {
  if constexpr (NT == 1) return v[0];
  else if constexpr (NT == 2) return m == 1 ? v[1] : v[0];
  else if constexpr (NT == 3) return m == 2 ? v[2] : (m == 1 ? v[1] : v[0]);
  else return m == 3 ? v[3] : (m == 2 ? v[2] : (m == 1 ? v[1] : v[0]));
}
template <int NT>
__device__ __forceinline__ TileDesc pick_tile(const TileDesc (&td)[NT], int m) {
  int ta[NT], tb[NT], sp[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    ta[i] = td[i].ta;
    tb[i] = td[i].tb;
    sp[i] = td[i].split;
  }
  TileDesc t;
  t.ta = pick_int(ta, m);
  t.tb = pick_int(tb, m);
  t.split = pick_int(sp, m);
  return t;
}

// bit I: rows 16 I .. 16 I + 15 of the tile hold a row of molecule A ([0, nA)) or B ([split, split + nB))
__device__ __forceinline__ int live16(int nA, int nB, int split) {
  const int g = row_group_mask(nA, nB, split);
  return ((g & 3) ? 1 : 0) | ((g & 12) ? 2 : 0);
}

// MODE 0 = forward (a.act_out: every layer's activations are stored as well — the training forward);
// MODE 1 = input-gradient pass (lnz_lanczosnet_input_grad): the same chained GEMMs run on dY with the
//          per-channel transposed weight packs, kernel iteration l = conv layer num_layer - 1 - l,
//          the epilogue masks with the stored activation instead of bias + ReLU and writes dY_{la-1}
//          (last iteration: dX_0, bwd_din0 columns — the waves beyond them only keep the barriers).
// FK  0 = diagonal spectral gains on Ritz vectors (LanczosNet): G [L][B][n_long][K], staged in LDS,
//          a long channel's C/D rows (= eigen slots) are scaled into T on the VALU;
//     2 = dense K x K filters on the Lanczos basis (AdaLanczosNet, model/ada_lanczos_net.py:280-281):
//          G [L][B][n_long][K][K]; T += DD_s Z_s is one more MFMA chain with the filter's fragments
//          as A operand (block diagonal per tile like the Laplacians), fetched under the channel's
//          own GEMM1.
// Short-diffusion channels c < n_short (powers p_c of the first Laplacian, model/lanczos_net.py:
// 172-174) run in node space: Z <- L_0^(p-1) Z in registers, then GEMM2 with L_0.
// SHORT: the model has short-diffusion channels (a template constant: their block would otherwise
// set the register allocation of the kernels that never run it).
template <int NT, int P, int MODE, int FK, bool SHORT>
__device__ __forceinline__ void forward16(KArgs& a, const TileDesc (&td)[NT], float* lds,
                                          const int tid, const int wave, const int part_slot = 0) {
  constexpr bool FWD = MODE == 0;
  constexpr bool DIAG = FK == 0;
#ifdef LNZ_F16_PHASES
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_all = clock64(), _t0 = t_all;
#define LNZ_PH(i) { const long long _t1 = clock64(); ph[i] += _t1 - _t0; _t0 = _t1; }
#else
#define LNZ_PH(i)
#endif
  const int lane = tid & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int N = a.N, K = a.K, B = a.B;
  const int ns = SHORT ? a.n_short : 0, nl = a.n_long, ne = a.n_edge;
  const int C = ns + nl + ne;
  constexpr int TILE = 32 * P;  // floats per tile buffer
  float* Xs = lds;                               // [2][NT][32][P]
  float* Vm = lds + 2 * NT * TILE;               // [NT][32][VP]   [node row][slot row]
  float* Gs = Vm + NT * 32 * VP;                 // [2][NT][nl][32]

  // ---- embedding gather / float features (model/lanczos_net.py:154, lanczos_net_general.py:156)
  {
    const int d4 = a.din0 >> 2;
    for (int idx = tid; idx < NT * 32 * d4; idx += 512) {
      const int m = idx / (32 * d4);
      const int rem = idx - m * 32 * d4;
      const int row = rem / d4, c4 = rem - row * d4;
      const TileDesc t = pick_tile(td, m);
      const bool first = row < t.split;
      const int mol = first ? t.ta : t.tb;
      const int lrow = first ? row : row - t.split;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!FWD) {  // the incoming gradient dY of the last conv layer
        if (mol >= 0)
          v = reinterpret_cast<const float4*>(
              a.dy + (((int64_t)(a.num_layer - 1) * B + mol) * 32 + lrow) * 128)[c4];
      } else if (lrow < N && mol >= 0) {
        if (a.node_feat) {
          int64_t id = a.node_feat[(int64_t)mol * N + lrow];
          id = id < 0 ? 0 : (id >= a.num_atom ? a.num_atom - 1 : id);
          v = reinterpret_cast<const float4*>(a.embedding + id * a.din0)[c4];
        } else {
          v = reinterpret_cast<const float4*>(a.node_feat_f + ((int64_t)mol * N + lrow) * a.din0)[c4];
        }
      }
      *reinterpret_cast<float4*>(&Xs[m * TILE + row * P + 4 * c4]) = v;
    }
  }

  // ---- extents, live 16-row subtiles (node rows / eigen-slot rows), identity channels
  int nA[NT], nB[NT], rl[NT], sl[NT], idm[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    tile_extents(a, td[m], lane, nA[m], nB[m]);
    rl[m] = live16(nA[m], nB[m], td[m].split);
    sl[m] = live16(nA[m] < K ? nA[m] : K, nB[m] < K ? nB[m] : K, td[m].split);
    unsigned v = (FWD && a.ident) ? a.ident[td[m].ta] : 0u;
    if (FWD && a.ident && td[m].tb >= 0) v &= a.ident[td[m].tb];
    idm[m] = __builtin_amdgcn_readfirstlane((int)v);
  }

  // ---- Ritz tiles [node row][slot row], block diagonal
  for (int idx = tid; idx < NT * 32 * 32; idx += 512) {
    const int m = idx >> 10, jj = (idx >> 5) & 31, rho = idx & 31;
    Vm[(m * 32 + jj) * VP + rho] = ritz_tile_elem(a, pick_tile(td, m), jj, rho);
  }
  // ---- spectral gains of a layer by slot row: Gs[l & 1][m][s][rho] (zero for unused slots).
  //      The loads of layer l + 1 are issued at the start of layer l and land in registers under its
  //      GEMMs; they go to LDS in front of the layer's last barrier (G is cold: staged in one go,
  //      every wave would sit out an HBM round trip per layer).
  constexpr int GREG = 3;  // NT * nl * 32 <= 512 * GREG  (nl <= 12 at NT = 4)
  float greg[GREG];
  // branch-free (raw buffer loads, offset beyond the buffer = 0.0): every vector load of the layer
  // loop is issued unconditionally, so the compiler's s_waitcnt vmcnt counts are exact and a wait
  // for a weight-ring slot never drains younger loads
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.G), 0, nl > 0 ? a.num_layer * B * nl * K * (DIAG ? 1 : K) * 4 : 0, 0x00020000);
  unsigned goff[GREG];  // byte offset of this thread's element(s) in layer 0
#pragma unroll
  for (int u = 0; u < GREG; ++u) {
    const int idx = tid + 512 * u;
    goff[u] = OOB;
    if (DIAG && idx < NT * nl * 32) {
      const int m = idx / (nl * 32);
      const int rem = idx - m * nl * 32;
      const int sc = rem >> 5, rho = rem & 31;
      const TileDesc t = pick_tile(td, m);
      const bool isA = rho < t.split;
      const int k = isA ? rho : rho - t.split;
      const bool ok = k < K && k < (isA ? pick_int(nA, m) : pick_int(nB, m));
      const int mol = isA ? t.ta : t.tb;
      if (ok) goff[u] = (unsigned)(((mol * nl + sc) * K + k) * 4);
    }
  }
  auto load_gains = [&](int l) {
#pragma unroll
    for (int u = 0; u < GREG; ++u)
      greg[u] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, goff[u], l * B * nl * K * 4, 0));
  };
  auto store_gains = [&](int l) {
    float* dst = Gs + (l & 1) * NT * nl * 32;
#pragma unroll
    for (int u = 0; u < GREG; ++u) {
      const int idx = tid + 512 * u;
      if (idx < NT * nl * 32) dst[idx] = greg[u];
    }
  };
  if (DIAG && nl > 0) {
    load_gains(FWD ? 0 : a.num_layer - 1);
    store_gains(FWD ? 0 : a.num_layer - 1);
  }

  // ---- per-lane addressing of the packed Laplacian: for (tile m, row subtile I) this lane's row
  //      16 I + j belongs to molecule A (local row = tile row, its column groups 0 .. g0 - 1) or B
  //      (local row = tile row - split, its groups follow A's)
  const int rt = wave >> 1;
  const int wlane = 64 * (kq >> 1) + 32 * (kq & 1) + 16 * (wave & 1) + j;  // float4 within a 16-k step
  unsigned loff[NT][2][2];  // byte offset of fragment (I, J) of edge type 0, OOB off the diagonal blocks
  const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.Lp), 0, B * ne * 4096, 0x00020000);
#pragma unroll
  for (int m = 0; m < NT; ++m) {
#pragma unroll
    for (int I = 0; I < 2; ++I) {
      const int row = 16 * I + j;
      const bool rowA = row < td[m].split;
      const int g0 = td[m].split >> 3;
      const int mol = rowA ? td[m].ta : td[m].tb;
      const int jl = rowA ? row : row - td[m].split;
#pragma unroll
      for (int J = 0; J < 2; ++J) {
        const int g = 2 * J + (kq >> 1);
        const bool ok = mol >= 0 && (rowA ? g < g0 : g >= g0);
        const int gl = rowA ? g : g - g0;
        loff[m][I][J] = ok ? (unsigned)((mol * ne * 256 + gl * 64 + 32 * (kq & 1) + jl) * 16) : OOB;
      }
    }
  }
  // FK = 2: fragment (I, J) of a dense filter, DDtile[16 I + j][16 J + 4 kq + 0..3]: slot row 16 I + j
  // is slot kr of molecule A (rows below the split) or B, its columns molecule-local; zero off the
  // diagonal blocks and beyond K (K % 4 == 0: a float4 never straddles K)
  unsigned doff[DIAG ? 1 : NT][2][2];
  if (!DIAG) {
#pragma unroll
    for (int m = 0; m < NT; ++m) {
#pragma unroll
      for (int I = 0; I < 2; ++I) {
        const int rho = 16 * I + j;
        const bool rowA = rho < td[m].split;
        const int mol = rowA ? td[m].ta : td[m].tb;
        const int kr = rowA ? rho : rho - td[m].split;
#pragma unroll
        for (int J = 0; J < 2; ++J) {
          const int c0 = 16 * J + 4 * kq;
          const int k0 = c0 - (rowA ? 0 : td[m].split);
          const bool ok = mol >= 0 && kr < K && (rowA ? c0 < td[m].split : c0 >= td[m].split) && k0 < K;
          doff[DIAG ? 0 : m][I][J] = ok ? (unsigned)((((mol * nl) * K + kr) * K + k0) * 4) : OOB;
        }
      }
    }
  }
  __syncthreads();

  LNZ_PH(7)  // prologue
  const lds_cptr xlane = (lds_cptr)(Xs + j * P + 4 * kq);  // this lane's A row of tile 0, subtile 0
  int cur = 0;
  for (int l = 0; l < a.num_layer; ++l) {
    const int la = FWD ? l : a.num_layer - 1 - l;  // conv layer of this iteration
    const int lg = FWD ? l + 1 : la - 1;            // layer whose gains are staged under it
    const bool more = l + 1 < a.num_layer;
    const int din = l == 0 ? a.din0 : 128;
    const int Q = din >> 3, Q16 = din >> 4;
    const int Gtot = C * Q;
    const float4* __restrict__ Wl = reinterpret_cast<const float4*>(a.Wp + a.w_off[l]);
    const float* gsl = Gs + (la & 1) * NT * nl * 32;
    const int nxt = cur ^ 1;
    // width this iteration produces: waves beyond it only keep the barriers
    const int wout = FWD ? 128 : (la == 0 ? a.bwd_din0 : 128);
    const bool active = FWD || 16 * wave < wout;

    // weight stream of this wave: contiguous over the layer's channels, 128 float4 per 16-k step,
    // 4-slot register ring (prefetch distance 3 steps = 24 NT MFMAs)
    const float4* __restrict__ wp = Wl + (int64_t)rt * Gtot * 64 + wlane;
    float4 ring[4];
    if (active) {
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) ring[s3] = wp[s3 * 128];
    }
    // (behind the ring prime: vector loads return in order, the first steps must not wait for G)
    if (DIAG && nl > 0 && more) load_gains(lg);

    f32x4 out[NT][2];
    {
      const float bv = FWD ? (a.bias + a.b_off[l])[16 * wave + j] : 0.0f;
#pragma unroll
      for (int m = 0; m < NT; ++m) out[m][0] = out[m][1] = splat4(bv);
    }

    // ---------------- first layer: Y = V^T X from LDS into the other buffer ----------------
    if (nl > 0 && l == 0) {
      if (16 * wave < din) {
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          f32x4 Y[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
          for (int J = 0; J < 2; ++J) {
            if ((rl[m] >> J) & 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int node = 16 * J + 4 * kq + r;
                const float xb = Xs[cur * NT * TILE + m * TILE + node * P + 16 * wave + j];
#pragma unroll
                for (int I = 0; I < 2; ++I)
                  Y[I] = mfma16(Vm[(m * 32 + node) * VP + 16 * I + j], xb, Y[I]);
              }
            }
          }
#pragma unroll
          for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Xs[nxt * NT * TILE + m * TILE + (16 * I + 4 * kq + r) * P + 16 * wave + j] = Y[I][r];
        }
      }
      __syncthreads();
    }

    LNZ_PH(0)  // layer head: gains staging, ring prime, first-layer projection
    // ---------------- GEMM1 of one channel: Zc[m][I] = A rows (X or Y) x W_c^T ----------------
    f32x4 Z[NT][2], acur[NT][2];
    auto load_first = [&](lds_cptr x0) {
#pragma unroll
      for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int I = 0; I < 2; ++I) acur[m][I] = lds4(x0 + (m * 32 + 16 * I) * P);
    };
    // four steps; `wrap`: the A prefetch of the last one fetches k = 0 again — the first fragments
    // of the NEXT channel (channels of a block read the same rows)
    auto steps4 = [&](f32x4 (&Zc)[NT][2], lds_cptr xb, lds_cptr x0, auto wrap) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#ifndef LNZ_F16_NO_WLOAD  // (experiment: the launch without its weight stream)
        ring[(u + 3) & 3] = wp[(u + 3) * 128];
#endif
        f32x4 anext[NT][2];
        const lds_cptr xn = (decltype(wrap)::value && u == 3) ? x0 : xb + 16 * (u + 1);
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
          for (int I = 0; I < 2; ++I) anext[m][I] = lds4(xn + (m * 32 + 16 * I) * P);
        const float4 bv = ring[u];
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
          for (int I = 0; I < 2; ++I) Zc[m][I] = mfma16(acur[m][I][0], bv.x, Zc[m][I]);
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
          for (int I = 0; I < 2; ++I) Zc[m][I] = mfma16(acur[m][I][1], bv.y, Zc[m][I]);
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
          for (int I = 0; I < 2; ++I) Zc[m][I] = mfma16(acur[m][I][2], bv.z, Zc[m][I]);
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
          for (int I = 0; I < 2; ++I) Zc[m][I] = mfma16(acur[m][I][3], bv.w, Zc[m][I]);
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
          for (int I = 0; I < 2; ++I) acur[m][I] = anext[m][I];
        // issue order: one LDS read behind every fourth MFMA, the ring load in the last gap
#pragma unroll
        for (int g = 0; g < 2 * NT; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (g == 2 * NT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
      wp += 4 * 128;
    };
    [[maybe_unused]] int chan = 0;  // channels done in this layer
    auto gemm1 = [&](f32x4 (&Zc)[NT][2], lds_cptr x0, auto&& before_last) {
#if LNZ_F16_PRIO == 1
      // the two waves of a SIMD (w, w + 4) take turns at the head of the matrix pipe: with a fixed
      // order the older wave runs every channel at full rate and the younger one finishes the layer
      // alone, behind the barrier
      if (((chan ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      ++chan;
#endif
#pragma unroll
      for (int m = 0; m < NT; ++m) Zc[m][0] = Zc[m][1] = splat4(0.f);
      lds_cptr xb = x0;
#pragma unroll 1
      for (int q0 = 4; q0 < Q16; q0 += 4) {
        steps4(Zc, xb, x0, std::false_type{});
        xb += 64;
      }
      before_last();
      steps4(Zc, xb, x0, std::true_type{});
    };

    // Laplacian fragments of one edge type for every (tile, I, J): they land under the last four
    // steps of the channel's own GEMM1 (identity channels of a tile: offsets beyond the buffer, no
    // memory traffic).  The same registers hold a dense filter's fragments in the long block.
    f32x4 mop[NT][2][2];
    auto fetch = [&](int e, bool use_ident) {
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const unsigned skip = (use_ident && ((idm[m] >> e) & 1)) ? OOB : 0u;
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
          for (int J = 0; J < 2; ++J)
            mop[m][I][J] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(l_rsrc, loff[m][I][J] | skip, e * 4096, 0));
      }
    };
    // acc[m][I] += M[I][J] Zp[m][J] over the live node subtiles J (GEMM2 / one power of L_0)
    auto apply_m = [&](f32x4 (&acc)[NT][2], const f32x4 (&Zp)[NT][2], int m) {
#pragma unroll
      for (int J = 0; J < 2; ++J) {
        if ((rl[m] >> J) & 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int I = 0; I < 2; ++I)
              acc[m][I] = mfma16(mop[m][I][J][r], Zp[m][J][r], acc[m][I]);
        }
      }
    };

    // ---------------- short-diffusion channels: out += L_0^p (X W_c^T) ----------------
    if (SHORT && active) {
      const lds_cptr x0 = xlane + cur * NT * TILE;
      load_first(x0);
      for (int c = 0; c < ns; ++c) {
        gemm1(Z, x0, [&] { fetch(0, false); });
        const int p = a.short_dist[c];
        for (int rep = 1; rep < p; ++rep) {
#pragma unroll
          for (int m = 0; m < NT; ++m) {  // tile by tile: one pair of temporaries
            f32x4 Zn[NT][2];
            Zn[m][0] = Zn[m][1] = splat4(0.f);
            apply_m(Zn, Z, m);
            Z[m][0] = Zn[m][0];
            Z[m][1] = Zn[m][1];
          }
        }
#pragma unroll
        for (int m = 0; m < NT; ++m) apply_m(out, Z, m);
      }
    }

    // ---------------- eigen-space block: out += V [ sum_s F_s (Y W_s^T) ] ----------------
    //   F_s = diag(g_s) (FK 0) or the dense filter DD_s (FK 2)
    if (nl > 0 && active) {
      const lds_cptr y0 = xlane + nxt * NT * TILE;
      load_first(y0);
      f32x4 T[NT][2];
#pragma unroll
      for (int m = 0; m < NT; ++m) T[m][0] = T[m][1] = splat4(0.f);
      // (deferring a channel's contribution to T into the NEXT channel's GEMM1, with two
      // accumulator sets, so that it does not wait for the channel's last MFMAs to retire:
      // measured, no gain — the matrix pipe is shared with the SIMD's other wave, which fills the gap)
      for (int s = 0; s < nl; ++s) {
        if constexpr (DIAG) {
          gemm1(Z, y0, [] {});
#pragma unroll
          for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int I = 0; I < 2; ++I) {
              const f32x4 gv = *reinterpret_cast<const f32x4*>(gsl + (m * nl + s) * 32 + 16 * I + 4 * kq);
              T[m][I] += gv * Z[m][I];
            }
        } else {
          gemm1(Z, y0, [&] {
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
              for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
                  mop[m][I][J] = __builtin_bit_cast(
                      f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 g_rsrc, doff[DIAG ? 0 : m][I][J], (la * B * nl + s) * K * K * 4, 0));
          });
#pragma unroll
          for (int m = 0; m < NT; ++m) {
#pragma unroll
            for (int J = 0; J < 2; ++J) {
              if ((sl[m] >> J) & 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                  for (int I = 0; I < 2; ++I)
                    T[m][I] = mfma16(mop[m][I][J][r], Z[m][J][r], T[m][I]);
              }
            }
          }
        }
      }
      LNZ_PH(1)
      // lift back: out[I] += V[rows of I][slots of J] T[J]
#pragma unroll
      for (int m = 0; m < NT; ++m) {
#pragma unroll
        for (int J = 0; J < 2; ++J) {
          if ((sl[m] >> J) & 1) {
            f32x4 v[2];
#pragma unroll
            for (int I = 0; I < 2; ++I)
              v[I] = *reinterpret_cast<const f32x4*>(&Vm[(m * 32 + 16 * I + j) * VP + 16 * J + 4 * kq]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int I = 0; I < 2; ++I) out[m][I] = mfma16(v[I][r], T[m][J][r], out[m][I]);
          }
        }
      }
    }

    LNZ_PH(3)  // lift
    // ---------------- node-space block: out += M_e (X W_e^T) per edge type ----------------
    if (active) {
      const lds_cptr x0 = xlane + cur * NT * TILE;
      load_first(x0);
      for (int e = 0; e < ne; ++e) {
        gemm1(Z, x0, [&] { fetch(e, true); });
        LNZ_PH(4)
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          if ((idm[m] >> e) & 1) {  // identity on the tile's molecules: out += Z
            out[m][0] += Z[m][0];
            out[m][1] += Z[m][1];
          } else {
            apply_m(out, Z, m);
          }
        }
        LNZ_PH(5)
      }
    }

    // ---------------- epilogue: X' where Y was, Y' = V^T X' where X was ----------
    //   forward: X' = relu(out) (+ the activation store training asks for)
    //   MODE 1:  dY_{la-1} = out * [X_la > 0] -> LDS and dy[la-1]; the last iteration writes dX_0
    if (DIAG && nl > 0 && more) store_gains(lg);  // (buffer last read two iterations ago)
    if (nl > 0) __syncthreads();  // every wave is through with X and Y
    if (active) {
      const int col = 16 * wave + j;
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        // (the tile's ids pass through an empty asm: row addresses are recomputed per layer instead
        // of being hoisted out of the layer loop, where they would hold registers across the GEMMs)
        int t_sp = td[m].split, t_a = td[m].ta, t_b = td[m].tb;
        asm volatile("" : "+s"(t_sp), "+s"(t_a), "+s"(t_b));
#pragma unroll
        for (int I = 0; I < 2; ++I) {
          // rows 16 I + 4 kq + r of the tile: one owner (the split row is a multiple of 8)
          const int row0 = 16 * I + 4 * kq;
          const bool first = row0 < t_sp;
          const int mol = first ? t_a : t_b;
          const int64_t rowbase = (int64_t)(mol >= 0 ? mol : 0) * 32 + (row0 - (first ? 0 : t_sp));
          f32x4 v = out[m][I];
          if (FWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            if (a.act_out && mol >= 0) {
              float* p = a.act_out + ((int64_t)l * B * 32 + rowbase) * 128 + col;
#pragma unroll
              for (int r = 0; r < 4; ++r) p[r * 128] = v[r];
            }
          } else if (la > 0) {
            const int64_t at = ((int64_t)(la - 1) * B * 32 + rowbase) * 128 + col;
            const float* xa = a.act + at;
            float* dyp = a.dy + at;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = (mol >= 0 && xa[r * 128] > 0.0f) ? v[r] : 0.0f;
              if (mol >= 0) dyp[r * 128] = v[r];
            }
          } else if (mol >= 0) {
            float* p = a.dx0 + rowbase * a.bwd_din0 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r * a.bwd_din0] = v[r];
          }
          out[m][I] = v;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Xs[nxt * NT * TILE + m * TILE + (row0 + r) * P + col] = v[r];
        }
        if (nl > 0 && more) {
          f32x4 Y[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
          for (int J = 0; J < 2; ++J) {
            if ((rl[m] >> J) & 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int I = 0; I < 2; ++I)
                  Y[I] = mfma16(Vm[(m * 32 + 16 * J + 4 * kq + r) * VP + 16 * I + j], out[m][J][r], Y[I]);
            }
          }
#pragma unroll
          for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Xs[cur * NT * TILE + m * TILE + (16 * I + 4 * kq + r) * P + col] = Y[I][r];
        }
      }
      if (MODE == 1 && la > 0 && (a.dy_compact || a.dbias_part)) {
        // What the weight / bias gradients of conv layer la - 1 need: dY_{la-1} in the COMPACT row
        // numbering of the message matrix (real nodes only) and this workgroup's column sums (rows
        // of padded nodes and of unowned tile rows are zero) — one writer per (workgroup, layer,
        // column): entry 2 * blockIdx.x + part_slot of dbias_part (the odd entry only where a
        // workgroup runs its four tiles as two passes; it stays zero otherwise).
        float colsum = 0.0f;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          int t_sp = td[m].split, t_a = td[m].ta, t_b = td[m].tb;
          asm volatile("" : "+s"(t_sp), "+s"(t_a), "+s"(t_b));
#pragma unroll
          for (int I = 0; I < 2; ++I) {
            const int row0 = 16 * I + 4 * kq;
            const bool first = row0 < t_sp;
            const int mol = first ? t_a : t_b;
            const int lrow0 = row0 - (first ? 0 : t_sp);
            const int nmol = first ? nA[m] : nB[m];
            const f32x4 v = out[m][I];
            colsum += (v[0] + v[1]) + (v[2] + v[3]);
            if (a.dy_compact && mol >= 0) {
              float* dc = a.dy_compact +
                          ((int64_t)(la - 1) * a.dy_compact_rows + a.row_off[mol] + lrow0) * 128 + col;
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (lrow0 + r < nmol) dc[r * 128] = v[r];
            }
          }
        }
        if (a.dbias_part) {
          colsum += __shfl_xor(colsum, 16, 64);
          colsum += __shfl_xor(colsum, 32, 64);
          if (kq == 0)
            a.dbias_part[(((int64_t)blockIdx.x * 2 + part_slot) * a.num_layer + (la - 1)) * 128 + col] = colsum;
        }
      }
    }
    __syncthreads();
    cur = nxt;
    LNZ_PH(6)
  }
#ifdef LNZ_F16_PHASES
  if (a.state_out && lane == 0 && blockIdx.x < 8) {
    float* d = a.state_out + ((int64_t)B * 32 * 128) + (blockIdx.x * 8 + wave) * 16;
    for (int i = 0; i < 8; ++i) d[i] = (float)ph[i];
    d[8] = (float)(clock64() - t_all);
    d[9] = (float)NT;
  }
#endif

  if (!FWD) return;

  // ---- optional debug/test output of the final node state (rows of the tile each molecule owns)
  if (a.state_out) {
    for (int idx = tid; idx < NT * 32 * 128; idx += 512) {
      const int m = idx >> 12, row = (idx >> 7) & 31, col = idx & 127;
      const TileDesc t = pick_tile(td, m);
      const bool first = row < t.split;
      const int mol = first ? t.ta : t.tb;
      const int lrow = first ? row : row - t.split;
      if (mol >= 0)
        a.state_out[((int64_t)mol * 32 + lrow) * 128 + col] = Xs[cur * NT * TILE + m * TILE + row * P + col];
    }
  }

  // ---- head (model/lanczos_net.py:185-194) on the 32x32x2 instruction, wave m = tile m: one
  //      32-column tile = [W_o ; w_a ; 0]; the gate logit is column dout of the same row
  if (wave < NT) {
    const int m = wave;
    const TileDesc t = pick_tile(td, m);
    const int jj = lane & 31, hh = lane >> 5;
    const int Pd = a.dout;
    f32x16 acc = lnz::splat16(a.bias_head[jj]);
    const float4* wh = reinterpret_cast<const float4*>(a.Wp_head) + lane;
    const float* xr = &Xs[cur * NT * TILE + m * TILE + jj * P + 4 * hh];
#pragma unroll 2
    for (int q = 0; q < 16; ++q) {
      const float4 av = *reinterpret_cast<const float4*>(xr + 8 * q);
      const float4 bv = wh[q * 64];
      acc = lnz::mfma32(av.x, bv.x, acc);
      acc = lnz::mfma32(av.y, bv.y, acc);
      acc = lnz::mfma32(av.z, bv.z, acc);
      acc = lnz::mfma32(av.w, bv.w, acc);
    }
    const bool pr = t.tb >= 0;
    const int64_t mol0 = t.ta, mol1 = pr ? t.tb : t.ta;
    const int g0 = t.split >> 3;
    float sum0 = 0.0f, sum1 = 0.0f, cnt0 = 0.0f, cnt1 = 0.0f;
    const int src = 32 * hh + Pd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float logit = __shfl(acc[r], src, 64);
      const float gate = 1.0f / (1.0f + __expf(-logit));
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
    if (hh == 0 && jj < Pd) {
      a.score[mol0 * Pd + jj] = sum0 / cnt0;
      if (pr) a.score[mol1 * Pd + jj] = sum1 / cnt1;
    }
  }
}

// LDS floats of a workgroup with NT tiles (row pitch P)
constexpr int lds_floats(int NT, int P, int nl) {
  return 2 * NT * 32 * P + NT * 32 * VP + 2 * NT * nl * 32;
}

// One workgroup = 8 waves on the 1..4 node tiles of its plan entry.  Three tiles or fewer: row
// pitch 136 (conflict-free A fragments); four tiles only fit in 160 KB at pitch 132 (one of the
// instruction's four lane groups then takes a 2-way conflict).
template <int MODE, int FK, bool SHORT>
__global__ __launch_bounds__(512) void lanczosnet_forward16_kernel(const lnz_forward_args) {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float lds16[];
  const int W = a.plan ? *a.n_wg : (a.B + 3) / 4;
  if ((int)blockIdx.x >= W) return;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the used slots of the entry, compacted (slot 0 is always used; scalar selects — a tile array
  // indexed by a running count would live in scratch memory)
  const TileDesc s0 = load_tile_desc(a, (int)blockIdx.x * 4 + 0);
  const TileDesc s1 = load_tile_desc(a, (int)blockIdx.x * 4 + 1);
  const TileDesc s2 = load_tile_desc(a, (int)blockIdx.x * 4 + 2);
  const TileDesc s3 = load_tile_desc(a, (int)blockIdx.x * 4 + 3);
  const bool u1 = s1.ta >= 0, u2 = s2.ta >= 0, u3 = s3.ta >= 0;
  const int nt = (s0.ta >= 0 ? 1 : 0) + (u1 ? 1 : 0) + (u2 ? 1 : 0) + (u3 ? 1 : 0);
  const TileDesc c1 = u1 ? s1 : (u2 ? s2 : s3);   // first used slot behind slot 0
  const TileDesc c2 = (u1 && u2) ? s2 : s3;       // second
#ifndef LNZ_F16_ONLY_NT  // (register / ISA studies of one variant: -DLNZ_F16_ONLY_NT=3)
#define LNZ_F16_ONLY_NT 0
#endif
  constexpr int ONLY = LNZ_F16_ONLY_NT;
  if (nt == 3 && (ONLY == 0 || ONLY == 3)) {
    const TileDesc t3[3] = {s0, c1, c2};
    forward16<3, 136, MODE, FK, SHORT>(a, t3, lds16, tid, wave);
  } else if (nt == 2 && (ONLY == 0 || ONLY == 2)) {
    const TileDesc t2[2] = {s0, c1};
    forward16<2, 136, MODE, FK, SHORT>(a, t2, lds16, tid, wave);
  } else if (nt == 4 && (ONLY == 0 || ONLY == 4)) {
    if constexpr (FK == 0) {
      const TileDesc t4[4] = {s0, s1, s2, s3};
      forward16<4, 132, MODE, FK, SHORT>(a, t4, lds16, tid, wave);
    } else {
      // dense filters: four tiles next to the filter fragments do not fit in 256 registers —
      // two passes of two tiles (batches beyond three tiles per CU only)
      const TileDesc ta[2] = {s0, s1}, tb[2] = {s2, s3};
      forward16<2, 136, MODE, FK, SHORT>(a, ta, lds16, tid, wave, 0);
      __syncthreads();
      forward16<2, 136, MODE, FK, SHORT>(a, tb, lds16, tid, wave, 1);
    }
  } else if (nt == 1 && (ONLY == 0 || ONLY == 1)) {
    const TileDesc t1[1] = {s0};
    forward16<1, 136, MODE, FK, SHORT>(a, t1, lds16, tid, wave);
  }
}

}  // namespace

namespace lnz {

// The forward (mode 0, with or without the activation store) and the input-gradient pass (mode 1)
// on 16 x 16 tiles where they are built (see the file comment): diagonal gains (filter_kind 0) and
// dense filters in eigen space (filter_kind 1).
bool forward16_eligible(const lnz_forward_args& a, int mode) {
  if (mode != 0 && mode != 1) return false;
  if (a.gemm_mode != 0 || (a.filter_kind != 0 && a.filter_kind != 1)) return false;
  if (a.dhid != 128 || a.din0 % 64 != 0 || a.din0 > 128) return false;
  if (mode == 1 && (a.din0 != 128 || a.bwd_din0 % 16 != 0)) return false;
  if (a.n_short + a.n_long + a.n_edge > 32 || a.n_edge < 1 || a.n_long > 12) return false;
  if (a.filter_kind == 1 && a.K % 4 != 0) return false;
  // 32-bit byte offsets into the packed Laplacian and the gains / filters (raw buffer loads)
  if ((int64_t)a.B * a.n_edge * 4096 >= (1ll << 31)) return false;
  const int64_t per_slot = a.filter_kind == 1 ? (int64_t)a.K * a.K : a.K;
  if ((int64_t)a.num_layer * a.B * a.n_long * per_slot * 4 >= (1ll << 31)) return false;
  return (size_t)lds_floats(4, 132, a.n_long) * sizeof(float) <= 160 * 1024;
}

int launch_forward16(const lnz_forward_args& a, int mode, hipStream_t s) {
  const int grid = a.plan ? a.plan_wg_cap : (a.B + 3) / 4;
  const int f3 = lds_floats(3, 136, a.n_long), f4 = lds_floats(4, 132, a.n_long);
  const size_t bytes = (size_t)(f3 > f4 ? f3 : f4) * sizeof(float);
  const void* fns[8] = {
      (const void*)lanczosnet_forward16_kernel<0, 0, false>, (const void*)lanczosnet_forward16_kernel<1, 0, false>,
      (const void*)lanczosnet_forward16_kernel<0, 2, false>, (const void*)lanczosnet_forward16_kernel<1, 2, false>,
      (const void*)lanczosnet_forward16_kernel<0, 0, true>,  (const void*)lanczosnet_forward16_kernel<1, 0, true>,
      (const void*)lanczosnet_forward16_kernel<0, 2, true>,  (const void*)lanczosnet_forward16_kernel<1, 2, true>};
  const int which = (a.n_short > 0 ? 4 : 0) + (a.filter_kind == 0 ? 0 : 2) + mode;
  const void* fn = fns[which];
  note_kernel("lanczosnet_forward16_kernel<%d,%d,%s>", mode, a.filter_kind == 0 ? 0 : 2,
              a.n_short > 0 ? "true" : "false");
  // per launch: the attribute is per device, and a process may drive several (DataParallel)
  LNZ_DYNAMIC_LDS(fn, 160 * 1024, "conv_forward16.hip");
  lnz_forward_args args = a;
  void* params[] = {&args};
  (void)hipLaunchKernel(fn, dim3(grid), dim3(512), params, bytes, s);
  return check_launch(mode == 0 ? "lnz_lanczosnet_forward (16x16 tiles)"
                                : "lnz_lanczosnet_input_grad (16x16 tiles)");
}

}  // namespace lnz
