// R7 + R9 + R10 on STRIPS of 16-row subtiles: the forward (inference and training), the
// input-gradient pass and the gain-gradient pass of the width-128 models.
//
// conv_forward16.hip runs 32-row node tiles (one molecule, or an 8 | 24 / 16 | 16 pair): the QM8
// bench batch rides in 744 tiles = 23.8 k rows for 17.3 k atoms, three tiles (96 rows) on the
// busiest compute unit.  Here a workgroup runs one strip of the plan of lnz_plan_strips — up to
// LNZ_STRIP_SUB subtiles of 16 rows in which every molecule takes ceil(n / 4) * 4 consecutive rows
// at a 4-aligned start and spans at most two subtiles: 18.8 k rows, five subtiles (80 rows) per
// compute unit.  GEMM1 (X W_c^T, 87 % of the matrix work) is row-proportional; the block-diagonal
// products (GEMM2 with the Laplacians, projection on / lift from the Ritz vectors) visit the
// subtile blocks (I, J), |I - J| <= 1, that some molecule touches.
//
// The block products are branch free: every neighbouring pair is multiplied (a pair no molecule
// touches has all-zero fragments), ordered so that consecutive MFMAs go to different accumulators.
//
// Everything else is conv_forward16.hip's scheme (same packs, same fragment indexing, see its file
// comment): wave w owns output columns [16 w, 16 w + 16) of every subtile; a 16-k step of GEMM1 is
// one weight float4 per lane (4-slot register ring), S A fragments (ds_read_b128, pitch 136) and
// 4 S MFMAs v_mfma_f32_16x16x4_f32; long channels in eigen space (Y = V^T X, rows scaled by the
// gains, one lift per layer), edge channels with the GEMM1 result chained into GEMM2 as B operand,
// the next layer's Y from the epilogue's C/D registers; all vector loads of the layer loop are
// issued unconditionally (raw buffer loads, out-of-range offset = 0.0) so that every s_waitcnt
// vmcnt is exact.  Laplacian fragment of block (I, J) for lane (j, kq): M[16 I + j][16 J + 4 kq +
// 0..3] — both the row and the 4-column group belong to one molecule each (4-row granularity), the
// fragment is real when they are the same molecule, and sits in that molecule's pack at its local
// (row, column group).
#include "common.hpp"
#include "conv_tiles.hpp"
#include <type_traits>

namespace {

constexpr int P = 136;     // row pitch of the node-state buffers (floats)
constexpr int VBP = 20;    // row pitch of a 16 x 16 Ritz block (floats: 16-byte rows)
constexpr int HP = 33;     // row pitch of the gated head outputs
constexpr int MAXMOL = 24; // molecules per strip (96 rows / 4)
typedef const __attribute__((address_space(3))) float* lds_cptr;
typedef const __attribute__((address_space(3))) f32x4* lds_c4ptr;

__device__ __forceinline__ f32x4 lds4(lds_cptr p) { return *(lds_c4ptr)p; }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }

// ---- split-precision node state (HALF, lnz_forward_args.gemm_mode 1): a row of the node-state
// buffers keeps its P floats of LDS, but a 32-column block (128 B) holds the 32 fp16 hi pieces of
// its columns, then the 32 lo pieces (x = hi + lo to 22 bits): lane (j, kq) of a wave reads the A
// operand of v_mfma_f32_16x16x32_f16 for columns 32 b + 8 kq .. + 7 as ONE ds_read_b128 per piece,
// at the same float offsets 32 b + 4 kq (hi) and + 16 (lo) as an fp32 fragment — the conflict-free
// pattern of the pitch.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma32h(f32x4 a, f32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b),
                                                c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma32h(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// four values -> elements [4 q, 4 q + 4) of an operand's hi and lo pieces
__device__ __forceinline__ void split_into(const f32x4 v, const int q, f16x8& h, f16x8& l) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 x = (_Float16)v[e];
    h[4 * q + e] = x;
    l[4 * q + e] = (_Float16)(v[e] - (float)x);
  }
}
__device__ __forceinline__ int half_at(int k) { return 64 * (k >> 5) + (k & 31); }  // hi piece; lo: + 32
template <bool HALF>
__device__ __forceinline__ void xs_put(float* rowp, int k, float x) {
  if constexpr (HALF) {
    _Float16* g = reinterpret_cast<_Float16*>(rowp) + half_at(k);
    const _Float16 h = (_Float16)x;
    g[0] = h;
    g[32] = (_Float16)(x - (float)h);
  } else {
    rowp[k] = x;
  }
}
template <bool HALF>
__device__ __forceinline__ float xs_get(const float* rowp, int k) {
  if constexpr (HALF) {
    const _Float16* g = reinterpret_cast<const _Float16*>(rowp) + half_at(k);
    return (float)g[0] + (float)g[32];
  } else {
    return rowp[k];
  }
}
// four consecutive columns 4 c4 .. + 3
template <bool HALF>
__device__ __forceinline__ void xs_put4(float* rowp, int c4, float4 v) {
  if constexpr (HALF) {
    f16x4 h, l;
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (_Float16)x[e];
      l[e] = (_Float16)(x[e] - (float)h[e]);
    }
    _Float16* g = reinterpret_cast<_Float16*>(rowp) + half_at(4 * c4);
    *reinterpret_cast<f16x4*>(g) = h;
    *reinterpret_cast<f16x4*>(g + 32) = l;
  } else {
    *reinterpret_cast<float4*>(rowp + 4 * c4) = v;
  }
}
template <bool HALF>
__device__ __forceinline__ f32x4 xs_get4(const float* rowp, int c4) {
  if constexpr (HALF) {
    const _Float16* g = reinterpret_cast<const _Float16*>(rowp) + half_at(4 * c4);
    const f16x4 h = *reinterpret_cast<const f16x4*>(g), l = *reinterpret_cast<const f16x4*>(g + 32);
    return f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                 (float)h[3] + (float)l[3]};
  } else {
    return *reinterpret_cast<const f32x4*>(rowp + 4 * c4);
  }
}

// LDS floats of a strip of S subtiles with nl long channels
constexpr int VH = 512;    // HALF: floats of a Ritz block's two split-precision operand fragments
constexpr int strip_lds_floats(int S, int nl, bool half = false) {
  return 2 * S * 16 * P + S * 3 * (half ? VH : 16 * VBP) + 2 * nl * S * 16 + 3 * S * 16 + 3 * MAXMOL + 8;
}


// MODE 0 = forward (a.act_out: the training forward's activation store); MODE 1 = the
// input-gradient pass; FK 0 = diagonal gains, 2 = dense K x K filters; SHORT = short-diffusion
// channels — all as in conv_forward16.hip.  HALF: GEMM1 in split precision (x_hi w_hi + x_lo w_hi +
// x_hi w_lo on v_mfma_f32_16x16x32_f16, fp32 accumulate: 3 x 16 cycles per 32-k block and subtile
// instead of 8 x 32), node state in LDS as fp16 pieces (above), weights from lnz_pack_rows_k8_split;
// the block products (Laplacian blocks from a pack converted by lnz_split_laplacian_pack, Ritz blocks
// pre-split in LDS; lift and projection) run in the same split on subtile pairs (apply_split below);
// gains, biases, ReLU, the head and all accumulation are the exact-fp32 code.
template <int S, int MODE, int FK, bool SHORT, bool HALF>
__device__ __forceinline__ void strip_forward(KArgs& a, const int32_t* __restrict__ ent, float* lds,
                                              const int tid, const int wave) {
  constexpr int R = 16 * S;
  constexpr bool FWD = MODE == 0;
  constexpr bool DIAG = FK == 0;
  static_assert(!HALF || (FWD && DIAG && !SHORT), "split precision: inference forward, diagonal gains, no short channels");
#ifdef LNZ_STRIP_PHASES  // per-wave clock64 stamps of the phases (tools/phase_probe16.py)
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
  float* Xs = lds;                                   // [2][R][P]
  float* Vb = Xs + 2 * R * P;                        // [S node subtile][3 = slot subtile - node subtile + 1][16 node][VBP]
  // HALF: Vb holds the A operands of the split-precision lift and projection instead — per block
  // (I, d), J = I + d - 1, and lane (j, kq) 16 bytes (4 hi | 4 lo pieces) of V[node 16 I + j][slot
  // 16 J + 4 kq + 0..3] (lift, first S * 3 * 64 entries) and of V[node 16 J + 4 kq + 0..3][slot 16 I
  // + j] (projection onto slot subtile I)
  float* Gs = Vb + S * 3 * (HALF ? VH : 16 * VBP);   // [2][nl][R]
  int* rowinfo = reinterpret_cast<int*>(Gs + 2 * nl * R);  // [R] molecule of the strip owning the row, or -1
  int* rowok = rowinfo + R;                          // [R] the row is a real (masked-in) node
  int* idI = rowok + R;                              // [S] identity-channel bits common to a subtile's molecules (+ pad to R)
  int* mstart = idI + R;                             // [MAXMOL] first row, node extent, molecule
  int* mext = mstart + MAXMOL;
  int* mid = mext + MAXMOL;
  const int nm = ent[0];

  // ---- the strip's molecules and the row map
  if (tid < MAXMOL) {
    const bool in = tid < nm;
    mid[tid] = in ? ent[2 + 3 * tid] : -1;
    mstart[tid] = in ? ent[3 + 3 * tid] : 0;
    mext[tid] = in ? ent[4 + 3 * tid] : 0;
  }
  if (tid < S) idI[tid] = (FWD && a.ident) ? -1 : 0;
  __syncthreads();
  if (tid < R) {
    int own = -1;
    for (int i = 0; i < nm; ++i) {
      const int n = mext[i];
      const int rows = n <= 4 ? 4 : (n + 3) & ~3;
      if (tid >= mstart[i] && tid < mstart[i] + rows) own = i;
    }
    rowinfo[tid] = own;
    int ok = 0;
    if (own >= 0) {
      const int lrow = tid - mstart[own];
      ok = (lrow < N && a.mask[(int64_t)mid[own] * N + lrow] != 0) ? 1 : 0;
      if (FWD && a.ident) atomicAnd(&idI[tid >> 4], (int)a.ident[mid[own]]);
    }
    rowok[tid] = ok;
  }
  __syncthreads();

  // ---- embedding gather / float features (model/lanczos_net.py:154, lanczos_net_general.py:156)
  //      Three entries per thread at a time, every load unconditional (a row nobody owns reads
  //      element 0 and stores zeros): the ids of all three are in flight together, then the three
  //      rows — one load after the other, each waited for, was 2 x 2.5 global round trips of the
  //      prologue's 8.7 us
  {
    const int d4 = a.din0 >> 2, total = R * d4;
    constexpr int UN = 3;
    for (int base = tid; base < total; base += 512 * UN) {
      int row[UN], c4[UN];
      bool in[UN], valid[UN];
      int64_t at[UN];   // element offset of the source row (or the id's position)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + 512 * u;
        in[u] = idx < total;
        row[u] = in[u] ? idx / d4 : 0;
        c4[u] = in[u] ? idx - row[u] * d4 : 0;
        const int own = rowinfo[row[u]];
        const int mol = own >= 0 ? mid[own] : 0, lrow = own >= 0 ? row[u] - mstart[own] : 0;
        valid[u] = in[u] && own >= 0 && (!FWD || lrow < N);
        at[u] = !valid[u] ? 0
                : !FWD   ? (((int64_t)(a.num_layer - 1) * B + mol) * 32 + lrow) * 128
                         : (int64_t)mol * N + lrow;
      }
      if (FWD && a.node_feat) {
        int64_t id[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) id[u] = a.node_feat[at[u]];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int64_t c = id[u] < 0 ? 0 : (id[u] >= a.num_atom ? a.num_atom - 1 : id[u]);
          at[u] = valid[u] ? c * a.din0 : 0;
        }
      } else if (FWD) {
#pragma unroll
        for (int u = 0; u < UN; ++u) at[u] *= a.din0;
      }
      const float* __restrict__ srcb = !FWD ? a.dy : (a.node_feat ? a.embedding : a.node_feat_f);
      float4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = reinterpret_cast<const float4*>(srcb + at[u])[c4[u]];
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (in[u]) xs_put4<HALF>(&Xs[row[u] * P], c4[u], valid[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
  // ---- Ritz blocks: Vb[Jn][d][nu][ro] = V[molecule][node][slot] when node row 16 Jn + nu and
  //      slot row 16 (Jn + d - 1) + ro belong to the same molecule (slot k of a molecule rides on
  //      its k-th row), zero elsewhere
  //      (every load of a thread unconditional and in flight before the first store: element 0 where
  //      the entry is zero)
  if constexpr (HALF) {
    constexpr int NV = (2 * S * 3 * 64 + 511) / 512;
    float vv[NV][4];
    bool ok[NV][4];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 512 * u;
      const bool in = idx < 2 * S * 3 * 64;
      const int proj = idx >= S * 3 * 64 ? 1 : 0;
      const int i0 = in ? idx - proj * S * 3 * 64 : 0;
      const int I = i0 / 192, rem = i0 - I * 192;
      const int d = rem >> 6, jj = rem & 15, kk = (rem >> 4) & 3;
      const int J = I + d - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nrow = proj ? 16 * J + 4 * kk + r : 16 * I + jj;
        const int srow = proj ? 16 * I + jj : 16 * J + 4 * kk + r;
        int64_t at = 0;
        ok[u][r] = false;
        if (in && J >= 0 && J < S) {
          const int own = rowinfo[nrow];
          if (own >= 0 && rowinfo[srow] == own) {
            const int lnode = nrow - mstart[own], k = srow - mstart[own];
            if (lnode < N && k < K) ok[u][r] = true, at = ((int64_t)mid[own] * N + lnode) * K + k;
          }
        }
        vv[u][r] = a.V[at];
      }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 512 * u;
      if (idx >= 2 * S * 3 * 64) continue;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = ok[u][r] ? finite_or_zero(vv[u][r]) : 0.0f;
      f16x8 h = f16x8{0, 0, 0, 0, 0, 0, 0, 0}, l = h;
      split_into(v, 0, h, l);
#pragma unroll
      for (int e = 0; e < 4; ++e) h[4 + e] = l[e];
      *reinterpret_cast<f16x8*>(Vb + 4 * idx) = h;
    }
  } else {
    constexpr int NV = (S * 3 * 256 + 511) / 512;
    float vv[NV];
    bool ok[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 512 * u;
      const bool in = idx < S * 3 * 256;
      const int Jn = idx / 768, rem = idx - Jn * 768;
      const int d = rem >> 8, nu = (rem >> 4) & 15, ro = rem & 15;
      const int nrow = 16 * Jn + nu, srow = 16 * (Jn + d - 1) + ro;
      int64_t at = 0;
      ok[u] = false;
      if (in && srow >= 0 && srow < R) {
        const int own = rowinfo[nrow];
        if (own >= 0 && rowinfo[srow] == own) {
          const int lnode = nrow - mstart[own], k = srow - mstart[own];
          if (lnode < N && k < K) ok[u] = true, at = ((int64_t)mid[own] * N + lnode) * K + k;
        }
      }
      vv[u] = a.V[at];
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 512 * u;
      if (idx >= S * 3 * 256) continue;
      const int Jn = idx / 768, rem = idx - Jn * 768;
      const int d = rem >> 8, nu = (rem >> 4) & 15, ro = rem & 15;
      Vb[((Jn * 3 + d) * 16 + nu) * VBP + ro] = ok[u] ? finite_or_zero(vv[u]) : 0.0f;
    }
  }
  // ---- spectral gains by slot row: Gs[l & 1][s][rho]; loads of layer l + 1 are issued at the
  //      start of layer l and go to LDS in front of the layer's last barrier
  constexpr int GREG = 3;  // nl * R <= 512 * GREG  (nl <= 12, R <= 96: 1152)
  constexpr unsigned OOB = 0x80000000u;
  float greg[GREG];
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.G), 0, nl > 0 ? a.num_layer * B * nl * K * (DIAG ? 1 : K) * 4 : 0, 0x00020000);
  unsigned goff[GREG];
#pragma unroll
  for (int u = 0; u < GREG; ++u) {
    const int idx = tid + 512 * u;
    goff[u] = OOB;
    if (DIAG && idx < nl * R) {
      const int sc = idx / R, rho = idx - sc * R;
      const int own = rowinfo[rho];
      if (own >= 0) {
        const int k = rho - mstart[own];
        if (k < K && k < mext[own]) goff[u] = (unsigned)(((mid[own] * nl + sc) * K + k) * 4);
      }
    }
  }
  auto load_gains = [&](int l) {
#pragma unroll
    for (int u = 0; u < GREG; ++u)
      greg[u] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, goff[u], l * B * nl * K * 4, 0));
  };
  auto store_gains = [&](int l) {
    float* dst = Gs + (l & 1) * nl * R;
#pragma unroll
    for (int u = 0; u < GREG; ++u) {
      const int idx = tid + 512 * u;
      if (idx < nl * R) dst[idx] = greg[u];
    }
  };
  if (DIAG && nl > 0) {
    load_gains(FWD ? 0 : a.num_layer - 1);
    store_gains(FWD ? 0 : a.num_layer - 1);
  }

  // ---- per-lane addressing of the packed Laplacian and the block masks
  const int rt = wave >> 1;
  const int wlane = 64 * (kq >> 1) + 32 * (kq & 1) + 16 * (wave & 1) + j;  // float4 within a 16-k step
  unsigned loff[S][3];  // byte offset of fragment (I, J = I + d - 1) of edge type 0, or OOB
  unsigned doff[DIAG ? 1 : S][3];  // FK 2: fragment (I, J) of a dense filter: slot row 16 I + j is
                                   // slot kr of its molecule, the 4-column group molecule-local
  const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.Lp), 0, B * ne * 4096, 0x00020000);
  int idm[S];
#pragma unroll
  for (int I = 0; I < S; ++I) {
    const int row = 16 * I + j;
    const int own = rowinfo[row];
    const int st = own >= 0 ? mstart[own] : 0;
    const int mol = own >= 0 ? mid[own] : 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int J = I + d - 1;
      bool ok = false;
      unsigned off = OOB;
      if (J >= 0 && J < S) {
        const int cg = 16 * J + 4 * kq;
        ok = own >= 0 && rowinfo[cg] == own;
        const int c = cg - st;
        if (ok) off = (unsigned)((mol * ne * 256 + (c >> 3) * 64 + ((c >> 2) & 1) * 32 + (row - st)) * 16);
        if (!DIAG)
          doff[DIAG ? 0 : I][d] = (ok && row - st < K && c < K)
                                      ? (unsigned)((((mol * nl) * K + (row - st)) * K + c) * 4) : OOB;
      } else if (!DIAG) {
        doff[DIAG ? 0 : I][d] = OOB;
      }
      loff[I][d] = off;
    }
    idm[I] = __builtin_amdgcn_readfirstlane(idI[I]);
  }
  __syncthreads();

  int chan_total = 0;
  LNZ_PH(0)  // prologue
  const lds_cptr xlane = (lds_cptr)(Xs + j * P + 4 * kq);  // this lane's A row of subtile 0
  int cur = 0;
  for (int l = 0; l < a.num_layer; ++l) {
    const bool more = l + 1 < a.num_layer;
    const int la = FWD ? l : a.num_layer - 1 - l;  // conv layer of this iteration
    const int lg = FWD ? l + 1 : la - 1;            // layer whose gains are staged under it
    const int din = l == 0 ? a.din0 : 128;
    const int Q = din >> 3, Q16 = din >> 4;
    const int Gtot = C * Q;
    const float4* __restrict__ Wl = reinterpret_cast<const float4*>(a.Wp + a.w_off[l]);
    const float* gsl = Gs + (la & 1) * nl * R;
    const int nxt = cur ^ 1;
    // width this iteration produces: waves beyond it only keep the barriers
    const int wout = FWD ? 128 : (la == 0 ? a.bwd_din0 : 128);
    const bool active = FWD || 16 * wave < wout;

    // weight stream of this wave: contiguous over the layer's channels, 128 float4 per 16-k step,
    // 4-slot register ring (prefetch distance 3 steps = 12 S MFMAs)
    const float4* __restrict__ wp = Wl + (int64_t)rt * Gtot * 64 + wlane;
    // (HALF: every layer is 128 wide = four 32-k blocks per channel = eight slots; a slot is reloaded
    // with the NEXT channel's block right behind its last use — a whole channel of prefetch distance:
    // with GEMM1 at a fifth of its fp32 time, one block of distance left the loads exposed)
    float4 ring[HALF ? 8 : 4];
    if (active) {
#pragma unroll
      for (int s3 = 0; s3 < (HALF ? 8 : 3); ++s3) ring[s3] = wp[s3 * 128];
    }
    // (behind the ring prime: vector loads return in order, the first steps must not wait for G)
    if (DIAG && nl > 0 && more) load_gains(lg);

    f32x4 out[S];
    {
      const float bv = FWD ? (a.bias + a.b_off[l])[16 * wave + j] : 0.0f;
#pragma unroll
      for (int I = 0; I < S; ++I) out[I] = splat4(bv);
    }

    // ---------------- first layer: Y = V^T X from LDS into the other buffer ----------------
    if (nl > 0 && l == 0) {
      if (16 * wave < din) {
        f32x4 Y[S];
#pragma unroll
        for (int I = 0; I < S; ++I) Y[I] = splat4(0.f);
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int I = 0; I < S; ++I) {  // slot subtile I, node subtile J
              const int J = I + d - 1;
              if (J < 0 || J >= S) continue;
              const int nu = 4 * kq + r;
              float vt;
              if constexpr (HALF) {  // (once per launch: the fp32 product on the rebuilt operands)
                const _Float16* f = reinterpret_cast<const _Float16*>(Vb + 4 * ((S * 3 + I * 3 + d) * 64 + lane));
                vt = (float)f[r] + (float)f[4 + r];
              } else {
                vt = Vb[((J * 3 + (2 - d)) * 16 + nu) * VBP + j];
              }
              Y[I] = mfma16(vt, xs_get<HALF>(&Xs[cur * R * P + (16 * J + nu) * P], 16 * wave + j), Y[I]);
            }
#pragma unroll
        for (int I = 0; I < S; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            xs_put<HALF>(&Xs[nxt * R * P + (16 * I + 4 * kq + r) * P], 16 * wave + j, Y[I][r]);
      }
      __syncthreads();
    }

    LNZ_PH(1)  // layer head: ring prime, gains loads, first-layer projection
    // ---------------- GEMM1 of one channel: Z[I] = A rows (X or Y) x W_c^T ----------------
    f32x4 Z[S], acur[S], alow[HALF ? S : 1];
    auto load_first = [&](lds_cptr x0) {
#pragma unroll
      for (int I = 0; I < S; ++I) {
        acur[I] = lds4(x0 + 16 * I * P);
        if constexpr (HALF) alow[I] = lds4(x0 + 16 * I * P + 16);
      }
    };
    // four steps; `wrap`: the A prefetch of the last one fetches k = 0 again — the first fragments
    // of the NEXT channel (channels of a block read the same rows)
    auto steps4 = [&](lds_cptr xb, lds_cptr x0, auto wrap) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ring[(u + 3) & 3] = wp[(u + 3) * 128];
        f32x4 anext[S];
        const lds_cptr xn = (decltype(wrap)::value && u == 3) ? x0 : xb + 16 * (u + 1);
#pragma unroll
        for (int I = 0; I < S; ++I) anext[I] = lds4(xn + 16 * I * P);
        const float4 bv = ring[u];
#pragma unroll
        for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][0], bv.x, Z[I]);
#pragma unroll
        for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][1], bv.y, Z[I]);
#pragma unroll
        for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][2], bv.z, Z[I]);
#pragma unroll
        for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][3], bv.w, Z[I]);
#pragma unroll
        for (int I = 0; I < S; ++I) acur[I] = anext[I];
        // issue order: one LDS read behind every fourth MFMA, the ring load in the last gap
#pragma unroll
        for (int g = 0; g < S; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (g == S - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
      wp += 4 * 128;
    };
    auto gemm1 = [&](lds_cptr x0, auto&& before_last) {
      // (HALF: a channel's GEMM1 is 12 S short MFMAs — the operator fragments are asked for at its
      // start, not under its last four steps)
      // the two waves of a SIMD (w, w + 4) take turns at the head of the matrix pipe.  (A feedback
      // version — each wave publishes the channels it has started in LDS and the one behind its
      // partner raises its priority — evens the two out, 548 | 596 k cycles in the long block
      // instead of 506 | 625, and is 1.3 % SLOWER over the launch.)
      if (((chan_total ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      ++chan_total;
#pragma unroll
      for (int I = 0; I < S; ++I) Z[I] = splat4(0.f);
      if constexpr (HALF) {
        before_last();
#pragma unroll
        for (int b = 0; b < 4; ++b) {  // 32-k block b: ring slots 2 b (hi pieces of the weights), 2 b + 1 (lo)
          f32x4 anext[S], lnext[S];
          const lds_cptr xn = b == 3 ? x0 : x0 + 32 * (b + 1);  // (wraps to the next channel's first block)
#pragma unroll
          for (int I = 0; I < S; ++I) {
            anext[I] = lds4(xn + 16 * I * P);
            lnext[I] = lds4(xn + 16 * I * P + 16);
          }
          const f32x4 bh = __builtin_bit_cast(f32x4, ring[2 * b]);
          const f32x4 bl = __builtin_bit_cast(f32x4, ring[2 * b + 1]);
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = mfma32h(acur[I], bl, Z[I]);
          ring[2 * b + 1] = wp[(8 + 2 * b + 1) * 128];
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = mfma32h(alow[I], bh, Z[I]);
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = mfma32h(acur[I], bh, Z[I]);
          ring[2 * b] = wp[(8 + 2 * b) * 128];
#pragma unroll
          for (int I = 0; I < S; ++I) {
            acur[I] = anext[I];
            alow[I] = lnext[I];
          }
          // (Issue order left to the compiler: it reads a block's fragments in one burst in front of
          // the block and sinks a channel's eight reloads behind its last MFMA.  Forcing one fragment
          // read behind each MFMA and each reload behind its slot's last sweep with
          // sched_group_barrier costs 48 spilled registers and measures the same: 0.219 | 0.222 ms.)
        }
        wp += 8 * 128;
        return;
      }
      lds_cptr xb = x0;
#pragma unroll 1
      for (int q0 = 4; q0 < Q16; q0 += 4) {
        steps4(xb, x0, std::false_type{});
        xb += 64;
      }
      before_last();
      steps4(xb, x0, std::true_type{});
    };

    // Laplacian fragments of one edge type for every block (I, J): they land under the last four
    // steps of the channel's own GEMM1 (identity channels of a subtile: offsets beyond the buffer,
    // no memory traffic).  The same registers hold a dense filter's fragments in the long block.
    f32x4 mop[S][3];
    auto fetch = [&](int e, bool use_ident) {
#pragma unroll
      for (int I = 0; I < S; ++I) {
        const unsigned skip = (use_ident && ((idm[I] >> e) & 1)) ? OOB : 0u;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (I + d - 1 < 0 || I + d - 1 >= S) continue;
          mop[I][d] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(l_rsrc, loff[I][d] | skip, e * 4096, 0));
        }
      }
    };
    // acc[I] += F[I][J] Zp[J] over the neighbouring blocks in split precision on
    // v_mfma_f32_16x16x32_f16 (HALF); frag(I, d, q, ah, al) puts the hi / lo pieces of lane (j, kq)'s
    // fragment of block (I, J = I + d - 1) into elements [4 q, 4 q + 4) of the A operands
    auto apply_split = [&](f32x4 (&acc)[S], const f32x4 (&Zp)[S], auto&& frag) {
      if constexpr (HALF) {
        // the same products in split precision on v_mfma_f32_16x16x32_f16 (16 cycles, like the
        // 16-k shape): K = 32 = the subtile pair (2 p, 2 p + 1) — lane (j, kq) supplies its own C/D
        // rows 4 kq .. + 3 of both Z blocks as B operand and the two operator fragments it holds as
        // A operand (a block the row subtile does not touch: zeros) — three products per pair, at
        // most two pairs per row subtile: 27 MFMAs of 16 cycles for S = 5 instead of 52 of 32
        constexpr int NP = (S + 1) / 2;
        f16x8 zh[NP], zl[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          zh[p] = zl[p] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (2 * p + q < S) split_into(Zp[2 * p + q], q, zh[p], zl[p]);
        }
        f16x8 ah[S][2], al[S][2];
#pragma unroll
        for (int I = 0; I < S; ++I)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int p = ((I > 0 ? I - 1 : 0) >> 1) + t;
            ah[I][t] = al[I][t] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int J = 2 * p + q, d = J - I + 1;
              if (J < S && d >= 0 && d < 3) frag(I, d, q, ah[I][t], al[I][t]);
            }
          }
#pragma unroll
        for (int kind = 0; kind < 3; ++kind)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int I = 0; I < S; ++I) {
              const int p = ((I > 0 ? I - 1 : 0) >> 1) + t;
              if (2 * p >= S || 2 * p > I + 1) continue;  // no block of this pair touches row subtile I
              acc[I] = mfma32h(kind == 1 ? al[I][t] : ah[I][t], kind == 0 ? zl[p] : zh[p], acc[I]);
            }
      }
    };
    // a Ritz block's stored operand pieces (which = 0: lift, 1: projection)
    auto ritz_frag = [&](int which) {
      return [&, which](int I, int d, int q, f16x8& ah, f16x8& al) {
        const f16x4* f = reinterpret_cast<const f16x4*>(Vb + 4 * ((which * S * 3 + I * 3 + d) * 64 + lane));
        const f16x4 h = f[0], l = f[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) ah[4 * q + e] = h[e], al[4 * q + e] = l[e];
      };
    };
    // acc[I] += M[I][J] Zp[J] over every neighbouring block (I, J): branch free — a block no
    // molecule touches has all-zero fragments (offsets beyond the buffer) — and ordered so that
    // consecutive MFMAs go to different accumulators.  (With the blocks masked by a wave-uniform
    // bit each, 8 % fewer MFMAs were issued in chains of four dependent ones behind a branch:
    // the block phases ran at 2 - 5 x their matrix time.)
    auto apply_all = [&](f32x4 (&acc)[S], const f32x4 (&Zp)[S]) {
      if constexpr (HALF) {
        apply_split(acc, Zp, [&](int I, int d, int q, f16x8& ah, f16x8& al) {
          // the pack holds a fragment as 4 hi | 4 lo pieces (lnz_spectral_gains_rows_split /
          // lnz_split_laplacian_pack): splitting the 3 S - 2 fragments of an edge type here, in every
          // one of the eight waves and in every layer, was 11 % of the launch
          const f16x8 f = __builtin_bit_cast(f16x8, mop[I][d]);
#pragma unroll
          for (int e = 0; e < 4; ++e) ah[4 * q + e] = f[e], al[4 * q + e] = f[4 + e];
        });
        return;
      }
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int I = 0; I < S; ++I) {
            const int J = I + d - 1;
            if (J < 0 || J >= S) continue;
            acc[I] = mfma16(mop[I][d][r], Zp[J][r], acc[I]);
          }
    };

    // ---------------- short-diffusion channels: out += L_0^p (X W_c^T) ----------------
    if (SHORT && active) {
      const lds_cptr x0 = xlane + cur * R * P;
      load_first(x0);
      for (int c = 0; c < ns; ++c) {
        gemm1(x0, [&] { fetch(0, false); });
        const int p = a.short_dist[c];
        for (int rep = 1; rep < p; ++rep) {
          f32x4 Zn[S];
#pragma unroll
          for (int I = 0; I < S; ++I) Zn[I] = splat4(0.f);
          apply_all(Zn, Z);
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = Zn[I];
        }
        apply_all(out, Z);
      }
    }

    // ---------------- eigen-space block: out += V [ sum_s F_s (Y W_s^T) ] ----------------
    //   F_s = diag(g_s) (FK 0) or the dense filter DD_s (FK 2)
    if (nl > 0 && active) {
      const lds_cptr y0 = xlane + nxt * R * P;
      load_first(y0);
      f32x4 T[S];
#pragma unroll
      for (int I = 0; I < S; ++I) T[I] = splat4(0.f);
      for (int s = 0; s < nl; ++s) {
        if constexpr (DIAG) {
          gemm1(y0, [] {});
#pragma unroll
          for (int I = 0; I < S; ++I) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gsl + s * R + 16 * I + 4 * kq);
            T[I] += gv * Z[I];
          }
        } else {
          gemm1(y0, [&] {
#pragma unroll
            for (int I = 0; I < S; ++I)
#pragma unroll
              for (int d = 0; d < 3; ++d) {
                if (I + d - 1 < 0 || I + d - 1 >= S) continue;
                mop[I][d] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                               g_rsrc, doff[DIAG ? 0 : I][d], (la * B * nl + s) * K * K * 4, 0));
              }
          });
          apply_all(T, Z);
        }
      }
      LNZ_PH(2)  // long block
      // lift back: out[I] (node rows) += V[I][J] T[J] over the slot subtiles J the block mask names
      if constexpr (HALF) {
        apply_split(out, T, ritz_frag(0));
      } else {
        f32x4 vl[S][3];
#pragma unroll
        for (int I = 0; I < S; ++I)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            if (I + d - 1 < 0 || I + d - 1 >= S) continue;
            vl[I][d] = *reinterpret_cast<const f32x4*>(&Vb[((I * 3 + d) * 16 + j) * VBP + 4 * kq]);
          }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int I = 0; I < S; ++I) {
              const int J = I + d - 1;
              if (J < 0 || J >= S) continue;
              out[I] = mfma16(vl[I][d][r], T[J][r], out[I]);
            }
      }
    }

    LNZ_PH(3)  // lift
    // ---------------- node-space block: out += M_e (X W_e^T) per edge type ----------------
    if (active) {
      const lds_cptr x0 = xlane + cur * R * P;
      load_first(x0);
      for (int e = 0; e < ne; ++e) {
        gemm1(x0, [&] { fetch(e, true); });
        LNZ_PH(4)  // edge GEMM1
        apply_all(out, Z);  // (identity subtiles: zero fragments)
#pragma unroll
        for (int I = 0; I < S; ++I)
          if ((idm[I] >> e) & 1) out[I] += Z[I];  // identity on every molecule of the subtile
        LNZ_PH(5)  // GEMM2
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
      for (int I = 0; I < S; ++I) {
        // rows 16 I + 4 kq + r: one owner (molecules start on multiples of 4).  The row map is
        // re-read per layer: addresses kept across the layer loop would hold registers under the GEMMs
        const int row0 = 16 * I + 4 * kq;
        int own = rowinfo[row0];
        asm volatile("" : "+v"(own));
        const int mol = own >= 0 ? mid[own] : 0;
        const int lrow0 = own >= 0 ? row0 - mstart[own] : 0;
        const int64_t rowbase = (int64_t)mol * 32 + lrow0;
        f32x4 v = out[I];
        if (FWD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
          if (a.act_out && own >= 0) {
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
            v[r] = (own >= 0 && xa[r * 128] > 0.0f) ? v[r] : 0.0f;
            if (own >= 0) dyp[r * 128] = v[r];
          }
        } else if (own >= 0) {
          float* p = a.dx0 + rowbase * a.bwd_din0 + col;
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r * a.bwd_din0] = v[r];
        }
        out[I] = v;
#pragma unroll
        for (int r = 0; r < 4; ++r) xs_put<HALF>(&Xs[nxt * R * P + (row0 + r) * P], col, v[r]);
      }
      if (nl > 0 && more) {
        f32x4 Y[S];
#pragma unroll
        for (int I = 0; I < S; ++I) Y[I] = splat4(0.f);
        if constexpr (HALF) {
          apply_split(Y, out, ritz_frag(1));
        } else {
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int I = 0; I < S; ++I) {  // slot subtile I, node subtile J
                const int J = I + d - 1;
                if (J < 0 || J >= S) continue;
                Y[I] = mfma16(Vb[((J * 3 + (2 - d)) * 16 + 4 * kq + r) * VBP + j], out[J][r], Y[I]);
              }
        }
#pragma unroll
        for (int I = 0; I < S; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) xs_put<HALF>(&Xs[cur * R * P + (16 * I + 4 * kq + r) * P], col, Y[I][r]);
      }
      if (MODE == 1 && la > 0 && (a.dy_compact || a.dbias_part)) {
        // What the weight / bias gradients of conv layer la - 1 need: dY_{la-1} in the COMPACT row
        // numbering of the message matrix (real nodes only) and this strip's column sums (rows of
        // padded nodes and unowned rows are zero) — one writer per (strip, layer, column): entry
        // blockIdx.x of dbias_part (zero-initialised, summed over all entries by the consumer).
        float colsum = 0.0f;
#pragma unroll
        for (int I = 0; I < S; ++I) {
          const int row0 = 16 * I + 4 * kq;
          int own = rowinfo[row0];
          asm volatile("" : "+v"(own));
          const f32x4 v = out[I];
          colsum += (v[0] + v[1]) + (v[2] + v[3]);
          if (a.dy_compact && own >= 0) {
            const int lrow0 = row0 - mstart[own], nmol = mext[own];
            float* dc = a.dy_compact +
                        ((int64_t)(la - 1) * a.dy_compact_rows + a.row_off[mid[own]] + lrow0) * 128 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (lrow0 + r < nmol) dc[r * 128] = v[r];
          }
        }
        if (a.dbias_part && (int)blockIdx.x < (a.dbias_part_cap > 0 ? a.dbias_part_cap : 2 * a.plan_wg_cap)) {
          colsum += __shfl_xor(colsum, 16, 64);
          colsum += __shfl_xor(colsum, 32, 64);
          if (kq == 0)
            a.dbias_part[((int64_t)blockIdx.x * a.num_layer + (la - 1)) * 128 + col] = colsum;
        }
      }
    }
    __syncthreads();
    cur = nxt;
    LNZ_PH(6)  // epilogue
  }
  __builtin_amdgcn_s_setprio(0);
  if (!FWD) return;

  // ---- optional debug/test output of the final node state
  if (a.state_out) {
    for (int idx = tid; idx < R * 128; idx += 512) {
      const int row = idx >> 7, col = idx & 127;
      const int own = rowinfo[row];
      if (own >= 0)
        a.state_out[((int64_t)mid[own] * 32 + (row - mstart[own])) * 128 + col] =
            xs_get<HALF>(&Xs[cur * R * P + row * P], col);
    }
  }

  // ---- head (model/lanczos_net.py:185-194): wave w = subtile w; one 32-column tile = [W_o ; w_a ;
  //      0] as two 16-column MFMA tiles; the gate logit is column dout of the same row.  The gated
  //      rows go to LDS (the free node-state buffer) and every (molecule, output) is summed over
  //      the molecule's rows in a fixed order.
  float* Hs = Xs + (cur ^ 1) * R * P;  // [R][HP]
  const int Pd = a.dout;
  if (wave < S) {
    const int I = wave;
    f32x4 acc[2] = {splat4(a.bias_head[j]), splat4(a.bias_head[16 + j])};
    const float4* __restrict__ wh = reinterpret_cast<const float4*>(a.Wp_head) + 64 * (kq >> 1) + 32 * (kq & 1) + j;
    const lds_cptr xr = xlane + cur * R * P + 16 * I * P;
    const float* xrow = Xs + cur * R * P + (16 * I + j) * P;
#pragma unroll 2
    for (int q = 0; q < 8; ++q) {
      const f32x4 av = HALF ? xs_get4<HALF>(xrow, 4 * q + kq) : lds4(xr + 16 * q);
      const float4 b0 = wh[q * 128], b1 = wh[q * 128 + 16];
      acc[0] = mfma16(av[0], b0.x, acc[0]);
      acc[1] = mfma16(av[0], b1.x, acc[1]);
      acc[0] = mfma16(av[1], b0.y, acc[0]);
      acc[1] = mfma16(av[1], b1.y, acc[1]);
      acc[0] = mfma16(av[2], b0.z, acc[0]);
      acc[1] = mfma16(av[2], b1.z, acc[1]);
      acc[0] = mfma16(av[3], b0.w, acc[0]);
      acc[1] = mfma16(av[3], b1.w, acc[1]);
    }
    const int glane = (lane & 48) | (Pd & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float logit = __shfl(Pd < 16 ? acc[0][r] : acc[1][r], glane, 64);
      const float gate = 1.0f / (1.0f + __expf(-logit));
      const int row = 16 * I + 4 * kq + r;
      const bool ok = rowok[row] != 0;
      Hs[row * HP + j] = ok ? gate * acc[0][r] : 0.0f;
      Hs[row * HP + 16 + j] = ok ? gate * acc[1][r] : 0.0f;
    }
  }
  __syncthreads();
  for (int t = tid; t < nm * Pd; t += 512) {
    const int i = t / Pd, c = t - i * Pd;
    const int st = mstart[i], n = mext[i];
    float sum = 0.0f, cnt = 0.0f;
    for (int r = 0; r < n; ++r) {
      sum += Hs[(st + r) * HP + c];
      cnt += rowok[st + r] ? 1.0f : 0.0f;
    }
    a.score[(int64_t)mid[i] * Pd + c] = sum / cnt;
  }
#ifdef LNZ_STRIP_PHASES
  LNZ_PH(7)  // head
  if (a.state_out && lane == 0 && blockIdx.x < 8) {
    float* d = a.state_out + ((int64_t)B * 32 * 128) + (blockIdx.x * 8 + wave) * 16;
    for (int i = 0; i < 8; ++i) d[i] = (float)ph[i];
    d[8] = (float)(clock64() - t_all);
    d[9] = (float)S;
  }
#endif
}

// -----------------------------------------------------------------------------------------
// Gradient of the loss w.r.t. the spectral gains on strips (training; conv_forward.hip's
// gain_grad_half on the tile plan):
//   dG[l][b][k][s] = sum_o P[k][o] Q_s[k][o],   P = V^T dY_l,   Q_s = (V^T X_l) W_s^T
// Every conv layer in turn: X_l and dY_l are staged in the two node-state buffers, projected (Y to
// LDS over X_l, P kept in C/D registers), then each long channel's GEMM1 runs on Y with the FORWARD
// weight pack and its result is multiplied with P and reduced over the wave's 16 columns (DPP row
// sums) and over the eight waves (LDS, fixed order: deterministic).
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
  return v;
}

template <int S>
__device__ __forceinline__ void strip_gain_grad(KArgs& a, const int32_t* __restrict__ ent, float* lds,
                                                const int tid, const int wave) {
  constexpr int R = 16 * S;
  const int lane = tid & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int N = a.N, K = a.K, B = a.B;
  const int nl = a.n_long;
  const int C = a.n_short + a.n_long + a.n_edge;
  float* Xs = lds;                                   // [2][R][P]: X_l / Y, dY_l / wave partials
  float* Vb = Xs + 2 * R * P;
  int* rowinfo = reinterpret_cast<int*>(Vb + S * 3 * 16 * VBP);
  int* mstart = rowinfo + R;
  int* mext = mstart + MAXMOL;
  int* mid = mext + MAXMOL;
  const int nm = ent[0];
  if (tid < MAXMOL) {
    const bool in = tid < nm;
    mid[tid] = in ? ent[2 + 3 * tid] : -1;
    mstart[tid] = in ? ent[3 + 3 * tid] : 0;
    mext[tid] = in ? ent[4 + 3 * tid] : 0;
  }
  __syncthreads();
  if (tid < R) {
    int own = -1;
    for (int i = 0; i < nm; ++i) {
      const int n = mext[i];
      const int rows = n <= 4 ? 4 : (n + 3) & ~3;
      if (tid >= mstart[i] && tid < mstart[i] + rows) own = i;
    }
    rowinfo[tid] = own;
  }
  __syncthreads();
  for (int idx = tid; idx < S * 3 * 256; idx += 512) {
    const int Jn = idx / 768, rem = idx - Jn * 768;
    const int d = rem >> 8, nu = (rem >> 4) & 15, ro = rem & 15;
    const int nrow = 16 * Jn + nu, srow = 16 * (Jn + d - 1) + ro;
    float v = 0.0f;
    if (srow >= 0 && srow < R) {
      const int own = rowinfo[nrow];
      if (own >= 0 && rowinfo[srow] == own) {
        const int lnode = nrow - mstart[own], k = srow - mstart[own];
        if (lnode < N && k < K) v = finite_or_zero(a.V[((int64_t)mid[own] * N + lnode) * K + k]);
      }
    }
    Vb[((Jn * 3 + d) * 16 + nu) * VBP + ro] = v;
  }
  const int rt = wave >> 1;
  const int wlane = 64 * (kq >> 1) + 32 * (kq & 1) + 16 * (wave & 1) + j;
  const lds_cptr ylane = (lds_cptr)(Xs + j * P + 4 * kq);

  for (int la = 0; la < a.num_layer; ++la) {
    const int din = la == 0 ? a.din0 : 128;
    const int Q = din >> 3, Q16 = din >> 4;
    // ---- stage X_la (buffer 0) and dY_la (buffer 1): rows of the strip's molecules, zero elsewhere
    {
      const float* xsrc = la == 0 ? a.x0 : a.act + (int64_t)(la - 1) * B * 32 * 128;
      const int d4 = din >> 2, e4 = 32;
      const int total = R * (d4 + e4);
      constexpr int UN = 4;  // loads in flight per thread
      for (int base = tid; base < total; base += 512 * UN) {
        float4 v[UN];
        int dst[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int idx = base + u * 512;
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[u] = -1;
          if (idx < total) {
            const int row = idx / (d4 + e4), c = idx - row * (d4 + e4);
            const int own = rowinfo[row];
            const int64_t mrow = own >= 0 ? (int64_t)mid[own] * 32 + (row - mstart[own]) : 0;
            if (c < d4) {
              if (own >= 0) v[u] = reinterpret_cast<const float4*>(xsrc + mrow * din)[c];
              dst[u] = row * P + 4 * c;
            } else {
              if (own >= 0)
                v[u] = reinterpret_cast<const float4*>(a.dy + ((int64_t)la * B * 32 + mrow) * 128)[c - d4];
              dst[u] = R * P + row * P + 4 * (c - d4);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
          if (dst[u] >= 0) *reinterpret_cast<float4*>(&Xs[dst[u]]) = v[u];
      }
    }
    __syncthreads();
    // ---- projections: Pb = V^T dY (registers), Y = V^T X (registers, then LDS over X)
    f32x4 Pb[S], Yb[S];
#pragma unroll
    for (int I = 0; I < S; ++I) Pb[I] = Yb[I] = splat4(0.f);
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int I = 0; I < S; ++I) {  // slot subtile I, node subtile J
          const int J = I + d - 1;
          if (J < 0 || J >= S) continue;
          const int nu = 4 * kq + r;
          const float vt = Vb[((J * 3 + (2 - d)) * 16 + nu) * VBP + j];
          Pb[I] = mfma16(vt, Xs[R * P + (16 * J + nu) * P + 16 * wave + j], Pb[I]);
          if (16 * wave < din) Yb[I] = mfma16(vt, Xs[(16 * J + nu) * P + 16 * wave + j], Yb[I]);
        }
    __syncthreads();  // every wave has read X and dY
    if (16 * wave < din) {
#pragma unroll
      for (int I = 0; I < S; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(16 * I + 4 * kq + r) * P + 16 * wave + j] = Yb[I][r];
    }
    __syncthreads();
    // ---- long channels: Q_s = Y W_s^T, row-wise <P, Q_s> over this wave's 16 columns
    const float4* __restrict__ Wl = reinterpret_cast<const float4*>(a.Wp + a.w_off[la]);
    const float4* __restrict__ wp = Wl + ((int64_t)rt * (C * Q) + (int64_t)a.n_short * Q) * 64 + wlane;
    float4 ring[4];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) ring[s3] = wp[s3 * 128];
    float* red = Xs + R * P;  // [8 waves][nl][R] partial sums (the dY buffer is free now)
    f32x4 acur[S];
#pragma unroll
    for (int I = 0; I < S; ++I) acur[I] = lds4(ylane + 16 * I * P);
    for (int s = 0; s < nl; ++s) {
      f32x4 Z[S];
#pragma unroll
      for (int I = 0; I < S; ++I) Z[I] = splat4(0.f);
      lds_cptr xb = ylane;
#pragma unroll 1
      for (int q0 = 0; q0 < Q16; q0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ring[(u + 3) & 3] = wp[(u + 3) * 128];
          f32x4 anext[S];
          // (the read one step past the channel's last wraps to k = 0: the next channel's first)
          const lds_cptr xn = (q0 + u + 1 == Q16) ? ylane : xb + 16 * (u + 1);
#pragma unroll
          for (int I = 0; I < S; ++I) anext[I] = lds4(xn + 16 * I * P);
          const float4 bv = ring[u];
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][0], bv.x, Z[I]);
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][1], bv.y, Z[I]);
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][2], bv.z, Z[I]);
#pragma unroll
          for (int I = 0; I < S; ++I) Z[I] = mfma16(acur[I][3], bv.w, Z[I]);
#pragma unroll
          for (int I = 0; I < S; ++I) acur[I] = anext[I];
#pragma unroll
          for (int g = 0; g < S; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g == S - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
        wp += 4 * 128;
        xb += 64;
      }
#pragma unroll
      for (int I = 0; I < S; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = row16_sum(Pb[I][r] * Z[I][r]);
          if (j == 0) red[(wave * nl + s) * R + 16 * I + 4 * kq + r] = v;
        }
    }
    __syncthreads();
    // ---- sum over the waves (fixed order) and store dG[la][mol][k][s]
    for (int idx = tid; idx < nl * R; idx += 512) {
      const int s = idx / R, rho = idx - s * R;
      const int own = rowinfo[rho];
      if (own >= 0) {
        const int k = rho - mstart[own];
        if (k < K) {
          float v = 0.0f;
          for (int w = 0; w < 8; ++w) v += red[(w * nl + s) * R + rho];
          a.dgains[(((int64_t)la * B + mid[own]) * K + k) * nl + s] = v;
        }
      }
    }
    __syncthreads();
  }
}

// -----------------------------------------------------------------------------------------
// The message pass on strips (training, lnz_lanczosnet_messages; conv_forward.hip's MODE 2 on the
// tile plan): every channel's M_c X_l of ONE conv layer, written in the compact row numbering of the
// message matrix — the reference's cat(msg) (model/lanczos_net.py:164-180), from which dW_l =
// dY_l^T msg_l is one GEMM.  No GEMM1 and no node-state buffers: wave w loads its 16 columns of X_l
// straight into C/D order (register q of subtile I = row 16 I + 4 kq + q), projects once (Y = V^T X),
// and per channel either scales Y by the gains and lifts it (long scales) or multiplies with the
// Laplacian blocks (edge types); the products are the forward's block loops.  FK 0: diagonal gains;
// FK 2: dense K x K filters in eigen space (AdaLanczosNet: V DD_s V^T X).  SHORT: short-diffusion
// channels L_0^p X in front (channel order [short][long][edge], as everywhere).
template <int S, int FK, bool SHORT>
__device__ __forceinline__ void strip_messages(KArgs& a, const int32_t* __restrict__ ent, float* lds,
                                               const int tid, const int wave) {
  constexpr int R = 16 * S;
  constexpr unsigned OOB = 0x80000000u;
  const int lane = tid & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int N = a.N, K = a.K, B = a.B;
  constexpr bool DIAG = FK == 0;
  const int ns = SHORT ? a.n_short : 0, nl = a.n_long, ne = a.n_edge, C = ns + nl + ne;
  const int la = a.msg_layer;
  const int d = la == 0 ? a.din0 : 128;
  const float* __restrict__ src = la == 0 ? a.x0 : a.act + (int64_t)(la - 1) * B * 32 * 128;
  float* Vb = lds;                                   // [S][3][16][VBP]
  float* Gs = Vb + S * 3 * 16 * VBP;                 // [nl][R]
  int* rowinfo = reinterpret_cast<int*>(Gs + nl * R);
  int* mstart = rowinfo + R;
  int* mext = mstart + MAXMOL;
  int* mid = mext + MAXMOL;
  int* drow = mid + MAXMOL + 8;                      // [R]
  constexpr int OP = 132;                            // row pitch of the staged channel tile
  float* Os = reinterpret_cast<float*>(drow + R);    // [2][R][OP]
  const int nm = ent[0];
  if (tid < MAXMOL) {
    const bool in = tid < nm;
    mid[tid] = in ? ent[2 + 3 * tid] : -1;
    mstart[tid] = in ? ent[3 + 3 * tid] : 0;
    mext[tid] = in ? ent[4 + 3 * tid] : 0;
  }
  __syncthreads();
  if (tid < R) {
    int own = -1;
    for (int i = 0; i < nm; ++i) {
      const int n = mext[i];
      const int rows = n <= 4 ? 4 : (n + 3) & ~3;
      if (tid >= mstart[i] && tid < mstart[i] + rows) own = i;
    }
    rowinfo[tid] = own;
  }
  __syncthreads();
  for (int idx = tid; idx < S * 3 * 256; idx += 512) {
    const int Jn = idx / 768, rem = idx - Jn * 768;
    const int dd = rem >> 8, nu = (rem >> 4) & 15, ro = rem & 15;
    const int nrow = 16 * Jn + nu, srow = 16 * (Jn + dd - 1) + ro;
    float v = 0.0f;
    if (srow >= 0 && srow < R) {
      const int own = rowinfo[nrow];
      if (own >= 0 && rowinfo[srow] == own) {
        const int lnode = nrow - mstart[own], k = srow - mstart[own];
        if (lnode < N && k < K) v = finite_or_zero(a.V[((int64_t)mid[own] * N + lnode) * K + k]);
      }
    }
    Vb[((Jn * 3 + dd) * 16 + nu) * VBP + ro] = v;
  }
  for (int idx = tid; idx < (DIAG ? nl * R : 0); idx += 512) {   // this layer's gains by slot row
    const int sc = idx / R, rho = idx - sc * R;
    const int own = rowinfo[rho];
    float g = 0.0f;
    if (own >= 0) {
      const int k = rho - mstart[own];
      if (k < K && k < mext[own]) g = a.G[(((int64_t)la * B + mid[own]) * nl + sc) * K + k];
    }
    Gs[idx] = g;
  }
  // per-lane addressing of the packed Laplacian (as in the forward)
  unsigned loff[S][3];
  unsigned doff[DIAG ? 1 : S][3];  // FK 2: fragment (I, J) of a dense filter (as in the forward)
  const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.Lp), 0, B * ne * 4096, 0x00020000);
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.G), 0, (!DIAG && nl > 0) ? a.num_layer * B * nl * K * K * 4 : 0, 0x00020000);
#pragma unroll
  for (int I = 0; I < S; ++I) {
    const int row = 16 * I + j;
    const int own = rowinfo[row];
    const int st = own >= 0 ? mstart[own] : 0;
    const int mol = own >= 0 ? mid[own] : 0;
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
      const int J = I + dd - 1;
      unsigned off = OOB, dof = OOB;
      if (J >= 0 && J < S) {
        const int cg = 16 * J + 4 * kq;
        const int c = cg - st;
        if (own >= 0 && rowinfo[cg] == own) {
          off = (unsigned)((mol * ne * 256 + (c >> 3) * 64 + ((c >> 2) & 1) * 32 + (row - st)) * 16);
          if (row - st < K && c < K) dof = (unsigned)((((mol * nl) * K + (row - st)) * K + c) * 4);
        }
      }
      loff[I][dd] = off;
      if constexpr (!DIAG) doff[I][dd] = dof;
    }
  }
  // this wave's 16 columns of X_l in C/D order
  const int col = 16 * wave + j;
  const bool active = 16 * wave < d;
  f32x4 Xc[S];
#pragma unroll
  for (int I = 0; I < S; ++I) {
    const int row0 = 16 * I + 4 * kq;
    const int own = rowinfo[row0];
    Xc[I] = splat4(0.f);
    if (own >= 0 && active) {
      const float* p = src + ((int64_t)mid[own] * 32 + (row0 - mstart[own])) * d + col;
#pragma unroll
      for (int q = 0; q < 4; ++q) Xc[I][q] = p[q * d];
    }
  }
  // where a strip row's message goes: its row of the (compact) message matrix, or -1
  if (tid < R) {
    const int own = rowinfo[tid];
    int dst = -1;
    if (own >= 0) {
      const int lrow = tid - mstart[own];
      if (lrow < (a.row_off ? mext[own] : 32))
        dst = a.row_off ? (int)a.row_off[mid[own]] + lrow : mid[own] * 32 + lrow;
    }
    drow[tid] = dst;
  }
  __syncthreads();
  const int64_t ld = (int64_t)C * d;
  // A channel leaves through LDS: a wave holds 16 columns of a row (64 bytes) — written from the
  // registers, half cache lines of a 124 MB stream arrive at different times (measured: the message
  // pass 3.5 x slower than the tile kernel's 128-byte rows) — so the tile is staged (two buffers, one
  // barrier per channel) and leaves as whole rows, 16 bytes per lane.
  int chan = 0;
  auto emit = [&](const f32x4 (&out)[S], bool have) {
    float* os = Os + (chan & 1) * R * OP;
    if (have) {
#pragma unroll
      for (int I = 0; I < S; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) os[(16 * I + 4 * kq + q) * OP + col] = out[I][q];
    }
    __syncthreads();   // (also: every thread is through with the stores out of the other buffer's previous use)
    const int d4 = d >> 2;
    for (int idx = tid; idx < R * d4; idx += 512) {
      const int r = idx / d4, c4 = idx - r * d4;
      const int dst = drow[r];
      if (dst >= 0)
        *reinterpret_cast<float4*>(a.msg + (int64_t)dst * ld + (int64_t)chan * d + 4 * c4) =
            *reinterpret_cast<const float4*>(os + r * OP + 4 * c4);
    }
    ++chan;
  };
  // ---- short-diffusion channels: L_0^p X_l
  if constexpr (SHORT) {
    f32x4 mop[S][3];
    if (active) {
#pragma unroll
      for (int I = 0; I < S; ++I)
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
          if (I + dd - 1 < 0 || I + dd - 1 >= S) continue;
          mop[I][dd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(l_rsrc, loff[I][dd], 0, 0));
        }
    }
    for (int c = 0; c < ns; ++c) {
      f32x4 Zc[S];
#pragma unroll
      for (int I = 0; I < S; ++I) Zc[I] = Xc[I];
      const int p = a.short_dist[c];
      if (active) {
        for (int rep = 0; rep < p; ++rep) {
          f32x4 Zn[S];
#pragma unroll
          for (int I = 0; I < S; ++I) Zn[I] = splat4(0.f);
#pragma unroll
          for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int I = 0; I < S; ++I) {
                const int J = I + dd - 1;
                if (J < 0 || J >= S) continue;
                Zn[I] = mfma16(mop[I][dd][r], Zc[J][r], Zn[I]);
              }
#pragma unroll
          for (int I = 0; I < S; ++I) Zc[I] = Zn[I];
        }
      }
      emit(Zc, active);
    }
  }
  // ---- long scales: V diag(g_s) V^T X_l  (FK 2: V DD_s V^T X_l)
  if (nl > 0) {
    f32x4 Y[S];
#pragma unroll
    for (int I = 0; I < S; ++I) Y[I] = splat4(0.f);
    f32x4 vl[S][3];
    if (active) {
#pragma unroll
      for (int dd = 0; dd < 3; ++dd)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int I = 0; I < S; ++I) {  // slot subtile I, node subtile J
            const int J = I + dd - 1;
            if (J < 0 || J >= S) continue;
            Y[I] = mfma16(Vb[((J * 3 + (2 - dd)) * 16 + 4 * kq + r) * VBP + j], Xc[J][r], Y[I]);
          }
#pragma unroll
      for (int I = 0; I < S; ++I)
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
          if (I + dd - 1 < 0 || I + dd - 1 >= S) continue;
          vl[I][dd] = *reinterpret_cast<const f32x4*>(&Vb[((I * 3 + dd) * 16 + j) * VBP + 4 * kq]);
        }
    }
    for (int s = 0; s < nl; ++s) {
      f32x4 out[S];
      if (active) {
        f32x4 T[S];
        if constexpr (DIAG) {
#pragma unroll
          for (int I = 0; I < S; ++I) T[I] = *reinterpret_cast<const f32x4*>(Gs + s * R + 16 * I + 4 * kq) * Y[I];
        } else {
          f32x4 dd_[S][3];
#pragma unroll
          for (int I = 0; I < S; ++I) {
            T[I] = splat4(0.f);
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
              if (I + dd - 1 < 0 || I + dd - 1 >= S) continue;
              dd_[I][dd] = __builtin_bit_cast(
                  f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, doff[DIAG ? 0 : I][dd],
                                                               (la * B * nl + s) * K * K * 4, 0));
            }
          }
#pragma unroll
          for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int I = 0; I < S; ++I) {
                const int J = I + dd - 1;
                if (J < 0 || J >= S) continue;
                T[I] = mfma16(dd_[I][dd][r], Y[J][r], T[I]);
              }
        }
#pragma unroll
        for (int I = 0; I < S; ++I) out[I] = splat4(0.f);
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int I = 0; I < S; ++I) {
              const int J = I + dd - 1;
              if (J < 0 || J >= S) continue;
              out[I] = mfma16(vl[I][dd][r], T[J][r], out[I]);
            }
      }
      emit(out, active);
    }
  }
  // ---- edge types: M_e X_l
  for (int e = 0; e < ne; ++e) {
    f32x4 out[S];
    if (active) {
      f32x4 mop[S][3];
#pragma unroll
      for (int I = 0; I < S; ++I) {
        out[I] = splat4(0.f);
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
          if (I + dd - 1 < 0 || I + dd - 1 >= S) continue;
          mop[I][dd] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(l_rsrc, loff[I][dd], e * 4096, 0));
        }
      }
#pragma unroll
      for (int dd = 0; dd < 3; ++dd)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int I = 0; I < S; ++I) {
            const int J = I + dd - 1;
            if (J < 0 || J >= S) continue;
            out[I] = mfma16(mop[I][dd][r], Xc[J][r], out[I]);
          }
    }
    emit(out, active);
  }
}

template <int FK, bool SHORT>
__global__ __launch_bounds__(512) void lanczosnet_strip_messages_kernel(const lnz_forward_args) {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float lds_strip[];
  if ((int)blockIdx.x >= *a.n_strips) return;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t* ent = a.strips + (int64_t)blockIdx.x * LNZ_STRIP_INTS;
  const int sub = __builtin_amdgcn_readfirstlane(ent[1]);
  switch (sub) {
    case 1: strip_messages<1, FK, SHORT>(a, ent, lds_strip, tid, wave); break;
    case 2: strip_messages<2, FK, SHORT>(a, ent, lds_strip, tid, wave); break;
    case 3: strip_messages<3, FK, SHORT>(a, ent, lds_strip, tid, wave); break;
    case 4: strip_messages<4, FK, SHORT>(a, ent, lds_strip, tid, wave); break;
    case 5: strip_messages<5, FK, SHORT>(a, ent, lds_strip, tid, wave); break;
    case 6: strip_messages<6, FK, SHORT>(a, ent, lds_strip, tid, wave); break;
    default: break;
  }
}

__global__ __launch_bounds__(512) void lanczosnet_strip_gain_grad_kernel(const lnz_forward_args) {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float lds_strip[];
  if ((int)blockIdx.x >= *a.n_strips) return;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t* ent = a.strips + (int64_t)blockIdx.x * LNZ_STRIP_INTS;
  const int sub = __builtin_amdgcn_readfirstlane(ent[1]);
  switch (sub) {
    case 1: strip_gain_grad<1>(a, ent, lds_strip, tid, wave); break;
    case 2: strip_gain_grad<2>(a, ent, lds_strip, tid, wave); break;
    case 3: strip_gain_grad<3>(a, ent, lds_strip, tid, wave); break;
    case 4: strip_gain_grad<4>(a, ent, lds_strip, tid, wave); break;
    case 5: strip_gain_grad<5>(a, ent, lds_strip, tid, wave); break;
    case 6: strip_gain_grad<6>(a, ent, lds_strip, tid, wave); break;
    default: break;
  }
}

// One workgroup = 8 waves on one strip of the plan.
template <int MODE, int FK, bool SHORT, bool HALF = false>
__global__ __launch_bounds__(512) void lanczosnet_strip_kernel(const lnz_forward_args) {
  KArgs& a = *(KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) float lds_strip[];
  if ((int)blockIdx.x >= *a.n_strips) return;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t* ent = a.strips + (int64_t)blockIdx.x * LNZ_STRIP_INTS;
  const int sub = __builtin_amdgcn_readfirstlane(ent[1]);
  switch (sub) {
    case 1: strip_forward<1, MODE, FK, SHORT, HALF>(a, ent, lds_strip, tid, wave); break;
    case 2: strip_forward<2, MODE, FK, SHORT, HALF>(a, ent, lds_strip, tid, wave); break;
    case 3: strip_forward<3, MODE, FK, SHORT, HALF>(a, ent, lds_strip, tid, wave); break;
    case 4: strip_forward<4, MODE, FK, SHORT, HALF>(a, ent, lds_strip, tid, wave); break;
    case 5: strip_forward<5, MODE, FK, SHORT, HALF>(a, ent, lds_strip, tid, wave); break;
    case 6: strip_forward<6, MODE, FK, SHORT, HALF>(a, ent, lds_strip, tid, wave); break;
    default: break;
  }
}

}  // namespace

namespace lnz {

// The forward (mode 0, with or without the activation store) and the input-gradient pass (mode 1)
// of a width-128 model on strips: diagonal gains or dense filters in eigen space.
bool strip_forward_eligible(const lnz_forward_args& a, int mode) {
  if (mode != 0 && mode != 1) return false;
  if (!a.strips || !a.n_strips || a.strip_cap <= 0) return false;
  if (a.filter_kind != 0 && a.filter_kind != 1) return false;
  // gemm_mode 1 (split-precision GEMM1): the inference forward with diagonal gains
  // (no short-diffusion channels there: that instantiation spills 36 registers, and no configuration of
  // the reference with diagonal gains has them)
  if (a.gemm_mode == 1 ? (mode != 0 || a.act_out || a.filter_kind != 0 || a.din0 != 128 || a.n_short != 0)
                       : a.gemm_mode != 0)
    return false;
  if (a.dhid != 128 || a.din0 % 64 != 0 || a.din0 > 128) return false;
  if (mode == 1 && (a.din0 != 128 || a.bwd_din0 % 16 != 0)) return false;
  // dbias_part is indexed by strip here: a strip beyond its entries (dbias_part_cap, or the tile
  // plan's 2 * plan_wg_cap) would drop its bias-gradient partial, so such a launch stays on the tile
  // kernel
  if (mode == 1 && a.dbias_part &&
      a.strip_cap > (a.dbias_part_cap > 0 ? a.dbias_part_cap : (a.plan ? 2 * a.plan_wg_cap : 0)))
    return false;
  if (a.n_short + a.n_long + a.n_edge > 32 || a.n_edge < 1 || a.n_long > 12 || a.dout > 31) return false;
  if (a.filter_kind == 1 && a.K % 4 != 0) return false;
  if ((int64_t)a.B * a.n_edge * 4096 >= (1ll << 31)) return false;
  const int64_t per_slot = a.filter_kind == 1 ? (int64_t)a.K * a.K : a.K;
  if ((int64_t)a.num_layer * a.B * a.n_long * per_slot * 4 >= (1ll << 31)) return false;
  return (size_t)strip_lds_floats(LNZ_STRIP_SUB, a.n_long, a.gemm_mode == 1) * sizeof(float) <= 160 * 1024;
}

// lnz_lanczosnet_gain_grad on strips (the arguments have passed that entry point's checks)
bool strip_gain_grad_eligible(const lnz_forward_args& a) {
  if (!a.strips || !a.n_strips || a.strip_cap <= 0) return false;
  if (a.din0 % 64 != 0 || a.n_long < 1 || a.n_long > 12) return false;
  // the eight waves' partial sums of a strip fit in its dY buffer
  return 8 * a.n_long <= P;
}

// lnz_lanczosnet_messages on strips (the arguments have passed that entry point's checks)
bool strip_messages_eligible(const lnz_forward_args& a) {
  if (!a.strips || !a.n_strips || a.strip_cap <= 0) return false;
  if (a.gemm_mode != 0 || (a.filter_kind != 0 && a.filter_kind != 1) || a.dhid != 128) return false;
  if (a.din0 % 16 != 0 || a.din0 > 128 || a.n_edge < 1 || a.n_long > 16) return false;
  if (a.filter_kind == 1 && a.K % 4 != 0) return false;
  const int64_t per_slot = a.filter_kind == 1 ? (int64_t)a.K * a.K : a.K;
  if ((int64_t)a.num_layer * a.B * a.n_long * per_slot * 4 >= (1ll << 31)) return false;
  return (int64_t)a.B * a.n_edge * 4096 < (1ll << 31);
}

int launch_strip_messages(const lnz_forward_args& a, hipStream_t s) {
  const size_t bytes = (size_t)(LNZ_STRIP_SUB * 3 * 16 * VBP + a.n_long * 16 * LNZ_STRIP_SUB +
                                2 * 16 * LNZ_STRIP_SUB + 3 * MAXMOL + 8 + 2 * 16 * LNZ_STRIP_SUB * 132) * sizeof(float);
  // per launch: the attribute is per device, and a process may drive several (DataParallel)
  const void* fns[4] = {(const void*)lanczosnet_strip_messages_kernel<0, false>,
                        (const void*)lanczosnet_strip_messages_kernel<0, true>,
                        (const void*)lanczosnet_strip_messages_kernel<2, false>,
                        (const void*)lanczosnet_strip_messages_kernel<2, true>};
  const void* fn = fns[(a.filter_kind == 0 ? 0 : 2) + (a.n_short > 0 ? 1 : 0)];
  LNZ_DYNAMIC_LDS(fn, 160 * 1024, "conv_strip.hip");
  lnz_forward_args args = a;
  void* params[] = {&args};
  (void)hipLaunchKernel(fn, dim3(a.strip_cap), dim3(512), params, bytes, s);
  return check_launch("lnz_lanczosnet_messages (strips)");
}

int launch_strip_gain_grad(const lnz_forward_args& a, hipStream_t s) {
  const size_t bytes = (size_t)strip_lds_floats(LNZ_STRIP_SUB, a.n_long) * sizeof(float);
  // per launch: the attribute is per device, and a process may drive several (DataParallel)
  LNZ_DYNAMIC_LDS(lanczosnet_strip_gain_grad_kernel, 160 * 1024, "conv_strip.hip");
  hipLaunchKernelGGL(lanczosnet_strip_gain_grad_kernel, dim3(a.strip_cap), dim3(512), bytes, s, a);
  return check_launch("lnz_lanczosnet_gain_grad (strips)");
}

int launch_strip_forward(const lnz_forward_args& a, int mode, hipStream_t s) {
  const size_t bytes = (size_t)strip_lds_floats(LNZ_STRIP_SUB, a.n_long, a.gemm_mode == 1) * sizeof(float);
  const void* fns[8] = {
      (const void*)lanczosnet_strip_kernel<0, 0, false>, (const void*)lanczosnet_strip_kernel<1, 0, false>,
      (const void*)lanczosnet_strip_kernel<0, 2, false>, (const void*)lanczosnet_strip_kernel<1, 2, false>,
      (const void*)lanczosnet_strip_kernel<0, 0, true>,  (const void*)lanczosnet_strip_kernel<1, 0, true>,
      (const void*)lanczosnet_strip_kernel<0, 2, true>,  (const void*)lanczosnet_strip_kernel<1, 2, true>};
  const int which = (a.n_short > 0 ? 4 : 0) + (a.filter_kind == 0 ? 0 : 2) + mode;
  const void* fn = fns[which];
  if (a.gemm_mode == 1) fn = (const void*)lanczosnet_strip_kernel<0, 0, false, true>;
  note_kernel("lanczosnet_strip_kernel<%d,%d,%s,%s>", mode, a.filter_kind == 0 ? 0 : 2,
              a.n_short > 0 && a.gemm_mode != 1 ? "true" : "false", a.gemm_mode == 1 ? "true" : "false");
  // per launch: the attribute is per device, and a process may drive several (DataParallel)
  LNZ_DYNAMIC_LDS(fn, 160 * 1024, "conv_strip.hip");
  lnz_forward_args args = a;
  void* params[] = {&args};
  (void)hipLaunchKernel(fn, dim3(a.strip_cap), dim3(512), params, bytes, s);
  return check_launch(mode == 0 ? "lnz_lanczosnet_forward (strips)" : "lnz_lanczosnet_input_grad (strips)");
}

}  // namespace lnz
