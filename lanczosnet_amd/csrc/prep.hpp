// Device bodies of the batch-preparation byte movers (Laplacian pack, batch plan), shared by the
// standalone kernels of pack.hip and the fused preparation launch of lanczos_ritz.hip.
#pragma once
#include "common.hpp"

// ---------------------------------------------------------------------------------------
// Lp[b][c][g][lane][u] = L[b][lane & 31][8 g + 4 (lane >> 5) + u][c]   (zero beyond N)
// The source is channels-last (stride_ch = 1 for the collate layout): a workgroup stages
// one molecule's whole [N, N, C] block with fully coalesced reads into LDS, then writes
// the C packed tiles with coalesced float4 stores — each HBM byte is touched once.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_laplacian_body(
    const float* __restrict__ L, int64_t sb, int64_t sr, int64_t sc, int64_t sch, int N, int C,
    float4* __restrict__ Lp, float* tile, const int b,  // tile: LDS [N*N*C], source order if dense
    uint32_t* __restrict__ ident = nullptr) {
  __shared__ unsigned not_ident;
  if (threadIdx.x == 0) not_ident = 0u;
  const float* Lb = L + (int64_t)b * sb;
  const bool dense_cl = (sch == 1 && sc == C && sr == (int64_t)N * C);
  const int total = N * N * C;
  if (dense_cl) {
    for (int i = threadIdx.x; i < total; i += blockDim.x) tile[i] = Lb[i];
  } else {
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      int c = i % C;
      int m = (i / C) % N;
      int r = i / (C * N);
      tile[i] = Lb[r * sr + m * sc + c * sch];
    }
  }
  __syncthreads();
  if (ident) {
    // channel c is an IDENTITY on this molecule when its tile is diag(0/1) — a bond type the
    // molecule does not contain (L4 of the empty graph) — and the forward adds Z instead of M Z
    unsigned bad = 0u;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int c = i % C, m = (i / C) % N, r = i / (C * N);
      const float v = tile[i];
      const bool ok = v == 0.0f || (r == m && v == 1.0f);
      bad |= ok ? 0u : (1u << c);
    }
    if (bad) atomicOr(&not_ident, bad);
    __syncthreads();
    if (threadIdx.x == 0) ident[b] = ~not_ident & (C >= 32 ? 0xffffffffu : ((1u << C) - 1u));
  }
  // C * 4 * 64 float4 outputs
  for (int o = threadIdx.x; o < C * 256; o += blockDim.x) {
    int lane = o & 63;
    int g = (o >> 6) & 3;
    int c = o >> 8;
    int row = lane & 31;
    int col0 = 8 * g + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int col = col0 + u;
      v[u] = (row < N && col < N) ? tile[(row * N + col) * C + c] : 0.0f;
    }
    Lp[((int64_t)b * C + c) * 256 + (o & 255)] = make_float4(v[0], v[1], v[2], v[3]);
  }
}


// ---- tile plan: pairing of small molecules + per-workgroup dealing ------------------------------
// The forward kernels work on 32-row node tiles.  Small molecules share a tile (block-diagonal
// operators): an (<= 8)-node molecule rides with a 17..24-node one (split row 8), two (<= 16)-node
// molecules split at row 16.  For a QM8-like size range that turns B molecules into ~0.74 B tiles
// (only when that shortens the busiest CU's queue: see depth_for below).
//
// A forward launch is ONE round of workgroups when B <= 4 tiles x CUs, so it lasts as long as the
// busiest CU: the plan therefore fixes the number of workgroups W (one per CU while the tiles fit
// in a single round of <= 4 per workgroup) and deals the tiles, in descending cost order, over the
// workgroups boustrophedon-wise — every workgroup gets floor or ceil of T / W tiles of balanced
// total cost (the r-th tile dealt to a workgroup sits in slot 0, 2, 1, 3: the halves fill
// alternately).  Slot s of workgroup g is plan[(4g + s) * 3 + {0,1,2}] = (molecule A, molecule B or
// -1, split row: rows < split belong to A; 32 for a single); an unused slot has A = -1.
//
// Stable counting sort on node extent in one workgroup; every molecule derives its slot from its
// rank, so the plan is deterministic.
__device__ __forceinline__ void plan_tiles_body(const uint8_t* __restrict__ mask, int B, int N,
                                                int n_cu, int allow_pairs, int wg_cap,
                                                int32_t* __restrict__ plan,
                                                int32_t* __restrict__ n_wg, int K,
                                                int32_t* __restrict__ gain_rows,
                                                int32_t* __restrict__ n_gain_rows) {
  __shared__ int cnt[LNZ_TILE + 2];
  __shared__ int cls[4];  // molecules with extent <= 8, <= 16, <= 24, <= 32 (cumulative)
  __shared__ int wcnt[16][LNZ_TILE + 2];
  const int NT = blockDim.x;  // multiple of 64, <= 1024
  __shared__ int first[LNZ_TILE + 2];  // first rank of each extent value
  __shared__ int rbase[LNZ_TILE + 2];  // first gain row of each extent value
  const int tid = threadIdx.x;
  if (tid < LNZ_TILE + 2) cnt[tid] = 0;
  for (int i = tid; i < wg_cap * 12; i += NT) plan[i] = -1;
  __syncthreads();
  auto extent = [&](int b) {  // last real node + 1 (what the forward kernels size their work by)
    int n = 0;
    for (int i = 0; i < N; ++i) n = mask[(int64_t)b * N + i] ? i + 1 : n;
    return n;
  };
  // B <= NT (one chunk): every thread owns one molecule and keeps its extent in a register
  const int n_own = tid < B ? extent(tid) : -1;
  if (n_own >= 0) atomicAdd(&cnt[n_own], 1);
  for (int b = tid + NT; b < B; b += NT) atomicAdd(&cnt[extent(b)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0, rrun = 0;
    for (int n = 0; n <= LNZ_TILE; ++n) {
      int c = n <= N ? cnt[n] : 0;
      cnt[n] = run;
      first[n] = run;
      rbase[n] = rrun;
      run += c;
      rrun += c * (n < K ? n : K);
      if ((n & 7) == 0 && n > 0) cls[n / 8 - 1] = run;
    }
    if (gain_rows) *n_gain_rows = rrun;
  }
  __syncthreads();
  const int c8 = cls[0], c16 = cls[1] - cls[0], c24 = cls[2] - cls[1], c32 = cls[3] - cls[2];
  // rounds R = ceil(T / (4 CUs)); R workgroups per CU hold floor/ceil(T / W) <= 4 tiles each
  auto n_wg_for = [&](int t) { return t < n_cu ? t : ((t + 4 * n_cu - 1) / (4 * n_cu)) * n_cu; };
  // tiles a CU runs one after the other: rounds x the fullest workgroup of a round
  auto depth_for = [&](int t) {
    const int w = n_wg_for(t);
    return ((w + n_cu - 1) / n_cu) * ((t + w - 1) / w);
  };
  int X = allow_pairs ? (c8 < c24 ? c8 : c24) : 0;             // 8|24 pairs
  int P2 = allow_pairs ? (c8 - X + c16) / 2 : 0;               // 16|16 pairs
  // A pair tile costs more than a single one (more node-row groups and eigen slots are live),
  // so pairing only pays when it takes a tile off the busiest CU's queue
  if (depth_for(B - X - P2) >= depth_for(B)) X = 0, P2 = 0;
  const int pool = c8 - X + c16;                               // remaining (<= 16)-node molecules
  const int lone = pool - 2 * P2;                              // small singles (0/1, or all)
  const int T = B - X - P2;
  const int W = n_wg_for(T);
  if (tid == 0) *n_wg = W;
  // ascending cost order: small singles, 17..24 singles, >= 25 singles, 16|16 pairs, 8|24 pairs
  const int o24 = lone, o32 = o24 + (c24 - X), oP2 = o32 + c32, oX = oP2 + P2;
  // Stable ranks (ties in batch order) so the plan — and with it the rounding of every score —
  // is a pure function of the batch: chunks of blockDim molecules, per-wave ballots per extent value.
  const int wv = tid >> 6, ln = tid & 63;
  for (int c0 = 0; c0 < B; c0 += NT) {
    const int b = c0 + tid;
    const int n = c0 == 0 ? n_own : (b < B ? extent(b) : -1);
    int lr = 0;
    for (int v = 0; v <= N; ++v) {
      const unsigned long long mk = __ballot(n == v);
      if (n == v) lr = __popcll(mk & ((1ull << ln) - 1ull));
      if (ln == 0) wcnt[wv][v] = __popcll(mk);
    }
    __syncthreads();
    int r = -1;
    if (b < B) {
      r = cnt[n] + lr;
      for (int w = 0; w < wv; ++w) r += wcnt[w][n];
    }
    __syncthreads();
    if (tid <= N) {
      int add = 0;
      for (int w = 0; w < NT / 64; ++w) add += wcnt[w][tid];
      cnt[tid] += add;
    }
    __syncthreads();
    if (b >= B) continue;
    if (gain_rows) {
      // the k < min(n, K) eigen slots of this molecule, as rows b*K + k of D, in extent-sorted
      // order: row tiles hold molecules of similar size, which finish their Lanczos/QL together
      const int c = n < K ? n : K;
      const int base = rbase[n] + (r - first[n]) * c;
      for (int k = 0; k < c; ++k) gain_rows[base + k] = b * K + k;
    }
    int tau, role, split = 32;            // role 0 = single, 1 = A of a pair, 2 = B of a pair
    if (r < X) {
      tau = oX + r, role = 1, split = 8;
    } else if (r < c8 + c16) {
      const int q = r - X;
      if (q < 2 * P2) tau = oP2 + (q >> 1), role = 1 + (q & 1), split = 16;
      else tau = q - 2 * P2, role = 0;
    } else if (r < c8 + c16 + c24) {
      const int q = r - (c8 + c16);
      if (q < X) tau = oX + q, role = 2;
      else tau = o24 + (q - X), role = 0;
    } else {
      tau = o32 + (r - (c8 + c16 + c24)), role = 0;
    }
    const int d = T - 1 - tau;  // position in descending cost order
    const int round = d / W, idx = d % W;
    const int wg = (round & 1) ? W - 1 - idx : idx;
    // rounds 0, 1, 2, 3 -> slots 0, 2, 1, 3: a workgroup's second tile goes to its other half
    // (two halves with one tile each beat one half with two), the third joins the first
    const int slot = round == 1 ? 2 : round == 2 ? 1 : round;
    int32_t* e = plan + ((int64_t)wg * 4 + slot) * 3;
    if (role == 2) {
      e[1] = b;
    } else {
      e[0] = b;
      e[2] = split;
    }
  }
}



// ---- strip plan: molecules packed at 4-row granularity into strips of 16-row subtiles -----------
// The 16 x 16-tile forward (conv_strip.hip) runs one workgroup on a STRIP of up to LNZ_STRIP_SUB
// subtiles (96 node rows).  A molecule takes ceil(n / 4) * 4 consecutive rows at a 4-aligned start
// and never spans more than two subtiles, so every operator of the strip is block diagonal with
// blocks on the subtile diagonal and its two neighbours.  Against the 32-row tiles of
// plan_tiles_body (rows of 8 | 24, 16 | 16 pairs or singles) the bench batch needs 18.8 k rows
// instead of 23.8 k — five subtiles per CU instead of six.
//
// Packing: first fit decreasing by size class (rows / 4 = 8 .. 1), a whole class at a time — the
// items of a class are identical, so a bin's capacity for the class is a count, the bins' counts
// are prefix-summed and item r of the class (stable rank: batch order) lands in the bin whose
// range holds r.  Deterministic: the plan is a pure function of the mask.  The strip height is the
// smallest number of subtiles (2 .. LNZ_STRIP_SUB) for which the batch fits R strips per CU, R =
// the rounds it needs at full height (one for the bench batch).
// strips[s * LNZ_STRIP_INTS + {0, 1}] = molecules, subtiles of strip s; + 2 + 3 i + {0, 1, 2} =
// (molecule, first row, node extent) of its i-th molecule, rows ascending.
// scratch: >= kStripScratch bytes of LDS; B <= LNZ_STRIP_MAX_B.
constexpr int kStripBins = 1024;
constexpr int kStripScratch = 7 * LNZ_STRIP_MAX_B + 5 * kStripBins + 64;

__device__ __forceinline__ int strip_place(int fill, int rows) {  // first row of the next molecule
  return ((fill & 15) + rows <= 32) ? fill : ((fill + 15) & ~15);
}

// One chunk of up to LNZ_STRIP_MAX_B molecules (mask rows of the chunk, molecule ids offset by
// mol0); returns the number of strips written at `strips`.
__device__ __forceinline__ int plan_strips_chunk(const uint8_t* __restrict__ mask, int B, int N,
                                                 int n_cu, int mol0, int32_t* __restrict__ strips,
                                                 unsigned char* scratch) {
  const int NT = blockDim.x, tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  uint8_t* ext = scratch;                                        // [B] node extent
  uint8_t* poff = ext + LNZ_STRIP_MAX_B;                         // [B] first row in its strip
  uint8_t* pslot = poff + LNZ_STRIP_MAX_B;                       // [B] position in its strip
  uint16_t* rnk = reinterpret_cast<uint16_t*>(pslot + LNZ_STRIP_MAX_B);  // [B] rank in its class
  uint16_t* pbin = rnk + LNZ_STRIP_MAX_B;                        // [B] strip
  uint16_t* pre = pbin + LNZ_STRIP_MAX_B;                        // [bins] exclusive capacity sums
  uint8_t* fill = reinterpret_cast<uint8_t*>(pre + kStripBins);  // [bins] rows in use
  uint8_t* nmol = fill + kStripBins;                             // [bins] molecules
  uint8_t* capk = nmol + kStripBins;                             // [bins] capacity for the class
  __shared__ int ccnt[9], cbase[9], wcnt[16][9], s_total, s_used, s_chunk[17];
  if (tid < 9) ccnt[tid] = 0, cbase[tid] = 0;
  if (tid == 0) s_total = 0;
  __syncthreads();
  // ---- extents, size classes, stable ranks within a class
  for (int c0 = 0; c0 < B; c0 += NT) {
    const int b = c0 + tid;
    int cl = 0;
    if (b < B) {
      int n = 0;
      for (int i = 0; i < N; ++i) n = mask[(int64_t)b * N + i] ? i + 1 : n;
      ext[b] = (uint8_t)n;
      cl = n <= 4 ? 1 : (n + 3) >> 2;
    }
    int lr = 0;
    for (int v = 1; v <= 8; ++v) {
      const unsigned long long mk = __ballot(cl == v);
      if (cl == v) lr = __popcll(mk & ((1ull << ln) - 1ull));
      if (ln == 0) wcnt[wv][v] = __popcll(mk);
    }
    __syncthreads();
    if (b < B) {
      int r = cbase[cl] + lr;
      for (int w = 0; w < wv; ++w) r += wcnt[w][cl];
      rnk[b] = (uint16_t)r;
    }
    __syncthreads();
    if (tid >= 1 && tid <= 8) {
      int add = 0;
      for (int w = 0; w < NT / 64; ++w) add += wcnt[w][tid];
      cbase[tid] += add;
    }
    __syncthreads();
  }
  if (tid == 0) {
    int t = 0;
    for (int v = 1; v <= 8; ++v) ccnt[v] = cbase[v], t += 4 * v * cbase[v];
    s_total = t;
  }
  __syncthreads();
  // rounds of n_cu strips the batch needs at full height; the strips are then made just high
  // enough for that many rounds, so that every compute unit carries the same number of subtiles
  const int total16 = (s_total + 15) >> 4;
  const int rounds = (total16 + LNZ_STRIP_SUB * n_cu - 1) / (LNZ_STRIP_SUB * n_cu);
  const int target = rounds * n_cu < kStripBins ? rounds * n_cu : kStripBins;
  int cap = (total16 + target - 1) / target;
  cap = cap < 2 ? 2 : (cap > LNZ_STRIP_SUB ? LNZ_STRIP_SUB : cap);
  for (;; ++cap) {
    for (int i = tid; i < kStripBins; i += NT) fill[i] = 0, nmol[i] = 0;
    __syncthreads();
    for (int v = 8; v >= 1; --v) {
      const int cnt = ccnt[v], rows = 4 * v;
      if (cnt == 0) continue;  // (uniform)
      // capacity of every bin for this class
      for (int i = tid; i < kStripBins; i += NT) {
        int f = fill[i], k = 0;
        for (;;) {
          const int off = strip_place(f, rows);
          if (off + rows > 16 * cap) break;
          f = off + rows;
          ++k;
        }
        capk[i] = (uint8_t)k;
      }
      __syncthreads();
      // exclusive prefix sums over the bins: chunks of 64 bins per thread group, then the chunks
      if (tid < kStripBins / 64) {
        int sum = 0;
        for (int i = 0; i < 64; ++i) sum += capk[tid * 64 + i];
        s_chunk[tid + 1] = sum;
      }
      __syncthreads();
      if (tid == 0) {
        s_chunk[0] = 0;
        for (int i = 1; i <= kStripBins / 64; ++i) s_chunk[i] += s_chunk[i - 1];
      }
      __syncthreads();
      if (tid < kStripBins / 64) {
        int run = s_chunk[tid];
        for (int i = 0; i < 64; ++i) {
          pre[tid * 64 + i] = (uint16_t)(run > 65535 ? 65535 : run);
          run += capk[tid * 64 + i];
        }
      }
      __syncthreads();
      // the class's molecules: strip by rank, row by replaying the strip's placements
      for (int b = tid; b < B; b += NT) {
        const int n = ext[b];
        if ((n <= 4 ? 1 : (n + 3) >> 2) != v) continue;
        const int r = rnk[b];
        int lo = 0, hi = kStripBins - 1;  // last bin with pre <= r
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (pre[mid] <= r) lo = mid; else hi = mid - 1;
        }
        const int k = r - pre[lo];
        int f = fill[lo], off = 0;
        for (int i = 0; i <= k; ++i) {
          off = strip_place(f, rows);
          f = off + rows;
        }
        pbin[b] = (uint16_t)lo;
        poff[b] = (uint8_t)off;
        pslot[b] = (uint8_t)(nmol[lo] + k);
      }
      __syncthreads();
      for (int i = tid; i < kStripBins; i += NT) {
        int take = cnt - (int)pre[i];
        take = take < 0 ? 0 : (take > capk[i] ? capk[i] : take);
        int f = fill[i];
        for (int k = 0; k < take; ++k) f = strip_place(f, rows) + rows;
        fill[i] = (uint8_t)f;
        nmol[i] = (uint8_t)(nmol[i] + take);
      }
      __syncthreads();
    }
    // strips in use form a prefix of the bins (first fit)
    if (tid == 0) s_used = 0;
    __syncthreads();
    for (int i = tid; i < kStripBins; i += NT)
      if (nmol[i] > 0) atomicMax(&s_used, i + 1);
    __syncthreads();
    if (s_used <= target || cap >= LNZ_STRIP_SUB) break;
    __syncthreads();
  }
  const int used = s_used;
  for (int i = tid; i < used; i += NT) {
    strips[i * LNZ_STRIP_INTS + 0] = nmol[i];
    strips[i * LNZ_STRIP_INTS + 1] = (fill[i] + 15) >> 4;
  }
  for (int b = tid; b < B; b += NT) {
    int32_t* e = strips + (int64_t)pbin[b] * LNZ_STRIP_INTS + 2 + 3 * pslot[b];
    e[0] = mol0 + b;
    e[1] = poff[b];
    e[2] = ext[b];
  }
  __syncthreads();  // (the scratch block is reused by the next chunk)
  return used;
}

// Batches beyond LNZ_STRIP_MAX_B molecules are planned chunk by chunk (the planner's working set
// lives in one workgroup's LDS); every chunk's strips are as high as ITS rows need for a whole
// number of rounds over the CUs, which for full chunks of QM8-sized molecules is the same height.
__device__ __forceinline__ void plan_strips_body(const uint8_t* __restrict__ mask, int B, int N,
                                                 int n_cu, int32_t* __restrict__ strips,
                                                 int32_t* __restrict__ n_strips,
                                                 unsigned char* scratch) {
  int total = 0;
  for (int c0 = 0; c0 < B; c0 += LNZ_STRIP_MAX_B) {
    const int nb = B - c0 < LNZ_STRIP_MAX_B ? B - c0 : LNZ_STRIP_MAX_B;
    total += plan_strips_chunk(mask + (int64_t)c0 * N, nb, N, n_cu, c0,
                               strips + (int64_t)total * LNZ_STRIP_INTS, scratch);
  }
  if (threadIdx.x == 0) *n_strips = total;
}
