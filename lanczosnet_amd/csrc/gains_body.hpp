// Device body of the spectral-gains MLP (R7a), shared by the standalone kernel of
// spectral_gains.hip and the fused batch-preparation launch of lanczos_ritz.hip.
#pragma once
#include "common.hpp"

namespace lnz_gains {

constexpr int HID = 128;           // hidden width of the reference's spectral_filter MLP
constexpr int HT = HID / 32;       // 4 feature tiles
constexpr int SMAX = 16;           // exponents supported (two k-halves of 8)
// pack layout (floats) per conv layer: scalars/biases first, then the three 128-wide weight
// streams CONTIGUOUS (W2 | W4 | W6) so the kernel reads them as one prefetched stream, then slack
// for the prefetch ring's over-read.
constexpr int OFF_W0 = 0;                       // [HT][8][64]          A scalars of Linear(S->128)
constexpr int OFF_B0 = OFF_W0 + HT * 8 * 64;    // [HT][64][16]
constexpr int OFF_B2 = OFF_B0 + HT * 1024;
constexpr int OFF_B4 = OFF_B2 + HT * 1024;
constexpr int OFF_B6 = OFF_B4 + HT * 1024;      // [1][64][16]
constexpr int OFF_W2 = OFF_B6 + 1024;           // rows_k8(128,128): [HT][16][64] float4
constexpr int OFF_W4 = OFF_W2 + HT * 16 * 256;
constexpr int OFF_W6 = OFF_W4 + HT * 16 * 256;  // rows_k8(32,128) (S rows zero padded to 32)
#ifndef LNZ_GAINS_RING
#define LNZ_GAINS_RING 8
#endif
#ifndef LNZ_GAINS_DIST
#define LNZ_GAINS_DIST (LNZ_GAINS_RING - 2)
#endif
constexpr int RING = LNZ_GAINS_RING;             // prefetch ring slots (distance LNZ_GAINS_DIST steps)
constexpr int PACK_SIZE = OFF_W6 + 16 * 256 + RING * 256;
constexpr int NSTEP = 2 * HT * 16 + 16;         // float4 steps of the W2|W4|W6 stream (144)

struct DistArr {
  int32_t v[SMAX];
};

__device__ inline float powi(float x, int p) {
  // integer power in fp64, rounded once to fp32 (torch.pow(D, ii), model/lanczos_net.py:148,
  // is a <= 1 ulp powf; this is the correctly rounded value)
  double base = (double)x, acc = 1.0;
  int e = p < 0 ? -p : p;
  while (e) {
    if (e & 1) acc *= base;
    base *= base;
    e >>= 1;
  }
  return (float)(p < 0 ? 1.0 / acc : acc);
}

__device__ inline f32x16 load_bias_frag(const float* __restrict__ bp, int lane) {
  const float4* p = reinterpret_cast<const float4*>(bp) + lane * 4;
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 v = p[g];
    acc[4 * g + 0] = v.x;
    acc[4 * g + 1] = v.y;
    acc[4 * g + 2] = v.z;
    acc[4 * g + 3] = v.w;
  }
  return acc;
}

__device__ inline f32x16 relu_bias16(f32x16 v, f32x16 b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i] + b[i], 0.0f);
  return v;
}

// One 32-row output tile of a 128-wide layer for RT row tiles at once: 16 stream steps, each
// weight fragment feeding 4 MFMAs per row tile.  F0 = index of the tile's first step in the
// W2|W4|W6 stream (compile time => static ring slots / registers).
template <int F0, int RT>
__device__ inline void dense_tile(const float4* __restrict__ wstream, float4 (&ring)[RING],
                                  const f32x16 (&Hin)[RT][HT], f32x16 (&res)[RT]) {
  // two accumulator chains per row tile (even / odd k-steps): consecutive MFMAs never depend on
  // each other
  f32x16 acc[RT], acc1[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    acc[t] = lnz::splat16(0.0f);
    acc1[t] = lnz::splat16(0.0f);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    constexpr int D = LNZ_GAINS_DIST;
#ifdef LNZ_GAINS_PROBE_NOLOAD   // probe: the MLP without its weight stream (what do the loads cost?)
    ring[(F0 + q + D) % RING] = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
#else
    ring[(F0 + q + D) % RING] = wstream[(F0 + q + D) * 64];  // over-read lands in the slack
#endif
    __builtin_amdgcn_sched_barrier(0);
    const float4 a = ring[(F0 + q) % RING];
    const int ti = q >> 2, g = q & 3;
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = lnz::mfma32(a.x, Hin[t][ti][4 * g + 0], acc[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc1[t] = lnz::mfma32(a.y, Hin[t][ti][4 * g + 1], acc1[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = lnz::mfma32(a.z, Hin[t][ti][4 * g + 2], acc[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc1[t] = lnz::mfma32(a.w, Hin[t][ti][4 * g + 3], acc1[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) res[t][i] = acc[t][i] + acc1[t][i];
  }
}

// rows / n_rows (optional): compact list of the (b*K + k) eigen slots that carry a Ritz pair
// (k < min(n_b, K), lnz_plan_batch) — the slots of zero-padded eigen columns are skipped: their
// gains never reach an output (the V column is zero, model/lanczos_net.py:114-117).
// One wavefront = RT 32-row tiles of eigen slots through the MLP of conv layer l (every weight
// fragment it streams feeds all RT tiles; a row's arithmetic does not depend on RT).  `row[t]` is
// this lane's eigen slot b*K + k in tile t (both lane halves hold the same 32 rows), `valid[t]`
// masks the ragged tail.
template <int RT>
__device__ __forceinline__ void gains_mlp_tiles(const float* __restrict__ D, const int (&row)[RT],
                                                const bool (&valid)[RT], const int l,
                                                const int lane, int B, int K, const DistArr& dist,
                                                int S, const float* __restrict__ mlp_pack,
                                                float* __restrict__ G) {
  const int hh = lane >> 5;
  const float* pk = mlp_pack + (int64_t)l * PACK_SIZE;
#ifdef LNZ_GAINS_PROBE_STAGGER   // probe: de-phase the waves that stream the same weights in lockstep
  for (int i = 0; i < (int)((blockIdx.x & (LNZ_GAINS_PROBE_STAGGER_N - 1)) * LNZ_GAINS_PROBE_STAGGER); ++i)
    __builtin_amdgcn_s_sleep(16);   // 16 x 64 clocks
#endif

  // start the 128-wide weight stream right away: it is independent of the first layer
  const float4* __restrict__ wstream = reinterpret_cast<const float4*>(pk + OFF_W2) + lane;
  float4 ring[RING];
#pragma unroll
  for (int f = 0; f < LNZ_GAINS_DIST; ++f) ring[f] = wstream[f * 64];

  const int steps0 = S < 8 ? S : 8;
  f32x16 h1[RT][HT], h2[RT][HT];
  {
    // first-layer A scalars (32 per lane) and features of this lane's k-half: f = 8 hh + t
    float w0[HT][8];
#pragma unroll
    for (int ot = 0; ot < HT; ++ot) {
#pragma unroll
      for (int t = 0; t < 8; ++t) w0[ot][t] = pk[OFF_W0 + (ot * 8 + t) * 64 + lane];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float dval = valid[rt] ? D[row[rt]] : 0.0f;
      float feat[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#ifdef LNZ_GAINS_PROBE_NOPOW    // probe: the features without the fp64 power loops
        float lo = t < S ? dval : 0.0f;
        float hi = (8 + t) < S ? dval : 0.0f;
#else
        float lo = t < S ? powi(dval, dist.v[t]) : 0.0f;
        float hi = (8 + t) < S ? powi(dval, dist.v[8 + t]) : 0.0f;
#endif
        feat[t] = hh ? hi : lo;
      }
      // Linear(S -> 128) + ReLU
#pragma unroll
      for (int ot = 0; ot < HT; ++ot) {
        f32x16 acc = lnz::splat16(0.0f);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < steps0) acc = lnz::mfma32(w0[ot][t], feat[t], acc);
        }
        h1[rt][ot] = relu_bias16(acc, load_bias_frag(pk + OFF_B0 + ot * 1024, lane));
      }
    }
  }
  // Linear(128 -> 128) + ReLU, twice (stream steps 0..63 and 64..127), then Linear(128 -> S)
  f32x16 res[RT];
#define LNZ_DENSE(F0, IN, OUT, OT, BOFF)                                   \
  {                                                                        \
    const f32x16 bb = load_bias_frag(pk + (BOFF), lane);                   \
    dense_tile<F0, RT>(wstream, ring, IN, res);                            \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) OUT[rt][OT] = relu_bias16(res[rt], bb); \
  }
  LNZ_DENSE(0, h1, h2, 0, OFF_B2 + 0 * 1024)
  LNZ_DENSE(16, h1, h2, 1, OFF_B2 + 1 * 1024)
  LNZ_DENSE(32, h1, h2, 2, OFF_B2 + 2 * 1024)
  LNZ_DENSE(48, h1, h2, 3, OFF_B2 + 3 * 1024)
  LNZ_DENSE(64, h2, h1, 0, OFF_B4 + 0 * 1024)
  LNZ_DENSE(80, h2, h1, 1, OFF_B4 + 1 * 1024)
  LNZ_DENSE(96, h2, h1, 2, OFF_B4 + 2 * 1024)
  LNZ_DENSE(112, h2, h1, 3, OFF_B4 + 3 * 1024)
#undef LNZ_DENSE
  {
    const f32x16 b6 = load_bias_frag(pk + OFF_B6, lane);
    dense_tile<128, RT>(wstream, ring, h1, res);  // no activation (stream steps 128..143)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (valid[rt]) {
        const int b = row[rt] / K, k = row[rt] - b * K;
        float* Gb = G + (((int64_t)l * B + b) * S) * K + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int sidx = lnz::cd_row(r, hh);
          if (sidx < S) Gb[(int64_t)sidx * K] = res[rt][r] + b6[r];
        }
      }
    }
  }
}

__device__ __forceinline__ void gains_mlp_tile(const float* __restrict__ D, const int row,
                                               const bool valid, const int l, const int lane,
                                               int B, int K, const DistArr& dist, int S,
                                               const float* __restrict__ mlp_pack,
                                               float* __restrict__ G) {
  const int rows1[1] = {row};
  const bool valid1[1] = {valid};
  gains_mlp_tiles<1>(D, rows1, valid1, l, lane, B, K, dist, S, mlp_pack, G);
}

}  // namespace lnz_gains
