// R7 (first half): per-eigenvalue spectral filter gains
//   G[l][b][s][k] = MLP_l([D[b,k]^p_1, ..., D[b,k]^p_S])_s        model/lanczos_net.py:110-113,146-149
//
// The MLP (S -> 128 -> 128 -> 128 -> S, ReLU) is evaluated TRANSPOSED on the matrix cores:
// one wavefront owns a tile of 32 eigenvalue rows (b,k); activations are H^T [features x 32 rows].
//   H^T_{i+1} = relu(W_{i+1} H^T_i + b_{i+1})
// With v_mfma_f32_32x32x2_f32 the C/D registers of one layer (lane = row column j, register r =
// feature cd_row(r, lane>>5)) ARE the B operand of the next layer if the k-steps are taken in the
// order k(r, hh) = cd_row(r, hh) — the weight (A operand) stream is pre-packed in exactly that
// order by lnz_pack_rows_k8().  So the whole chain runs out of registers: no LDS, no barriers,
// no transposes; HBM/L2 traffic is the packed weights (209 KB per layer) and 4 B in / 4*S B out
// per row.  grid = (row tiles, conv layers): B*K/32 * L workgroups (4480 for B=1024) >> 256 CUs.
#include "common.hpp"
#include "gains_body.hpp"
#include "split_pack.hpp"
#include <stdlib.h>

namespace {

using namespace lnz_gains;

// W0p[ot][t][lane] = W0[32 ot + (lane&31)][8 (lane>>5) + t]  (zero beyond S)
__global__ void pack_w0_kernel(const float* __restrict__ W0, int S, float* __restrict__ out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= HT * 8 * 64) return;
  int lane = idx & 63, t = (idx >> 6) & 7, ot = idx >> 9;
  int f = 8 * (lane >> 5) + t;
  out[idx] = f < S ? W0[(32 * ot + (lane & 31)) * S + f] : 0.0f;
}

// A byte mover that rides along (lnz_spectral_gains_rows_split): the workgroups behind the MLP's
// (blockIdx.x >= main_x) turn the batch's packed Laplacian into the split-precision strip kernel's
// form (into `dst`: another buffer, or the pack itself), one 16 KiB chunk each — the gains launch sits between the pack and the forward
// anyway, is bound by the matrix pipe and touches no memory to speak of.
struct SplitRide {
  const float4* pack;  // NULL: nothing rides along
  float4* dst;
  int64_t n4;       // float4 words
  int main_x;       // gridDim.x of the MLP part
};
__device__ __forceinline__ bool ride_along(const SplitRide& r, const int lane) {
  if (!r.pack || (int)blockIdx.x < r.main_x) return false;
  const int64_t chunk = (int64_t)(blockIdx.x - r.main_x) * gridDim.y + blockIdx.y;
  if (chunk * lnz::kSplitChunk < r.n4) lnz::split_pack_chunk<64>(r.pack, r.dst, r.n4, chunk, lane);
  return true;
}

// rows / n_rows (optional): compact list of the (b*K + k) eigen slots that carry a Ritz pair
// (k < min(n_b, K), lnz_plan_batch) — the slots of zero-padded eigen columns are skipped: their
// gains never reach an output (the V column is zero, model/lanczos_net.py:114-117).
__global__ __launch_bounds__(64) void spectral_gains_mlp_kernel(
    const float* __restrict__ D, int R, int B, int K, DistArr dist, int S,
    const float* __restrict__ mlp_pack, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ n_rows, float* __restrict__ G, const SplitRide ride) {
  const int lane = threadIdx.x;
  if (ride_along(ride, lane)) return;
  if (rows) R = *n_rows;
  if ((int)blockIdx.x * 32 >= R) return;
  const int idx = blockIdx.x * 32 + (lane & 31);
  const bool valid = idx < R;
  const int row = valid ? (rows ? rows[idx] : idx) : 0;
  gains_mlp_tile(D, row, valid, blockIdx.y, lane, B, K, dist, S, mlp_pack, G);
}

// Two row tiles per wavefront: every weight fragment streamed from L2 feeds 8 MFMAs instead of 4
// (the one-tile kernel pulls 144 KB per 592 MFMAs through the CU's vector-memory path), four
// independent accumulator chains, one wave per SIMD (the whole 512-register file).
__global__ __launch_bounds__(64) void spectral_gains_mlp2_kernel(
    const float* __restrict__ D, int R, int B, int K, DistArr dist, int S,
    const float* __restrict__ mlp_pack, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ n_rows, float* __restrict__ G, const SplitRide ride) {
  const int lane = threadIdx.x;
  if (ride_along(ride, lane)) return;
  if (rows) R = *n_rows;
  if ((int)blockIdx.x * 64 >= R) return;
  int row[2];
  bool valid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int idx = blockIdx.x * 64 + 32 * t + (lane & 31);
    valid[t] = idx < R;
    row[t] = valid[t] ? (rows ? rows[idx] : idx) : 0;
  }
  gains_mlp_tiles<2>(D, row, valid, blockIdx.y, lane, B, K, dist, S, mlp_pack, G);
}

// non-MLP branch (model/lanczos_net.py:118-121): G[l][b][s][k] = D[b,k]^p_s for every layer
__global__ void spectral_gains_pow_kernel(const float* __restrict__ D, int B, int K, DistArr dist,
                                          int S, int num_layer, float* __restrict__ G) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)num_layer * B * S * K;
  if (idx >= total) return;
  int k = (int)(idx % K);
  int s = (int)((idx / K) % S);
  int b = (int)((idx / ((int64_t)K * S)) % B);
  G[idx] = powi(D[(int64_t)b * K + k], dist.v[s]);
}

}  // namespace

extern "C" int64_t lnz_spectral_mlp_pack_size(int S) {
  (void)S;
  return PACK_SIZE;
}

extern "C" int lnz_pack_spectral_mlp(const float* W0, const float* b0, const float* W2,
                                     const float* b2, const float* W4, const float* b4,
                                     const float* W6, const float* b6, int S, float* pack,
                                     lnz_stream_t stream) {
  LNZ_REQUIRE(W0 && b0 && W2 && b2 && W4 && b4 && W6 && b6 && pack, LNZ_EINVAL,
              "lnz_pack_spectral_mlp: null pointer");
  LNZ_REQUIRE(S >= 1 && S <= SMAX, LNZ_ENOTSUP, "lnz_pack_spectral_mlp: S=%d not in 1..%d", S,
              SMAX);
  hipLaunchKernelGGL(pack_w0_kernel, dim3((HT * 8 * 64 + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, W0, S, pack + OFF_W0);
  int rc = lnz::check_launch("lnz_pack_spectral_mlp(W0)");
  if (rc) return rc;
  if ((rc = lnz_pack_bias_rows(b0, HID, pack + OFF_B0, stream))) return rc;
  if ((rc = lnz_pack_rows_k8(W2, HID, HID, HID, pack + OFF_W2, stream))) return rc;
  if ((rc = lnz_pack_bias_rows(b2, HID, pack + OFF_B2, stream))) return rc;
  if ((rc = lnz_pack_rows_k8(W4, HID, HID, HID, pack + OFF_W4, stream))) return rc;
  if ((rc = lnz_pack_bias_rows(b4, HID, pack + OFF_B4, stream))) return rc;
  if ((rc = lnz_pack_rows_k8(W6, S, HID, HID, pack + OFF_W6, stream))) return rc;
  if ((rc = lnz_pack_bias_rows(b6, S, pack + OFF_B6, stream))) return rc;
  return LNZ_OK;
}

// All conv layers' MLP packs in ONE launch (a training step re-packs every layer after the
// optimizer update: 8 launches per layer otherwise).  grid.y = layer, one thread per pack float.
struct MlpLayerPtrs {
  const float* p[16][8];  // [layer][W0, b0, W2, b2, W4, b4, W6, b6]
};

__device__ inline float rows_k8_elem(const float* __restrict__ W, int rows, int cols, int ld,
                                     int Q, int f) {
  const int idx4 = f >> 2, u = f & 3;
  const int lane = idx4 & 63, q = (idx4 >> 6) % Q, rt = (idx4 >> 6) / Q;
  const int row = 32 * rt + (lane & 31), col = 8 * q + 4 * (lane >> 5) + u;
  return (row < rows && col < cols) ? W[(int64_t)row * ld + col] : 0.0f;
}

__device__ inline float bias_rows_elem(const float* __restrict__ b, int rows, int f) {
  const int r = f & 15, lane = (f >> 4) & 63, rt = f >> 10;
  const int row = 32 * rt + lnz::cd_row(r, lane >> 5);
  return row < rows ? b[row] : 0.0f;
}

__global__ void pack_spectral_mlp_layers_kernel(MlpLayerPtrs ptrs, int S, float* __restrict__ pack) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= PACK_SIZE) return;
  const int l = blockIdx.y;
  const float* const* pl = ptrs.p[l];
  float v = 0.0f;  // the prefetch slack behind W6 stays zero
  if (f < OFF_B0) {
    const int lane = f & 63, t = (f >> 6) & 7, ot = f >> 9;
    const int feat = 8 * (lane >> 5) + t;
    v = feat < S ? pl[0][(32 * ot + (lane & 31)) * S + feat] : 0.0f;
  } else if (f < OFF_B2) {
    v = bias_rows_elem(pl[1], HID, f - OFF_B0);
  } else if (f < OFF_B4) {
    v = bias_rows_elem(pl[3], HID, f - OFF_B2);
  } else if (f < OFF_B6) {
    v = bias_rows_elem(pl[5], HID, f - OFF_B4);
  } else if (f < OFF_W2) {
    v = bias_rows_elem(pl[7], S, f - OFF_B6);
  } else if (f < OFF_W4) {
    v = rows_k8_elem(pl[2], HID, HID, HID, HID / 8, f - OFF_W2);
  } else if (f < OFF_W6) {
    v = rows_k8_elem(pl[4], HID, HID, HID, HID / 8, f - OFF_W4);
  } else if (f < OFF_W6 + 16 * 256) {
    v = rows_k8_elem(pl[6], S, HID, HID, HID / 8, f - OFF_W6);
  }
  pack[(int64_t)l * PACK_SIZE + f] = v;
}

extern "C" int lnz_pack_spectral_mlp_layers(const float* const* ptrs, int num_layer, int S,
                                            float* pack, lnz_stream_t stream) {
  LNZ_REQUIRE(ptrs && pack && num_layer >= 1 && num_layer <= 16, LNZ_EINVAL,
              "lnz_pack_spectral_mlp_layers: bad arguments (L=%d)", num_layer);
  LNZ_REQUIRE(S >= 1 && S <= SMAX, LNZ_ENOTSUP, "lnz_pack_spectral_mlp_layers: S=%d not in 1..%d",
              S, SMAX);
  MlpLayerPtrs mp;
  for (int l = 0; l < 16; ++l)
    for (int i = 0; i < 8; ++i) {
      mp.p[l][i] = l < num_layer ? ptrs[l * 8 + i] : nullptr;
      LNZ_REQUIRE(l >= num_layer || mp.p[l][i], LNZ_EINVAL,
                  "lnz_pack_spectral_mlp_layers: null pointer (layer %d, tensor %d)", l, i);
    }
  hipLaunchKernelGGL(pack_spectral_mlp_layers_kernel, dim3((PACK_SIZE + 255) / 256, num_layer),
                     dim3(256), 0, (hipStream_t)stream, mp, S, pack);
  return lnz::check_launch("lnz_pack_spectral_mlp_layers");
}

extern "C" int lnz_spectral_gains_rows_split_to(const float* D, int B, int K, const int32_t* dist_host,
                                                int S, int num_layer, int kind, const float* mlp_pack,
                                                const int32_t* rows, const int32_t* n_rows, float* G,
                                                const float* Lp_split, uint16_t* Lp_dst, int64_t lp_floats,
                                                lnz_stream_t stream) {
  LNZ_REQUIRE(!Lp_split == !Lp_dst, LNZ_EINVAL, "lnz_spectral_gains_rows_split_to: pack and destination come together");
  LNZ_REQUIRE(!rows || n_rows, LNZ_EINVAL, "lnz_spectral_gains_rows: rows without n_rows");
  LNZ_REQUIRE(!Lp_split || (kind == 0 && lp_floats > 0 && lp_floats % 4 == 0), LNZ_EINVAL,
              "lnz_spectral_gains_rows_split: the pack rides along with the MLP launch only (kind 0), "
              "lp_floats a positive multiple of 4");
  LNZ_REQUIRE(D && dist_host && G && B > 0 && K > 0 && num_layer > 0, LNZ_EINVAL,
              "lnz_spectral_gains: bad arguments (B=%d K=%d L=%d)", B, K, num_layer);
  LNZ_REQUIRE(S >= 1 && S <= SMAX, LNZ_ENOTSUP, "lnz_spectral_gains: S=%d not in 1..%d", S, SMAX);
  DistArr dist;
  for (int i = 0; i < SMAX; ++i) dist.v[i] = i < S ? dist_host[i] : 0;
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) {
    LNZ_REQUIRE(mlp_pack, LNZ_EINVAL, "lnz_spectral_gains: kind=MLP needs mlp_pack");
    int R = B * K;
    // two row tiles per wave once the launch fills every SIMD twice over (a two-tile wave lasts
    // twice as long: on a part-filled chip the one-tile kernel finishes first)
    static const int forced = [] {
      const char* e = getenv("LNZ_GAINS_TILES");
      return e ? atoi(e) : 0;
    }();
    const bool two = forced ? forced == 2 : (int64_t)((R + 31) / 32) * num_layer >= 2048;
    SplitRide ride = {reinterpret_cast<const float4*>(Lp_split), reinterpret_cast<float4*>(Lp_dst), lp_floats / 4,
                      (R + (two ? 63 : 31)) / (two ? 64 : 32)};
    const int64_t chunks = Lp_split ? (ride.n4 + lnz::kSplitChunk - 1) / lnz::kSplitChunk : 0;
    const int64_t extra_x = (chunks + num_layer - 1) / num_layer;
    LNZ_REQUIRE(ride.main_x + extra_x < (1ll << 31), LNZ_ENOTSUP, "lnz_spectral_gains_rows_split: pack too large");
    dim3 grid((unsigned)(ride.main_x + extra_x), num_layer);
    if (two) {
      hipLaunchKernelGGL(spectral_gains_mlp2_kernel, grid, dim3(64), 0, s, D, R, B, K, dist, S,
                         mlp_pack, rows, n_rows, G, ride);
    } else {
      hipLaunchKernelGGL(spectral_gains_mlp_kernel, grid, dim3(64), 0, s, D, R, B, K, dist, S,
                         mlp_pack, rows, n_rows, G, ride);
    }
  } else {
    int64_t total = (int64_t)num_layer * B * S * K;
    hipLaunchKernelGGL(spectral_gains_pow_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0,
                       s, D, B, K, dist, S, num_layer, G);
  }
  return lnz::check_launch("lnz_spectral_gains");
}

extern "C" int lnz_spectral_gains_rows_split(const float* D, int B, int K, const int32_t* dist_host,
                                             int S, int num_layer, int kind, const float* mlp_pack,
                                             const int32_t* rows, const int32_t* n_rows, float* G,
                                             float* Lp_split, int64_t lp_floats, lnz_stream_t stream) {
  return lnz_spectral_gains_rows_split_to(D, B, K, dist_host, S, num_layer, kind, mlp_pack, rows, n_rows, G,
                                          Lp_split, reinterpret_cast<uint16_t*>(Lp_split), lp_floats, stream);
}

extern "C" int lnz_spectral_gains_rows(const float* D, int B, int K, const int32_t* dist_host,
                                       int S, int num_layer, int kind, const float* mlp_pack,
                                       const int32_t* rows, const int32_t* n_rows, float* G,
                                       lnz_stream_t stream) {
  return lnz_spectral_gains_rows_split_to(D, B, K, dist_host, S, num_layer, kind, mlp_pack, rows, n_rows, G,
                                          nullptr, nullptr, 0, stream);
}

extern "C" int lnz_spectral_gains(const float* D, int B, int K, const int32_t* dist_host, int S,
                                  int num_layer, int kind, const float* mlp_pack, float* G,
                                  lnz_stream_t stream) {
  return lnz_spectral_gains_rows(D, B, K, dist_host, S, num_layer, kind, mlp_pack, nullptr,
                                 nullptr, G, stream);
}
