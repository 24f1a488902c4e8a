// Readout of the large-graph path (reference model/lanczos_net_general.py:185-201, model/lanczos_net.py:
// 185-194): y = (W_h x + b_h) * sigmoid(w_g x + b_g) per node, mean over the graph's real nodes —
// on the last conv state X [B,N,128] in ONE pass (the library form is two thin GEMMs over the 268 MB
// state of config 5 and five elementwise / reduction launches: 0.3 ms; this is the state read once).
// One workgroup of 16 waves per graph: a wave takes every 16th row, lanes along the 128 features
// (two per lane), P + 1 <= 32 dot products per row as lane partials + one butterfly each; the masked
// sums stay in registers and meet in LDS in a fixed order (deterministic).
#include "common.hpp"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int HW = 16;   // waves per workgroup

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 64 lanes, the same bits in every lane
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp<0x141>(v);   // row_half_mirror
  v += dpp<0x140>(v);   // row_mirror
  const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (s0 + s1) + (s2 + s3);
}

template <int PT>   // outputs (P <= PT); the gate is row P of Whead
__global__ __launch_bounds__(64 * HW) void large_head_kernel(
    const float* __restrict__ X, const unsigned char* __restrict__ mask, const float* __restrict__ Whead,
    const float* __restrict__ bhead, int N, int P, float* __restrict__ score) {
  __shared__ float part[HW][PT + 1];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x2 w[PT + 1];
#pragma unroll
  for (int p = 0; p <= PT; ++p)
    w[p] = p <= P ? *reinterpret_cast<const f32x2*>(Whead + (p < P ? p : P) * 128 + 2 * lane) : f32x2{0.0f, 0.0f};
  // (row P of Whead is the gate: slot PT of w)
  if (P < PT) {
    w[PT] = w[P];
    w[P] = f32x2{0.0f, 0.0f};
  }
  float acc[PT];
#pragma unroll
  for (int p = 0; p < PT; ++p) acc[p] = 0.0f;
  float cnt = 0.0f;
  const float bg = bhead[P];
  const float* Xb = X + (int64_t)b * N * 128 + 2 * lane;
  const unsigned char* mb = mask + (int64_t)b * N;
  for (int r0 = wave; r0 < N; r0 += 8 * HW) {   // eight rows in flight (four: the same 0.093 ms for 268 MB — the butterflies, not the loads)
    f32x2 x[8];
    bool live[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + HW * u;
      live[u] = r < N && mb[r < N ? r : 0] != 0;   // (uniform)
      x[u] = r < N ? *reinterpret_cast<const f32x2*>(Xb + (int64_t)r * 128) : f32x2{0.0f, 0.0f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (!live[u]) continue;
      const float g = wave_sum(fmaf(x[u][0], w[PT][0], x[u][1] * w[PT][1])) + bg;
      const float sg = 1.0f / (1.0f + __expf(-g));
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        if (p < P) {
          const float y = wave_sum(fmaf(x[u][0], w[p][0], x[u][1] * w[p][1])) + bhead[p];
          acc[p] = fmaf(y, sg, acc[p]);
        }
      }
      cnt += 1.0f;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int p = 0; p < PT; ++p) part[wave][p] = acc[p];
    part[wave][PT] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < P) {
    float s = 0.0f, c = 0.0f;
    for (int k = 0; k < HW; ++k) s += part[k][threadIdx.x], c += part[k][PT];
    score[(int64_t)b * P + threadIdx.x] = s / c;   // (no real node: 0 / 0, as the reference's mean of nothing)
  }
}

}  // namespace

extern "C" int lnz_large_head(const float* X, const uint8_t* mask, const float* Whead,
                              const float* bhead, int B, int N, int P, float* score,
                              lnz_stream_t stream) {
  LNZ_REQUIRE(X && mask && Whead && bhead && score && B > 0 && N > 0 && P > 0, LNZ_EINVAL,
              "lnz_large_head: bad arguments");
  LNZ_REQUIRE(P <= 16, LNZ_ENOTSUP, "lnz_large_head: %d outputs > 16", P);
  LNZ_REQUIRE((((uintptr_t)X) & 7) == 0 && (((uintptr_t)Whead) & 7) == 0, LNZ_EINVAL,
              "lnz_large_head: X / Whead must be 8-byte aligned");
  if (P <= 2)
    hipLaunchKernelGGL(large_head_kernel<2>, dim3(B), dim3(64 * HW), 0, (hipStream_t)stream, X, mask, Whead,
                       bhead, N, P, score);
  else if (P <= 8)
    hipLaunchKernelGGL(large_head_kernel<8>, dim3(B), dim3(64 * HW), 0, (hipStream_t)stream, X, mask, Whead,
                       bhead, N, P, score);
  else
    hipLaunchKernelGGL(large_head_kernel<16>, dim3(B), dim3(64 * HW), 0, (hipStream_t)stream, X, mask, Whead,
                       bhead, N, P, score);
  return lnz::check_launch("lnz_large_head");
}
