// The packed Laplacian in the form the split-precision strip kernel reads (lnz_forward_args.gemm_mode
// 1): a fragment float4 — four consecutive columns of one row — becomes 4 fp16 hi pieces | 4 lo pieces
// (x = hi + lo to 22 bits) in 16 bytes at the same position of the destination — another buffer (the
// product path: the result is a float16-typed tensor, its format travels with its type) or the source
// itself (in place: a thread reads all of its fragments before it writes any).  Shared by the
// standalone launch (pack.hip) and the gains launch that carries the conversion along
// (spectral_gains.hip).
#pragma once
#include "common.hpp"

namespace lnz {

constexpr int kSplitChunk = 1024;  // float4 per block

template <int NT>
__device__ __forceinline__ void split_pack_chunk(const float4* p, float4* dst, const int64_t n4,
                                                 const int64_t chunk, const int tid) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  constexpr int PER = kSplitChunk / NT;
  const int64_t base = chunk * kSplitChunk + tid;
  float4 v[PER];
  // (all loads of the chunk in flight before the first store: one wave, nobody to hide behind)
#pragma unroll
  for (int u = 0; u < PER; ++u)
    v[u] = base + u * NT < n4 ? p[base + u * NT] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
    h8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const _Float16 h = (_Float16)x[e];
      o[e] = h;
      o[4 + e] = (_Float16)(x[e] - (float)h);
    }
    if (base + u * NT < n4) dst[base + u * NT] = __builtin_bit_cast(float4, o);
  }
}

}  // namespace lnz
