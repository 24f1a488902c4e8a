// R1: batched L4 = D^-1/2 (I + A) D^-1/2 (simple graph + one channel per bond type), and
// R12: unsorted_segment_sum forward / backward.  Both are small HBM-bound byte movers.
#include "common.hpp"

// One workgroup per molecule.  adjs [N,N,E] channels-last is read once (coalesced) into
// registers/LDS-free form: thread t walks (i, j) pairs, all E channels of a pair are
// contiguous.  Degrees are accumulated per row in LDS in fp64 (the reference builds L4 in
// float64: utils/data_helper.py:99-114 after `np.eye(n) + adj`).
template <bool STAGED>
__global__ __launch_bounds__(256) void laplacian_l4_kernel(const float* __restrict__ adjs,
                                                           const int32_t* __restrict__ n_nodes,
                                                           int N, int E, float* __restrict__ L) {
  extern __shared__ __attribute__((aligned(16))) double sdeg[];  // [(E+1) * N] -> d^-1/2
  const int b = blockIdx.x;
  const int n = n_nodes[b];
  const int E1 = E + 1;
  const float* Ab = adjs + (int64_t)b * N * N * E;
  float* Lb = L + (int64_t)b * N * N * E1;
  // STAGED: the graph's whole [N, N, E] block goes to LDS first, by 16-byte loads that are all in
  // flight together (eight per thread and round).  Read from global memory where it is used, the
  // block was ~23 rounds of loads in series — the degree sums column by column, then the pairs —
  // at ~2 us each with one workgroup per compute unit: 43 us for a graph of 100 nodes.
  const float* As = Ab;
  if (STAGED) {
    float* stage = reinterpret_cast<float*>(sdeg + ((E1 * N + 1) & ~1));
    const int total4 = N * N * E / 4;   // (the launcher checks the divisibility and the alignment)
    const float4* src = reinterpret_cast<const float4*>(Ab);
    for (int base = threadIdx.x; base < total4; base += 8 * blockDim.x) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = base + u * blockDim.x;
        v[u] = src[q < total4 ? q : total4 - 1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = base + u * blockDim.x;
        if (q < total4) reinterpret_cast<float4*>(stage)[q] = v[u];
      }
    }
    __syncthreads();
    As = stage;
  }
  // degree of row i in channel ch (ch 0 = simple graph = sum over bond types) of (I + A)
  for (int t = threadIdx.x; t < E1 * N; t += blockDim.x) {
    int i = t % N, ch = t / N;
    double deg = 1.0;  // the identity's diagonal
    if (i < n) {
      for (int j = 0; j < n; ++j) {
        const float* a = As + ((int64_t)i * N + j) * E;
        if (ch == 0) {
          for (int e = 0; e < E; ++e) deg += (double)a[e];
        } else {
          deg += (double)a[ch - 1];
        }
      }
    }
    double s = 1.0 / sqrt(deg);  // rowsum ** -0.5 ; deg >= 1 - sum|negatives|, inf -> 0 guard
    if (!(deg > 0.0) ) s = 0.0;  // utils/data_helper.py:106 (inf -> 0); negative degrees -> nan in ref
    sdeg[t] = s;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < N * N; p += blockDim.x) {
    int i = p / N, j = p % N;
    const float* a = As + (int64_t)p * E;
    float* out = Lb + (int64_t)p * E1;
    if (i >= n || j >= n) {
      for (int ch = 0; ch < E1; ++ch) out[ch] = 0.0f;  // dataset/qm8.py:225-260 zero padding
      continue;
    }
    double id = (i == j) ? 1.0 : 0.0;
    double asum = 0.0;
    for (int e = 0; e < E; ++e) {
      double v = (double)a[e];
      asum += v;
      // np.diag(r).dot(A).dot(np.diag(r)): (s_i * a_ij) * s_j
      out[1 + e] = (float)((sdeg[(1 + e) * N + i] * (id + v)) * sdeg[(1 + e) * N + j]);
    }
    out[0] = (float)((sdeg[i] * (id + asum)) * sdeg[j]);
  }
}

extern "C" int lnz_laplacian_l4(const float* adjs, const int32_t* n_nodes, int B, int N, int E,
                                float* L, lnz_stream_t stream) {
  LNZ_REQUIRE(adjs && n_nodes && L && B > 0 && N > 0 && E > 0, LNZ_EINVAL,
              "lnz_laplacian_l4: bad arguments (B=%d N=%d E=%d)", B, N, E);
  size_t lds = (size_t)(E + 1) * N * sizeof(double);
  LNZ_REQUIRE(lds <= 64 * 1024, LNZ_ENOTSUP, "lnz_laplacian_l4: (E+1)*N too large");
  // the adjacency block of a graph staged in LDS when it fits (next to the degrees, 64 KB in all)
  // and every graph's block starts on a 16-byte boundary
  const size_t staged = (((size_t)(E + 1) * N + 1) & ~(size_t)1) * sizeof(double) +
                        (size_t)N * N * E * sizeof(float);
  if (staged <= 64 * 1024 && ((size_t)N * N * E) % 4 == 0 && (((uintptr_t)adjs) & 15) == 0) {
    hipLaunchKernelGGL(laplacian_l4_kernel<true>, dim3(B), dim3(256), staged, (hipStream_t)stream,
                       adjs, n_nodes, N, E, L);
  } else {
    hipLaunchKernelGGL(laplacian_l4_kernel<false>, dim3(B), dim3(256), lds, (hipStream_t)stream, adjs,
                       n_nodes, N, E, L);
  }
  return lnz::check_launch("lnz_laplacian_l4");
}

// ---------------------------------------------------------------------------------------
// R1, every kind of utils/data_helper.py:119-166 (get_laplacian 'L1' .. 'L7' on normalize_adj
// :92-116), same batch layout as the L4 kernel: channel 0 = simple graph (sum over bond types),
// channel 1 + e = bond type e.  With M = A (kinds 1-3, 6, 7) or I + A (kinds 4, 5) and d = rowsum(M):
//   L1 = diag(d) - A          L2 = I - d^-1/2 A d^-1/2      L3 = I - d^-1 A
//   L4 = d^-1/2 M d^-1/2      L5 = d^-1 M                    L6 = d^-alpha A d^-alpha     L7 = d^-1 A
// d^-x with the reference's guard: inf -> 0 (an isolated node's row is zero, :106).  fp64 inside.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void laplacian_kind_kernel(const float* __restrict__ adjs,
                                                             const int32_t* __restrict__ n_nodes,
                                                             int N, int E, int kind, double alpha,
                                                             float* __restrict__ L) {
  extern __shared__ __attribute__((aligned(16))) double sdeg[];  // [(E+1) * N]: d, then d^-x
  const int b = blockIdx.x;
  const int n = n_nodes[b];
  const int E1 = E + 1;
  const float* Ab = adjs + (int64_t)b * N * N * E;
  float* Lb = L + (int64_t)b * N * N * E1;
  const bool plus_identity = kind == 4 || kind == 5;
  const double expo = (kind == 2 || kind == 4) ? 0.5 : (kind == 6 ? alpha : 1.0);
  for (int t = threadIdx.x; t < E1 * N; t += blockDim.x) {
    int i = t % N, ch = t / N;
    double deg = plus_identity ? 1.0 : 0.0;
    if (i < n) {
      for (int j = 0; j < n; ++j) {
        const float* a = Ab + ((int64_t)i * N + j) * E;
        if (ch == 0) {
          for (int e = 0; e < E; ++e) deg += (double)a[e];
        } else {
          deg += (double)a[ch - 1];
        }
      }
    }
    double s = deg;                              // L1 keeps the degree itself
    if (kind != 1) {
      s = pow(deg, -expo);                       // np.power(rowsum, -exponent)
      if (isinf(s)) s = 0.0;                     // r_inv[np.isinf(r_inv)] = 0
    }
    sdeg[t] = s;
  }
  __syncthreads();
  const bool sym = kind == 2 || kind == 4 || kind == 6;
  for (int p = threadIdx.x; p < N * N; p += blockDim.x) {
    int i = p / N, j = p % N;
    const float* a = Ab + (int64_t)p * E;
    float* out = Lb + (int64_t)p * E1;
    if (i >= n || j >= n) {
      for (int ch = 0; ch < E1; ++ch) out[ch] = 0.0f;
      continue;
    }
    const double id = (i == j) ? 1.0 : 0.0;
    double asum = 0.0;
    for (int e = 0; e < E; ++e) asum += (double)a[e];
    for (int ch = 0; ch < E1; ++ch) {
      const double v = ch == 0 ? asum : (double)a[ch - 1];
      const double si = sdeg[ch * N + i], sj = sdeg[ch * N + j];
      double r;
      if (kind == 1) {
        r = id * si - v;                                   // np.diag(rowsum) - adj
      } else {
        const double m = plus_identity ? id + v : v;
        r = sym ? (si * m) * sj : si * m;                  // R.dot(A).dot(R) / R.dot(A)
        if (kind == 2 || kind == 3) r = id - r;
      }
      out[ch] = (float)r;
    }
  }
}

extern "C" int lnz_laplacian(const float* adjs, const int32_t* n_nodes, int B, int N, int E, int kind,
                             double alpha, float* L, lnz_stream_t stream) {
  LNZ_REQUIRE(adjs && n_nodes && L && B > 0 && N > 0 && E > 0, LNZ_EINVAL,
              "lnz_laplacian: bad arguments (B=%d N=%d E=%d)", B, N, E);
  LNZ_REQUIRE(kind >= 1 && kind <= 7, LNZ_EINVAL, "lnz_laplacian: kind %d not in 1..7", kind);
  size_t lds = (size_t)(E + 1) * N * sizeof(double);
  LNZ_REQUIRE(lds <= 64 * 1024, LNZ_ENOTSUP, "lnz_laplacian: (E+1)*N too large");
  hipLaunchKernelGGL(laplacian_kind_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, adjs,
                     n_nodes, N, E, kind, alpha, L);
  return lnz::check_launch("lnz_laplacian");
}

// ---------------------------------------------------------------------------------------
// unsorted_segment_sum.  data [B, D1, D2]; ids [B, D1]; out [B, S, D2].
//   forward : out[b, ids[b,c], x] += data[b, c, x]      (operators/src/cuda/segment_reduction.cu:39-53)
//   backward: gdata[b, c, x] = gout[b, ids[b,c], x]     (:55-69)
// The reference is one thread per ELEMENT with a global atomicAdd each.  Here a scatter-add is a
// privatised reduction: a workgroup owns (batch b, a block of VEC*64 feature columns) and every
// lane owns VEC columns of it for ALL segments, so the accumulators — [S][VEC*64] floats in LDS —
// have exactly one writer per address: plain ds_read/ds_write read-modify-write, no atomics, a
// fixed summation order (source-row order within a wave's row range: bit-reproducible, and for
// one wave identical to the CPU loop of operators/src/segment_reduction.cpp:6-30).  The segment
// id of a source row is wave uniform (one scalar load per row instead of one per element); the
// row's data is one coalesced dwordx4 (VEC = 4) or dword load per lane, software-pipelined UNR
// rows deep because consecutive rows of one segment form a dependent LDS chain.  With few
// workgroups (small B) the D1 rows are split over NW waves, each with a private accumulator
// copy, combined in wave order at the end; `out` is read-modify-written once per element, so the
// "+=" onto the caller's pre-zeroed (or not) buffer is kept.  Fallback when S*VEC*64*4 B does
// not fit LDS: the element-wise fp32 atomics (fire-and-forget L2 atomics).
// ---------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void segsum_fwd_lds_kernel(
    const float* __restrict__ data, const int64_t* __restrict__ ids, int D1, int D2, int S,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sacc[];  // [NW][S][VEC*64]
  constexpr int CW = VEC * 64, UNR = 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int b = blockIdx.y;
  const int x0 = blockIdx.x * CW + lane * VEC;
  const bool live = x0 < D2;  // D2 % VEC == 0 on the VEC = 4 path: a lane is all in or all out
  float* acc = sacc + (size_t)w * S * CW + lane * VEC;
  for (int sgm = 0; sgm < S; ++sgm)
#pragma unroll
    for (int u = 0; u < VEC; ++u) acc[(size_t)sgm * CW + u] = 0.0f;
  const int per = (D1 + NW - 1) / NW;
  const int c0 = w * per, c1 = min(D1, c0 + per);
  const int64_t* idb = ids + (int64_t)b * D1;
  const float* db = data + (int64_t)b * D1 * D2 + x0;
  for (int c = c0; c < c1; c += UNR) {
    float v[UNR][VEC];
    int sg[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      const int cc = c + k;
      int64_t s64 = cc < c1 ? idb[cc] : -1;
      sg[k] = __builtin_amdgcn_readfirstlane((s64 < 0 || s64 >= S) ? -1 : (int)s64);
      if (sg[k] >= 0 && live) {
        if constexpr (VEC == 4) {
          f32x4 t = *reinterpret_cast<const f32x4*>(db + (int64_t)cc * D2);
          v[k][0] = t[0]; v[k][1] = t[1]; v[k][2] = t[2]; v[k][3] = t[3];
        } else {
          v[k][0] = db[(int64_t)cc * D2];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      if (sg[k] < 0 || !live) continue;  // out-of-range ids are dropped, never written OOB
      float* a = acc + (size_t)sg[k] * CW;
#pragma unroll
      for (int u = 0; u < VEC; ++u) a[u] += v[k][u];
    }
  }
  __syncthreads();
  // combine the NW private copies in wave order; one read-modify-write of `out` per element
  float* ob = out + (int64_t)b * S * D2 + blockIdx.x * CW;
  const int width = min(CW, D2 - blockIdx.x * CW);
  for (int t = threadIdx.x; t < S * CW; t += blockDim.x) {
    const int sgm = t / CW, x = t % CW;
    if (x >= width) continue;
    float sum = sacc[t];
    for (int k = 1; k < NW; ++k) sum += sacc[(size_t)k * S * CW + t];
    ob[(int64_t)sgm * D2 + x] += sum;
  }
}

__global__ void segsum_fwd_atomic_kernel(const float* __restrict__ data,
                                         const int64_t* __restrict__ ids, int64_t total, int D1,
                                         int D2, int S, float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int x = (int)(idx % D2);
    int64_t bc = idx / D2;
    int b = (int)(bc / D1);
    int64_t seg = ids[bc];
    if (seg < 0 || seg >= S) continue;
    atomicAdd(out + ((int64_t)b * S + seg) * D2 + x, data[idx]);
  }
}

// Gather: one wave per source row (its id is one scalar load), VEC columns per lane.
template <int VEC>
__global__ __launch_bounds__(256) void segsum_bwd_kernel(const float* __restrict__ gout,
                                                         const int64_t* __restrict__ ids,
                                                         int64_t rows, int D1, int D2, int S,
                                                         float* __restrict__ gdata) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t bc = wave; bc < rows; bc += nwave) {
    const int64_t s64 = ids[bc];
    const int seg = __builtin_amdgcn_readfirstlane((s64 < 0 || s64 >= S) ? -1 : (int)s64);
    const int64_t b = bc / D1;
    const float* src = gout + (b * S + (seg < 0 ? 0 : seg)) * D2;
    float* dst = gdata + bc * D2;
    for (int x = lane * VEC; x < D2; x += 64 * VEC) {
      if constexpr (VEC == 4) {
        f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
        if (seg >= 0) t = *reinterpret_cast<const f32x4*>(src + x);
        *reinterpret_cast<f32x4*>(dst + x) = t;
      } else {
        dst[x] = seg >= 0 ? src[x] : 0.0f;
      }
    }
  }
}

extern "C" int lnz_unsorted_segment_sum_forward(const float* data, const int64_t* segment_ids,
                                                int B, int dim1, int dim2, int num_segments,
                                                float* out, lnz_stream_t stream) {
  LNZ_REQUIRE(data && segment_ids && out && B > 0 && dim1 > 0 && dim2 > 0 && num_segments > 0,
              LNZ_EINVAL, "lnz_unsorted_segment_sum_forward: bad arguments");
  const bool vec4 = dim2 % 4 == 0 && dim2 >= 128 && (((uintptr_t)data | (uintptr_t)out) & 15) == 0;
  const int cw = vec4 ? 256 : 64;
  const size_t copy = (size_t)num_segments * cw * sizeof(float);
  const size_t lds_max = 64 * 1024;
  if (copy <= lds_max) {
    const int blocks = (dim2 + cw - 1) / cw;
    // row-split waves only while the grid does not fill the 256 CUs and a copy each still fits
    int nw = 1;
    while (nw < 4 && (int64_t)B * blocks * nw < 512 && copy * (nw * 2) <= lds_max &&
           dim1 >= 16 * nw * 2)
      nw *= 2;
    dim3 grid(blocks, B);
    if (vec4)
      hipLaunchKernelGGL(segsum_fwd_lds_kernel<4>, grid, dim3(64 * nw), copy * nw,
                         (hipStream_t)stream, data, segment_ids, dim1, dim2, num_segments, out);
    else
      hipLaunchKernelGGL(segsum_fwd_lds_kernel<1>, grid, dim3(64 * nw), copy * nw,
                         (hipStream_t)stream, data, segment_ids, dim1, dim2, num_segments, out);
  } else {
    int64_t total = (int64_t)B * dim1 * dim2;
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(segsum_fwd_atomic_kernel, dim3((int)(g > 2048 ? 2048 : g)), dim3(256), 0,
                       (hipStream_t)stream, data, segment_ids, total, dim1, dim2, num_segments, out);
  }
  return lnz::check_launch("lnz_unsorted_segment_sum_forward");
}

extern "C" int lnz_unsorted_segment_sum_backward(const float* grad_out, const int64_t* segment_ids,
                                                 int B, int dim1, int dim2, int num_segments,
                                                 float* grad_data, lnz_stream_t stream) {
  LNZ_REQUIRE(grad_out && segment_ids && grad_data && B > 0 && dim1 > 0 && dim2 > 0 &&
                  num_segments > 0,
              LNZ_EINVAL, "lnz_unsorted_segment_sum_backward: bad arguments");
  const int64_t rows = (int64_t)B * dim1;
  const bool vec4 = dim2 % 4 == 0 && (((uintptr_t)grad_out | (uintptr_t)grad_data) & 15) == 0;
  int64_t g = (rows + 3) / 4;
  const int grid = (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
  if (vec4)
    hipLaunchKernelGGL(segsum_bwd_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out,
                       segment_ids, rows, dim1, dim2, num_segments, grad_data);
  else
    hipLaunchKernelGGL(segsum_bwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out,
                       segment_ids, rows, dim1, dim2, num_segments, grad_data);
  return lnz::check_launch("lnz_unsorted_segment_sum_backward");
}

// ---------------------------------------------------------------------------------------
// Training: the embedding table's gradient (model/lanczos_net.py:154: node_feat -> nn.Embedding)
//   dE[a][c] = sum over the node rows (b, i) with id[b][i] == a of dX0[b][i][c]
// The reference gets it from autograd's embedding backward (atomics); as one-hot^T dX0 it was a
// library GEMM with K = B N rows for a 70 x 64 output plus the one-hot matrix itself (0.13 ms of the
// step).  Here workgroup (a, chunk) scans a slice of the rows: a wave reads 64 ids at a time (lane =
// row, coalesced), then visits the matching rows one by one with lane = column and adds them in row
// order; the four waves meet in LDS: one partial per (chunk, a), summed by the caller in a fixed
// order.  No atomics, no shuffles.  (A first form — lane = row, 64 accumulators per lane, a shuffle
// tree at the end — spent 51 us in 384 ds_bpermutes per wave.)
// ---------------------------------------------------------------------------------------
template <int CPL>   // columns per lane: width = 64 CPL (CPL = 1: widths up to 64, lanes beyond it idle)
__global__ __launch_bounds__(256) void embedding_grad_kernel(
    const int64_t* __restrict__ ids, const float* __restrict__ dx, int64_t rows, int N, int64_t mol_stride,
    int64_t row_stride, int width, int num_atom, float* __restrict__ part) {
  __shared__ float red[4][64 * CPL];
  const int a = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t lo = chunk * per, hi = lo + per < rows ? lo + per : rows;
  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
  const bool mine = lane * CPL < width;
  // 64 ids per wave and step (lane = row, coalesced); the matching rows are then visited one by one
  // with lane = column(s): no cross-lane reduction at the end
  for (int64_t r0 = lo + 64 * wave; r0 < hi; r0 += 256) {
    const int64_t r = r0 + lane;
    int64_t id = r < hi ? ids[r] : -1;
    const bool hit = r < hi && (id < 0 ? 0 : (id >= num_atom ? num_atom - 1 : id)) == a;   // (the forward's clamp)
    unsigned long long m = __ballot(hit);
    while (m) {
      // up to eight matching rows in flight (padding rows carry id 0: a third of all rows for that
      // atom — one dependent load after the other made its workgroups the launch's long pole);
      // added in row order, a missing one adds 0.0
      float v[8][CPL];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool have = m != 0;
        const int k = have ? __builtin_ctzll(m) : 0;
        m &= m - 1;
        const int64_t rr = r0 + k, b = rr / N, i = rr - b * N;
        const float* __restrict__ src = dx + b * mol_stride + i * row_stride + lane * CPL;
#pragma unroll
        for (int c = 0; c < CPL; ++c) v[u][c] = (have && mine) ? src[c] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] += v[u][c];
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) red[wave][lane * CPL + c] = acc[c];
  __syncthreads();
  if (tid < width)
    part[((int64_t)chunk * num_atom + a) * width + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

extern "C" int lnz_embedding_grad(const int64_t* ids, int B, int N, const float* dx, int64_t mol_stride,
                                  int64_t row_stride, int width, int num_atom, int chunks, float* partials,
                                  lnz_stream_t stream) {
  LNZ_REQUIRE(ids && dx && partials && B > 0 && N > 0 && num_atom > 0 && chunks > 0 && chunks <= 65535,
              LNZ_EINVAL, "lnz_embedding_grad: bad arguments (B=%d N=%d atoms=%d chunks=%d)", B, N, num_atom, chunks);
  LNZ_REQUIRE((width == 16 || width == 32 || width == 64 || width == 128) && mol_stride % 4 == 0 &&
                  row_stride % 4 == 0 && ((uintptr_t)dx & 15) == 0,
              LNZ_ENOTSUP, "lnz_embedding_grad: width %d not in {16, 32, 64, 128} or rows not 16-byte aligned", width);
  const dim3 grid(num_atom, chunks);
  const int64_t rows = (int64_t)B * N;
  hipStream_t s = (hipStream_t)stream;
#define LNZ_EG(CPL) hipLaunchKernelGGL(embedding_grad_kernel<CPL>, grid, dim3(256), 0, s, ids, dx, rows, N, \
                                       mol_stride, row_stride, width, num_atom, partials)
  if (width <= 64) LNZ_EG(1);
  else LNZ_EG(2);
#undef LNZ_EG
  return lnz::check_launch("lnz_embedding_grad");
}
