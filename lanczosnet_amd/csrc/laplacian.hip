// R1: batched L4 = D^-1/2 (I + A) D^-1/2 (simple graph + one channel per bond type), and
// R12: unsorted_segment_sum forward / backward.  Both are small HBM-bound byte movers.
#include "common.hpp"

// One workgroup per molecule.  adjs [N,N,E] channels-last is read once (coalesced) into
// registers/LDS-free form: thread t walks (i, j) pairs, all E channels of a pair are
// contiguous.  Degrees are accumulated per row in LDS in fp64 (the reference builds L4 in
// float64: utils/data_helper.py:99-114 after `np.eye(n) + adj`).
__global__ __launch_bounds__(256) void laplacian_l4_kernel(const float* __restrict__ adjs,
                                                           const int32_t* __restrict__ n_nodes,
                                                           int N, int E, float* __restrict__ L) {
  extern __shared__ __attribute__((aligned(16))) double sdeg[];  // [(E+1) * N] -> d^-1/2
  const int b = blockIdx.x;
  const int n = n_nodes[b];
  const int E1 = E + 1;
  const float* Ab = adjs + (int64_t)b * N * N * E;
  float* Lb = L + (int64_t)b * N * N * E1;
  // degree of row i in channel ch (ch 0 = simple graph = sum over bond types) of (I + A)
  for (int t = threadIdx.x; t < E1 * N; t += blockDim.x) {
    int i = t % N, ch = t / N;
    double deg = 1.0;  // the identity's diagonal
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
    double s = 1.0 / sqrt(deg);  // rowsum ** -0.5 ; deg >= 1 - sum|negatives|, inf -> 0 guard
    if (!(deg > 0.0) ) s = 0.0;  // utils/data_helper.py:106 (inf -> 0); negative degrees -> nan in ref
    sdeg[t] = s;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < N * N; p += blockDim.x) {
    int i = p / N, j = p % N;
    const float* a = Ab + (int64_t)p * E;
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
  hipLaunchKernelGGL(laplacian_l4_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, adjs,
                     n_nodes, N, E, L);
  return lnz::check_launch("lnz_laplacian_l4");
}

// ---------------------------------------------------------------------------------------
// unsorted_segment_sum.  data [B, D1, D2]; ids [B, D1]; out [B, S, D2] (pre-zeroed).
// One thread per float4 (or scalar tail) of `data`; consecutive threads walk dim2, so loads
// and the atomics of one source row are contiguous.  fp32 atomicAdd returns nothing ->
// fire-and-forget L2 atomics (operators/src/cuda/segment_reduction.cu:39-53 semantics).
// ---------------------------------------------------------------------------------------
__global__ void segsum_fwd_kernel(const float* __restrict__ data, const int64_t* __restrict__ ids,
                                  int64_t total, int D1, int D2, int S, float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int x = (int)(idx % D2);
    int64_t bc = idx / D2;
    int b = (int)(bc / D1);
    int64_t seg = ids[bc];
    if (seg < 0 || seg >= S) continue;  // out-of-range ids are dropped, never written OOB
    atomicAdd(out + ((int64_t)b * S + seg) * D2 + x, data[idx]);
  }
}

__global__ void segsum_bwd_kernel(const float* __restrict__ gout, const int64_t* __restrict__ ids,
                                  int64_t total, int D1, int D2, int S, float* __restrict__ gdata) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int x = (int)(idx % D2);
    int64_t bc = idx / D2;
    int b = (int)(bc / D1);
    int64_t seg = ids[bc];
    gdata[idx] = (seg < 0 || seg >= S) ? 0.0f : gout[((int64_t)b * S + seg) * D2 + x];
  }
}

static int segsum_grid(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));  // grid-stride beyond 256 CUs x 8
}

extern "C" int lnz_unsorted_segment_sum_forward(const float* data, const int64_t* segment_ids,
                                                int B, int dim1, int dim2, int num_segments,
                                                float* out, lnz_stream_t stream) {
  LNZ_REQUIRE(data && segment_ids && out && B > 0 && dim1 > 0 && dim2 > 0 && num_segments > 0,
              LNZ_EINVAL, "lnz_unsorted_segment_sum_forward: bad arguments");
  int64_t total = (int64_t)B * dim1 * dim2;
  hipLaunchKernelGGL(segsum_fwd_kernel, dim3(segsum_grid(total)), dim3(256), 0,
                     (hipStream_t)stream, data, segment_ids, total, dim1, dim2, num_segments, out);
  return lnz::check_launch("lnz_unsorted_segment_sum_forward");
}

extern "C" int lnz_unsorted_segment_sum_backward(const float* grad_out, const int64_t* segment_ids,
                                                 int B, int dim1, int dim2, int num_segments,
                                                 float* grad_data, lnz_stream_t stream) {
  LNZ_REQUIRE(grad_out && segment_ids && grad_data && B > 0 && dim1 > 0 && dim2 > 0 &&
                  num_segments > 0,
              LNZ_EINVAL, "lnz_unsorted_segment_sum_backward: bad arguments");
  int64_t total = (int64_t)B * dim1 * dim2;
  hipLaunchKernelGGL(segsum_bwd_kernel, dim3(segsum_grid(total)), dim3(256), 0,
                     (hipStream_t)stream, grad_out, segment_ids, total, dim1, dim2, num_segments,
                     grad_data);
  return lnz::check_launch("lnz_unsorted_segment_sum_backward");
}
