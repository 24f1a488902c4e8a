// SURVEY.md §8(f) rank 1 + 3: device-side collate from a packed molecule shard.
//
// The reference keeps one pickle per molecule holding dense float64 Laplacians and an offline
// eigendecomposition (dataset/get_qm8_data.py:56-96), loads 21k of them per epoch
// (dataset/qm8.py:41-47) and pads/concatenates them in Python (dataset/qm8.py:57-100,220-262).
// Here a shard is four flat arrays (atom ids, bond list, labels, offsets: ~160 B per molecule, the
// whole QM8 set is 3.5 MB and stays resident in HBM) and a batch is built by ONE launch:
// workgroup b scatters molecule ids[b]'s bonds into an LDS adjacency, forms
//   L4 = D^-1/2 (I + A) D^-1/2      (utils/data_helper.py:92-116,155-156; fp64 like the reference)
// for the simple graph A = sum_e A_e (get_qm8_data.py:62) and for every bond type
// (get_multi_graph_laplacian_eigs, utils/data_helper.py:261-291), and writes the padded
// channels-last batch of dataset/qm8.py:220-262: L [B,N,N,E+1], node_feat [B,N] (pad 0,
// qm8.py:71-77), node_mask (qm8.py:80-86), label, plus n_nodes for lnz_lanczos_ritz, which
// replaces the pickled (D_simple, V_simple).  The arithmetic is the one of lnz_laplacian_l4, so
// the result is bit-identical to running that kernel on the dense adjacency.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void collate_qm8_kernel(
    const int64_t* __restrict__ mol_off, const int64_t* __restrict__ edge_off,
    const uint8_t* __restrict__ atoms, const uint32_t* __restrict__ edges,
    const float* __restrict__ labels, const int64_t* __restrict__ ids, int n_mol, int N, int E,
    int P, int64_t* __restrict__ node_feat, uint8_t* __restrict__ mask,
    float* __restrict__ label_out, float* __restrict__ L, int32_t* __restrict__ n_nodes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* sdeg = reinterpret_cast<double*>(smem);                      // [(E+1) * N] -> d^-1/2
  int* adj = reinterpret_cast<int*>(smem + (size_t)(E + 1) * N * 8);   // [N][N][E] bond counts
  const int b = blockIdx.x, tid = threadIdx.x;
  const int E1 = E + 1;
  int64_t mol = ids ? ids[b] : b;
  const bool known = mol >= 0 && mol < n_mol;
  mol = known ? mol : 0;
  const int64_t a0 = mol_off[mol], e0 = edge_off[mol];
  int n = known ? (int)(mol_off[mol + 1] - a0) : 0;
  const int ne = known ? (int)(edge_off[mol + 1] - e0) : 0;
  n = n < N ? n : N;  // the host sizes N as the batch maximum; never write out of the tile

  for (int t = tid; t < N * N * E; t += 256) adj[t] = 0;
  __syncthreads();
  for (int t = tid; t < ne; t += 256) {
    const uint32_t w = edges[e0 + t];
    const int u = w & 0xff, v = (w >> 8) & 0xff, ty = (w >> 16) & 0xff;
    if (u < n && v < n && ty < E) {
      atomicAdd(&adj[(u * N + v) * E + ty], 1);
      if (u != v) atomicAdd(&adj[(v * N + u) * E + ty], 1);
    }
  }
  __syncthreads();
  // degree of row i in channel ch (ch 0 = simple graph = sum over bond types) of (I + A)
  for (int t = tid; t < E1 * N; t += 256) {
    const int i = t % N, ch = t / N;
    double deg = 1.0;
    if (i < n) {
      for (int j = 0; j < n; ++j) {
        const int* a = adj + (i * N + j) * E;
        if (ch == 0) {
          for (int e = 0; e < E; ++e) deg += (double)a[e];
        } else {
          deg += (double)a[ch - 1];
        }
      }
    }
    sdeg[t] = 1.0 / sqrt(deg);  // deg >= 1: the inf -> 0 guard of data_helper.py:106 cannot fire
  }
  __syncthreads();
  float* Lb = L + (int64_t)b * N * N * E1;
  for (int p = tid; p < N * N; p += 256) {
    const int i = p / N, j = p % N;
    float* out = Lb + (int64_t)p * E1;
    if (i >= n || j >= n) {
      for (int ch = 0; ch < E1; ++ch) out[ch] = 0.0f;  // dataset/qm8.py:225-260 zero padding
      continue;
    }
    const int* a = adj + p * E;
    const double id = (i == j) ? 1.0 : 0.0;
    double asum = 0.0;
    for (int e = 0; e < E; ++e) {
      const double v = (double)a[e];
      asum += v;
      out[1 + e] = (float)((sdeg[(1 + e) * N + i] * (id + v)) * sdeg[(1 + e) * N + j]);
    }
    out[0] = (float)((sdeg[i] * (id + asum)) * sdeg[j]);
  }
  for (int i = tid; i < N; i += 256) {
    node_feat[(int64_t)b * N + i] = i < n ? (int64_t)atoms[a0 + i] : 0;
    mask[(int64_t)b * N + i] = i < n ? 1 : 0;
  }
  for (int p = tid; p < P; p += 256)
    label_out[(int64_t)b * P + p] = known ? labels[mol * P + p] : 0.0f;
  if (tid == 0) n_nodes[b] = n;
}

}  // namespace

extern "C" int lnz_collate_qm8(const int64_t* mol_off, const int64_t* edge_off,
                               const uint8_t* atoms, const uint32_t* edges, const float* labels,
                               const int64_t* ids, int64_t n_mol, int B, int N, int E, int P,
                               int64_t* node_feat, uint8_t* mask, float* label, float* L,
                               int32_t* n_nodes, lnz_stream_t stream) {
  LNZ_REQUIRE(mol_off && edge_off && atoms && labels && node_feat && mask && label && L && n_nodes,
              LNZ_EINVAL, "lnz_collate_qm8: null pointer");
  LNZ_REQUIRE(B > 0 && N > 0 && N <= 255 && E > 0 && E <= 255 && P > 0 && n_mol > 0 &&
                  n_mol <= INT32_MAX,
              LNZ_EINVAL, "lnz_collate_qm8: bad sizes (B=%d N=%d E=%d P=%d)", B, N, E, P);
  const size_t lds = (size_t)(E + 1) * N * 8 + (size_t)N * N * E * 4;
  LNZ_REQUIRE(lds <= 64 * 1024, LNZ_ENOTSUP,
              "lnz_collate_qm8: N=%d with %d bond types needs %zu B of LDS (> 64 KiB)", N, E, lds);
  hipLaunchKernelGGL(collate_qm8_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, mol_off,
                     edge_off, atoms, edges, labels, ids, (int)n_mol, N, E, P, node_feat, mask,
                     label, L, n_nodes);
  return lnz::check_launch("lnz_collate_qm8");
}
