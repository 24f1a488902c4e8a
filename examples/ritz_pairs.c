/* The C ABI without Python: Ritz pairs (D, V) of the L4 Laplacians of a small batch of graphs.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/ritz_pairs.c \
 *       -Llanczosnet_amd/csrc -llanczosnet_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/lanczosnet_amd/csrc -Wl,-rpath,/opt/rocm/lib -o ritz_pairs
 *   ./ritz_pairs            # prints, per graph, n and its K leading eigenvalues by |lambda|
 *
 * Graph b is the path graph on n_b = 4 + b nodes: the L4 eigenvalues are known in closed form only
 * for regular graphs, so tests/test_c_example.py checks the printed values against numpy instead.
 * What this file shows is the boundary: plain device pointers, sizes and a stream; no torch. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lanczosnet_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_LNZ(x) do { int rc_ = (x); if (rc_ != LNZ_OK) { \
  fprintf(stderr, "%s: %d %s\n", #x, rc_, lnz_last_error()); return 1; } } while (0)

int main(void) {
  enum { B = 5, N = 8, E = 1, K = 6 };
  float adj[B][N][N][E];
  int32_t n_nodes[B];
  memset(adj, 0, sizeof(adj));
  for (int b = 0; b < B; ++b) {
    n_nodes[b] = 4 + b;
    for (int i = 0; i + 1 < n_nodes[b]; ++i) adj[b][i][i + 1][0] = adj[b][i + 1][i][0] = 1.0f;
  }
  if (lnz_abi_version() != LNZ_ABI_VERSION) { fprintf(stderr, "unexpected ABI version\n"); return 1; }

  float *d_adj, *d_L, *d_D, *d_V;
  int32_t* d_n;
  CHECK_HIP(hipMalloc((void**)&d_adj, sizeof(adj)));
  CHECK_HIP(hipMalloc((void**)&d_n, sizeof(n_nodes)));
  CHECK_HIP(hipMalloc((void**)&d_L, sizeof(float) * B * N * N * (E + 1)));
  CHECK_HIP(hipMalloc((void**)&d_D, sizeof(float) * B * K));
  CHECK_HIP(hipMalloc((void**)&d_V, sizeof(float) * B * N * K));
  CHECK_HIP(hipMemcpy(d_adj, adj, sizeof(adj), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_n, n_nodes, sizeof(n_nodes), hipMemcpyHostToDevice));

  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  /* L [B,N,N,E+1] channels-last; channel 0 = the simple graph */
  CHECK_LNZ(lnz_laplacian_l4(d_adj, d_n, B, N, E, d_L, (lnz_stream_t)stream));
  /* Ritz pairs of channel 0, addressed in place through element strides */
  CHECK_LNZ(lnz_lanczos_ritz(d_L, (int64_t)N * N * (E + 1), (int64_t)N * (E + 1), (int64_t)(E + 1),
                             d_n, B, N, K, d_D, d_V, NULL, (lnz_stream_t)stream));
  CHECK_HIP(hipStreamSynchronize(stream));

  float D[B][K], V[B][N][K];
  CHECK_HIP(hipMemcpy(D, d_D, sizeof(D), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(V, d_V, sizeof(V), hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b) {
    printf("n=%d D=", n_nodes[b]);
    for (int k = 0; k < K; ++k) printf(" %.7f", D[b][k]);
    double vnorm = 0.0;  /* first Ritz vector has unit norm; rows >= n are zero */
    for (int i = 0; i < N; ++i) vnorm += (double)V[b][i][0] * V[b][i][0];
    printf(" |v0|^2=%.6f\n", vnorm);
  }
  /* error reporting: an unsupported size comes back as a code + message, nothing is thrown */
  int rc = lnz_lanczos_ritz(d_L, 1, 1, 1, d_n, B, 100000, K, d_D, d_V, NULL, (lnz_stream_t)stream);
  printf("N=100000 -> rc=%d (%s)\n", rc, lnz_last_error());
  hipFree(d_adj); hipFree(d_n); hipFree(d_L); hipFree(d_D); hipFree(d_V);
  hipStreamDestroy(stream);
  return 0;
}
