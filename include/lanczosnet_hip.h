/*
 * lanczosnet_hip.h — C ABI of the MI355X (gfx950) LanczosNet hot path.
 *
 * The reference (lrjconan/LanczosNetwork) has no FFI in front of this path: the boundary
 * is the Python nn.Module contract.  This header is what a binding for
 * that path binds to; `lanczosnet_amd/_lib.py` is the ctypes binding, INTEGRATION.md shows
 * the reference-side stub.  All pointers are DEVICE pointers unless the name ends in
 * `_host`; all tensors are dense row-major float32 unless stated; `stream` is a
 * hipStream_t (NULL = default stream).  Every entry point is asynchronous on `stream`,
 * returns 0 on success and a negative LNZ_E* code on error (message via lnz_last_error()).
 * Nothing here allocates device memory or synchronises the device.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference
 * repository root).
 */
#ifndef LANCZOSNET_HIP_H_
#define LANCZOSNET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7: lnz_midgraph_forward's `sync` grew to B * (num_layer + 1) words (placement words behind the
 *    arrival counters) and its launch is chunked to the resident workgroups; + the K-step entry
 *    (lnz_lanczos_ritz_kstep[_image], _workspace_bytes), the sparse large-graph conv
 *    (lnz_large_sparse_image, lnz_large_pack_vectors, lnz_large_gemm1_rows, lnz_large_sparse_conv[_f32];
 *    lnz_large_conv accepts C = 0), lnz_head_backward, lnz_node_extents, the out-of-place split-pack
 *    entries (lnz_split_laplacian_pack_to, lnz_spectral_gains_rows_split_to), lnz_last_kernel,
 *    lnz_stream_create_cu_masked;
 * 6: lnz_forward_args lost Wp16 / w16_off / Wp16_head / Lp16 (gemm_mode 1 is now the split precision
 *    inside the strip kernel: lnz_pack_rows_k8_split; lnz_pack_rows_f16x2 and
 *    lnz_pack_laplacian_f16x2 are gone) and gained dbias_part_cap; + lnz_midgraph_forward,
 *    lnz_spectral_mlp_grad;  5: + the strip plan (lnz_plan_strips, strips / n_strips / strip_cap);
 * 4: + lnz_large_pack_operators_fold, lnz_f32_linear_workspace_floats (stream-K inside
 *    lnz_f32_linear: the partials argument changed its size and gained a zero-on-entry tail);
 * 3: + lnz_f32_linear, lnz_laplacian, the fp64 training kernels of the AdaLanczosNet spectrum
 *    (lnz_ada_graph_laplacian_f64, lnz_ada_lanczos_layer_f64, lnz_ada_t_powers_f64 and their
 *    _backward);  2: + lnz_lanczos_ritz_ws / _workspace_bytes, lnz_f16x3_*. */
#define LNZ_ABI_VERSION 7
#define LNZ_OK 0
#define LNZ_EINVAL (-1)   /* bad argument (shape/limit)            */
#define LNZ_ELAUNCH (-2)  /* HIP launch / runtime error            */
#define LNZ_ENOTSUP (-3)  /* valid request outside the built range */

#define LNZ_TILE 32       /* node tile: one v_mfma_f32_32x32x2_f32 tile per molecule */
#define LNZ_MAX_CHANNELS 32
/* strip plan (lnz_plan_strips): subtiles of 16 node rows per strip, int32 words per strip entry,
 * molecules the planner packs together (larger batches: chunk by chunk) */
#define LNZ_STRIP_SUB 6
#define LNZ_STRIP_INTS 80
#define LNZ_STRIP_MAX_B 2048

typedef void* lnz_stream_t;

int lnz_abi_version(void);
const char* lnz_last_error(void);
/* The kernel (name with template arguments) the calling thread's last lnz_lanczosnet_forward /
 * _input_grad launch selected — the launchers choose between the strip, 16 x 16-tile and 32 x 32-tile
 * kernels by shape and plan; measurements name what ran instead of restating the rule. */
const char* lnz_last_kernel(void);

/* ---- R1: graph Laplacian ------------------------------------------------------------
 * L4 = D^-1/2 (I + A) D^-1/2 for the simple graph (channel 0, A = sum_e A_e) and for every
 * bond-type channel e (channel 1+e), written channels-last like the collate output.
 * Replaces utils/data_helper.py:92-116,155-156 (normalize_adj / get_laplacian 'L4'),
 * dataset/get_qm8_data.py:62-75 and the L part of dataset/qm8.py:225-262.
 * adjs [B,N,N,E]; n_nodes [B] (rows/cols >= n are written as exact zeros); L [B,N,N,E+1].
 * Arithmetic in fp64 (the reference builds L4 in float64), stored as float32. */
int lnz_laplacian_l4(const float* adjs, const int32_t* n_nodes, int B, int N, int E,
                     float* L, lnz_stream_t stream);
/* Every kind of get_laplacian (utils/data_helper.py:119-166 over normalize_adj :92-116), same
 * layouts: kind 1 .. 7 = 'L1' .. 'L7' (L1 = D - A, L2 = I - D^-1/2 A D^-1/2, L3 = I - D^-1 A,
 * L4 = D^-1/2 (I+A) D^-1/2, L5 = D^-1 (I+A), L6 = D^-alpha A D^-alpha, L7 = D^-1 A; D = row sums of
 * the normalised matrix, D^-x of an isolated node = 0 like the reference's inf guard), alpha only
 * for kind 6 (reference default 0.5).  The reference's offline scripts store L6 / L7 of the simple
 * graph for the ChebyNet / DCNN baselines (dataset/get_qm8_data.py:73-77,
 * dataset/get_graph_data.py:70-75).  kind 4 is lnz_laplacian_l4 bit for bit. */
int lnz_laplacian(const float* adjs, const int32_t* n_nodes, int B, int N, int E, int kind,
                  double alpha, float* L, lnz_stream_t stream);

/* ---- R2 + R6: Lanczos tridiagonalisation -> tridiagonal eigensolve -> Ritz select ----
 * Per molecule: full-length (m = n) Lanczos with twice-iterated classical Gram-Schmidt and
 * restart on breakdown on the n x n leading block of A; then the eigendecomposition of T:
 *   N <= 32 (the QM8 regime): lane-parallel — T split into unreduced blocks, one eigenvalue per
 *            lane by section search on Sturm counts, its eigenvector by the twisted factorisation
 *            of T - lambda (dlar1v / MRRR vector), near-degenerate copies re-orthogonalised in
 *            lane order, Ritz vectors V = Q*S; the implicit-shift QL sweep only as the fallback
 *            (info += 256);
 *   32 < N <= 192 (the reference's synthetic-graph configuration, dataset/get_graph_data.py:15-49
 *            with config/graph_lanczos_net.yaml: n in [20,100]): one 512-thread workgroup per
 *            graph, A staged once into LDS, the fp64 basis in LDS up to N = 108 and in a workspace
 *            above (lnz_lanczos_ritz_workspace_bytes; without one lnz_lanczos_ritz takes a
 *            stream-ordered allocation for the launch — LNZ_ENOTSUP while the stream is being
 *            captured: use lnz_lanczos_ritz_ws there); eigenvalues by Sturm-count section search
 *            (one per thread), the K selected eigenvectors by twisted factorisation, V = Q S;
 *            the QL sweep, run barrier-free on per-wave copies of T, as the fallback (info += 256);
 * then stable ordering by descending |lambda| (ties: ascending lambda), cut / zero-pad to K.
 * Produces exactly
 * the (D, V) that utils/data_helper.py:197-223 (np.linalg.eigh + mergesort on -|eig|)
 * followed by dataset/qm8.py:264-291 (pad rows to N, cut/pad to K, cast fp32) produce,
 * up to the basis of degenerate eigenspaces and eigenvector sign.  fp64 arithmetic.
 * A is addressed as A[b*stride_b + r*stride_r + c*stride_c] (elements) so channel 0 of a
 * channels-last L [B,N,N,E+1] can be passed without a copy.  N <= 192 (LNZ_ENOTSUP above: only
 * the K-step streamed kernels below apply there, a different function — SURVEY.md F8).
 * D [B,K], V [B,N,K].  info [B] (optional, may be NULL): number of Lanczos restarts, + 256 when
 * the N <= 32 kernel fell back from its parallel tridiagonal eigensolver to the QL sweep. */
int lnz_lanczos_ritz(const float* A, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                     const int32_t* n_nodes, int B, int N, int K, float* D, float* V,
                     int32_t* info, lnz_stream_t stream);

/* The same with an explicit workspace for the fp64 Krylov basis of graphs too large to keep it in
 * LDS next to A (N > 108): lnz_lanczos_ritz_workspace_bytes(B, N) bytes (0 when none is needed).
 * Always takes the workgroup-per-graph kernel (any N <= 192); flags bit 0 places the basis in the
 * workspace even when it would fit in LDS, bit 1 takes the QL sweep instead of the parallel
 * tridiagonal eigensolver, bit 2 the eight-wave Lanczos phase where the wave-level one would run
 * (basis in LDS), bits 3-4 fix the wave-level phase's parts per row group (1: one, 2: two, 3: four)
 * — all used by the tests to cover every variant at one size.
 * The workgroup kernel's eigensolver: T split into unreduced blocks, one eigenvalue per thread by
 * section search on Sturm counts, the eigenvectors of the K SELECTED eigenvalues by the twisted
 * factorisation of T - lambda (one thread per vector), V = Q S; two eigenvalues of one block closer
 * than 1e-8 |T| send the graph to the QL sweep (info += 256), as does K > ~N/2 (no room for the
 * vectors next to T in LDS). */
int64_t lnz_lanczos_ritz_workspace_bytes(int B, int N);
int lnz_lanczos_ritz_ws(const float* A, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                        const int32_t* n_nodes, int B, int N, int K, float* D, float* V,
                        int32_t* info, void* workspace, int64_t workspace_bytes, int flags,
                        lnz_stream_t stream);


/* ---- R9 / R11 beyond the 32-node tile: streamed spectral convolution for large dense graphs ----
 * (BASELINE config 5: LanczosNetGeneral, N = 2048, K = 64, batch 256, bf16 operands / fp32
 * accumulate).  One conv layer of model/lanczos_net_general.py:157-182,
 *     X' = relu( sum_e L_e (X W_e^T) + V [ sum_s diag(g_s) (V^T X) W_s^T ] + b ),
 * as four launches; hidden width 128, input width <= 128, K <= 64, any N.
 * planes = 1: bf16 operands (8-bit mantissa: ~1e-2 after 7 layers).  planes = 3: every fp32
 * operand travels as three bf16 pieces and every product as its six piece products of order
 * <= 2, fp32 accumulate: fp32-grade results from the same kernels (the parity mode).
 * planes = 2: two fp16 pieces per operand and the three products hi hi + hi lo + lo hi (~2^-22;
 * 2/3 of the bytes and half the matrix work of planes = 3).  In this mode the A operands carry a
 * factor 2^10 (lnz_large_pack_operators applies it to L and V; the caller applies it to the
 * weight pieces of Wf) which lnz_large_gemm1 / lnz_large_conv divide out; B operands (X, Zt, Tt)
 * are plain fp16 pieces: |activations| must stay below 65504.
 *   Nk = lnz_large_nk(N) = N rounded up to 64 (the k extent of the packed operands).
 *   lnz_large_pack_operators  once per batch.  L [B,N,N,C] fp32 addressed by element strides
 *                             (channels-last collate layout, dataset/graph_data.py collate) ->
 *                             Lb [planes][B][C][RT][Nk/64][4][64][8] bf16 in fragment-tile order
 *                             (RT = ceil(N/32) row groups; chunk (rg, kb) = 32 rows x 64 k as the
 *                             four v_mfma_f32_16x16x32_bf16 A fragments f = 2 rt + ks, lane =
 *                             16 kq + r15 holding row 32 rg + 16 rt + r15, k 64 kb + 32 ks + 8 kq
 *                             .. + 7);  V [B,N,K] -> Vb [planes][B][RT][4][64][8] likewise.
 *   lnz_large_gemm1           Zt [planes][B][C][128][Nk] bf16 = (X W_c^T)^T for the C node-space
 *                             channels; X [B,N,ldx] fp32 (first din columns used);
 *                             Wf [planes][C][4][dinp/16][64][8] bf16 = the channel blocks of the
 *                             mix weight (dinp = din rounded up to 16, zero padded) in
 *                             v_mfma_f32_32x32x16_bf16 A-fragment order: fragment (c, mt, ks),
 *                             lane l holds W_c[32 mt + (l & 31)][16 ks + 8 (l >> 5) .. + 7].
 *                             Columns n >= N of Zt are NOT written: the caller provides a
 *                             zero-initialised buffer (it can be reused for every layer).
 *   lnz_large_spectral        Tt [planes][B][128][64] bf16 = (sum_s diag(g_s) (V^T X) W_s^T)^T;
 *                             exact fp32 inside (two launches: row-chunked projection V^T X with
 *                             fp32 atomics into Ybuf, then the per-graph mix).  G [B,S,K] = this
 *                             layer's gains (one [B,S,K] slice of lnz_spectral_gains' output);
 *                             Wt = lnz_pack_rows_k8 of W_long [128][S*dinp] fp32, the long-scale
 *                             column blocks of the mix weight, each zero padded to dinp columns;
 *                             Ybuf [B][64][128] fp32 work buffer, ZERO on entry, zero on return.
 *   lnz_large_conv            Xout [B,N,128] fp32 = act( sum_c Lb_c Zt_c^T + Vb Tt^T + bias ).
 * Replaces the per-slice bmm / cat / Linear of model/lanczos_net_general.py:161-182 (and
 * model/lanczos_net.py:157-182 for N > 32). */
int64_t lnz_large_nk(int N);
int lnz_large_pack_operators(const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                             int64_t stride_ch, const float* V, int B, int N, int C, int K,
                             int planes, uint16_t* Lb, uint16_t* Vb, lnz_stream_t stream);
/* The same with CHANNEL FOLDING.  With one edge type (config/graph_lanczos_net.yaml:14) the
 * bond-type channel of the collated L is the simple-graph channel again (dataset/graph_data.py:
 * 225-262), and sum_c L_c X W_c^T = L (X (sum_c W_c)^T) over a class of equal operators: only the
 * n_src DISTINCT channels chan_src_host[0..n_src) (ascending) are packed, Lb [planes][B][n_src]...;
 * lnz_large_gemm1 / lnz_large_conv then run with C = n_src and the caller sums the mix-weight
 * blocks of a class.  chan_rep_host[c] (c < C) = packed slot whose source channel c is CLAIMED to equal
 * (chan_src_host[chan_rep_host[c]] == c: packed itself).  With neq != NULL (one device uint64, zeroed by the
 * caller) the kernel — which has all C values of every entry in hand anyway — verifies: bit
 * 8 c + c' is set when channel c differs from channel c' somewhere, compared for every folded
 * channel against its representative and for every packed channel against the channels packed
 * before it (chan_check_host[c] == 0 exempts channel c, e.g. a zero channel stride; NULL = all).  A
 * failed claim means the caller must repack; packed channels that never differed may be folded in
 * the next batch.  chan_src_host == NULL: identity (C <= 8). */
int lnz_large_pack_operators_fold(const float* L, int64_t stride_b, int64_t stride_r,
                                  int64_t stride_c, int64_t stride_ch, const float* V, int B, int N,
                                  int C, int K, int planes, const int32_t* chan_src_host, int n_src,
                                  const int32_t* chan_rep_host, const int32_t* chan_check_host,
                                  unsigned long long* neq, uint16_t* Lb, uint16_t* Vb,
                                  lnz_stream_t stream);
int lnz_large_gemm1(const float* X, int ldx, int din, const uint16_t* Wf, int B, int N, int C,
                    int planes, uint16_t* Zt, lnz_stream_t stream);
int lnz_large_spectral(const float* X, int ldx, int din, const float* V, const float* G,
                       const float* Wt, int B, int N, int K, int S, int planes, float* Ybuf,
                       uint16_t* Tt, lnz_stream_t stream);
int lnz_large_conv(const uint16_t* Lb, const uint16_t* Vb, const uint16_t* Zt, const uint16_t* Tt,
                   const float* bias, int B, int N, int C, int planes, int relu, float* Xout,
                   lnz_stream_t stream);

/* ---- the same layer on the NONZEROS of the Laplacian (csrc/conv_sparse.hip) --------------------
 * The normalised Laplacian of a large sparse graph (config 5: G(n = 2048, p = 0.01), 99 % zeros) is
 * read from HBM once per batch instead of once per layer; planes = 1 (bf16 operands, fp32
 * accumulate: the streamed form's products without the zeros), one operator class (every channel
 * of L equal to channel 0: the reference's single-edge-type collate, dataset/graph_data.py:225-262).
 * A layer = gemm1_rows + spectral + conv(C = 0, relu = 0) + sparse_conv(relu).
 *   lnz_large_sparse_image   L [B,N,N,C] fp32 by element strides -> entries [B][N][row_cap] u32 =
 *                            bf16(value) << 16 | column (rounded to nearest even, as
 *                            lnz_large_pack_operators rounds), counts [B][N] i32: the nonzeros of
 *                            channel 0 row by row (fixed order; zero padded to a multiple of 8 per
 *                            row); values (optional, NULL to skip): the same entries' fp32
 *                            values [B][N][row_cap].  flags (one device int32, cleared by the call): bit 0 = a
 *                            channel differs from channel 0 somewhere, bit 1 = a row holds more
 *                            than row_cap nonzeros (row_cap a multiple of 8, >= 32; N <= 65536).
 *                            A non-zero flag means the image must NOT be used: the caller reads it
 *                            back and takes lnz_large_pack_operators / lnz_large_conv instead.
 *   lnz_large_pack_vectors   Vb alone, exactly as lnz_large_pack_operators writes it.
 *   lnz_large_gemm1_rows     Z [B][N][128] bf16 = X W^T, row major (Wf: the C = 1, planes = 1 form
 *                            of lnz_large_gemm1's fragments, the class's weight blocks summed).
 *   lnz_large_conv, C = 0    the lift alone: Xout = act( Vb Tt^T + bias ) (Lb, Zt may be NULL).
 *   lnz_large_sparse_conv    X [B,N,128] in place: X[r] = act( X[r] + sum_k value[r][k] Z[column[r][k]] ),
 *                            accumulated in fp32 in entry order.  An exact zero of L contributes
 *                            nothing (the streamed kernels multiply it: 0 x inf = NaN there). */
int lnz_large_sparse_image(const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                           int64_t stride_ch, int B, int N, int C, int row_cap, uint32_t* entries,
                           float* values, int32_t* counts, int32_t* flags, lnz_stream_t stream);
int lnz_large_pack_vectors(const float* V, int B, int N, int K, int planes, uint16_t* Vb,
                           lnz_stream_t stream);
int lnz_large_gemm1_rows(const float* X, int ldx, int din, const uint16_t* Wf, int B, int N,
                         uint16_t* Z, lnz_stream_t stream);
int lnz_large_sparse_conv(const uint32_t* entries, const int32_t* counts, int row_cap,
                          const uint16_t* Z, int B, int N, int relu, float* X, lnz_stream_t stream);
/* The readout on the last large-graph conv state (model/lanczos_net_general.py:185-201 /
 * model/lanczos_net.py:185-194): score [B,P] = mean over the real nodes (mask [B,N] != 0) of
 * (W_h x + b_h) * sigmoid(w_g x + b_g); X [B,N,128] fp32, Whead [P + 1][128] = the output Linear's
 * rows then the gate's, bhead [P + 1]; P <= 16.  One pass over X, deterministic. */
int lnz_large_head(const float* X, const uint8_t* mask, const float* Whead, const float* bhead, int B,
                   int N, int P, float* score, lnz_stream_t stream);
/* lnz_large_spectral (planes = 1) and lnz_large_gemm1_rows in one pass over X: the projection's
 * X tile in LDS also yields Z (the same bits as lnz_large_gemm1_rows). */
int lnz_large_spectral_gemm1_rows(const float* X, int ldx, int din, const float* V, const float* G,
                                  const float* Wt, const uint16_t* Wf, int B, int N, int K, int S,
                                  float* Ybuf, uint16_t* Tt, uint16_t* Z, lnz_stream_t stream);
/* The node-space term of the split-precision modes (planes = 2, 3) in EXACT fp32 on the same image:
 * values [B][N][row_cap] fp32 (lnz_large_sparse_image's optional output: the unrounded entries, in
 * the entries' order), Zf [B][N][128] fp32 = X W^T (lnz_f32_linear on the [B N, din] states):
 * X[r] = act( X[r] + sum_k values[r][k] Zf[column[r][k]] ), fp32 fma in entry order.  A layer =
 * lnz_f32_linear + lnz_large_spectral(planes) + lnz_large_conv(C = 0, planes) + this. */
int lnz_large_sparse_conv_f32(const uint32_t* entries, const float* values, const int32_t* counts,
                              int row_cap, const float* Zf, int B, int N, int relu, float* X,
                              lnz_stream_t stream);

/* ---- R6 standalone: batched symmetric tridiagonal eigensolver --------------------------------
 * The step the reference leaves to LAPACK (inside np.linalg.eigh, utils/data_helper.py:201) /
 * ARPACK (:208).  diag [B,M], offdiag [B,M-1] (fp64) -> R [B,M] ascending, Bm [B,M,M] with
 * eigenvectors in columns.  Implicit-shift QL (tql2 recurrences), M <= 64. */
int lnz_tridiag_eigh(const double* diag, const double* offdiag, int B, int M, double* R,
                     double* Bm, lnz_stream_t stream);

/* ---- R2 + R6, large graphs (BASELINE config 5: N = 2048, K = 64) -------------------------------
 * M-step Lanczos (full re-orthogonalisation: classical Gram-Schmidt against every previous vector,
 * a second pass when the first cancelled |w| below 1e-3; stops early if the Krylov space becomes
 * invariant) -> QL on the M x M tridiagonal -> Ritz vectors V = Q S, top-K by |theta|, zero
 * padded.  The dense A (rows contiguous, 16-byte aligned, row stride stride_r, N %% 4 == 0,
 * N <= 2048) is re-streamed from HBM every step: this is the HBM-bound regime of the path.
 * The reference's counterpart is scipy.sparse.linalg.eigsh(L, k, which='LM')
 * (utils/data_helper.py:205-208, ARPACK, implicitly restarted): converged leading pairs agree,
 * unconverged ones are a different function — see oracle/lanczos_kstep.py.
 * workspace: lnz_lanczos_ritz_large_workspace_bytes(B, N) bytes of device memory (the fp64 Krylov
 * basis, B x 64 x N).  D [B,K], V [B,N,K]; info [B] (optional) = Lanczos steps actually taken. */
int64_t lnz_lanczos_ritz_large_workspace_bytes(int B, int N);
int lnz_lanczos_ritz_large(const float* A, int64_t stride_b, int64_t stride_r, int B, int N,
                           int M, int K, void* workspace, float* D, float* V, int32_t* info,
                           lnz_stream_t stream);
/* The same for a SYMMETRIC A (every Laplacian of utils/data_helper.py:92-130 is), reading only
 * the 256 x 256 chunk blocks (I, J >= I): an off-diagonal block serves w_I += A_IJ q_J and
 * w_J += A_IJ^T q_I, 56 %% of the bytes per Lanczos step at N = 2048.  What lies below the diagonal
 * chunk blocks is never read (numpy.linalg.eigh's UPLO convention, upper).  Deterministic (no
 * atomics); same workspace. */
int lnz_lanczos_ritz_large_sym(const float* A, int64_t stride_b, int64_t stride_r, int B, int N,
                               int M, int K, void* workspace, float* D, float* V, int32_t* info,
                               lnz_stream_t stream);

/* The K-step entry of the product surface: what get_graph_laplacian_eigs computes with
 * use_eigen_decomp=False (utils/data_helper.py:205-208: `eigsh(L, k, which='LM')`, a K-dimensional
 * Krylov method instead of the full decomposition) for graphs beyond the 192 nodes
 * lnz_lanczos_ritz serves, ragged batches included: n_nodes [B] (optional; NULL = every graph has
 * N nodes) are the real node counts of zero-padded matrices (dataset/graph_data.py:222-260); the
 * Krylov vectors stay zero on the padding and the recurrence stops after n_b steps by itself.
 * flags:
 *   LNZ_KSTEP_SYMMETRIC  the dense stream reads only the upper 256 x 256 chunk blocks
 *                        (= lnz_lanczos_ritz_large_sym; otherwise = lnz_lanczos_ritz_large);
 *   LNZ_KSTEP_COMPACT    A is read from HBM ONCE: a first launch (one wavefront per row) gathers the
 *                        nonzeros of every 64-row slab into a sliced-ELL image in the workspace (row_cap entries per
 *                        row at most, a multiple of 8), and the M steps multiply by that image —
 *                        the Laplacian of a G(n, 0.01) graph is 99 %% zeros and skipping an exact
 *                        zero changes no sum.  A graph with a longer row is computed by the dense
 *                        stream in the same call (dense_fallback [B], optional output: 1 for such
 *                        a graph).  The FULL matrix is read (no UPLO convention in this mode).
 * stride_c (elements between the columns of a row): 1, or — with LNZ_KSTEP_COMPACT and a
 * dense_fallback output — 2: A is channel 0 of a channels-last [N][N][2] block, the collated
 * `L[..., 0]` of dataset/graph_data.py:225-262 read in place (N even; one float4 = two columns x
 * two channels).  The dense streams need contiguous rows: with stride_c = 2 a graph flagged in
 * dense_fallback is NOT computed — the caller copies such a batch and calls again.
 * Same algorithm, arithmetic and outputs as lnz_lanczos_ritz_large; the three SpMV variants differ
 * in the order of their fp64 additions only.  Deterministic. */
#define LNZ_KSTEP_SYMMETRIC 1
#define LNZ_KSTEP_COMPACT 2
int64_t lnz_lanczos_ritz_kstep_workspace_bytes(int B, int N, int flags, int row_cap);
int lnz_lanczos_ritz_kstep(const float* A, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                           const int32_t* n_nodes, int B, int N, int M, int K, int flags,
                           int row_cap, void* workspace, int64_t workspace_bytes, float* D, float* V,
                           int32_t* info, int32_t* dense_fallback, lnz_stream_t stream);

/* The same call leaving the image of lnz_large_sparse_image behind (csrc/conv_sparse.hip): the
 * compaction pass has every entry of the row in hand (and, with stride_c = 2, the second channel of
 * the collated pair), so conv_entries [B][N][conv_row_cap] (+ conv_values, optional: the unrounded
 * fp32 values of the exact form) / conv_counts [B][N] / conv_flags (one
 * int32: bit 0 = the two channels differ somewhere — stride_c = 2 only; with stride_c = 1 the
 * caller vouches for a single operator —, bit 1 = a row beyond conv_row_cap) cost no second read of
 * L: one pass over the collated Laplacian per batch serves the Ritz pairs and all conv layers.
 * LNZ_KSTEP_COMPACT required; conv_row_cap a multiple of 8, at least 32. */
int lnz_lanczos_ritz_kstep_image(const float* A, int64_t stride_b, int64_t stride_r,
                                 int64_t stride_c, const int32_t* n_nodes, int B, int N, int M, int K,
                                 int flags, int row_cap, void* workspace, int64_t workspace_bytes,
                                 float* D, float* V, int32_t* info, int32_t* dense_fallback,
                                 uint32_t* conv_entries, float* conv_values, int32_t* conv_counts,
                                 int conv_row_cap, int32_t* conv_flags, lnz_stream_t stream);

/* A HIP stream whose kernels run on compute units [first_cu, end_cu) of the current device only
 * (hipExtStreamCreateWithCUMask).  For latency-chain launches that fill a fraction of the chip —
 * the workgroup-per-graph Ritz launch of the reference's graph configuration keeps 64 of 256
 * compute units busy for 0.45 ms — next to the previous batch's forward: on shared compute units
 * the forward's waves stretch the chain to 0.60 ms and the overlap buys nothing; on disjoint ones a
 * batch of the stream takes 0.50 instead of 0.67 ms.  The stream is the caller's
 * (hipStreamDestroy). */
int lnz_stream_create_cu_masked(int first_cu, int end_cu, lnz_stream_t* stream);

/* ---- operand packing (MFMA fragment order) -------------------------------------------
 * W [rows, cols] (leading dimension ld) -> Wp[rt][q][lane][u] =
 *   W[32*rt + (lane&31)][8*q + 4*(lane>>5) + u], zero padded to rows%32==0, cols%8==0.
 * One float4 per lane per (rt, q): the A- or B-operand stream of v_mfma_f32_32x32x2_f32
 * for four consecutive k-steps.  Wp holds lnz_packed_rows_k8_size(rows, cols) floats. */
int64_t lnz_packed_rows_k8_size(int rows, int cols);
int lnz_pack_rows_k8(const float* W, int rows, int cols, int64_t ld, float* Wp,
                     lnz_stream_t stream);
/* The same stream for the split-precision GEMM1 of the strip kernel (lnz_forward_args.gemm_mode =
 * 1): same size (lnz_packed_rows_k8_size; cols a multiple of 32) and the same (rt, 16-k step, lane)
 * indexing, but the two 16-byte slots a lane owns in a 32-k block b hold eight fp16 hi pieces, then
 * the eight lo pieces (x = hi + lo to 22 bits) of W[32 rt + wj][32 b + 8 kq + 0..7] (lane slot
 * 64 (kq >> 1) + 32 (kq & 1) + wj) — the B operands of v_mfma_f32_16x16x32_f16. */
int lnz_pack_rows_k8_split(const float* W, int rows, int cols, int64_t ld, float* Wp,
                           lnz_stream_t stream);
/* bias [rows] -> bp[rt][lane][r] = bias[32*rt + (r&3) + 8*(r>>2) + 4*(lane>>5)] (the C/D
 * row of accumulator register r); holds 32*ceil(rows/32)*32 floats. */
int lnz_pack_bias_rows(const float* bias, int rows, float* bp, lnz_stream_t stream);
/* L [B,N,N,C] addressed by element strides -> Lp[b][c][g][lane][u] =
 *   L[b][lane&31][8*g + 4*(lane>>5) + u][c], zero padded to the 32 x 32 tile (N <= 32).
 * Replaces the per-slice `L[:, :, :, ii]` strided-view clones of model/lanczos_net.py:172-178. */
int lnz_pack_laplacian(const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                       int64_t stride_ch, int B, int N, int C, float* Lp, lnz_stream_t stream);
/* Same, plus ident [B] uint32 (may be NULL): bit c is set when channel c of molecule b is an
 * identity on its nodes — the tile is diag(0/1): L4 of a bond type the molecule does not contain
 * (utils/data_helper.py:92-116 on an empty A_e) — so that lnz_lanczosnet_forward can add Z_c
 * instead of multiplying by M_c (lnz_forward_args.ident).  The ident argument of
 * lnz_pack_laplacian_plan / lnz_prepare_batch[_gains] is the same array. */
int lnz_pack_laplacian_ident(const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                             int64_t stride_ch, int B, int N, int C, float* Lp, uint32_t* ident,
                             lnz_stream_t stream);

/* ---- R7 (first half): per-eigenvalue spectral filter gains ----------------------------
 * G[l][b][s][k] = MLP_l([D^p_1 .. D^p_S])_s for every conv layer l — the
 * `spectral_filter[layer_idx]` Sequential (Linear S->128, ReLU, 128->128, ReLU, 128->128,
 * ReLU, 128->S) of model/lanczos_net.py:48-60,110-113, with the powers of :146-149.
 * mlp_pack: per layer, lnz_spectral_mlp_pack_size(S) floats written by
 * lnz_pack_spectral_mlp().  dist_host: S integer exponents (host memory).
 * kind 0 = 'MLP'; kind 1 = plain powers G = D^p (the non-MLP branch, :118-121; mlp_pack unused).
 * D [B,K]; G [num_layer,B,S,K]. */
int64_t lnz_spectral_mlp_pack_size(int S);
int lnz_pack_spectral_mlp(const float* W0, const float* b0, const float* W2, const float* b2,
                          const float* W4, const float* b4, const float* W6, const float* b6,
                          int S, float* pack, lnz_stream_t stream);
/* Same packs for ALL conv layers in one launch: ptrs[l*8 + {0..7}] = W0, b0, W2, b2, W4, b4, W6, b6
 * of layer l (host array of device pointers); pack holds num_layer consecutive layer packs. */
int lnz_pack_spectral_mlp_layers(const float* const* ptrs, int num_layer, int S, float* pack,
                                 lnz_stream_t stream);
int lnz_spectral_gains(const float* D, int B, int K, const int32_t* dist_host, int S,
                       int num_layer, int kind, const float* mlp_pack, float* G,
                       lnz_stream_t stream);
/* Same, but (kind 0) only for the rows listed in rows[0 .. *n_rows) (device memory, from
 * lnz_plan_batch); every other entry of G is left untouched — zero-initialise G.  rows NULL =
 * lnz_spectral_gains. */
int lnz_spectral_gains_rows(const float* D, int B, int K, const int32_t* dist_host, int S,
                            int num_layer, int kind, const float* mlp_pack, const int32_t* rows,
                            const int32_t* n_rows, float* G, lnz_stream_t stream);

/* Training: parameter gradients of the spectral-filter MLPs of ALL conv layers from dG (the output of
 * lnz_lanczosnet_gain_grad viewed as [num_layer][B*K][S]) in one launch — replaces autograd through
 * the `spectral_filter[l]` Sequentials (model/lanczos_net.py:95-123,146-149: Linear(S,128), ReLU,
 * Linear(128,128), ReLU, Linear(128,128), ReLU, Linear(128,S)).  ptrs[l*8 + {0..7}] = W0, b0, W2, b2,
 * W4, b4, W6, b6 of layer l as for lnz_pack_spectral_mlp_layers (raw row-major parameters; b6 is not
 * read).  rows / n_rows: the live eigen rows of lnz_plan_batch (NULL, NULL = every row of D); rows
 * of dG outside that list are not read.  Every one of the `parts` workgroups of a layer (parts <=
 * lnz_spectral_mlp_grad_parts(rows upper bound, num_layer, n_cu)) writes ONE partial of T =
 * lnz_spectral_mlp_grad_floats(S) floats into partials [L][parts][T]:
 *   dW0 [128][S] | dW2 [128][128] | dW4 [128][128] | dW6 [S][128] | db0, db2, db4 [3][128] | db6 [S] | pad
 * and the gradients are the sum of a layer's partials over the `parts` index (any fixed order: one
 * reduction for all eight).  S <= 8; exact fp32. */
int lnz_spectral_mlp_grad_floats(int S);
int lnz_spectral_mlp_grad_parts(int n_rows_max, int num_layer, int n_cu);
int lnz_spectral_mlp_grad(const float* D, int B, int K, const int32_t* dist_host, int S,
                          int num_layer, const int32_t* rows, const int32_t* n_rows, const float* dG,
                          const float* const* ptrs, int parts, float* partials, lnz_stream_t stream);

/* Training: the embedding table's gradient dE[a][c] = sum over the node rows (b, i) with ids[b][i] == a
 * (clamped to [0, num_atom) as the forward clamps) of dx[b * mol_stride + i * row_stride + c], c <
 * width — autograd's embedding backward (model/lanczos_net.py:154), without atomics: workgroup
 * (a, chunk) writes partials[chunk][a][width], the gradient is their sum over `chunks` (any fixed
 * order).  ids [B, N] int64; width in {16, 32, 64, 128}; strides in floats, multiples of 4. */
int lnz_embedding_grad(const int64_t* ids, int B, int N, const float* dx, int64_t mol_stride,
                       int64_t row_stride, int width, int num_atom, int chunks, float* partials,
                       lnz_stream_t stream);

/* ... and, riding along with the MLP launch (kind 0), the conversion of the batch's packed Laplacian
 * (lp_floats floats at Lp_split, e.g. B * C * 1024 of lnz_pack_laplacian*) into the form the
 * split-precision forward reads (lnz_forward_args.gemm_mode = 1), IN PLACE: every fragment float4
 * becomes 4 fp16 hi pieces | 4 lo pieces.  The launch sits between the pack and the forward anyway
 * and is bound by the matrix pipe; the conversion's workgroups run behind the MLP's.  Lp_split NULL
 * = lnz_spectral_gains_rows.  lnz_split_laplacian_pack is the same conversion as a launch of its
 * own (models without spectral MLPs).  A converted pack is of no use to gemm_mode 0. */
int lnz_spectral_gains_rows_split(const float* D, int B, int K, const int32_t* dist_host, int S,
                                  int num_layer, int kind, const float* mlp_pack, const int32_t* rows,
                                  const int32_t* n_rows, float* G, float* Lp_split, int64_t lp_floats,
                                  lnz_stream_t stream);
int lnz_split_laplacian_pack(float* Lp, int64_t n_floats, lnz_stream_t stream);
/* The same two conversions OUT OF PLACE, into `dst` (lp_floats * 4 bytes; the same bytes are moved):
 * the fp32 pack stays what it was, and a host that types `dst` as 2-byte elements carries the format
 * in the data's type instead of in a flag beside it (what lanczosnet_amd/ops.py does since r06).
 * dst == Lp is the in-place form above. */
int lnz_spectral_gains_rows_split_to(const float* D, int B, int K, const int32_t* dist_host, int S,
                                     int num_layer, int kind, const float* mlp_pack,
                                     const int32_t* rows, const int32_t* n_rows, float* G,
                                     const float* Lp_split, uint16_t* Lp_dst, int64_t lp_floats,
                                     lnz_stream_t stream);
int lnz_split_laplacian_pack_to(const float* Lp, int64_t n_floats, uint16_t* dst, lnz_stream_t stream);

/* ---- R7 (second half) + R9 + R10: fused LanczosNet forward ------------------------------
 * One workgroup per molecule runs the whole network on chip: embedding gather, then per conv
 * layer  X' = relu( sum_c M_c X W_c^T + b )  with M_c in message order
 *   short  : L[...,0]^p                       (model/lanczos_net.py:164-169)
 *   long   : V diag(G[l,:,s,:]) V^T           (:114-117 + :172-174)
 *   edge   : L[...,e], e = 0..E               (:177-178)
 * evaluated as M_c (X W_c^T) so that `cat` (:180) and the [B,N,N,S] filter stack (:123) never
 * exist, then the head  y = sigmoid(X w_a + b_a) * (X W_o^T + b_o)  (:185-188) and the masked
 * mean over real nodes (:190-194).  fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate.
 * Also serves LanczosNetGeneral (model/lanczos_net_general.py:156): pass node_feat_f. */
typedef struct lnz_forward_args {
  int32_t B, N, K;            /* batch, padded node count (<= 32), eigen slots (<= 32)      */
  int32_t num_layer;          /* conv layers, head excluded (<= 16)                          */
  int32_t din0, dhid, dout;   /* input width (%8==0, <=128), hidden width (64|128), head P (<=31) */
  int32_t n_short, n_long, n_edge;      /* channel counts (n_edge = E+1)                    */
  int32_t short_dist[8];      /* powers p of the short channels                              */
  const int64_t* node_feat;   /* [B,N] atom ids (embedding path) or NULL                     */
  const float* node_feat_f;   /* [B,N,din0] float features (General path) or NULL           */
  const float* embedding;     /* [num_atom, din0]                                            */
  int32_t num_atom;
  int32_t filter_kind;        /* 0 = LanczosNet diagonal gains, 1 = AdaLanczosNet dense filters    */
  const uint8_t* mask;        /* [B,N] 1 = real node                                         */
  const float* Lp;            /* lnz_pack_laplacian output, C = n_edge                       */
  const float* V;             /* [B,N,K]                                                     */
  const float* G;             /* filter_kind 0: [num_layer,B,n_long,K] from lnz_spectral_gains (V = Ritz
                                 vectors); filter_kind 1: dense [num_layer,B,n_long,K,K] from
                                 lnz_ada_symmetrize_filters (V = Lanczos basis Q): M = Q DD Q^T     */
  const float* Wp;            /* packed conv weights: layer l at Wp + w_off[l]               */
  const float* bias;          /* conv biases: layer l at bias + b_off[l], dhid floats        */
  int64_t w_off[16];
  int64_t b_off[16];
  const float* Wp_head;       /* packed [32, dhid]: rows 0..P-1 = filter[-1], row P = att    */
  const float* bias_head;     /* [32]: b_o (P), b_a, zeros                                   */
  float* score;               /* [B,dout]                                                    */
  float* state_out;           /* optional [B,32,dhid] final node state (debug/tests) or NULL */
  /* ---- gemm_mode = 0 (default): the exact fp32 path.  gemm_mode = 1: opt-in split precision on the
   * STRIP plan (inference forward only; needs `strips`, dhid 128, din0 128 — zero-pad narrower inputs
   * and the layer-0 weight columns —, diagonal gains, no short-diffusion channels): X W_c^T as x_hi
   * w_hi + x_lo w_hi + x_hi w_lo with fp16 pieces (x = hi + lo to 22 bits) on
   * v_mfma_f32_16x16x32_f16, fp32 accumulate; Wp = lnz_pack_rows_k8_split of the same matrices at the
   * same offsets, followed by 32 KiB of slack (the weight ring over-reads 16 KiB); the node state
   * lives in LDS as fp16 hi | lo blocks; the products with the Laplacian blocks and the Ritz blocks
   * (projection, lift) run in the same three-product split: V is the exact kernel's (split once per
   * launch), Lp is the exact kernel's pack converted in place by lnz_spectral_gains_rows_split /
   * lnz_split_laplacian_pack; gains, biases, activations and the head are exact fp32.  Measured
   * 1e-6 .. 2e-6 against float64 (exact path: 2e-7 .. 3e-7); parity bar 1e-5. */
  int32_t gemm_mode;
  const int32_t* plan;        /* optional tile plan (lnz_plan_tiles): [plan_wg_cap][4][3] int32, slot s
                                 of workgroup g = (molecule A, molecule B or -1, split row); NULL =
                                 one tile per molecule, 4 per workgroup, in batch order.  The plan
                                 changes the time, and the scores only by fp32 re-association
                                 (<= 1e-6 relative; tested at 1e-5).  Pair tiles (B >= 0):
                                 gemm_mode 0 with filter_kind 0 only.                               */
  const int32_t* n_wg;        /* device scalar written by lnz_plan_tiles: workgroups in use          */
  int plan_wg_cap;            /* lnz_plan_wg_cap(B, n_cu): workgroup entries in `plan` (= grid size) */
  /* ---- training (gemm_mode 0, filter_kind 0, dhid 128).  All buffers are
   * molecule-major [.., B, 32, dhid] fp32 and must be ZERO-INITIALISED by the caller (rows a
   * molecule does not own in its tile are never written). */
  float* act_out;             /* lnz_lanczosnet_forward: optional [num_layer,B,32,dhid]; slot l receives
                                 X_{l+1} = relu(conv layer l) — the activations the backward needs  */
  const float* act;           /* input_grad / messages: the act_out of the forward                   */
  float* dy;                  /* input_grad: [num_layer,B,32,dhid]; slot num_layer-1 holds, on entry,
                                 dLoss/dY of the last conv layer (pre-activation); slots l-1 are
                                 written with dLoss/dY_{l-1} for l = num_layer-1 .. 1              */
  float* dx0;                 /* input_grad: [B,32,bwd_din0] dLoss/dX_0 (embedding gradient rows)    */
  int32_t bwd_din0;           /* input_grad: width of X_0 (the model's input_dim, multiple of 32)    */
  const float* x0;            /* messages, layer 0: X_0 [B,32,din0] (rows >= n zero)                 */
  float* msg;                 /* messages: [B*32, C*d] with d = din0 (layer 0) or dhid: row
                                 (b*32 + node), column c*d + i = (M_c X_l)[node][i] — the reference's
                                 cat(msg) (model/lanczos_net.py:164-180)                           */
  int32_t msg_layer;          /* messages: the conv layer l whose messages are built                 */
  const uint32_t* ident;      /* forward, optional [B] (lnz_pack_laplacian_ident): edge-type channels
                                 that are identities on a molecule skip their Laplacian fragments
                                 and GEMM2 (out += Z_c; same bits on the real nodes)               */
  const int64_t* row_off;     /* messages, optional [B]: first row of each molecule in a COMPACT msg
                                 (real nodes only: row_off = exclusive scan of the node counts, rows
                                 >= n are not written); NULL = row b*32 + node                     */
  float* dgains;              /* gain_grad: [num_layer,B,K,n_long] dLoss/dG, ZERO-INITIALISED by the
                                 caller (slots beyond a molecule's row block in a shared tile are
                                 dead, k >= n, and are not written)                                */
  /* ---- input_grad, optional (ABI 4): what the weight / bias gradients need, produced where the
   * values are in registers anyway instead of by separate gather and reduction launches */
  float* dy_compact;          /* [num_layer][dy_compact_rows][dhid]: slots 0 .. num_layer-2 receive
                                 dLoss/dY_l in the COMPACT row numbering of `row_off` (real nodes
                                 only: row_off[molecule] + node) — the row order of the compact
                                 message matrix, so dW_l = dy_compact[l]^T msg_l needs no gather.
                                 Needs row_off.  Slot num_layer-1 (the incoming gradient) is the
                                 caller's.                                                          */
  int64_t dy_compact_rows;    /* rows per slot of dy_compact                                         */
  float* dbias_part;          /* [2 * plan_wg_cap][num_layer][dhid], ZERO-INITIALISED: entry (2 g + h,
                                 l) = column sums of dLoss/dY_l over the node tiles of half h of
                                 workgroup g, for l = 0 .. num_layer-2; their sum over the first
                                 index (any fixed order) is the bias gradient of conv layer l      */
  int32_t dbias_part_cap;     /* entries (first index) of dbias_part; 0 = 2 * plan_wg_cap.  The pass
                                 runs on the strip plan (one entry per strip) when strip_cap fits,
                                 on 32-row tiles (two entries per workgroup) otherwise: size it
                                 max(2 * plan_wg_cap, strip_cap) to get the strips                 */
  /* ---- optional (ABI 5): strip plan of lnz_plan_strips.  With it the width-128 launches run on
   * strips of 16-row subtiles (conv_strip.hip) instead of 32-row tiles: the forward (inference and
   * training, both GEMM modes; gemm_mode 1 exists on strips only), the input-gradient pass (see
   * dbias_part_cap), the message pass (diagonal gains, no short-diffusion channels) and the
   * gain-gradient pass. */
  const int32_t* strips;      /* [strip_cap][LNZ_STRIP_INTS] int32                                   */
  const int32_t* n_strips;    /* device scalar: strips in use                                        */
  int strip_cap;              /* lnz_strip_cap(B): entries in `strips` (= grid size)                 */
} lnz_forward_args;
int lnz_lanczosnet_forward(const lnz_forward_args* args, lnz_stream_t stream);
/* Backward of the conv stack w.r.t. its node-state inputs, in the forward's own structure:
 *   dX_l = sum_c M_c (dY_l W_c),   dY_{l-1} = dX_l * [X_l > 0]
 * (M_c symmetric).  Wp / w_off must hold, per KERNEL layer t = num_layer-1-l, pack_rows_k8 of the
 * per-channel TRANSPOSED mix Wb[i][c*dhid + o] = W_l[o][c*d_l + i] ([d_l, C*dhid]); din0 = dhid;
 * Lp, V, G, mask, plan as in the forward.  Writes dy[0..num_layer-2] and dx0. */
int lnz_lanczosnet_input_grad(const lnz_forward_args* args, lnz_stream_t stream);
/* Gradient w.r.t. the spectral gains (the input of the filter MLPs' backward,
 * model/lanczos_net.py:95-123), in eigen space:
 *   dG[l][b][k][s] = sum_o (V^T dY_l)[k][o] * ((V^T X_l) W_{l,s}^T)[k][o]
 * Reads dy (all num_layer slots, as left by lnz_lanczosnet_input_grad), act / x0 (X_l), V, mask,
 * plan and the FORWARD weight packs Wp / w_off (not the transposed ones); writes dgains. */
int lnz_lanczosnet_gain_grad(const lnz_forward_args* args, lnz_stream_t stream);
/* Messages of conv layer msg_layer, msg = cat_c(M_c X_l): with dY_l they give the weight gradient
 * of the reference's Linear(15 d -> 128) as ONE library GEMM, dW_l = dY_l^T msg
 * (model/lanczos_net.py:180-182).  Reads act (or x0 for layer 0), Lp, V, G, mask, plan. */
int lnz_lanczosnet_messages(const lnz_forward_args* args, lnz_stream_t stream);
/* Tile plan for lnz_forward_args.plan.  The forward kernels work on 32-row node tiles; with
 * allow_pairs small molecules (extent of mask [B,N] u8) share a tile: one of <= 8 nodes with one
 * of 17..24 (split row 8), two of <= 16 (split row 16).  The tiles are dealt over W workgroups —
 * W = R * n_cu with R = ceil(tiles / (4 n_cu)) rounds (one workgroup per compute unit and round;
 * W = tiles when there are fewer than n_cu) — in descending cost order, boustrophedon, so every workgroup of the launch gets
 * floor/ceil(tiles/W) tiles of balanced cost.  Pair tiles rely on V[:, :, k] = 0 and D[:, k] = 0
 * for k >= n, which the reference's collate guarantees (dataset/qm8.py:264-291) and
 * lnz_lanczos_ritz produces.  No reference counterpart (the reference pads every molecule to N,
 * dataset/qm8.py:232-262).
 * plan: [lnz_plan_wg_cap(B, n_cu) * 12] int32; n_wg: [1] int32 (device). */
int lnz_plan_wg_cap(int B, int n_cu);
int lnz_plan_tiles(const uint8_t* mask, int B, int N, int n_cu, int allow_pairs, int32_t* plan,
                   int32_t* n_wg, lnz_stream_t stream);
/* lnz_plan_tiles plus the list of eigen slots that carry a Ritz pair: gain_rows [<= B*K] int32 =
 * b*K + k for k < min(n_b, K) (any order), n_gain_rows [1] — input of lnz_spectral_gains_rows,
 * which then skips the MLP for the zero-padded eigen columns (~18 % of the rows for QM8 sizes). */
int lnz_plan_batch(const uint8_t* mask, int B, int N, int n_cu, int allow_pairs, int32_t* plan,
                   int32_t* n_wg, int K, int32_t* gain_rows, int32_t* n_gain_rows,
                   int32_t* strips, int32_t* n_strips, lnz_stream_t stream);
/* Strip plan for lnz_forward_args.strips (the 16 x 16-tile inference forward, conv_strip.hip): a
 * workgroup runs a strip of up to LNZ_STRIP_SUB subtiles of 16 node rows; a molecule takes
 * ceil(n / 4) * 4 consecutive rows at a 4-aligned start and spans at most two subtiles, so the
 * strip's operators are block diagonal on the subtile diagonal and its neighbours.  Packing: first
 * fit decreasing by size class, stable in batch order (a pure function of the mask); the strip
 * height is the smallest for which the batch fits R strips per compute unit, R = the rounds it
 * needs at full height (one round for B = 1024 QM8 molecules on 256 units).  Against the 32-row
 * tiles of lnz_plan_tiles a QM8-sized batch needs 21 % fewer rows.
 * strips: [lnz_strip_cap(B) * LNZ_STRIP_INTS] int32 — words 0, 1 of an entry = molecules and
 * subtiles of the strip, words 2 + 3 i + {0,1,2} = (molecule, first row, node extent) of its i-th
 * molecule; n_strips [1] int32 (device): strips in use.  N <= 32; batches beyond
 * LNZ_STRIP_MAX_B molecules are planned in chunks of that many (consecutive strip ranges).
 * The strips / n_strips arguments of lnz_plan_batch, lnz_prepare_batch[_prev_gains] and
 * lnz_pack_laplacian_plan (may be NULL) make the same plan inside those launches. */
int lnz_strip_cap(int B);
int lnz_plan_strips(const uint8_t* mask, int B, int N, int n_cu, int32_t* strips,
                    int32_t* n_strips, lnz_stream_t stream);
/* ---- R10 backward: the readout head (model/lanczos_net.py:185-194 under loss.backward(),
 * runner/qm8_runner.py:247).  y = (W_o x + b_o) * sigmoid(w_g x + b_g), score = masked mean of y.
 *   X_last     [B,32,dhid]  the stored last conv state (post ReLU; lnz_lanczosnet_forward act_out)
 *   mask       [B,N] uint8;  grad_score [B,P] = dL/dscore
 *   Whead      [P+1,dhid], bhead [P+1]: output rows, then the gate row — or [P,dhid], [P] with the
 *              gate row given on its own as Wgate [1,dhid], bgate [1] (the module keeps them in two
 *              Linears, model/lanczos_net.py:57,61)
 *   row_off    [B] int64 (optional): first row of molecule b in the compact numbering of dY_compact
 * ->
 *   dY         [B,32,dhid]  dL/d(pre-activation of the last conv layer) (zero on padding / masked rows)
 *   dY_compact [R,dhid]     (optional) the rows below a molecule's node extent, compact
 *   dWhead     [P+1,dhid], dbhead [P+1], dbias_last [dhid] (column sums of dY)
 * workspace: lnz_head_backward_workspace_floats(P, n_wg) floats; n_wg workgroups walk the molecules
 * (the partial sums are added in workgroup order: deterministic).  dhid = 128, N <= 32, P <= 31. */
int64_t lnz_head_backward_workspace_floats(int P, int n_wg);
/* Node extents (last real node + 1; the training kernels size a molecule by it), their exclusive
 * prefix sums (first row of a molecule in the compact numbering of the message matrix,
 * lnz_lanczosnet_messages row_off) and their total, from mask [B,N] uint8: extent, row_off [B] int64,
 * total [1] int64.  One launch. */
int lnz_node_extents(const uint8_t* mask, int B, int N, int64_t* extent, int64_t* row_off,
                     int64_t* total, lnz_stream_t stream);
int lnz_head_backward(const float* X_last, const uint8_t* mask, const float* grad_score,
                      const float* Whead, const float* bhead, const float* Wgate, const float* bgate,
                      const int64_t* row_off, int B, int N,
                      int P, int dhid, int n_wg, float* workspace, float* dY, float* dY_compact,
                      float* dWhead, float* dbhead, float* dbias_last, lnz_stream_t stream);
/* ---- R9 + R10 + R11 for graphs of 33..128 nodes: the reference's own graph configuration
 * (config/graph_lanczos_net.yaml, dataset/get_graph_data.py:15-49: n in [20, 100]) — every conv
 * layer of model/lanczos_net_general.py:157-182 (model/lanczos_net.py:157-182), the head and the
 * gated masked mean (:185-194) in ONE launch, exact fp32 (v_mfma_f32_16x16x4_f32).  A graph is
 * spread over four workgroups by output columns (32 each); the layer's state goes through Xwork
 * behind a counter in `sync` the four spin on (csrc/conv_mid.hip).  The four must be resident
 * together: a batch larger than the device holds at once (occupancy x compute units; 64 graphs on
 * an idle MI355X) is issued as consecutive launches.  Whether the four share one L2 is read from
 * the hardware at run time (fence-free exchange only then, agent-scope release / acquire
 * otherwise); every spin is bounded and ends in a trap, i.e. a failed launch, not a hang.
 *   X0     [B,N,din0] fp32   layer-0 state (node features / embedding rows), din0 a multiple of
 *                            16 (zero-padded columns), <= 128
 *   L      [B,N,N,C] fp32    by element strides; C = edge types + 1 channels, 1..2
 *   V, G, mask               as for lnz_lanczosnet_forward: [B,N,K], [num_layer,B,S,K], [B,N]
 *   W      fp32              per layer [128][S + C][din_l] (din_0 = din0, else 128): the mix
 *                            weight's column blocks in the reference's order — long scales, then
 *                            edge types (no short-diffusion channels) —, the layers behind each other
 *   bias   [num_layer,128];  Whead [dout + 1,128], bhead [dout + 1]: head rows, then the gate row
 *   Xwork  [lnz_midgraph_workspace_floats(B, N, num_layer)] fp32 scratch
 *   sync   [B * (num_layer + 1)] int32, ZERO on entry (arrival counters, then placement words)
 *   score  [B,dout]
 * N <= 128, K <= 32, S <= 16, dout <= 31, hidden width 128. */
int64_t lnz_midgraph_workspace_floats(int B, int N, int num_layer);
int lnz_midgraph_forward(const float* X0, const float* L, int64_t stride_b, int64_t stride_r,
                         int64_t stride_c, int64_t stride_ch, const float* V, const float* G,
                         const uint8_t* mask, const float* W, const float* bias, const float* Whead,
                         const float* bhead, int B, int N, int K, int C, int S, int num_layer,
                         int din0, int dout, float* Xwork, int32_t* sync, float* score,
                         lnz_stream_t stream);
/* The whole batch preparation in ONE launch: lnz_plan_batch (workgroup 0), lnz_lanczos_ritz on
 * channel 0 of L (workgroups 1..B, dispatched first: they are the long, latency-bound pole) and
 * lnz_pack_laplacian (workgroups B+1..2B) — the two byte movers run in the shadow of the Lanczos
 * wavefronts instead of in front of them.  L [B,N,N,C] with strides (elements); n_nodes [B] as
 * for lnz_lanczos_ritz; outputs as in the three functions.  N <= 32.
 * Lp == NULL (then ident must be NULL too): plan + Ritz pairs only, B + 1 workgroups — the pack is
 * then the caller's lnz_pack_laplacian_ident on a second stream, under the spectral-gains launch
 * (matrix-pipe work that leaves the memory path idle) instead of next to the Lanczos wavefronts. */
int lnz_prepare_batch(const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                      int64_t stride_ch, int B, int N, int C, float* Lp, const uint8_t* mask,
                      const int32_t* n_nodes, int n_cu, int allow_pairs, int32_t* plan,
                      int32_t* n_wg, int K, int32_t* gain_rows, int32_t* n_gain_rows, float* D,
                      float* V, int32_t* info, uint32_t* ident, int32_t* strips, int32_t* n_strips,
                      lnz_stream_t stream);
/* Software pipeline over a STREAM of batches: lnz_prepare_batch of batch k+1 and the spectral
 * gains (kind 0, live rows) of batch k in one launch.  The two are independent — no flags — and
 * complementary: the Lanczos / eigensolve wavefronts are latency bound (one per SIMD, matrix pipes
 * idle), the MLP is matrix-pipe work.  D_prev [B_prev,K], rows_prev / n_rows_prev: outputs of the
 * previous preparation; G_prev [num_layer,B_prev,S,K]: live slots written.  Everything else as in
 * lnz_prepare_batch. */
int lnz_prepare_batch_prev_gains(
    const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c, int64_t stride_ch, int B,
    int N, int C, float* Lp, const uint8_t* mask, const int32_t* n_nodes, int n_cu, int allow_pairs,
    int32_t* plan, int32_t* n_wg, int K, int32_t* gain_rows, int32_t* n_gain_rows, float* D,
    float* V, uint32_t* ident, const float* D_prev, int B_prev, const int32_t* rows_prev,
    const int32_t* n_rows_prev, const int32_t* dist_host, int S, int num_layer,
    const float* mlp_pack, float* G_prev, int32_t* strips, int32_t* n_strips, lnz_stream_t stream);
/* lnz_pack_laplacian + lnz_plan_batch in ONE launch (workgroup B plans while 0..B-1 pack): the two
 * byte movers in front of the Lanczos kernel are independent, and the planner is a single
 * latency-bound workgroup.  Arguments as in the two functions; gain_rows may be NULL. */
int lnz_pack_laplacian_plan(const float* L, int64_t stride_b, int64_t stride_r, int64_t stride_c,
                            int64_t stride_ch, int B, int N, int C, float* Lp,
                            const uint8_t* mask, int n_cu, int allow_pairs, int32_t* plan,
                            int32_t* n_wg, int K, int32_t* gain_rows, int32_t* n_gain_rows,
                            uint32_t* ident, int32_t* strips, int32_t* n_strips,
                            lnz_stream_t stream);
/* sizeof(lnz_forward_args) as compiled into the library — lets a foreign-language binding verify
 * its struct layout before the first call. */
int64_t lnz_forward_args_size(void);

/* ---- R4, R5, R8: AdaLanczosNet stages --------------------------------------------------------
 * lnz_ada_graph_laplacian: Gaussian-kernel learned Laplacian of model/ada_lanczos_net.py:101-137
 *   (dist2 over node embeddings, sigma2 = mean over all N^2 pairs, exp(-d2/sigma2) * adj,
 *   D^-1/2 A D^-1/2 with the zero-row guard); adj = (L0 != 0) as :310-311.  Node features are either
 *   embedding rows gathered by id (node_feat + embedding[num_atom, D]) or float rows node_feat_f
 *   [B,N,D].  L0 addressed by element strides (pass L[...,0] of the channels-last Laplacian).
 *   Le [B,N,N].
 * lnz_ada_lanczos_layer: the in-model Lanczos layer of :139-247, reference exact: fp32,
 *   T_it = min(N,K) steps, sequential Gram-Schmidt applied twice, breakdown mask beta < 1e-4, and
 *   the three quirks of SURVEY.md F6 (alpha of the breakdown step zeroed; its Q column dropped;
 *   NODE rows >= idx_mask zeroed).  q1 [B,N] is the raw start vector (the reference draws
 *   torch.randn(B,N,1) on the CPU generator, :161).  mask [B,N] uint8 or NULL.  T [B,K,K], Q [B,N,K].
 * lnz_ada_t_powers: T^p for the S exponents in dist_host by sequential TT = TT*T (:262-270),
 *   written as torch.cat(T_list, dim=2): Tcat [B, K, S*K] (= the [B, K*K*S] MLP input).
 * lnz_ada_symmetrize_filters: DD [B,K,K,S] (MLP output view, :274-275) ->
 *   DDp [B,S,K,K] = (DD + DD^T)/2 (:278), the dense-filter operand of lnz_lanczosnet_forward. */
int lnz_ada_graph_laplacian(const int64_t* node_feat, const float* embedding, int num_atom,
                            const float* node_feat_f, int D, const float* L0, int64_t stride_b,
                            int64_t stride_r, int64_t stride_c, int B, int N, float* Le,
                            lnz_stream_t stream);
int lnz_ada_lanczos_layer(const float* A, const uint8_t* mask, const float* q1, int B, int N,
                          int K, float* T, float* Q, lnz_stream_t stream);
int lnz_ada_t_powers(const float* T, int B, int K, const int32_t* dist_host, int S, float* Tcat,
                     lnz_stream_t stream);
int lnz_ada_symmetrize_filters(const float* DD, int B, int K, int S, float* DDp,
                               lnz_stream_t stream);

/* Opt-in split-precision operand of the filter MLPs' GEMMs (model/ada_lanczos_net.py:271-272):
 * v = [relu](alpha * X[m][k] + bias[k]), X [M, K] fp32 (leading dimension ldx) ->
 * out [M, 3 Kp] fp16 = [ hi | hi | lo ], hi = fp16(v), lo = fp16(v - hi), columns >= K zero.
 * Against weights laid out [ w_hi | w_lo | w_hi ] ONE fp16 GEMM of depth 3 Kp (fp32 accumulate)
 * gives hi w_hi + hi w_lo + lo w_hi = v w to ~2^-22 relative. */
int lnz_split_f16x3(const float* X, int M, int K, int ldx, const float* bias, float alpha, int relu,
                    int Kp, void* out, lnz_stream_t stream);

/* R5 for training (csrc/ada_lanczos_grad.hip): the same Lanczos layer on an fp64 Laplacian
 * A [B,N,N] (N, K <= 32), outputs T [B,K,K] and Q [B,N,K] in fp64, plus the state its backward needs
 * in `ws` (lnz_ada_lanczos_f64_workspace_doubles(B) doubles: basis vectors, alpha, beta, the valid
 * flags and every Gram-Schmidt coefficient), and the backward: given dLoss/dT, dLoss/dQ (fp64,
 * same shapes) the reverse sweep of the recurrence -> dA [B,N,N] = dLoss/dA.  Replaces torch
 * autograd through model/ada_lanczos_net.py:139-247 (the reference differentiates its own loop of
 * small ATen ops).  mask may be NULL (all nodes real); masks / flags are constants of the gradient. */
/* The stages either side of it for training, fp64 as well.  Learned Laplacian (:101-137) from the
 * embedded node states X [B,N,D] fp32 and the adjacency mask L0 != 0 (strided [B,N,N] view):
 * Le [B,N,N] fp64 + `state` (lnz_ada_laplacian_f64_state_doubles(B, N) doubles: A, dist2, D^-1/2,
 * sigma2); backward: dLe -> dX [B,N,D] fp64.  T powers (:262-270) of the fp64 T: Tcat [B,K,S K] fp32
 * as lnz_ada_t_powers + every power P [B,pmax,K,K] fp64; backward: dTcat (fp32) -> dT [B,K,K] fp64. */
int64_t lnz_ada_laplacian_f64_state_doubles(int B, int N);
int lnz_ada_graph_laplacian_f64(const float* X, int D, const float* L0, int64_t stride_b,
                                int64_t stride_r, int64_t stride_c, int B, int N, double* Le,
                                double* state, lnz_stream_t stream);
int lnz_ada_graph_laplacian_f64_backward(const float* X, int D, int B, int N, const double* state,
                                         const double* dLe, double* dX, lnz_stream_t stream);
int lnz_ada_t_powers_f64(const double* T, int B, int K, const int32_t* dist_host, int S, float* Tcat,
                         double* P, lnz_stream_t stream);
int lnz_ada_t_powers_f64_backward(const double* T, int B, int K, const int32_t* dist_host, int S,
                                  const float* dTcat, const double* P, double* dT, lnz_stream_t stream);
int64_t lnz_ada_lanczos_f64_workspace_doubles(int B);
int lnz_ada_lanczos_layer_f64(const double* A, const uint8_t* mask, const float* q1, int B, int N,
                              int K, double* T, double* Q, double* ws, lnz_stream_t stream);
int lnz_ada_lanczos_layer_f64_backward(const double* A, int B, int N, int K, const double* ws,
                                       const double* dT, const double* dQ, double* dA,
                                       lnz_stream_t stream);

/* The same filter MLPs, hand-written (csrc/f16x3_linear.hip): one launch per Linear,
 *   out = [relu]( alpha * (X W^T) + bias ),   X [M, K], W [N, K],
 * both operands as (hi, lo) fp16 PLANES (x = x_hi + x_lo, w = w_hi + w_lo) and the product as
 * x_hi w_hi + x_hi w_lo + x_lo w_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation — the four
 * pieces of a k-slice are staged once (global_load_lds, two 64 KB LDS buffers) and feed all three
 * products.  K %% 64 == 0; x planes hold ceil(M / 128) * 128 rows, w planes ceil(N / 128) * 128 rows
 * (zero padded), leading dimensions ldx / ldw in elements (multiples of 8).  Output either as the
 * next Linear's operand, (out_hi, out_lo) fp16 planes with leading dimension ldo (rows >= M are
 * not written), or — the last Linear — as fp32 out_f32 [M, ldo].  lnz_f16x3_split writes the
 * planes of an fp32 matrix (scale * X; rows >= M and columns >= K zero). */
int lnz_f16x3_split(const float* X, int M, int K, int64_t ldx, float scale, int Mp, int Kp,
                    uint16_t* hi, uint16_t* lo, lnz_stream_t stream);
int lnz_f16x3_linear(const uint16_t* x_hi, const uint16_t* x_lo, int ldx, const uint16_t* w_hi,
                     const uint16_t* w_lo, int ldw, const float* bias, float alpha, int relu, int M,
                     int N, int K, uint16_t* out_hi, uint16_t* out_lo, float* out_f32, int ldo,
                     float* partials, lnz_stream_t stream);
/* Split-K for shapes with too few 128 x 128 output tiles to fill the chip (the last Linear, N =
 * 1056: 72 tiles): with `partials` = lnz_f16x3_linear_splits(M, N, K) * M * N floats of scratch the K
 * range is divided over that many workgroups per tile and a second kernel adds the partial tiles in
 * a fixed order (bit-reproducible) before alpha / bias / ReLU; partials = NULL: no split. */
int lnz_f16x3_linear_splits(int M, int N, int K);

/* The filter MLPs in the default exact-fp32 mode, hand-written (csrc/f32_linear.hip; replaces
 * torch.nn.functional.linear + relu of model/ada_lanczos_net.py:271-272 = nn.Sequential of
 * nn.Linear / nn.ReLU): one launch per Linear,
 *   out = [relu]( X W^T + bias ),   X [M, K] (leading dimension ldx), W [N, K] (ldw), out [M, ldo],
 * fp32 accumulation, bias + ReLU in the epilogue;
 * K %% 32 == 0, ldx / ldw multiples of 4 and the base pointers 16-byte aligned (LNZ_ENOTSUP
 * otherwise); any M, N (partial tiles re-read the last row, never store).  bias may be NULL.
 * fp32 operands on v_mfma_f32_16x16x4_f32 (a k-ordered fma chain per output element: the result
 * is bit-identical to hipBLASLt's fp32 GEMM on these shapes), operand slices copied global -> LDS
 * by buffer_load ... lds.  Outputs with too few 128 x 128 tiles to fill the chip (the last Linear,
 * N = 1056: 72 tiles) run STREAM-K inside the one launch: (tile, k-slice) units dealt evenly over
 * lnz_f32_linear_splits(M, N, K) workgroups, partial tiles added by the tile's owner in a fixed
 * order (bit-reproducible).  For those shapes `partials` = lnz_f32_linear_workspace_floats(M, N, K)
 * floats, 16-byte aligned, whose LAST tiles-many words (the tile counters) must be ZERO on entry;
 * they are zero again on return, so one zero-initialised workspace serves every later call on
 * the same stream.  partials = NULL (or _splits == 1): one workgroup per tile. */
int lnz_f32_linear(const float* x, int ldx, const float* w, int ldw, const float* bias, int relu,
                   int M, int N, int K, float* out, int ldo, float* partials, lnz_stream_t stream);
int lnz_f32_linear_splits(int M, int N, int K);
int64_t lnz_f32_linear_workspace_floats(int M, int N, int K);


/* ---- next row (SURVEY.md 8f rank 1 + 3): device-side collate from a packed molecule shard ----
 * Replaces the per-molecule pickles of dataset/get_qm8_data.py:56-96 (dense float64 Laplacians +
 * offline eigendecomposition), their loading (dataset/qm8.py:41-47) and the Python pad/concat of
 * QM8Data.collate_fn (dataset/qm8.py:57-100,220-262).  A shard is four flat device arrays:
 *   mol_off  [n_mol+1] int64   atom offsets          atoms  [total_atoms] u8  atom ids
 *   edge_off [n_mol+1] int64   bond offsets          edges  [total_bonds] u32 = u | v<<8 | type<<16
 *   labels   [n_mol, P] fp32                         (each undirected bond listed once)
 * One launch builds the batch of molecules ids[0..B) (NULL = 0..B-1), padded to N nodes
 * (N >= the largest molecule of the batch, qm8.py:66):
 *   L [B,N,N,E+1] fp32 channels-last, channel 0 = L4 of the simple graph sum_e A_e
 *     (get_qm8_data.py:62), channel 1+e = L4 of bond type e (utils/data_helper.py:92-116,
 *     155-156, 261-291), computed in fp64 and rounded once — bit-identical to lnz_laplacian_l4
 *     on the dense adjacency;
 *   node_feat [B,N] int64 (pad 0), mask [B,N] u8, label [B,P], n_nodes [B] int32.
 * (D, V) then come from lnz_lanczos_ritz on L[..., 0] instead of the pickled D_simple/V_simple.
 * An id outside [0, n_mol) yields an empty molecule (n_nodes = 0).  N*N*E*4 + (E+1)*N*8 <= 64 KiB. */
int lnz_collate_qm8(const int64_t* mol_off, const int64_t* edge_off, const uint8_t* atoms,
                    const uint32_t* edges, const float* labels, const int64_t* ids, int64_t n_mol,
                    int B, int N, int E, int P, int64_t* node_feat, uint8_t* mask, float* label,
                    float* L, int32_t* n_nodes, lnz_stream_t stream);

/* ---- R12: unsorted_segment_sum -----------------------------------------------------------
 * out[b, ids[b,c], x] += data[b,c,x]  /  grad_data[b,c,x] = grad_out[b, ids[b,c], x].
 * Replaces operators/src/cuda/segment_reduction.cu:39-69 (+ launchers :72-95) behind
 * operators/functions/unsorted_segment_sum.py:8-44.  `out` must be pre-zeroed by the caller
 * (as :25-26 does).  The output batch stride is num_segments*dim2 (the reference kernel's
 * dim1*dim2 stride, :48, is only in-bounds when num_segments == dim1; identical there). */
int lnz_unsorted_segment_sum_forward(const float* data, const int64_t* segment_ids, int B,
                                     int dim1, int dim2, int num_segments, float* out,
                                     lnz_stream_t stream);
int lnz_unsorted_segment_sum_backward(const float* grad_out, const int64_t* segment_ids, int B,
                                      int dim1, int dim2, int num_segments, float* grad_data,
                                      lnz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LANCZOSNET_HIP_H_ */
