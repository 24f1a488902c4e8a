// torch extension over the C ABI (include/lanczosnet_hip.h): the ops of the LanczosNet forward step
// registered with the dispatcher as torch.ops.lanczosnet.* (BASELINE.json north_star: "hand-written
// HIP C++ kernels bound as a torch extension").  Host code only — the kernels live in
// liblanczosnet_hip.so; this file adds what ATen brings: tensor / dtype / device checks, the device
// guard, the CURRENT HIP stream of the calling thread, output allocation from the caching
// allocator, and no Python-side marshalling on the step's critical path.
//
// Reference native op surface mirrored here as well: `unsorted_segment_sum_forward/_backward`
// taking tensors (operators/src/segment_reduction_cuda.cpp:8-38).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/lanczosnet_hip.h"

namespace {

using at::Tensor;

void check(int code, const char* what) {
  TORCH_CHECK(code == LNZ_OK, what, ": lanczosnet_hip error ", code, ": ", lnz_last_error());
}

// A ROCm build of torch presents its HIP devices under the "cuda" device type: the current stream
// of the calling thread is the masquerading one, the guard the type-erased c10::DeviceGuard.
lnz_stream_t cur_stream() {
  return (lnz_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
}

void need(const Tensor& t, at::ScalarType dt, const char* name, bool contiguous = true) {
  TORCH_CHECK(t.is_cuda(), "lanczosnet: ", name, " must be a HIP (cuda) tensor; there is no CPU path");
  TORCH_CHECK(t.scalar_type() == dt, "lanczosnet: ", name, " must be ", dt, ", got ", t.scalar_type());
  TORCH_CHECK(!contiguous || t.is_contiguous(), "lanczosnet: ", name, " must be contiguous");
}

const void* optr(const c10::optional<Tensor>& t) { return t.has_value() ? t->data_ptr() : nullptr; }

// ---- R1 ------------------------------------------------------------------------------------
Tensor laplacian_l4(const Tensor& adjs, const Tensor& n_nodes) {
  need(adjs, at::kFloat, "adjs");
  need(n_nodes, at::kInt, "n_nodes");
  TORCH_CHECK(adjs.dim() == 4 && adjs.size(1) == adjs.size(2) && n_nodes.numel() == adjs.size(0));
  const c10::DeviceGuard guard(adjs.device());
  const int B = adjs.size(0), N = adjs.size(1), E = adjs.size(3);
  Tensor L = at::empty({B, N, N, E + 1}, adjs.options());
  check(lnz_laplacian_l4(adjs.data_ptr<float>(), n_nodes.data_ptr<int32_t>(), B, N, E,
                         L.data_ptr<float>(), cur_stream()), "laplacian_l4");
  return L;
}

// ---- R2 + R6 -------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> lanczos_ritz(const Tensor& A, const Tensor& n_nodes, int64_t K) {
  need(A, at::kFloat, "A", /*contiguous=*/false);  // any strides: channel 0 of a channels-last L
  need(n_nodes, at::kInt, "n_nodes");
  TORCH_CHECK(A.dim() == 3 && A.size(1) == A.size(2) && n_nodes.numel() == A.size(0) && K > 0);
  const c10::DeviceGuard guard(A.device());
  const int B = A.size(0), N = A.size(1);
  Tensor D = at::empty({B, K}, A.options());
  Tensor V = at::empty({B, N, K}, A.options());
  Tensor info = at::empty({B}, n_nodes.options());
  const int64_t need_ws = lnz_lanczos_ritz_workspace_bytes(B, N);
  if (N > 32) {
    Tensor ws = at::empty({need_ws > 0 ? need_ws : 1}, A.options().dtype(at::kByte));
    check(lnz_lanczos_ritz_ws(A.data_ptr<float>(), A.stride(0), A.stride(1), A.stride(2),
                              n_nodes.data_ptr<int32_t>(), B, N, (int)K, D.data_ptr<float>(),
                              V.data_ptr<float>(), info.data_ptr<int32_t>(),
                              need_ws > 0 ? ws.data_ptr() : nullptr, need_ws, 0, cur_stream()),
          "lanczos_ritz");
  } else {
    check(lnz_lanczos_ritz(A.data_ptr<float>(), A.stride(0), A.stride(1), A.stride(2),
                           n_nodes.data_ptr<int32_t>(), B, N, (int)K, D.data_ptr<float>(),
                           V.data_ptr<float>(), info.data_ptr<int32_t>(), cur_stream()),
          "lanczos_ritz");
  }
  return {D, V, info};
}

// ---- batch preparation: pack + plan + Ritz pairs in one launch ------------------------------
// returns (Lp [B,C,4,64,4], ident [B], plan [12 cap + 2 + B K] = tile plan | n_wg | n_rows | rows,
//          D [B,K], V [B,N,K])
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> prepare_batch(const Tensor& L, const Tensor& mask,
                                                                 const Tensor& n_nodes, int64_t K,
                                                                 int64_t n_cu, bool allow_pairs) {
  need(L, at::kFloat, "L", /*contiguous=*/false);
  need(mask, at::kByte, "mask");
  need(n_nodes, at::kInt, "n_nodes");
  TORCH_CHECK(L.dim() == 4 && L.size(1) == L.size(2) && mask.size(0) == L.size(0) &&
              mask.size(1) == L.size(1));
  const c10::DeviceGuard guard(L.device());
  const int B = L.size(0), N = L.size(1), C = L.size(3);
  const int cap = lnz_plan_wg_cap(B, (int)n_cu);
  auto iopt = n_nodes.options();
  Tensor Lp = at::empty({B, C, 4, 64, 4}, L.options());
  Tensor ident = at::empty({B}, iopt);
  Tensor plan = at::empty({12 * (int64_t)cap + 2 + (int64_t)B * K}, iopt);
  Tensor D = at::empty({B, K}, L.options());
  Tensor V = at::empty({B, N, K}, L.options());
  int32_t* pb = plan.data_ptr<int32_t>();
  check(lnz_prepare_batch(L.data_ptr<float>(), L.stride(0), L.stride(1), L.stride(2), L.stride(3), B,
                          N, C, Lp.data_ptr<float>(), mask.data_ptr<uint8_t>(),
                          n_nodes.data_ptr<int32_t>(), (int)n_cu, allow_pairs ? 1 : 0, pb,
                          pb + 12 * cap, (int)K, pb + 12 * cap + 2, pb + 12 * cap + 1,
                          D.data_ptr<float>(), V.data_ptr<float>(), nullptr,
                          (uint32_t*)ident.data_ptr<int32_t>(), cur_stream()),
        "prepare_batch");
  return {Lp, ident, plan, D, V};
}

// ---- R7a: spectral gains ---------------------------------------------------------------------
Tensor spectral_gains(const Tensor& D, at::IntArrayRef dist, int64_t num_layer,
                      const c10::optional<Tensor>& mlp_pack, const c10::optional<Tensor>& rows,
                      const c10::optional<Tensor>& n_rows, bool zero_fill) {
  need(D, at::kFloat, "D");
  TORCH_CHECK(D.dim() == 2 && !dist.empty());
  if (mlp_pack.has_value()) need(*mlp_pack, at::kFloat, "mlp_pack");
  const bool use_rows = rows.has_value() && n_rows.has_value() && mlp_pack.has_value();
  if (use_rows) {
    need(*rows, at::kInt, "rows");
    need(*n_rows, at::kInt, "n_rows");
  }
  const c10::DeviceGuard guard(D.device());
  const int B = D.size(0), K = D.size(1), S = dist.size();
  const int64_t n = num_layer * (int64_t)B * S * K;
  // + 64 B of slack: the split-precision forward reads gains as whole dwordx4 groups
  Tensor buf = (use_rows && zero_fill) ? at::zeros({n + 16}, D.options()) : at::empty({n + 16}, D.options());
  Tensor G = buf.narrow(0, 0, n).view({num_layer, B, S, K});
  std::vector<int32_t> d32(dist.begin(), dist.end());
  check(lnz_spectral_gains_rows(D.data_ptr<float>(), B, K, d32.data(), S, (int)num_layer,
                                mlp_pack.has_value() ? 0 : 1, (const float*)optr(mlp_pack),
                                use_rows ? rows->data_ptr<int32_t>() : nullptr,
                                use_rows ? n_rows->data_ptr<int32_t>() : nullptr, G.data_ptr<float>(),
                                cur_stream()),
        "spectral_gains");
  return G;
}

// ---- R7b + R9 + R10: the fused forward (exact-fp32 kernel) -------------------------------------
// dims = [num_layer, din0, dhid, dout, n_long, n_edge, filter_kind]
Tensor forward(const Tensor& node_feat, const c10::optional<Tensor>& embedding, const Tensor& Lp,
               const c10::optional<Tensor>& ident, const Tensor& V, const c10::optional<Tensor>& G,
               const Tensor& mask, const Tensor& Wp, const Tensor& bias, at::IntArrayRef w_off,
               at::IntArrayRef b_off, const Tensor& Wp_head, const Tensor& bias_head,
               const c10::optional<Tensor>& plan, int64_t plan_cap, at::IntArrayRef dims,
               at::IntArrayRef short_dist) {
  TORCH_CHECK(dims.size() == 7, "lanczosnet::forward: dims = [num_layer, din0, dhid, dout, n_long, "
                                "n_edge, filter_kind]");
  need(Lp, at::kFloat, "Lp");
  need(V, at::kFloat, "V");
  need(mask, at::kByte, "mask");
  need(Wp, at::kFloat, "Wp");
  need(bias, at::kFloat, "bias");
  need(Wp_head, at::kFloat, "Wp_head");
  need(bias_head, at::kFloat, "bias_head");
  TORCH_CHECK(V.dim() == 3 && (int64_t)w_off.size() >= dims[0] && (int64_t)b_off.size() >= dims[0] &&
              dims[0] <= 16 && short_dist.size() <= 8);
  const c10::DeviceGuard guard(V.device());
  lnz_forward_args a = {};
  a.B = V.size(0), a.N = V.size(1), a.K = V.size(2);
  a.num_layer = dims[0], a.din0 = dims[1], a.dhid = dims[2], a.dout = dims[3];
  a.n_short = short_dist.size(), a.n_long = dims[4], a.n_edge = dims[5];
  a.filter_kind = dims[6];
  for (size_t i = 0; i < short_dist.size(); ++i) a.short_dist[i] = (int32_t)short_dist[i];
  if (node_feat.scalar_type() == at::kLong) {
    need(node_feat, at::kLong, "node_feat");
    TORCH_CHECK(embedding.has_value(), "lanczosnet::forward: atom ids need the embedding table");
    need(*embedding, at::kFloat, "embedding");
    a.node_feat = node_feat.data_ptr<int64_t>();
    a.embedding = embedding->data_ptr<float>();
    a.num_atom = embedding->size(0);
  } else {
    need(node_feat, at::kFloat, "node_feat");
    TORCH_CHECK(node_feat.size(-1) == a.din0, "lanczosnet::forward: float features must be padded "
                                              "to din0 columns");
    a.node_feat_f = node_feat.data_ptr<float>();
  }
  a.mask = mask.data_ptr<uint8_t>();
  a.V = V.data_ptr<float>();
  a.Lp = Lp.data_ptr<float>();
  if (ident.has_value()) {
    need(*ident, at::kInt, "ident");
    a.ident = (const uint32_t*)ident->data_ptr<int32_t>();
  }
  if (G.has_value()) {
    need(*G, at::kFloat, "G");
    a.G = G->data_ptr<float>();
  }
  a.Wp = Wp.data_ptr<float>(), a.bias = bias.data_ptr<float>();
  for (int i = 0; i < a.num_layer; ++i) a.w_off[i] = w_off[i], a.b_off[i] = b_off[i];
  a.Wp_head = Wp_head.data_ptr<float>(), a.bias_head = bias_head.data_ptr<float>();
  if (plan.has_value()) {
    need(*plan, at::kInt, "plan");
    a.plan = plan->data_ptr<int32_t>();
    a.n_wg = plan->data_ptr<int32_t>() + 12 * plan_cap;
    a.plan_wg_cap = (int)plan_cap;
  }
  Tensor score = at::empty({a.B, a.dout}, V.options());
  a.score = score.data_ptr<float>();
  check(lnz_lanczosnet_forward(&a, cur_stream()), "forward");
  return score;
}

// ---- R12: the reference's native op surface, on tensors ----------------------------------------
Tensor segment_sum_forward(const Tensor& data, const Tensor& segment_ids, int64_t num_segments) {
  need(data, at::kFloat, "data");
  need(segment_ids, at::kLong, "segment_ids");
  TORCH_CHECK(data.dim() == 3 && segment_ids.dim() == 2 && segment_ids.size(0) == data.size(0) &&
              segment_ids.size(1) == data.size(1));
  const c10::DeviceGuard guard(data.device());
  Tensor out = at::zeros({data.size(0), num_segments, data.size(2)}, data.options());
  check(lnz_unsorted_segment_sum_forward(data.data_ptr<float>(), segment_ids.data_ptr<int64_t>(),
                                         data.size(0), data.size(1), data.size(2), (int)num_segments,
                                         out.data_ptr<float>(), cur_stream()),
        "unsorted_segment_sum_forward");
  return out;
}

Tensor segment_sum_backward(const Tensor& grad_out, const Tensor& segment_ids, int64_t dim1) {
  need(grad_out, at::kFloat, "grad_out");
  need(segment_ids, at::kLong, "segment_ids");
  TORCH_CHECK(grad_out.dim() == 3 && segment_ids.dim() == 2);
  const c10::DeviceGuard guard(grad_out.device());
  Tensor gd = at::zeros({grad_out.size(0), dim1, grad_out.size(2)}, grad_out.options());
  check(lnz_unsorted_segment_sum_backward(grad_out.data_ptr<float>(), segment_ids.data_ptr<int64_t>(),
                                          grad_out.size(0), (int)dim1, grad_out.size(2),
                                          grad_out.size(1), gd.data_ptr<float>(), cur_stream()),
        "unsorted_segment_sum_backward");
  return gd;
}

}  // namespace

TORCH_LIBRARY(lanczosnet, m) {
  m.def("abi_version() -> int", []() -> int64_t { return lnz_abi_version(); });
  m.def("laplacian_l4(Tensor adjs, Tensor n_nodes) -> Tensor");
  m.def("lanczos_ritz(Tensor A, Tensor n_nodes, int K) -> (Tensor, Tensor, Tensor)");
  m.def("prepare_batch(Tensor L, Tensor mask, Tensor n_nodes, int K, int n_cu, bool allow_pairs) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("spectral_gains(Tensor D, int[] dist, int num_layer, Tensor? mlp_pack, Tensor? rows, "
        "Tensor? n_rows, bool zero_fill) -> Tensor");
  m.def("forward(Tensor node_feat, Tensor? embedding, Tensor Lp, Tensor? ident, Tensor V, Tensor? G, "
        "Tensor mask, Tensor Wp, Tensor bias, int[] w_off, int[] b_off, Tensor Wp_head, "
        "Tensor bias_head, Tensor? plan, int plan_cap, int[] dims, int[] short_dist) -> Tensor");
  m.def("unsorted_segment_sum_forward(Tensor data, Tensor segment_ids, int num_segments) -> Tensor");
  m.def("unsorted_segment_sum_backward(Tensor grad_out, Tensor segment_ids, int dim1) -> Tensor");
}

TORCH_LIBRARY_IMPL(lanczosnet, CUDA, m) {  // the CUDA dispatch key is HIP on a ROCm build
  m.impl("laplacian_l4", laplacian_l4);
  m.impl("lanczos_ritz", lanczos_ritz);
  m.impl("prepare_batch", prepare_batch);
  m.impl("spectral_gains", spectral_gains);
  m.impl("forward", forward);
  m.impl("unsorted_segment_sum_forward", segment_sum_forward);
  m.impl("unsorted_segment_sum_backward", segment_sum_backward);
}
