// Binding (ours) around the reference's CPU operator, compiled from the reference source where it
// lies (/root/reference/operators/src/segment_reduction.cpp, unmodified) by oracle/ref_build.py.
// Exposes the two functions of operators/src/segment_reduction.h:1-5 to Python so the oracle's
// restatement (oracle/segment_sum.py) and the HIP operator can be pinned on the real thing.
#include <torch/extension.h>

#include <vector>

#include "segment_reduction.h"  // the reference's header (found via -I /root/reference/operators/src)

static int fwd(at::Tensor data, at::Tensor ids, std::vector<int> shape, at::Tensor out) {
  return unsorted_segment_sum_forward(data, ids, shape.data(), out);
}
static int bwd(at::Tensor gout, at::Tensor ids, std::vector<int> shape, at::Tensor gdata) {
  return unsorted_segment_sum_backward(gout, ids, shape.data(), gdata);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("unsorted_segment_sum_forward", &fwd);
  m.def("unsorted_segment_sum_backward", &bwd);
}
