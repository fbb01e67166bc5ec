// Binding TU (ours) for oracle/_ref: exposes the reference's own CPU RoIAlign
// (compiled from /root/reference/pet/lib/ops/csrc/ROIAlign/ROIAlign_cpu.cpp where
// it lies) to Python.  The reference's vision.cpp cannot be used because it pulls
// in CUDA-only headers (NMS/ml_nms.h:3).  TEST INFRASTRUCTURE ONLY.
#include <torch/extension.h>

namespace pet {
at::Tensor ROIAlign_forward_cpu(const at::Tensor&, const at::Tensor&, const float, const int, const int,
                                const int, const bool, const int);
at::Tensor ROIAlign_backward_cpu(const at::Tensor&, const at::Tensor&, const float, const int, const int,
                                 const int, const int, const int, const int, const int, const bool, const int);
}  // namespace pet

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("roi_align_forward", &pet::ROIAlign_forward_cpu);
  m.def("roi_align_backward", &pet::ROIAlign_backward_cpu);
}
