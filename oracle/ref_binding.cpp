// Binding TU (ours) for oracle/_ref: exposes the reference's own CPU RoIAlign and soft-NMS
// (compiled from /root/reference/pet/lib/ops/csrc/{ROIAlign/ROIAlign_cpu.cpp,NMS/soft_nms.cpp,NMS/ml_soft_nms.cpp} where
// they lie) to Python.  The reference's vision.cpp cannot be used because it pulls
// in CUDA-only headers (NMS/ml_nms.h:3).  TEST INFRASTRUCTURE ONLY.
#include <torch/extension.h>

namespace pet {
at::Tensor ROIAlign_forward_cpu(const at::Tensor&, const at::Tensor&, const float, const int, const int,
                                const int, const bool, const int);
at::Tensor ROIAlign_backward_cpu(const at::Tensor&, const at::Tensor&, const float, const int, const int,
                                 const int, const int, const int, const int, const int, const bool, const int);
std::tuple<at::Tensor, at::Tensor, at::Tensor> soft_nms_cpu(const at::Tensor&, const at::Tensor&, const float,
                                                            const int, const float, const float);
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> ml_soft_nms_cpu(const at::Tensor&, const at::Tensor&,
                                                                           const at::Tensor&, const float, const int,
                                                                           const float, const float, const int);
}  // namespace pet

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("roi_align_forward", &pet::ROIAlign_forward_cpu);
  m.def("roi_align_backward", &pet::ROIAlign_backward_cpu);
  m.def("soft_nms_cpu", &pet::soft_nms_cpu);      // (dets, scores, threshold, method, sigma, min_score)
  m.def("ml_soft_nms_cpu", &pet::ml_soft_nms_cpu);  // (dets, scores, labels, threshold, method, sigma, min_score, topk)
}
