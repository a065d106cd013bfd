// oracle/ref_bind.cpp -- C-ABI shim around the reference's OWN rotated-IoU CPU code
// (/root/reference/models/bbox_post_process/src/iou3d_cpu.cpp, compiled from where it lies by
// oracle/make_ref.py).  This file contains no reference code: it only declares the reference's
// entry point (iou3d_cpu.h:8) and wraps raw pointers into at::Tensor views.
// TEST INFRASTRUCTURE ONLY -- output goes to oracle/_ref/ (git-ignored).
#include <torch/extension.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

extern "C" int ref_boxes_iou_bev(const float* a, int na, const float* b, int nb, float* out) {
    auto opts = at::TensorOptions().dtype(at::kFloat);
    at::Tensor ta = at::from_blob(const_cast<float*>(a), {na, 7}, opts);
    at::Tensor tb = at::from_blob(const_cast<float*>(b), {nb, 7}, opts);
    at::Tensor to = at::from_blob(out, {na, nb}, opts);
    return boxes_iou_bev_cpu(ta, tb, to);
}
