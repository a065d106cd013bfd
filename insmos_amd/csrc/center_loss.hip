// insmos_amd/csrc/center_loss.hip -- training side of the BEV CenterHead:
//
//   targets   CenterHead.get_targets_single (models/backbones_2d/center_head.py:170-249): per GT box the heat-map cell,
//             the gaussian radius (gaussian_radius :395-424), the gaussian splat (draw_heatmap_gaussian :364-393 with
//             gaussian_2d :346-362) and the regression row [dx, dy, z, log dims, sin, cos].  One workgroup per object; the
//             splat is an element-wise MAX, so integer atomicMax on the (non-negative) float bits is order-independent
//             and the result is deterministic (the reference walks the boxes serially on the host, with a device
//             round trip per box and element).
//   loss      CenterHead.get_loss (:279-331): clip_sigmoid (:333-344) + gaussian_focal_loss (:597-616) summed over the
//             map / max(#cells == 1, 1), masked L1 on the gathered regression rows (:618-631) / (#masked + 1e-4), and
//             the gradient of cls_weight * cls + loc_weight * loc with respect to both head maps, from the same pass.
//             Reductions are fixed-order (no float atomics): loss and gradient are bit-reproducible run to run.
#include "common.h"

extern "C" size_t insmos_col_sum_ws_floats(int64_t n, int c);
extern "C" int insmos_col_sum(const float* a, int ld, int c, int64_t n, float* out, int accumulate, float* ws, void* stream);

namespace insmos {

struct CenterTargetCfg {
    int n_gt, max_objs, num_class, fm_w, fm_h, min_radius, range_f64;
    double x0, y0, fac64;
    float vx, vy, fac;
    float c_1m, c_1p, c_m2, c_m1, c_4a3;  // (1 - ov), (1 + ov), (-2 ov), (ov - 1), 4 * (4 ov) rounded to float like torch does
};

// gaussian_radius (:395-424) on fp32 scalars, operation for operation (the library is built with -ffp-contract=off)
__device__ __forceinline__ float gaussian_radius_f32(float h, float w, const CenterTargetCfg& c) {
    const float b1 = h + w;
    const float c1 = w * h * c.c_1m / c.c_1p;
    const float r1 = (b1 + sqrtf(b1 * b1 - 4.f * c1)) / 2.f;
    const float b2 = 2.f * (h + w);
    const float c2 = c.c_1m * w * h;
    const float r2 = (b2 + sqrtf(b2 * b2 - 16.f * c2)) / 2.f;
    const float b3 = c.c_m2 * (h + w);
    const float c3 = c.c_m1 * w * h;
    const float r3 = (b3 + sqrtf(b3 * b3 - c.c_4a3 * c3)) / 2.f;
    return fminf(fminf(r1, r2), r3);
}

__global__ void __launch_bounds__(64) k_center_targets(const float* __restrict__ gt, CenterTargetCfg c, float* __restrict__ heat,
                                                       float* __restrict__ anno, int64_t* __restrict__ ind,
                                                       uint8_t* __restrict__ mask) {
    const int k = blockIdx.x;  // object slot
    bool ok = k < c.n_gt;
    float cx = 0.f, cy = 0.f;
    int x = 0, y = 0, radius = 0, cls = -1;
    const float* b = gt + (int64_t)k * 8;
    if (ok) {
        const float lab = b[7] - 1.f;
        ok = lab > -1.f && lab < (float)c.num_class;      // (label - 1).int() > -1; labels beyond the map would index out of it
        cls = ok ? (int)lab : -1;
        const float width = b[3] / c.vx / c.fac;
        const float length = b[4] / c.vy / c.fac;
        ok = ok && width > 0.f && length > 0.f;
        if (ok) {
            const float r = gaussian_radius_f32(length, width, c);
            radius = max(c.min_radius, (r < 1e9f) ? (int)r : 1000000000);
            if (c.range_f64) {  // float-valued POINT_CLOUD_RANGE: torch promotes this expression to float64
                cx = (float)(((double)b[0] - c.x0) / (double)c.vx / c.fac64);
                cy = (float)(((double)b[1] - c.y0) / (double)c.vy / c.fac64);
            } else {
                cx = (b[0] - (float)c.x0) / c.vx / c.fac;
                cy = (b[1] - (float)c.y0) / c.vy / c.fac;
            }
            // .to(torch.int32) truncates: cells 0 <= trunc(c) < size  <=>  -1 < c < size
            ok = cx > -1.f && cx < (float)c.fm_w && cy > -1.f && cy < (float)c.fm_h;
            if (ok) {
                x = (int)cx;
                y = (int)cy;
            }
        }
    }
    if (threadIdx.x == 0) {
        ind[k] = ok ? (int64_t)y * c.fm_w + x : 0;
        mask[k] = ok ? 1 : 0;
        float* a = anno + (int64_t)k * 8;
        if (ok) {
            a[0] = cx - (float)x;
            a[1] = cy - (float)y;
            a[2] = b[2];
            a[3] = logf(b[3]);
            a[4] = logf(b[4]);
            a[5] = logf(b[5]);
            a[6] = sinf(b[6]);
            a[7] = cosf(b[6]);
        } else {
            for (int j = 0; j < 8; ++j) a[j] = 0.f;
        }
    }
    if (!ok) return;
    const int left = min(x, radius), right = min(c.fm_w - x, radius + 1);
    const int top = min(y, radius), bottom = min(c.fm_h - y, radius + 1);
    const int ww = left + right, hh = top + bottom;
    const double sigma = (double)(2 * radius + 1) / 6.0;
    const double den = 2.0 * sigma * sigma;
    float* hm = heat + (int64_t)cls * c.fm_h * c.fm_w;
    for (int e = threadIdx.x; e < ww * hh; e += 64) {
        const int dy = e / ww - top, dx = e % ww - left;
        double g = exp(-(double)(dx * dx + dy * dy) / den);
        if (g < 2.220446049250313e-16) g = 0.0;  // h[h < eps * h.max()] = 0 (h.max() is the centre value, 1)
        const float v = (float)g;
        atomicMax(reinterpret_cast<int*>(hm + (int64_t)(y + dy) * c.fm_w + (x + dx)), __float_as_int(v));
    }
}

// per element of the (HW, C) class map: focal-loss term, positive flag, unnormalised d term / d logit
__global__ void k_center_focal(const float* __restrict__ cls, int ld, const float* __restrict__ heat, int64_t hw, int nc,
                               float* __restrict__ term, float* __restrict__ posf, float* __restrict__ grad, int ld_g) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= hw * nc) return;
    const int64_t cell = e / nc;
    const int ch = (int)(e % nc);
    const float z = cls[cell * ld + ch];
    const float t = heat[(int64_t)ch * hw + cell];
    const float sg = 1.f / (1.f + expf(-z));
    const float lo = 1e-4f, hi = 1.f - 1e-4f;
    const float p = fminf(fmaxf(sg, lo), hi);
    const bool inside = sg >= lo && sg <= hi;
    const float eps = 1e-12f;
    const float pos = (t == 1.f) ? 1.f : 0.f;
    const float omt = 1.f - t;
    const float negw = (omt * omt) * (omt * omt);
    const float lp = logf(p + eps), lq = logf(1.f - p + eps);
    const float omp = 1.f - p;
    term[e] = -lp * (omp * omp) * pos + -lq * (p * p) * negw;
    posf[e] = pos;
    if (grad) {
        const float dpos = (-(omp * omp) / (p + eps) + 2.f * omp * lp) * pos;
        const float dneg = ((p * p) / (omp + eps) - 2.f * p * lq) * negw;
        grad[cell * ld_g + ch] = inside ? (dpos + dneg) * (p * omp) : 0.f;
    }
}

struct CenterLossCfg {
    int max_objs;
    float cls_weight, loc_weight;
    float code_w[8];
};

// one workgroup: finishes both losses, scatters the regression gradient in slot order (duplicate cells add up in a
// fixed order), and leaves the class-gradient scale in sums[2]
__global__ void __launch_bounds__(64) k_center_finish(const float* __restrict__ box, int ld_box, const float* __restrict__ anno,
                                                      const int64_t* __restrict__ ind, const uint8_t* __restrict__ mask,
                                                      CenterLossCfg c, float* __restrict__ sums, float* __restrict__ losses,
                                                      float* __restrict__ gbox, int ld_gbox) {
    __shared__ float part[8];
    const int j = threadIdx.x;
    float num = 0.f;
    for (int k = 0; k < c.max_objs; ++k) num += mask[k] ? 1.f : 0.f;
    const float avg_loc = num + 1e-4f;
    if (j < 8) {
        float acc = 0.f;
        for (int k = 0; k < c.max_objs; ++k) {
            const float tb = anno[(int64_t)k * 8 + j];
            const float w = (mask[k] ? 1.f : 0.f) * (isnan(tb) ? 0.f : 1.f) * c.code_w[j];
            const float d = box[ind[k] * ld_box + j] - tb;
            acc += fabsf(d) * w;
            if (gbox) {
                const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                gbox[ind[k] * ld_gbox + j] += sgn * w * (c.loc_weight / avg_loc);
            }
        }
        part[j] = acc;
    }
    __syncthreads();
    if (j == 0) {
        float loc = 0.f;
        for (int q = 0; q < 8; ++q) loc += part[q];
        const float avg = fmaxf(sums[1], 1.f);
        const float lc = sums[0] / avg * c.cls_weight;
        const float ll = loc / avg_loc * c.loc_weight;
        losses[0] = lc;
        losses[1] = ll;
        losses[2] = lc + ll;
        sums[2] = c.cls_weight / avg;
    }
}

__global__ void k_center_scale(float* __restrict__ g, int ld, int nc, int64_t hw, const float* __restrict__ sums) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= hw * nc) return;
    g[(e / nc) * ld + (e % nc)] *= sums[2];
}

}  // namespace insmos

using namespace insmos;

extern "C" int insmos_center_assign_targets(const float* gt_boxes8, int n_gt, int max_objs, int num_class, int fm_w, int fm_h,
                                            double range_x0, double range_y0, int range_is_f64, float voxel_x, float voxel_y,
                                            int out_size_factor, double gaussian_overlap, int min_radius, float* heatmap,
                                            float* anno_box, int64_t* ind, uint8_t* mask, void* stream) {
    if (n_gt < 0 || max_objs <= 0 || num_class <= 0 || fm_w <= 0 || fm_h <= 0 || out_size_factor <= 0 || !heatmap ||
        !anno_box || !ind || !mask || (n_gt > 0 && !gt_boxes8) || !(voxel_x > 0.f) || !(voxel_y > 0.f))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    CenterTargetCfg c;
    c.n_gt = n_gt < max_objs ? n_gt : max_objs;
    c.max_objs = max_objs;
    c.num_class = num_class;
    c.fm_w = fm_w;
    c.fm_h = fm_h;
    c.min_radius = min_radius;
    c.range_f64 = range_is_f64 ? 1 : 0;
    c.x0 = range_x0;
    c.y0 = range_y0;
    c.fac64 = (double)out_size_factor;
    c.vx = voxel_x;
    c.vy = voxel_y;
    c.fac = (float)out_size_factor;
    const double ov = gaussian_overlap;
    c.c_1m = (float)(1.0 - ov);
    c.c_1p = (float)(1.0 + ov);
    c.c_m2 = (float)(-2.0 * ov);
    c.c_m1 = (float)(ov - 1.0);
    c.c_4a3 = (float)(4.0 * (4.0 * ov));
    ProfScope ps(KK_FILL, s);
    HIP_TRY(hipMemsetAsync(heatmap, 0, (size_t)num_class * fm_h * fm_w * sizeof(float), s));
    INSMOS_LAUNCH(k_center_targets, dim3(max_objs), dim3(64), 0, s, gt_boxes8, c, heatmap, anno_box, ind, mask);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}

// ws: term (hw * nc) | positive flags (hw * nc) | column-sum scratch | sums (4)
extern "C" size_t insmos_center_head_loss_ws_floats(int64_t hw, int num_class) {
    const int64_t n = hw * num_class;
    return (size_t)2 * (size_t)n + insmos_col_sum_ws_floats(n, 1) + 16;
}

extern "C" int insmos_center_head_loss(const float* cls_preds, int ld_cls, const float* box_preds, int ld_box, int64_t hw,
                                       int num_class, const float* heatmap, const float* anno_box, const int64_t* ind,
                                       const uint8_t* mask, int max_objs, float cls_weight, float loc_weight,
                                       const float* code_weights_host, float* losses, float* grad_cls, int ld_gcls,
                                       float* grad_box, int ld_gbox, float* ws, void* stream) {
    if (hw <= 0 || num_class <= 0 || max_objs <= 0 || !cls_preds || !box_preds || !heatmap || !anno_box || !ind || !mask ||
        !code_weights_host || !losses || !ws || ld_cls < num_class || ld_box < 8 || (grad_cls && ld_gcls < num_class) ||
        (grad_box && ld_gbox < 8))
        return INSMOS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = hw * num_class;
    float* term = ws;
    float* posf = ws + n;
    float* cs = ws + 2 * n;
    float* sums = cs + insmos_col_sum_ws_floats(n, 1);
    CenterLossCfg c;
    c.max_objs = max_objs;
    c.cls_weight = cls_weight;
    c.loc_weight = loc_weight;
    for (int j = 0; j < 8; ++j) c.code_w[j] = code_weights_host[j];
    ProfScope ps(KK_CONFUSION, s);
    INSMOS_LAUNCH(k_center_focal, dim3(cdiv(n, 256)), dim3(256), 0, s, cls_preds, ld_cls, heatmap, hw, num_class, term, posf,
                  grad_cls, ld_gcls);
    int rc = insmos_col_sum(term, 1, 1, n, sums, 0, cs, stream);
    if (rc) return rc;
    rc = insmos_col_sum(posf, 1, 1, n, sums + 1, 0, cs, stream);
    if (rc) return rc;
    if (grad_box) HIP_TRY(hipMemset2DAsync(grad_box, (size_t)ld_gbox * 4, 0, (size_t)8 * 4, (size_t)hw, s));
    INSMOS_LAUNCH(k_center_finish, dim3(1), dim3(64), 0, s, box_preds, ld_box, anno_box, ind, mask, c, sums, losses, grad_box,
                  ld_gbox);
    if (grad_cls) INSMOS_LAUNCH(k_center_scale, dim3(cdiv(n, 256)), dim3(256), 0, s, grad_cls, ld_gcls, num_class, hw, sums);
    HIP_TRY(hipGetLastError());
    return INSMOS_OK;
}
