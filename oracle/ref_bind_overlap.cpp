// oracle/ref_bind_overlap.cpp -- C-ABI shim exposing the reference's rotated-rectangle OVERLAP AREA
// (`box_overlap`, /root/reference/models/bbox_post_process/src/iou3d_cpu.cpp:128-206).  That function is `inline` in the
// reference translation unit, so this shim is compiled as ONE unit with that file, which make_ref.py hands to the
// compiler from where it lies (-include, nothing is copied); the only code here is the wrapper below.
// TEST INFRASTRUCTURE ONLY -- output goes to oracle/_ref/ (git-ignored).
extern "C" int ref_boxes_overlap_bev(const float* a, int na, const float* b, int nb, float* out) {
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(long)i * nb + j] = box_overlap(a + i * 7, b + j * 7);
    return 1;
}
