/*
 * include/insmos_hip.h -- C ABI of libinsmos_hip.so: the MI355X (gfx950) kernels behind the InsMOS
 * sparse-voxel inference hot path.  Plain pointers and sizes only; every pointer is DEVICE memory
 * unless its name ends in _host; every call is asynchronous on `stream` (a hipStream_t passed as
 * void*) and returns 0 or a negative INSMOS_E* code -- never exit() like the reference's extensions
 * (iou3d_nms.cpp:14-38).  Counts are produced in device memory (int32 slots of `counts`); the caller
 * reads them when it needs them.  Scratch comes from a caller-provided workspace (`ws`, `ws_bytes`);
 * there is no hidden allocation and no global state except the optional profiler.
 *
 * Each entry point names the reference interface it replaces (file:line under /root/reference).
 *
 * Coordinate conventions
 *   4D (MotionNet / MinkowskiEngine side): coords int32 (n,4) [x,y,z,t] in units of the finest
 *     voxel, rows in ascending key order, key = (t+32768)<<48 | morton3(x+32768,y+32768,z+32768).
 *   3D (UNetV2 / spconv side): coords int32 (n,4) [b=0,z,y,x] (spconv's indices layout), keys are
 *     linear (z*H + y)*W + x in the level's spatial shape [D,H,W].
 * Feature matrices are row-major fp32 with an explicit leading dimension (`ld_*`, in floats) so that
 * channel concatenation is free: producers write into column slices of a wider row.
 */
#ifndef INSMOS_HIP_H
#define INSMOS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INSMOS_OK 0
#define INSMOS_EINVAL (-1)    /* bad argument */
#define INSMOS_EHIP (-2)      /* a HIP runtime call failed; see insmos_last_hip_error() */
#define INSMOS_EWORKSPACE (-3) /* workspace too small */
#define INSMOS_EBATCH (-4)     /* launch set too large (32-bit table offsets, or a window's time range beside B - 1 others): run it as smaller sets */

int insmos_version(void);
int insmos_last_hip_error(void);

/* ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ---- */
int insmos_prof_enable(int on);
int insmos_prof_reset(void);
/* per-launch durations (ms) of one kernel kind in record order + the four integers its launch site attached (convolutions:
 * K, Cin, Cout, rows computed); returns the number of entries written (<= max) */
int insmos_prof_read_spans(int kind_id, int max, double* ms_host, int64_t* meta4_host);
/* Synchronises, then writes up to `max` entries; returns the number of kernel kinds. */
int insmos_prof_read(int max, int* kind_ids_host, double* total_ms_host, int64_t* launches_host);
const char* insmos_prof_name(int kind_id);
/* One kernel kind when launches overlap on several streams: length of the union of the launches' event intervals,
 * the plain sum of their durations, and their number. */
int insmos_prof_read_union(int kind_id, double* union_ms_host, double* sum_ms_host, int64_t* launches_host);

/* ------------------------------------------------------------------------------------------------
 * insmos_quantize4d -- replaces ME.utils.sparse_collate + ME.TensorField(...).sparse() as called
 * at models/backbones_3d/motionnet.py:22-36, plus the t==0 row selection of :42-45.
 *   points (n,ld_pts) fp32 rows [x,y,z,intensity,t]; quant[4] = [ds,ds,ds,dt] (host values).
 *   coords = floor(fp32 [x,y,z,t] / fp32 quant) (IEEE division), unique, canonical key order.
 * Outputs: keys (cap n) u64, coords (cap n,4) i32, inverse (n) i32 point->voxel row,
 *          cur_index (cap n) i32: ascending point indices whose quantised t == 0,
 *          counts[0] = #voxels, counts[1] = #current points, counts[2] = #points outside the
 *          +-32768-voxel key window (must be 0).
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_quantize4d_ws_bytes(int64_t n);
/* insmos_quantize4d_ex: compact_keys != 0 sorts 40-bit keys (5 radix passes instead of 8) that order exactly like the
 * canonical ones when every point lies within +-2048 voxels and 16 time steps; counts[3] = #points outside that box --
 * if it is not 0 the outputs are invalid and the call must be repeated with compact_keys = 0. */
int insmos_quantize4d_ex(const float* points, int64_t n, int ld_pts, const float* quant_host, uint64_t* keys,
                         int32_t* coords, int32_t* inverse, int32_t* cur_index, int32_t* counts, void* ws,
                         size_t ws_bytes, int compact_keys, void* stream);
int insmos_quantize4d(const float* points, int64_t n, int ld_pts, const float* quant_host, uint64_t* keys,
                      int32_t* coords, int32_t* inverse, int32_t* cur_index, int32_t* counts, void* ws,
                      size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_level_down4d -- the coordinate map MinkowskiConvolution(kernel [2,2,2,1], stride [2,2,2,1])
 * creates (models/MinkowskiEngine/minkunet.py:63-89): floor(c / 2s) * 2s in x,y,z, t untouched.
 * Because keys are Morton-in-space this is a prefix de-duplication of the sorted key array.
 *   shift = log2 of the OUTPUT tensor stride (1,2,3).  parent (n) i32 = fine row -> coarse row.
 *   child_start (cap n) i32 = first fine row of each coarse voxel, child_mask (cap n) u32 = which of its
 *   8 octants (x | y<<1 | z<<2) are occupied, in its LOW BYTE; bits 8..31 repeat child_start when n < 2^24 (0 otherwise), so
 *   that the table kernels fetch one word per coarse neighbour (both arrays optional).  counts[0] = #coarse voxels.
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_level_down4d_ws_bytes(int64_t n);
int insmos_level_down4d(const uint64_t* keys, int64_t n, int shift, uint64_t* out_keys, int32_t* out_coords,
                        int32_t* parent, int32_t* child_start, uint32_t* child_mask, int32_t* counts, void* ws,
                        size_t ws_bytes, void* stream);
/* Levels 1 .. n_levels (<= 3) of that hierarchy -- the coordinate maps conv1p1s2, conv2p2s2, conv3p4s2 create one from the other
 * (minkunet.py:139-160) -- in ONE chain of launches: every level's row count stays on the device and is what the next level's
 * kernels read, so a caller waits once for all counts instead of once per level.  Host pointer lists, entry l - 1 = the arrays of
 * level l, each with room for n0 rows; chain (device, 4 + 16 * (n_levels + 1) int32): [l - 1] = rows of level l, [4 + 16 * l + d] =
 * insmos_tslice_starts_batched(level l's keys, max_d 16)[d] for l = 0 .. n_levels (0 = keys0).  n0 < 2^24; ws:
 * insmos_level_down4d_ws_bytes(n0).  Same arrays (below each count) as the per-level calls. */
int insmos_level_down4d_chain(const uint64_t* keys0, int64_t n0, int n_levels, int B, uint64_t* const* out_keys,
                              int32_t* const* out_coords, int32_t* const* parent, int32_t* const* child_start,
                              uint32_t* const* child_mask, int32_t* chain, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Search-free kernel maps from the Morton hierarchy (same tables insmos_build_nbr would produce, same
 * reference call sites: minkunet.py:55-124).  A coarse voxel's children are contiguous in key order,
 * so with child_start / child_mask (the 8-bit octant occupancy) from insmos_level_down4d:
 *   insmos_nbr_from_coarse: fine-level table for taps delta_host (K,4) ([x,y,z,t] in finest-voxel units,
 *     |spatial offset| <= 2 fine strides, |dt| <= 1) from the COARSE level's 81-tap (3x3x3x3, x fastest)
 *     table: neighbour = child of the matching neighbour block -- no key search at all.
 *   insmos_nbr_down_up: the kernel [2,2,2,1] / stride [2,2,2,1] conv map (dn: (8, n_c), tap = octant)
 *     and its transpose (up: (8, n_f)).
 *   fine_shift = log2 of the fine level's tensor stride.  mask16 outputs as in insmos_build_nbr.
 * ---------------------------------------------------------------------------------------------- */
int insmos_nbr_from_coarse(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                           const int32_t* coarse_nbr81, int64_t n_c, const int32_t* child_start,
                           const uint32_t* child_mask, const int32_t* delta_host, int K, int32_t* nbr,
                           uint32_t* mask16, void* stream);
/* 3x3x3x3 table specialisation of insmos_nbr_from_coarse: one thread per voxel fetches its <= 24 coarse
 * entries once (LDS-cached) and resolves all 81 taps with bit arithmetic; masks need no atomics. */
int insmos_nbr81_from_coarse(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                             const int32_t* coarse_nbr81, int64_t n_c, const int32_t* child_start,
                             const uint32_t* child_mask, int32_t* nbr, uint32_t* mask16, void* stream);
/* First MotionNet layer (minkunet.py:55-60, kernel [5,5,5,1], 1 -> 8 channels) for a CONSTANT input feature
 * (motionnet.py:29-32 feeds 0.5 on every point): out[o, 0:8] = relu?(bias8 + sum over existing taps k of
 * w125x8[k, 0:8]) with w125x8 = value * BN-folded kernel.  No 125-tap table is built and nothing is gathered. */
/* rows [row0, n_f) of the same table only (row0 rounded down to a multiple of 16; rows below are left untouched):
 * the level-0 table of MotionNet is only read by block8, i.e. at the last two scans' voxels */
int insmos_nbr81_from_coarse_rows(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent,
                                  int fine_shift, const int32_t* coarse_nbr81, int64_t n_c, const int32_t* child_start,
                                  const uint32_t* child_mask, int32_t* nbr, uint32_t* mask16, void* stream);
/* insmos_nbr81_from_coarse_rows with SPARSE stores: the 16 entries of a (16-row group, tap) pair that is not in the group's mask16
 * are left unwritten -- insmos_sparse_conv (16-row tiles) reads a group's taps through its mask only; half of the table's bytes.
 * mask16 is required.  Not for consumers that read every entry: the training kernels, multi-group tiles, and the derivation of the
 * next finer level's table / the first layer's tap resolver (insmos_nbr81_from_coarse*, insmos_const_conv125_from_coarse) -- i.e. the
 * finest level's table only. */
int insmos_nbr81_from_coarse_rows_sparse(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent,
                                         int fine_shift, const int32_t* coarse_nbr81, int64_t n_c, const int32_t* child_start,
                                         const uint32_t* child_mask, int32_t* nbr, uint32_t* mask16, void* stream);
int insmos_const_conv125_from_coarse(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                                     const int32_t* coarse_nbr81, int64_t n_c, const int32_t* child_start,
                                     const uint32_t* child_mask, const float* w125x8, const float* bias8, float* out,
                                     int ld_out, int relu, void* stream);
/* ... on occupancy cubes (coords.hip: k_parent_cubes / k_const_conv125): cubes_ws = 48 bytes of scratch per COARSE voxel, 16-byte
 * aligned (null = the per-tap resolver above); same bits, 1.5x faster on the S0 windows. */
int insmos_const_conv125_cubes(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift,
                               const int32_t* coarse_nbr81, const uint32_t* coarse_mask16 /* null = fully written table */,
                               int64_t n_c, const int32_t* child_start, const uint32_t* child_mask, const float* w125x8,
                               const float* bias8, float* out, int ld_out, int relu, void* cubes_ws, void* stream);
/* insmos_nbr81_from_coarse_rows from a coarse table that was itself written with SPARSE stores: coarse_mask16 = that table's
 * mask array (its entries outside a group's mask are unwritten memory and count as "no neighbour"; null = fully written);
 * sparse_stores != 0 writes this table sparsely too (mask16 required).  Same tables wherever they are defined. */
int insmos_nbr81_from_coarse_rows_masked(const int32_t* fine_coords, int64_t n_f, int64_t row0, const int32_t* parent,
                                         int fine_shift, const int32_t* coarse_nbr81, const uint32_t* coarse_mask16, int64_t n_c,
                                         const int32_t* child_start, const uint32_t* child_mask, int32_t* nbr, uint32_t* mask16,
                                         int sparse_stores, void* stream);
int insmos_nbr_down_up(const int32_t* fine_coords, int64_t n_f, const int32_t* parent, int fine_shift, int64_t n_c,
                       const int32_t* child_start, const uint32_t* child_mask, int32_t* dn, uint32_t* dn_mask16,
                       int32_t* up, uint32_t* up_mask16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_build_nbr -- the kernel map / indice pairs ("rulebook") in output-stationary form:
 *   nbr[k*n_out + o] = input row read by output o through tap k, or -1.
 * Query coordinate per (o,k):  q = out_coords[o]*mul + delta[k];  if div>1: require q % div == 0,
 * q /= div  (component-wise over the 4 coordinate slots; slot meaning per key_mode).
 * Replaces: ME kernel maps behind MinkowskiConvolution / MinkowskiConvolutionTranspose
 * (minkunet.py:55-124) and spconv's indice-pair generation behind SubMConv3d / SparseConv3d /
 * SparseInverseConv3d (models/backbones_3d/spconv_unet.py:120-207).
 *   key_mode 0: 4D Morton keys, coords [x,y,z,t];  key_mode 1: 3D linear keys, coords [b,z,y,x],
 *   in_shape_host = [D,H,W] of the INPUT level.
 *   in_keys ascending (n_in); in_perm (n_in) maps sorted position -> input row (NULL = identity;
 *   an entry of -1 marks a voxel dropped by the max-voxel cap).
 *   delta_host (K,4) i32, mul_host[4], div_host[4] are HOST arrays (tiny, copied by value).
 *   mask16 (optional, (ceil(n_out/16), 4) u32): bit k of group g is set iff some row of the 16-row
 *   group g has a neighbour on tap k -- the active-tap sets insmos_sparse_conv iterates over.
 * ---------------------------------------------------------------------------------------------- */
int insmos_build_nbr(const int32_t* out_coords, int64_t n_out, const uint64_t* in_keys, const int32_t* in_perm,
                     int64_t n_in, int key_mode, const int32_t* in_shape_host, const int32_t* delta_host, int K,
                     const int32_t* mul_host, const int32_t* div_host, int32_t* nbr, uint32_t* mask16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_voxelize_mean -- replaces spconv PointToVoxel.generate_voxel_with_id
 * (models/backbones_3d/voxel_generate.py:19-28) fused with MeanVFE (models/backbones_2d/mean_vfe.py:47-52).
 * Deterministic restatement of the CPU semantics: first-come voxel ids (cap max_voxels), first
 * max_pts points of each voxel averaged, pc_voxel_id = -1 for dropped / out-of-range points.
 *   points (n,ld_pts) fp32, first 3 columns x,y,z, first n_feat columns are the features (<= 8).
 *   range_host[6] = [x0,y0,z0,x1,y1,z1], vsize_host[3].  Grid = round((hi-lo)/vsize) per axis.
 * Outputs: feat (cap,ld_feat) fp32 (columns >= n_feat zeroed), coords (cap,4) [0,z,y,x],
 *          num_points (cap) i32, pc_voxel_id (n) i64, and the level-1 search structure
 *          ukeys (cap_seg) u64 ascending linear keys + uperm (cap_seg) i32 (voxel row or -1).
 *          counts[0] = #voxels kept, counts[1] = #distinct occupied cells (= len of ukeys),
 *          counts[2] = #points in range.     cap = max_voxels rows, cap_seg = n rows.
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_voxelize_mean_ws_bytes(int64_t n);
int insmos_voxelize_mean(const float* points, int64_t n, int ld_pts, int n_feat, const float* range_host,
                         const float* vsize_host, int max_voxels, int max_pts, float* feat, int ld_feat,
                         int32_t* coords, int32_t* num_points, int64_t* pc_voxel_id, uint64_t* ukeys,
                         int32_t* uperm, int32_t* counts, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_down_coords3d -- output coordinate set of spconv.SparseConv3d(kernel, stride, padding)
 * (spconv_unet.py:135-160: k3 s2 p1 x3, (3,1,1) s(2,1,1) p0): o active iff some tap reads an active
 * input (i = o*stride - pad + k), 0 <= o < out_shape.  Canonical order: ascending linear index.
 *   out_keys/out_coords capacity = min(n_in*K, prod(out_shape)) rows; counts[0] = #outputs.
 *   Implementation: occupancy bitmap of the output grid + popcount rank scan (no sort).
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_down_coords3d_ws_bytes(const int32_t* out_shape_host);
int insmos_down_coords3d(const int32_t* in_coords, int64_t n_in, const int32_t* ksize_host,
                         const int32_t* stride_host, const int32_t* pad_host, const int32_t* out_shape_host,
                         uint64_t* out_keys, int32_t* out_coords, int32_t* counts, void* ws, size_t ws_bytes,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_sparse_conv -- gather -> MFMA (v_mfma_f32_16x16x4_f32, exact fp32) -> fused epilogue.
 *   out[o, 0:cout] = epi( sum_k in[nbr[k][o], 0:cin] @ W[k] + bias )
 *   epi(v): if relu_pre v=max(v,0); if res_mode==1 v+=res[o,c]; if res_mode==2 v+=res[o,2c]+res[o,2c+1];
 *           if relu_post v=max(v,0)
 * Replaces the gather-GEMM-scatter inside MinkowskiConvolution(+Transpose) (minkunet.py:139-181),
 * spconv SubMConv3d/SparseConv3d/SparseInverseConv3d (spconv_unet.py:297-402) with eval-mode
 * BatchNorm folded into W/bias, ReLU, the residual adds of BasicBlock / SparseBasicBlock
 * (spconv_unet.py:86-106) and UR_block's channel_reduction add (spconv_unet.py:213-238); with a
 * full-grid nbr table it also serves the dense BEV convs (base_bev_backbone.py:33-61) and the 1x1
 * heads (center_head.py:47-54) in NHWC.
 *   nbr == NULL  -> K must be 1, identity map (1x1 conv / Linear).
 *   mask16: the table's active-tap sets from insmos_build_nbr, or NULL (every tap visited).
 *   in has n_in rows; cin must be 4, 8 or a multiple of 16 (pad with ZERO columns); ld_in % 4 == 0, `in`
 *   16-byte aligned, n_in*ld_in*4 < 2^31 (gathers are 32-bit-offset buffer loads).
 *   wpacked: tap-major MFMA-fragment layout produced by insmos_pack_weights_host (below).
 *   bias: (cout_pad16) fp32.
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_packed_weight_floats(int K, int cin, int cout);
/* Host helper: taps (K,cin_real,cout_real) fp32 row-major -> packed layout for (cin,cout) padded. */
int insmos_pack_weights_host(const float* taps_host, int K, int cin_real, int cout_real, int cin, int cout,
                             float* packed_host);
int insmos_sparse_conv(const float* in, int64_t n_in, int ld_in, int cin, const int32_t* nbr, const uint32_t* mask16,
                       int K, int64_t n_out, const float* wpacked, const float* bias, float* out, int ld_out, int cout,
                       const float* res, int ld_res, int res_mode, int relu_pre, int relu_post, void* stream);

/* Tuning hook (tools/conv_tune.py): force generic non-identity layers onto tile shape (16*cot channels x
 * 16*jt rows) with an operand ring of depth `ring`; cot == 0 restores the built-in cost model. */
/* insmos_sparse_conv on the output rows [row0, n_out) only (row0 is rounded down to a multiple of 16); pointers and
 * n_out describe the FULL arrays / table.  Used to skip rows nothing consumes: MotionNet's features are read at the
 * current scan's voxels only (motionnet.py:38-48), rows are ordered by time first, and a 3^4 convolution widens the
 * needed time range by one scan per layer -- so the decoder-side layers of the finer levels run on a suffix of their
 * rows.  insmos_tslice_starts: starts[d] = first row with t >= t_last - d of a sorted 4D key array. */
int insmos_sparse_conv_rows(const float* in, int64_t n_in, int ld_in, int cin, const int32_t* nbr, const uint32_t* mask16,
                            int K, int64_t n_out, int64_t row0, const float* wpacked, const float* bias, float* out,
                            int ld_out, int cout, const float* res, int ld_res, int res_mode, int relu_pre, int relu_post,
                            void* stream);
int insmos_tslice_starts(const uint64_t* keys, int64_t n, int max_d, int32_t* starts, void* stream);
/* ---- tap-compacted form of the 81-tap 4D layers (csrc/spconv_tapc.hip; MinkowskiConvolution kernel_size 3, dimension 4:
 * models/MinkowskiEngine/minkunet.py:63-124, resnet.py:110-119).  The 16-row tiles of insmos_sparse_conv pay a full MFMA pass
 * for every (16-row group, tap) slot ANY row of the group uses -- about twice the useful passes on the 4D levels.  Here the rows of
 * a 128-row block that HAVE a tap are packed into dense groups of 16 per tap ("items"), built once per neighbour table:
 *   insmos_tapc_blocks(n_out)        128-row blocks of a table
 *   insmos_tapc_words(K, n_out, c)   uint32 words of the item table for c tap classes (its counts: blocks * c int32)
 *   insmos_tapc_build                the item table of a DENSE (every entry written) K x n_out neighbour table, blocks from row0 / 128
 *                                    on (layers that run on a row suffix: the blocks below stay unwritten)
 *   insmos_tapc_build_masked         the same for a SPARSE table (mask16 = its active-tap bits per 16-row group, as insmos_build_nbr*
 *                                    write them: entries outside a group's mask are unwritten memory and are not read) -- the 27-tap
 *                                    SubMConv3d tables of the 3D UNet (models/backbones_3d/spconv_unet.py:120-207); mask16 NULL = dense
 *   insmos_conv_tap_classes          the partial chains insmos_sparse_conv sums a layer's taps in: 1 = one chain over all taps,
 *                                    4 = tap-split tiles (taps k % 4 == 0..3, summed ((c0 + c1) + c2) + c3), 0 = neither (chunk-split
 *                                    tiles, probe settings); a function of the layer's shape only.  The item table must be built
 *                                    with that class count: the class count IS the summation order.
 *   insmos_sparse_conv_tapc_rows     insmos_sparse_conv_rows on the item table: the SAME BITS (accumulators parked in LDS between
 *                                    taps, the same fmaf chain per output element).  Cin in {8, 16, 32, 48}, Cout <= 32,
 *                                    n_in < 2^23 - 1; EINVAL otherwise (the caller stays on insmos_sparse_conv_rows). */
size_t insmos_tapc_blocks(int64_t n_out);
size_t insmos_tapc_words(int K, int64_t n_out, int n_classes);
int insmos_tapc_build(const int32_t* nbr, int K, int64_t n_out, int64_t row0, int n_classes, uint32_t* items, int32_t* n_items,
                      void* stream);
int insmos_tapc_build_masked(const int32_t* nbr, const uint32_t* mask16, int K, int64_t n_out, int64_t row0, int n_classes,
                             uint32_t* items, int32_t* n_items, void* stream);
int insmos_conv_tap_classes(int K, int cin, int cout, int masked);
int insmos_sparse_conv_tapc_rows(const float* in, int64_t n_in, int ld_in, int cin, const uint32_t* items, const int32_t* n_items,
                                 int n_classes, int K, int64_t n_out, int64_t row0, const float* wpacked, const float* bias,
                                 float* out, int ld_out, int cout, const float* res, int ld_res, int res_mode, int relu_pre,
                                 int relu_post, void* stream);
/* ---- B windows in ONE set of launches (replaces the per-item loop of InsMOS_Model.forward, models/models.py:313) -------
 * The batch entry points below (suffix _windows / _b) are the single-window ones with a leading window dimension; B = 1
 * gives exactly the single-window results, and every window of a batch gets the bits it gets alone.  B <= 16.
 *   4D branch: the window index is folded into the time coordinate, t' = floor(t / dt) * B + b (the t column of coords and
 *   the time field of keys), so every table / convolution entry point works unchanged when the SEARCHED table is built on
 *   time offsets scaled by B.  3D branch: column 0 of the (n, 4) indices is the window (spconv's batch column), keys are
 *   b * cells + (z*H + y)*W + x, rows are window-major.
 * insmos_quantize4d_windows: points_host[b] / n_points_host[b] = device pointer / point count of window b (host arrays);
 *   outputs as insmos_quantize4d_ex over the concatenation of the windows (window-major point index);
 *   counts (5 + B int32): [0] voxels, [1] current points, [2] outside the key window, [3] outside the compact-key box,
 *   [4 + b] first current point of window b in cur_index ([4 + B] = all).
 * insmos_tslice_starts_batched: starts[d] = first row with floor(t' / B) >= tq_last - d. */
int insmos_quantize4d_windows(const float* const* points_host, const int64_t* n_points_host, int B, int ld_pts,
                              const float* quant_host, uint64_t* keys, int32_t* coords, int32_t* inverse, int32_t* cur_index,
                              int32_t* counts, void* ws, size_t ws_bytes, int compact_keys, void* stream);
int insmos_build_current_points_windows(const float* const* points_host, const int64_t* n_points_host, int B, int ld_pts,
                                        const float* motion, int ld_motion, const int32_t* inverse, const int32_t* cur_index,
                                        int64_t n_cur, float* cur, int ld_cur, void* stream);
/* the same in parts: part 0 = whole rows; part 1 = [x, y, z, r, 0, ..] (motion may be NULL: known before MotionNet has run, all the
 * voxeliser's coordinate phase reads); part 2 = the motion columns 4..6 into rows part 1 wrote (motionnet.py:42-48). */
int insmos_build_current_points_part(const float* const* points_host, const int64_t* n_points_host, int B, int ld_pts,
                                     const float* motion, int ld_motion, const int32_t* inverse, const int32_t* cur_index,
                                     int64_t n_cur, float* cur, int ld_cur, int part, void* stream);
/* win_start (device, B + 1 int32): first point of each window in the window-major point array (null: one window).  Every
 * window is voxelised in its own first-seen order and capped at max_voxels on its own (the reference calls VoxelGenerate
 * per batch item, models/models.py:326); voxel rows are window-major, pc_voxel_id holds batch-wide rows.
 * ukeys = b * key_cells + cell: key_cells = D*H*W of the spatial shape the level-1 tables are searched with (one cell
 * deeper than the voxel grid, spconv_unet.py:114); 0 = the voxel grid's own cell count.
 * counts: [0] voxel rows, [1] occupied cells, [2] in-range points, with win_start also [4 + b] = first row of window b. */
int insmos_voxelize_mean_windows(const float* points, int64_t n, int ld_pts, int n_feat, const int32_t* win_start, int B,
                                 int64_t key_cells, const float* range_host, const float* vsize_host, int max_voxels, int max_pts, float* feat,
                                 int ld_feat, int32_t* coords, int32_t* num_points, int64_t* pc_voxel_id, uint64_t* ukeys,
                                 int32_t* uperm, int32_t* counts, void* ws, size_t ws_bytes, void* stream);
/* The voxeliser in two phases (round 3: the 3D coordinate branch runs beside MotionNet).  phase 1 = everything that depends on the
 * points' POSITIONS only (coords, num_points, pc_voxel_id, ukeys / uperm, counts); phase 2 (same arguments and workspace, the
 * workspace untouched in between) = the MeanVFE feature means (mean_vfe.py:47-52), which need the motion columns.  phase 0 = both =
 * insmos_voxelize_mean_windows.  phase 1 + phase 2 give phase 0's bits. */
int insmos_voxelize_windows_phased(const float* points, int64_t n, int ld_pts, int n_feat, const int32_t* win_start, int B,
                                   int64_t key_cells, const float* range_host, const float* vsize_host, int max_voxels, int max_pts,
                                   float* feat, int ld_feat, int32_t* coords, int32_t* num_points, int64_t* pc_voxel_id,
                                   uint64_t* ukeys, int32_t* uperm, int32_t* counts, void* ws, size_t ws_bytes, int phase,
                                   void* stream);
size_t insmos_down_coords3d_ws_bytes_b(const int32_t* out_shape_host, int B);
int insmos_down_coords3d_b(const int32_t* in_coords, int64_t n_in, const int32_t* ksize_host, const int32_t* stride_host,
                           const int32_t* pad_host, const int32_t* out_shape_host, int B, uint64_t* out_keys,
                           int32_t* out_coords, int32_t* counts, void* ws, size_t ws_bytes, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Rank maps (round 3): the 3D kernel maps WITHOUT a key search.  A coordinate set of the (B, D, H, W) grid of a level as an
 * occupancy bitmap in 256-bit blocks (bits: insmos_rankmap_words u64) + the inclusive number of set bits up to each block
 * (blk_incl: a quarter as many i32): a cell's sorted position is two independent loads and three popcounts.
 *   insmos_rankmap_from_keys: from ascending unique cell keys b * cells + (z*H + y)*W + x (the voxeliser's ukeys: level 1).
 *   insmos_down_coords3d_rank = insmos_down_coords3d_b whose bitmap stays valid, in this form (the generated levels).
 *   insmos_build_nbr_rank = insmos_build_nbr(key_mode 1) over the INPUT level's rank map; in_perm as there.  Same table and
 *   same mask16 as the searched builder (tests/test_gpu_coords.py).  Same reference call sites: spconv's indice-pair
 *   generation behind SubMConv3d / SparseConv3d / SparseInverseConv3d (models/backbones_3d/spconv_unet.py:120-207).
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_rankmap_words(const int32_t* shape_host, int B);
size_t insmos_rankmap_ws_bytes(const int32_t* shape_host, int B);
int insmos_rankmap_from_keys(const uint64_t* keys, int64_t n, const int32_t* shape_host, int B, uint64_t* bits,
                             int32_t* blk_incl, void* ws, size_t ws_bytes, void* stream);
int insmos_down_coords3d_rank(const int32_t* in_coords, int64_t n_in, const int32_t* ksize_host, const int32_t* stride_host,
                              const int32_t* pad_host, const int32_t* out_shape_host, int B, uint64_t* out_keys,
                              int32_t* out_coords, int32_t* counts, uint64_t* bits, int32_t* blk_incl, void* ws,
                              size_t ws_bytes, void* stream);
int insmos_build_nbr_rank(const int32_t* out_coords, int64_t n_out, const uint64_t* bits, const int32_t* blk_incl,
                          const int32_t* in_perm, const int32_t* in_shape_host, const int32_t* delta_host, int K,
                          const int32_t* mul_host, const int32_t* div_host, int32_t* nbr, uint32_t* mask16, void* stream);
int insmos_build_nbr_rank_sparse(const int32_t* out_coords, int64_t n_out, const uint64_t* bits, const int32_t* blk_incl,
                                 const int32_t* in_perm, const int32_t* in_shape_host, const int32_t* delta_host, int K,
                                 const int32_t* mul_host, const int32_t* div_host, int32_t* nbr, uint32_t* mask16, void* stream);
                                 /* (sparse stores, see insmos_nbr81_from_coarse_rows_sparse; mask16 required) */
/* Several kernel maps in ONE launch (round 5; the native runner builds the 13 maps of the 3D branch this way): every job is one
 * insmos_build_nbr_rank call -- same entries, same masks.  All pointers of a job are device pointers except in_shape / delta / mul /
 * div (host; mul / div may be null = ones).  <= 16 jobs, K <= 32, <= 4 distinct offset sets, |offset| <= 127; a job with
 * n_out == 0 is skipped.  sparse_stores as insmos_build_nbr_rank_sparse (then every job needs mask16). */
typedef struct InsmosRankJob {
    const int32_t* out_coords;   /* (n_out, 4) [b, z, y, x] */
    const uint64_t* bits;        /* the INPUT level's rank map */
    const int32_t* blk_incl;
    const int32_t* in_perm;      /* sorted position -> row of the input level, or null */
    int32_t* nbr;                /* (K, n_out) */
    uint32_t* mask16;            /* ((n_out + 15) / 16, 4) or null */
    const int32_t* in_shape;     /* host, 3 */
    const int32_t* delta;        /* host, (K, 4) */
    const int32_t* mul;          /* host, 4, or null */
    const int32_t* div;          /* host, 4, or null */
    int64_t n_out;
    int32_t K;
    int32_t reserved;
} InsmosRankJob;
int insmos_build_nbr_rank_multi(const InsmosRankJob* jobs_host, int n_jobs, int sparse_stores, void* stream);
/* Row regrouping (no reference counterpart: spconv's rulebooks are pair lists, the row order of a level is an implementation
 * detail there too -- spconv_unet.py:120-207 never looks at it).  The output-stationary kernels pay one 16-row MFMA pass per
 * (16-row group, tap) slot any row of the group uses; rows are re-ordered inside blocks of block_rows (256 / 1024 / 4096)
 * consecutive rows by their 27-bit submanifold tap signature so that a group's rows want the same taps.
 *   insmos_regroup_rows3d: coords (n, 4) (b, z, y, x), bits = the level's rank-map bitmap -> new_coords (n, 4), new_of_old (n),
 *   old_of_new (n).  Sort key inside a block: (window, signature, parity class of (z, y, x), old row) -- window-major rows stay
 *   window-major; rows of equal signature are grouped by parity, which decides the valid taps of the strided / inverse maps.
 *   block_rows < 0: (window, parity class, signature, old row) in blocks of -block_rows.
 *   insmos_regroup_apply_voxels: the voxeliser's level-1 arrays under that renaming (num_points copied to its new rows,
 *   uperm and pc_voxel_id renamed in place; -1 entries stay). */
size_t insmos_regroup_ws_bytes(int64_t n);   /* workspace of either form */
int insmos_regroup_rows3d(const int32_t* coords, int64_t n, const uint64_t* bits, const int32_t* shape_host, int block_rows,
                          int32_t* new_coords, int32_t* new_of_old, int32_t* old_of_new /* optional */, void* ws, size_t ws_bytes,
                          void* stream);
/* ... over whole windows: one stable sort of (window, signature), rows of equal signature in their old order */
int insmos_regroup_rows3d_global(const int32_t* coords, int64_t n, const uint64_t* bits, const int32_t* shape_host,
                                 int32_t* new_coords, int32_t* new_of_old, int32_t* old_of_new /* optional */, void* ws,
                                 size_t ws_bytes, void* stream);
int insmos_regroup_apply_voxels(const int32_t* new_of_old, int64_t n_rows, const int32_t* num_points_old, int32_t* num_points_new,
                                int32_t* uperm, int64_t n_cells, int64_t* pc_voxel_id, int64_t n_points, void* stream);
int insmos_dense_nbr2d_b(int H, int W, int B, int32_t* nbr, void* stream);            /* B images stacked along the rows */
int insmos_sparse_to_bev_b(const float* feat, int ld_feat, int C, const int32_t* coords, int64_t n, int D, int H, int W, int B,
                           float* bev, void* stream);                                    /* bev (B, H, W, C*D) */
/* head rows (B * H * W, ld_head); candidate arrays (B, pre_max, .); counts (B, 4): [b][0] kept, [b][1] above threshold */
int insmos_center_decode_select_b(const float* head, int ld_head, int ncls, int H, int W, int up, int B, float out_factor,
                                  float vx, float vy, float x0, float y0, float score_thresh, int pre_max, float* cand_boxes,
                                  float* cand_scores, int32_t* cand_labels, int32_t* cand_cell, int32_t* counts, void* ws,
                                  size_t ws_bytes, void* stream);
size_t insmos_nms_ws_bytes_b(int max_n, int B);
int insmos_nms_rotated_bev_b(const float* boxes, const int32_t* n_dev, int max_n, float thresh, int post_max, int B,
                             int32_t* keep, int32_t* counts, void* ws, size_t ws_bytes, void* stream);
int insmos_gather_preds_b(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels, const int32_t* keep,
                          const int32_t* n_keep_dev, int pre_max, int post_max, int B, float* pred_boxes, float* pred_scores,
                          int64_t* pred_labels, void* stream);
size_t insmos_boxes_to_onehot_scratch_ints_b(int max_boxes, int B, int64_t n);
int insmos_boxes_to_onehot_b(const float* pred_boxes, const int64_t* pred_labels, const int32_t* n_boxes_dev, int max_boxes,
                             int B, const float* range_lo_host, const float* vsize_host, float stride, float mult,
                             const int32_t* coords, int64_t n, int ncls, int pad_to, int quirk_exact, float* out, int ld_out,
                             int32_t* scratch, void* stream);
/* ... over rows re-ordered by insmos_regroup_rows3d: orig_of_row / row_of_orig (both or neither; null = reference order) keep the
 * order-dependent part of Array_Index.cpp:40-56 -- the FIRST voxel inside a box -- in the reference's row order. */
int insmos_boxes_to_onehot_rows(const float* pred_boxes, const int64_t* pred_labels, const int32_t* n_boxes_dev, int max_boxes,
                                int B, const float* range_lo_host, const float* vsize_host, float stride, float mult,
                                const int32_t* coords, const int32_t* orig_of_row, const int32_t* row_of_orig, int64_t n, int ncls,
                                int pad_to, int quirk_exact, float* out, int ld_out, int32_t* scratch, void* stream);
int insmos_tslice_starts_batched(const uint64_t* keys, int64_t n, int max_d, int B, int32_t* starts, void* stream);
/* Dense BEV backbone convolution (base_bev_backbone.py:33-61: ZeroPad2d(1)+Conv2d(3x3) / Conv2d(3x3, padding 1), each with
 * BatchNorm2d + ReLU) as an LDS-tiled implicit GEMM: x (B, H, W, cin) NHWC with row pitch ld_x -> out (B, H, W, cout) with row
 * pitch ld_out; 3x3, stride 1, zero padding 1, + bias (folded BN) + optional ReLU.  wpacked / bias as
 * insmos_pack_weights_host(taps (9, cin, cout)), tap = ky*3 + kx.  cin a multiple of 16, cout 64 or 128 (else EINVAL: use
 * insmos_sparse_conv over insmos_dense_nbr2d).  Same result as that table path up to fp32 summation order. */
int insmos_bev_conv3x3(const float* x, int B, int H, int W, int ld_x, int cin, const float* wpacked, const float* bias,
                       float* out, int ld_out, int cout, int relu, void* stream);
/* The same layer with CONSTANT-REGION SKIPPING (round 4; csrc/bev.hip): the empty part of a BEV map (86 % of the sites of the S0
 * window) stays one constant vector per layer through the 3x3 stack of base_bev_backbone.py:33-61 -- layer 0 maps an all-zero
 * neighbourhood to relu(bias), layer l a neighbourhood that is c_{l-1} everywhere to a fixed c_l -- so 16-site row groups that
 * hold only such sites skip their matrix work and store c_l.  dist = insmos_bev_distance_map of the scattered voxels, layer =
 * position in the stack (0 reads the scattered map: its padding value IS the constant; later layers treat sites within layer - 1
 * of the image border as non-constant), cvec = insmos_bev_constant of this layer (cout floats).  Output bits are those of
 * insmos_bev_conv3x3 (tests/test_gpu_conv.py). */
int insmos_bev_conv3x3_skip(const float* x, int B, int H, int W, int ld_x, int cin, const float* wpacked, const float* bias,
                            float* out, int ld_out, int cout, int relu, const uint8_t* dist, int layer, const float* cvec,
                            void* stream);
/* insmos_bev_conv3x3_skip over COMPACTED row groups (round 4): the layer's non-constant 16-site row groups are listed per image
 * (ascending; ws: insmos_bev_skip_ws_bytes) and the workgroups walk the list four groups at a time, wherever they lie, so every
 * wave of every workgroup has matrix work; the other groups get the constant.  Same arguments, same output bits. */
size_t insmos_bev_skip_ws_bytes(int B, int H, int W);
int insmos_bev_conv3x3_skip_ws(const float* x, int B, int H, int W, int ld_x, int cin, const float* wpacked, const float* bias,
                               float* out, int ld_out, int cout, int relu, const uint8_t* dist, int layer, const float* cvec,
                               void* ws, size_t ws_bytes, void* stream);
/* Launch shape knob of the three entry points above (process-wide; the output bits do not depend on it, tests/test_gpu_conv.py): a
 * 128-channel layer whose launch would have fewer than max_wgs workgroups -- one to three windows of base_bev_backbone.py's
 * 150 x 125 map -- runs as two 64-channel workgroups per patch.  -1 = default (environment variable INSMOS_BEV_COSPLIT, else
 * 1024), 0 = never. */
int insmos_bev_cosplit(int max_wgs);
size_t insmos_bev_distance_map_ws_bytes(int B, int H, int W);
/* coords (n, 4) int32 [b, z, y, x] (spconv indices of the voxels HeightCompression scatters, height_compression.py:24-31) ->
 * dist (B * H * W bytes): Chebyshev distance of every site to the nearest occupied site of its image, capped at cap + 1. */
int insmos_bev_distance_map(const int32_t* coords, int64_t n, int B, int H, int W, int cap, uint8_t* dist, void* ws, size_t ws_bytes,
                            void* stream);
/* Accounting (bench.py's executed-flop count): the (site, tap) pairs insmos_bev_conv3x3_skip computes for `layer`; *pairs_dev =
 * 8 bytes of device memory. */
int insmos_bev_skip_executed_pairs(const uint8_t* dist, int B, int H, int W, int layer, unsigned long long* pairs_dev, void* stream);
size_t insmos_bev_constant_ws_floats(int cin, int cout);
/* c_out (cout floats) = what the layer produces at a site whose whole 3x3 neighbourhood is c_in (cin floats; null = zeros),
 * evaluated by the product kernel itself (same bits); ws: insmos_bev_constant_ws_floats floats, 16-byte aligned. */
int insmos_bev_constant(const float* wpacked, const float* bias, int cin, int cout, int relu, const float* c_in, float* c_out,
                        float* ws, void* stream);
/* Fused BEV deblock + heads (base_bev_backbone.py:104-115, center_head.py:65-72): x (n_site, cin) NHWC BEV features;
 * wd_packed / bd = the ConvTranspose2d(k=2,s=2)+BN as a 1x1 layer with 4*cup outputs laid out [ky][kx][co] (cup = 256);
 * wh_packed / bh = the merged 1x1 heads (cup -> head_cout <= 16).  head ((4*n_site), ld_head): row site*4 + ky*2 + kx.
 * The (4*n_site, cup) deconv output is never written: it is consumed from the accumulators.  Bit-identical to
 * insmos_sparse_conv(deconv, relu) followed by insmos_sparse_conv(head). */
int insmos_deconv_head(const float* x, int64_t n_site, int ld_x, int cin, const float* wd_packed, const float* bd, int cup,
                       const float* wh_packed, const float* bh, int head_cout, float* head, int ld_head, void* stream);
/* ... with the constant-region skipping of the 3x3 stack carried one step further (round 4): behind n_stack skipping-capable 3x3
 * layers (base_bev_backbone.py:33-61) a site further than n_stack from every occupied site (dist, insmos_bev_distance_map with cap >=
 * n_stack) and further than n_stack - 2 from the image border carries the stack's constant, so its deblock + head result is ONE
 * vector per sub-site: chead (4 x 16 floats, insmos_deconv_head_constant of the stack's last constant).  16-site groups of such
 * sites store it instead of computing.  x = B images of H x W sites.  Output bits are insmos_deconv_head's (tests/test_gpu_conv.py). */
int insmos_deconv_head_skip(const float* x, int64_t n_site, int ld_x, int cin, const float* wd_packed, const float* bd, int cup,
                            const float* wh_packed, const float* bh, int head_cout, float* head, int ld_head, const uint8_t* dist,
                            int H, int W, int n_stack, const float* chead, void* stream);
size_t insmos_deconv_head_constant_ws_floats(int cin);
int insmos_deconv_head_constant(const float* wd_packed, const float* bd, int cin, int cup, const float* wh_packed, const float* bh,
                                int head_cout, const float* c_in, float* chead, float* ws, void* stream);
/* Accounting (bench.py's executed-flop count): the sites insmos_deconv_head_skip computes; *sites_dev = 8 bytes of device memory. */
int insmos_deconv_head_skip_active_sites(const uint8_t* dist, int64_t n_site, int H, int W, int n_stack, unsigned long long* sites_dev,
                                         void* stream);
/* ------------------------------------------------------------------------------------------------
 * Reduced-precision convolution modes -- opt-in, process-wide, NEVER the default.  The inference path of this library is
 * exact fp32 (v_mfma_f32_16x16x4_f32) and every parity claim and benchmark line is made in mode 0.
 *   mode 0: exact fp32.
 *   mode 1: operands rounded to bf16, fp32 accumulate (v_mfma_f32_16x16x16_bf16) on the layers whose Cin is a multiple of
 *           16 and that have a neighbour table -- the opt-in for the TRAINING convolutions (no reference counterpart: the
 *           reference trains in fp32; config/config.yaml has no precision switch).
 *   mode 3: the split-bf16 x 3 experiment (x = hi + lo, three bf16 MFMAs per chunk; ~2^-16 relative error per product) on
 *           the layers whose split weights were registered; all others stay fp32.
 * insmos_split_weights_bf16: out (n_floats * 4 bytes) = the packed fp32 weights with every lane's 4 floats replaced by
 * (hi4 | lo4) bf16; insmos_register_split_weights(wpacked, wsplit): table entry (wsplit = NULL removes it). */
int insmos_conv_precision(int mode);
/* The same switch for the CALLING HOST THREAD only (mode -1 removes the override; while set it wins over the process-wide mode):
 * what the training convolutions use around their own launches, so that an inference forward issued from another host thread at
 * the same time stays exact fp32. */
int insmos_conv_precision_thread(int mode);
int insmos_split_weights_bf16(const float* wpacked, int64_t n_floats, void* out, void* stream);
int insmos_register_split_weights(const float* wpacked, const void* wsplit);
int insmos_debug_conv_force(int cot, int jt, int ring);
/* test hook: single-chunk layers (Cin 8 / 16, unsplit) on the quad-index kernel (1, the default) or on the generic one (0);
 * both produce the same bits (tests/test_gpu_conv.py). */
int insmos_debug_conv_quad(int on);
/* test hook: row-group thresholds below which the chunk-split tiles of a wide layer run at HALF width (two blocks per tile, half of
 * the channel tiles each) -- `wide` for Cout >= 128, `c64` for Cout 64; -1 = the environment / the defaults (4096, 1536), 0 = never.
 * Every output channel keeps its summation chain: both widths produce the same bits (tests/test_gpu_conv.py). */
int insmos_debug_conv_split_half(int wide, int c64);
/* test hook: the Cin = 32 layers whose input rows are whole 128-byte lines on the whole-row gather kernel (csrc/spconv_row32.hip: 1, the
 * default: where it pays -- the LDS-staged form for Cout <= 16 with K >= 16, the half-swizzled form for the 27-tap Cout 32 layers; 2 / 3: the
 * staged / the half-swizzled form on every shape it is built for; -1 = INSMOS_CONV_ROW32) or on the generic tiles (0); all produce the same
 * bits (tests/test_gpu_conv.py). */
int insmos_debug_conv_row32(int on);
/* test / tuning hook: the chunk-split layers (Cin 64 / 128 / 256 -> Cout 64 / 128: spconv_unet.py:146-160, 181-200) on the staged
 * 32-row kernel (csrc/spconv_wide.hip: 1 = on, the default; also INSMOS_CONV_WIDE) or on the chunk-split 16-row tiles (0); -1 = back to
 * the environment / default.  Same bits (tests/test_gpu_conv.py). */
int insmos_debug_conv_wide(int on);
/* The small-channel layers (Cin, Cout in {8, 16}: MotionNet's 81-tap BasicBlocks at 8 / 16 channels, minkunet.py:55-69,
 * resnet.py:110-119, and the k2s2 maps between them) on the row-per-lane VALU kernel (csrc/spconv_rowlane.hip): mode bit 0 =
 * 8 x 8 layers with K >= 16, bit 1 = K < 16 (Cin x Cout <= 128), bit 2 = 8 x 16 / 16 x 8 with K >= 16, bit 3 = 16 x 16; 0 = off
 * (MFMA tiles), -1 = default (INSMOS_CONV_ROWLANE, else 3).  rows_per_lane 1 or 2 (0 = INSMOS_CONV_ROWLANE_RPL, else 1).  Same
 * bits as the MFMA kernels whatever the mode (tests/test_gpu_conv.py).  mode | dbg << 4 (dbg > 0) selects a probe build of
 * the kernel (tools/batch_layers.py; results are wrong by construction). */
int insmos_debug_conv_rowlane(int mode, int rows_per_lane);
/* test / tuning hook: the 81-tap single-chunk layers (Cin 8 / 16, contiguous rows, masked table) on the LDS-staged kernel (1;
 * csrc/spconv_lds.hip; also INSMOS_CONV_LDS=1) or on the generic kernels (0, the default: the staged kernel is bit-identical but
 * slower in its first form); same bits as the unsplit generic kernels (tests/test_gpu_conv.py). */
int insmos_debug_conv_lds(int on);
/* probe build of that kernel (insmos_debug_conv_lds(2) / INSMOS_CONV_LDS=2): per-phase cycle sums and counters, 16 u64 (host array);
 * [0..5] wave 0's cycles in phases A, barrier, C, D, barrier, E; [8] groups, [9] raw groups, [10] overflow rows, [11] taps, [12] workgroups */
int insmos_debug_conv_lds_stats(unsigned long long* out16_host, int reset);
/* test / tuning hook: the d/dW kernel of insmos_sparse_conv_backward_weight -- 2 = row-compacting MFMA kernel (default),
 * 1 = first MFMA design (also INSMOS_DW_MFMA=1), 0 = LDS slabs.  All three are deterministic; they differ in summation order.
 * insmos_sparse_conv_backward_weight_ws_floats follows the mode: size the workspace after switching. */
int insmos_debug_dw_kernel(int mode);

/* ------------------------------------------------------------------------------------------------
 * insmos_dense_nbr2d -- full-grid 3x3 (pad 1) neighbour table for an H x W NHWC map, so that the
 * BEV Conv2d layers (base_bev_backbone.py:33-47; ZeroPad2d(1)+pad 0 == pad 1) run on insmos_sparse_conv.
 *   nbr (9, H*W) i32, tap = ky*3 + kx reads (y+ky-1, x+kx-1).
 * ---------------------------------------------------------------------------------------------- */
int insmos_dense_nbr2d(int H, int W, int32_t* nbr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_sparse_to_bev -- replaces SparseConvTensor.dense() + HeightCompression view
 * (models/backbones_2d/height_compression.py:24-31) in NHWC: bev[(y*W+x), c*D+d] = feat[i,c] for
 * coords[i] = [0,d,y,x]; bev (H*W, C*D) is zero-filled first.
 * ---------------------------------------------------------------------------------------------- */
int insmos_sparse_to_bev(const float* feat, int ld_feat, int C, const int32_t* coords, int64_t n, int D, int H, int W,
                         float* bev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_center_decode_select -- CenterHead.generate_predicted_boxes (center_head.py:251-276) +
 * the class-agnostic candidate selection of post_processing (models/post_process.py:182-192, :5-15):
 * sigmoid, max over classes (first max wins), score >= thresh, descending-score order (ties: ascending
 * cell index), at most pre_max candidates.
 *   head (n_cells, ld_head) fp32 rows [cls(ncls) | box(8)] in the deconv's [y][x][ky][kx] sub-site
 *   order: row r -> map row 2*(r/4/W0)+((r%4)/2), col 2*((r/4)%W0)+(r%2), W0 = half-resolution width.
 *   (up == 1: rows are already [row][col] over H x W.)
 * Outputs: cand_boxes (pre_max,7), cand_scores (pre_max), cand_labels (pre_max) i32 in {1..ncls},
 *          cand_cell (pre_max) i32 canonical cell index row*W+col; counts[0] = #candidates.
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_center_decode_select_ws_bytes(int64_t n_cells);
int insmos_center_decode_select(const float* head, int ld_head, int ncls, int H, int W, int up, float out_factor,
                                float vx, float vy, float x0, float y0, float score_thresh, int pre_max,
                                float* cand_boxes, float* cand_scores, int32_t* cand_labels, int32_t* cand_cell,
                                int32_t* counts, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_nms_rotated_bev -- replaces iou3d_nms_cuda.nms_gpu (models/bbox_post_process/src/
 * iou3d_nms.cpp:90-136 + nms_kernel iou3d_nms_kernel.cu:267-311): 64x64-block bitmask of
 * rotated-BEV IoU > thresh, then the greedy reduce ON DEVICE (no D2H of the mask).
 *   boxes (n_dev[0],7) sorted by descending score, n_dev: device count (<= max_n).
 *   keep (post_max) i32 ascending indices; counts[0] = #kept (<= post_max).
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_nms_ws_bytes(int max_n);
int insmos_nms_rotated_bev(const float* boxes, const int32_t* n_dev, int max_n, float thresh, int post_max,
                           int32_t* keep, int32_t* counts, void* ws, size_t ws_bytes, void* stream);
/* pairwise rotated BEV IoU (a: na x 7, b: nb x 7) -> out (na, nb); parity probe for the predicate. */
int insmos_iou_bev(const float* a, int na, const float* b, int nb, float* out, void* stream);
/* pairwise 3D IoU (iou3d_nms_utils.boxes_iou3d_gpu, iou3d_nms_utils.py:28-61: BEV overlap x height overlap over the
 * union volume) -- the eval-only kernel behind generate_recall_record (post_process.py:67-110). */
int insmos_iou3d(const float* a, int na, const float* b, int nb, float* out, void* stream);

/* Gather the kept candidates into the final prediction arrays (post_process.py:204-216):
 * pred_boxes (post_max,7) fp32, pred_scores (post_max) fp32, pred_labels (post_max) i64. */
int insmos_gather_preds(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                        const int32_t* keep, const int32_t* n_keep_dev, int post_max, float* pred_boxes,
                        float* pred_scores, int64_t* pred_labels, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_boxes_to_onehot -- replaces Array_Index.find_features_by_bbox_with_yaw
 * (models/utils/src/Array_Index.cpp:14-79) and the box rescaling at its 4 call sites
 * (spconv_unet.py:322-331,358,373,388), without the 4 host round trips.
 *   pred_boxes (m,7) metric boxes, pred_labels (m) i64, n_boxes_dev device count.
 *   Box in voxel units of this level: centre = (c - range_lo) * (1/vsize) * (1/stride) * mult, size
 *   likewise (fp32, same operation order as the reference's torch ops; mult = 1,2,4,8).
 *   coords (n,4) [b,z,y,x] int voxel indices of the level; the test uses the integer index (no +0.5).
 *   quirk_exact != 0 reproduces the order-dependent early-skip of Array_Index.cpp:48-51.
 *   onehot: fp32 written into out[i*ld_out + c], c < ncls (+ zero pad up to pad_to columns).
 *   scratch: insmos_boxes_to_onehot_scratch_ints(max_boxes, n) i32 device scratch (per-box first-hit voxel, box in voxel
 *   units, per-voxel class bits, per-voxel inside masks of each 64-box chunk).
 * ---------------------------------------------------------------------------------------------- */
size_t insmos_boxes_to_onehot_scratch_ints(int max_boxes, int64_t n);
int insmos_boxes_to_onehot(const float* pred_boxes, const int64_t* pred_labels, const int32_t* n_boxes_dev,
                           int max_boxes, const float* range_lo_host, const float* vsize_host, float stride,
                           float mult, const int32_t* coords, int64_t n, int ncls, int pad_to, int quirk_exact,
                           float* out, int ld_out, int32_t* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row gathers.
 * insmos_gather_rows: out[i,0:c] = idx[i] >= 0 ? src[idx[i],0:c] : 0   (idx i64)  -- replaces
 *   spconv gather_features_by_pc_voxel_id (spconv_unet.py:410).
 * insmos_build_current_points: cur[j] = [x,y,z,r | motion[inverse[p],0:3]] for p = cur_index[j]
 *   -- replaces predicted_sparse_tensor.slice(tensor_field) + the hstack of motionnet.py:38-48.
 * ---------------------------------------------------------------------------------------------- */
int insmos_gather_rows(const float* src, int ld_src, int c, const int64_t* idx, int64_t n, float* out, int ld_out,
                       void* stream);
int insmos_build_current_points(const float* points, int ld_pts, const float* motion, int ld_motion,
                                const int32_t* inverse, const int32_t* cur_index, int64_t n_cur, float* cur,
                                int ld_cur, void* stream);
/* fill rows: dst[i*ld + c0 .. c0+c) = value  (constant input features 0.5, motionnet.py:29-32) */
int insmos_fill_cols(float* dst, int64_t n, int ld, int c0, int c, float value, void* stream);
/* dst[i][0:c] = src[i][0:c] for n rows (column-slice copy: the stride-1 instance one-hots feed two layers,
 * spconv_unet.py:388,401) */
int insmos_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int64_t n, int c, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Caller-side stages ("next" rows of the scope table; scripts/predict_mos.py).
 * insmos_stack_scan: pose-align one raw scan (n,4) [x,y,z,intensity] into the current frame with the float64 3x4
 *   transform T_host (row-major, first 12 entries of inv(to_pose) @ from_pose), append timestamp t, write rows
 *   [x',y',z',intensity,t] at `out` (predict_mos.py:131-166).
 * insmos_output_stage: ignored classes -> -inf, softmax, confidence (n, ncls-1) = softmax[:,1:], argmax (first max),
 *   labels (n) = lut[argmax] (learning_map_inv), predict_mos.py:440-453.
 * ---------------------------------------------------------------------------------------------- */
int insmos_stack_scan(const float* scan, int64_t n, const double* T_host, float t, float* out, int ld_out, void* stream);
int insmos_output_stage(const float* logits, int ld, int64_t n, int ncls, unsigned ignore_mask, const int32_t* lut,
                        int32_t* labels, float* confidence, void* stream);

/* ------------------------------------------------------------------------------------------------
 * insmos_confusion3 -- ClassificationMetrics.compute_confusion_matrix (models/metrics.py:16-30):
 * ignored class columns of the logits are treated as -inf, argmax (first max), cm[pred, gt] += 1.
 *   cm (ncls*ncls) i64 is ACCUMULATED into (zero it once per evaluation).  ignore_mask bit c set =>
 *   class c ignored.  gt i64.
 * ---------------------------------------------------------------------------------------------- */
int insmos_confusion3(const float* logits, int ld, const int64_t* gt, int64_t n, int ncls, unsigned ignore_mask,
                      int64_t* cm, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Refine stage (scripts/refine.py:167-302), the step after the forward in the published pipeline.
 * insmos_points_in_instance_boxes -- Array_Index.find_point_in_instance_bbox_with_yaw (models/utils/src/
 *   Array_Index.cpp:85-154, called at refine.py:196): points (n, ld >= 3) fp32 xyz, boxes (m, 7) fp32, labels (m) i64,
 *   all device; index (n, ncls) i32 out: index[j][c-1] = 1-based number of the class-c box containing point j (0 =
 *   none).  Box z is lifted by ground_offset; quirk_exact = 1 reproduces the order-dependent early skip (:124-127).
 *   Where two boxes of one class share a point the reference's OpenMP loop races; here the larger box number wins
 *   (= the sequential walk).  scratch: 20*m i32.
 * insmos_instance_stats -- per instance of class column `col`: {points, points labelled 2 (moving), points with
 *   confidence[:,1] >= 1e-5} (refine.py:210-217); mos (n) i32 in {0,1,2}; conf (n,2) fp32 or null; stats (m,3) i32 out.
 * insmos_instance_relabel -- mos[j] = decision[id-1] for points of instance id with decision > 0 (refine.py:243-285).
 * ---------------------------------------------------------------------------------------------- */
int insmos_points_in_instance_boxes(const float* points, int64_t n, int ld_pts, const float* boxes,
                                    const int64_t* labels, int m, float ground_offset, int ncls, int quirk_exact,
                                    int32_t* index, int32_t* scratch, void* stream);
int insmos_instance_stats(const int32_t* index, int ncls, int col, const int32_t* mos, const float* conf, int64_t n,
                          int m, int32_t* stats, void* stream);
int insmos_instance_relabel(const int32_t* index, int ncls, int col, const int32_t* decision, int64_t n, int m,
                            int32_t* mos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-step pieces (models/models.py:61-98 training_step, models/loss.py:20-34) -- the
 * gradients behind `loss.backward()` for the sparse convolutions and the MOS loss.  A FIRST SLICE: BatchNorm in
 * training mode and the CenterHead targets / losses follow below; the optimiser stays with the caller (torch.optim).
 *   insmos_pack_weights_device: insmos_pack_weights_host on the device, from taps (K, cin_real, cout_real) that change
 *     every step; transpose = 1 packs W[k]^T (a layer mapping cout_real-wide rows to cin_real-wide ones: pass
 *     cin = padded cout_real, cout = cin_real) and mirror_taps = 1 reads tap K-1-k -- together the d/dx layer of a
 *     submanifold convolution on its own table; strided layers pass their transposed table (down <-> inverse) instead.
 *   insmos_sparse_conv_backward_weight: dW[k] = sum_o x[nbr[k][o]]^T (x) dy[o]  (K, cin, cout) fp32, deterministic
 *     (slab partial sums + fixed-order reduction; ws from *_ws_floats); accumulate != 0 adds to dW.
 *   insmos_col_sum: out[c] (+)= sum_rows a[row][c] (bias gradient), fixed order.
 *   insmos_mos_loss: MOSLoss.compute_loss -- ignored classes -> -inf, softmax, log(clamp(., 1e-8)), weighted NLL:
 *     loss_sums[0] = sum_i w[gt_i] * -log p_i, loss_sums[1] = sum_i w[gt_i] (loss = [0] / [1], device floats);
 *     grad (n, ncls) = d loss / d logits (already divided by loss_sums[1]) or null.
 * ---------------------------------------------------------------------------------------------- */
int insmos_pack_weights_device(const float* taps, int K, int cin_real, int cout_real, int cin, int cout, int transpose,
                               int mirror_taps, float* packed, void* stream);
size_t insmos_sparse_conv_backward_weight_ws_floats(int64_t n_out, int K, int cin, int cout);
int insmos_sparse_conv_backward_weight(const float* x, int64_t n_in, int ld_x, int cin, const float* dy, int ld_dy, int cout,
                                       const int32_t* nbr, int K, int64_t n_out, float* dw, int accumulate, float* ws,
                                       void* stream);
size_t insmos_col_sum_ws_floats(int64_t n, int c);
int insmos_col_sum(const float* a, int ld, int c, int64_t n, float* out, int accumulate, float* ws, void* stream);
/* BatchNorm1d over rows in TRAINING mode (the nn.BatchNorm1d / MinkowskiBatchNorm layers of spconv_unet.py and
 * minkunet.py under model.train()): batch mean / biased variance per channel, y = relu?(xhat * gamma + beta);
 * stats = [mean | invstd | biased var] (3c floats); xhat (n, c) is kept for the backward, which returns dx, dgamma,
 * dbeta (through the ReLU when relu != 0).  Running-statistics updates stay with the caller (two axpys). */
/* Segmented, fused BatchNorm in training mode (round 3: B windows per training step with the reference's per-item statistics --
 * models/models.py:313 walks the batch item by item, so a BatchNorm layer sees one window's rows at a time).  chunks: n_chunks x 4
 * int32 (row_start, row_end, segment, 0), <= 1024 rows of ONE segment each, sorted by segment; seg_first (S + 1): chunk ranges;
 * seg_rows (S): rows per segment; stats: S x 3c ([mean | invstd | biased var] per segment).  All device arrays.  S = 1 is the plain
 * layer.  ticket: one zero int32 on the device (the SCALAR statistics kernels -- widths the 16-byte kernels do not take -- fold their
 * partials in the block that finishes last; it is left zero).  The 16-byte kernels (c / 4 a power of two <= 64: every layer of the two
 * networks) run forward = 3 launches: statistics partials per chunk quarter, their merge (+ running statistics, item after item),
 * apply; backward = 3: both sums' partials, merge, dx (round 6: the last-block ticket cost ~0.1 us per block, serialised, and the
 * merge ran in one block: 58 -> 28 us for a 1.57 M x 8 layer's statistics, 430 -> 128 us for the 300 k x 256 deblock layer). */
size_t insmos_batchnorm_seg_ws_floats(int n_chunks, int c, int S);
int insmos_batchnorm_seg_forward(const float* x, int ld_x, int c, int64_t n, const int32_t* chunks, int n_chunks,
                                 const int32_t* seg_first, const int32_t* seg_rows, int S, const float* gamma, const float* beta,
                                 float eps, int relu, float* y, int ld_y, float* xhat, float* stats, float* running_mean,
                                 float* running_var, float momentum, int32_t* ticket, float* ws, void* stream);
int insmos_batchnorm_seg_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* xhat, int c, int64_t n,
                                  const int32_t* chunks, int n_chunks, const int32_t* seg_first, const int32_t* seg_rows, int S,
                                  const float* gamma, const float* stats, int relu, float* dx, int ld_dx, float* dgamma,
                                  float* dbeta, int32_t* ticket, float* ws, void* stream);
/* The same layer WITHOUT the stored x^ (round 6): insmos_batchnorm_seg_forward accepts xhat = NULL when
 * insmos_batchnorm_seg_recompute_ok(c, ld_x, ld_dy, ld_dx) (the 16-byte kernels: c / 4 a power of two <= 64, pitches multiples of 4
 * floats, 16-byte aligned pointers); the caller keeps the layer's input x instead and calls insmos_batchnorm_seg_backward_x, which
 * recomputes x^ = (x - mean) * invstd and the ReLU mask (gamma * x^ + beta > 0) from it: the same gradients bit for bit, 9 instead
 * of 12 passes over the layer's elements per forward + backward. */
int insmos_batchnorm_seg_recompute_ok(int c, int ld_x, int ld_dy, int ld_dx);
int insmos_batchnorm_seg_backward_x(const float* dy, int ld_dy, const float* x, int ld_x, int c, int64_t n, const int32_t* chunks,
                                    int n_chunks, const int32_t* seg_first, const int32_t* seg_rows, int S, const float* gamma,
                                    const float* beta, const float* stats, int relu, float* dx, int ld_dx, float* dgamma,
                                    float* dbeta, int32_t* ticket, float* ws, void* stream);
size_t insmos_batchnorm_ws_floats(int64_t n, int c);
int insmos_batchnorm_train_forward(const float* x, int ld_x, int c, int64_t n, const float* gamma, const float* beta, float eps,
                                   int relu, float* y, int ld_y, float* xhat, float* stats, float* ws, void* stream);
int insmos_batchnorm_train_backward(const float* dy, int ld_dy, const float* y, int ld_y, const float* xhat, int c, int64_t n,
                                    const float* gamma, const float* stats, int relu, float* dx, int ld_dx, float* dgamma,
                                    float* dbeta, float* g_ws, float* ws, void* stream);
size_t insmos_mos_loss_ws_floats(int64_t n);
int insmos_mos_loss(const float* logits, int ld, const int64_t* gt, int64_t n, int ncls, unsigned ignore_mask,
                    const float* class_weights, float* loss_sums, float* grad, int ld_grad, float* ws, void* stream);
/* CenterHead training side (models/backbones_2d/center_head.py).
 *   insmos_center_assign_targets: get_targets_single (:170-249) for ONE batch item.  gt_boxes8 (n_gt, 8) fp32 device =
 *     [x, y, z, dx, dy, dz, yaw, label]; only the first max_objs rows are looked at (:201).  A row is used when
 *     dx / voxel_x / factor > 0, dy / voxel_y / factor > 0, trunc(label - 1) > -1 and the truncated centre cell lies in
 *     the (fm_h, fm_w) map (:211,:230-232).  radius = max(min_radius, int(gaussian_radius((length, width), overlap)))
 *     (:212-215, :395-424, fp32); the gaussian (:346-393) is max-merged into heatmap (num_class, fm_h, fm_w), which this
 *     call zeroes first.  anno_box (max_objs, 8) = [cx - x, cy - y, z, log dx, log dy, log dz, sin yaw, cos yaw],
 *     ind (max_objs) int64 = y * fm_w + x, mask (max_objs) uint8; unused slots are zero.  range_is_f64: the reference
 *     evaluates (x - range_x0) / voxel / factor in float32 when POINT_CLOUD_RANGE is an integer list (the shipped
 *     config, config/config.yaml:6) and in float64 when it holds floats (torch 0-dim promotion) -- pass which.
 *   insmos_center_head_loss: get_loss (:279-331) for one item.  cls_preds (hw, >= num_class) raw logits and box_preds
 *     (hw, >= 8), row = y * fm_w + x (the NHWC maps of :71-72).  losses[3] (device) = [cls_weight * focal,
 *     loc_weight * l1, their sum]; grad_cls / grad_box (or null) = d losses[2] / d cls_preds, d box_preds (all rows of
 *     grad_box's 8 columns are written).  code_weights_host: 8 floats in HOST memory.  Fixed-order reductions. */
int insmos_center_assign_targets(const float* gt_boxes8, int n_gt, int max_objs, int num_class, int fm_w, int fm_h,
                                 double range_x0, double range_y0, int range_is_f64, float voxel_x, float voxel_y,
                                 int out_size_factor, double gaussian_overlap, int min_radius, float* heatmap, float* anno_box,
                                 int64_t* ind, uint8_t* mask, void* stream);
size_t insmos_center_head_loss_ws_floats(int64_t hw, int num_class);
int insmos_center_head_loss(const float* cls_preds, int ld_cls, const float* box_preds, int ld_box, int64_t hw, int num_class,
                            const float* heatmap, const float* anno_box, const int64_t* ind, const uint8_t* mask, int max_objs,
                            float cls_weight, float loc_weight, const float* code_weights_host, float* losses, float* grad_cls,
                            int ld_gcls, float* grad_box, int ld_gbox, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Native window runner -- InsMOS_Model.forward(list, 'test') for ONE batch item (models/models.py:313-364) as one
 * foreign call: the same operator sequence insmos_amd/engine.py issues step by step (MotionNet -> voxelise ->
 * UNetV2 encoder -> BEV CenterHead -> NMS -> instance-fused decoder -> per-point logits), driven from C++ so that
 * windows can be in flight on several HIP streams from several host threads without interpreter work in between.
 *   insmos_ctx_create: `layers[i]` (device pointers to weights packed by insmos_pack_weights_host + padded bias) under
 *     `names[i]` = the engine's layer names ("block1.0.conv1", "conv_up_t4.conv2", "bev0", "head", ...); a missing
 *     layer makes insmos_forward_window return INSMOS_EINVAL.  The context is immutable after creation and may be
 *     shared by threads; each concurrent window needs its own arena and stream.
 *   insmos_forward_window: points (N, ld >= 5) fp32 device [x, y, z, intensity, t]; every intermediate and the
 *     outputs live in `arena` (device, bump-allocated); INSMOS_EWORKSPACE => out->arena_needed holds a size that is
 *     enough for this window's allocations so far (grow and retry).  Outputs (byte offsets into the arena, valid
 *     until the arena's next use): logits (n_cur, 3) fp32, boxes (n_boxes, 7) fp32, scores (n_boxes) fp32,
 *     labels (n_boxes) i64.  Synchronises the stream (the count read-backs size the next launches).
 * ---------------------------------------------------------------------------------------------- */
typedef struct InsmosConvW {
    const float* w; /* packed taps */
    const float* b; /* bias, padded to a multiple of 16 */
    int32_t K, cin, cout, reserved;
} InsmosConvW;
typedef struct InsmosNetCfg {
    const float* w0_const;   /* (125, 8): 0.5 * BN-folded conv0 taps (constant-input first layer, motionnet.py:29-32) */
    const float* b0_const;   /* (8) */
    const int32_t* nbr_bev;  /* insmos_dense_nbr2d(bevH, bevW) */
    float vs[3], dt, range[6], score_thresh, nms_thresh, out_factor, tvs[2];
    int32_t in_ch, ncls, max_voxels, max_points;
    int32_t shape[6][3]; /* spconv spatial shapes [z, y, x] of levels 1..5 (index 0 unused), spconv_unet.py:114 */
    int32_t bevD, bevH, bevW, nbev, n_bev_layers, up_ch, head_ld, pre_max, post_max, quirk_exact;
} InsmosNetCfg;
typedef struct InsmosForwardOut {
    int64_t me_voxels[4], n_cur, unet_voxels[5], n_candidates, n_boxes, n_out_of_window;
    int64_t logits_off, boxes_off, scores_off, labels_off, arena_needed;
    int64_t cur_points_off; /* current_point (n_cur, 8) fp32 = [x, y, z, r, m0, m1, m2, 0] (motionnet.py:42-48): the motion
                               features the 'eval' mode's motion loss is taken on (models/models.py:321-323) */
    int64_t batch;          /* windows that shared this launch set: me_voxels and unet_voxels[1..4] are totals of the batch */
} InsmosForwardOut;
int insmos_ctx_create(const InsmosNetCfg* cfg, const char* const* names, const InsmosConvW* layers, int n_layers,
                      void** ctx_out);
int insmos_ctx_destroy(void* ctx);
int insmos_forward_window(void* ctx, const float* points, int64_t n, int ld_pts, void* arena, size_t arena_bytes,
                          void* stream, InsmosForwardOut* out);
/* The whole batch list of InsMOS_Model.forward in one launch set: points_host[b] (n_points_host[b], ld_pts) device arrays,
 * outs[b] as above for window b (offsets into the shared arena).  B = 1 is insmos_forward_window.  INSMOS_EBATCH: the
 * batch's finest 81-tap table would pass 2 GiB (kernels address tables with 32-bit byte offsets) -- split the list. */
/* The native runner puts the work that is not on the convolution chain on a second stream of the calling host thread (created on
 * first use; ordered against the caller's stream by events; joined before the function returns): bit 0 = the level-0 81-tap
 * table, bit 1 = the 3D branch's coordinate sets and kernel maps (beside MotionNet's convolutions), bit 2 = inv_conv_out (beside
 * the BEV head), bit 3 = the finer levels' one-hot passes (beside the level-4 decoder).  Same bits either way.  mask -1 = default
 * (15); the environment variable INSMOS_TWO_STREAMS overrides both.  Single-window latency 3.8 -> 3.4 ms; with several launch
 * sets in flight the sets already overlap each other and the caller switches it off (insmos_amd/models.py). */
int insmos_forward_streams(int mask);
/* Releases the calling host thread's second stream and events (created on first use by insmos_forward_windows, on the device
 * current at that time; a thread that later runs on another device gets new ones automatically).  Worker threads call it
 * before they end. */
int insmos_forward_thread_release(void);
/* Host timeline of the calling thread's last insmos_forward_window(s) call: "stage:microseconds since the call began;..." -- when
 * the host finished enqueueing each section and when each count read-back returned (the reference's caller hands over ONE window
 * per call, scripts/predict_mos.py:290,434: that latency is a chain of these).  Recorded only with INSMOS_HOST_MARKS=1 in the
 * environment (empty string otherwise); tools/b1_host_marks.py prints it. */
int insmos_forward_host_marks(char* buf, size_t cap);
/* Row regrouping of the runner's 3D levels 1..4, one decimal digit per level (level 1 = units): 0 = off, 1 = blocks of 256 rows,
 * 2 = 1024, 3 = 4096 (insmos_regroup_rows3d), 4 = whole windows (insmos_regroup_rows3d_global), 5 = 4096-row blocks with the
 * coordinate parity class above the signature (block_rows -4096); -1 = default (environment
 * variable INSMOS_REGROUP_ROWS, else 3553 for launch sets of two or more windows and off for a single window, whose latency
 * the block sorts would cost more than the convolutions gain).  Process-wide.  The outputs do not depend on it. */
int insmos_forward_regroup(int modes);
int insmos_debug_table_limit(int64_t bytes); /* tests only: lower the table size at which a batch is refused (0 = default) */
int insmos_forward_windows(void* ctx, const float* const* points_host, const int64_t* n_points_host, int B, int ld_pts,
                           void* arena, size_t arena_bytes, void* stream, InsmosForwardOut* outs);

#ifdef __cplusplus
}
#endif
#endif
