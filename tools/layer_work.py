#!/usr/bin/env python3
"""tools/layer_work.py -- algorithmic work of every convolution of one window, from the ORACLE's kernel maps (CPU only):
pairs = valid (tap, output) entries, flops = 2 * pairs * Cin * Cout, compulsory bytes = 4 * (N_in * Cin + N_out * Cout) +
4 * K * N_out (table), gather bytes = 4 * pairs * (Cin + Cout) + 8 * pairs  (SURVEY.md 8d / DESIGN.md 3).  For the MotionNet
decoder it also counts the rows the HIP path actually computes (dead-row elimination).

    python tools/layer_work.py [n_az] > profiles/r01_layer_work_s0.csv      # default 1886 = S0 (about two minutes)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import params as P  # noqa: E402
from insmos_amd.synth import make_window  # noqa: E402
from oracle import ref_ops as R  # noqa: E402

n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 1886
cfg = P.default_cfg()
w = make_window(seed=0, n_scans=10, n_az=n_az)
rows = []


def add(branch, name, K, ci, co, n_in, n_out, pairs, pairs_exec=None, fill=1.0):
    pe = pairs if pairs_exec is None else pairs_exec
    rows.append((branch, name, K, ci, co, n_in, n_out, pairs, pe, 2 * pairs * ci * co, 2 * pe * ci * co,
                 4 * (n_in * ci + n_out * co) + 4 * K * n_out, 4 * pairs * (ci + co) + 8 * pairs, round(fill, 4)))


def group_fill(nbr, keep=None):
    """Useful fraction of the MFMA columns the kernel issues: valid entries / (16 x active (16-row group, tap) slots) --
    the kernel skips a tap for a whole 16-row group only; `keep` = rows actually computed."""
    v = nbr >= 0
    if keep is not None:
        first = int(np.argmax(keep)) // 16 * 16      # executed rows are a suffix, rounded down to a group boundary
        v = v[:, first:]
    K, nn = v.shape
    pad = (-nn) % 16
    if pad:
        v = np.concatenate([v, np.zeros((K, pad), bool)], 1)
    g = v.reshape(K, -1, 16)
    slots = int(g.any(2).sum())
    return float(v.sum()) / max(16 * slots, 1)


# ---------------- MotionNet (minkunet.py:139-181)
pts4 = np.concatenate([w[:, 0:3], w[:, 4:5]], 1).astype(np.float32)
coords, keys, inverse = R.me_quantize(pts4, np.array([0.1, 0.1, 0.1, 0.1], np.float32))
lv = [(coords, keys)]
for L in (1, 2, 3):
    pc, pk, _ = R.me_stride_down(coords, keys, L)
    lv.append((pc, pk))
n = [len(c) for c, _ in lv]


def valid_per_row(nbr):
    return (nbr >= 0).sum(0)


t81 = [R.me_nbr(c, k, R.me_kernel_offsets([3, 3, 3, 3], [1 << L, 1 << L, 1 << L, 1])) for L, (c, k) in enumerate(lv)]
v81 = [valid_per_row(t) for t in t81]
v125 = valid_per_row(R.me_nbr(coords, keys, R.me_kernel_offsets([5, 5, 5, 1], [1, 1, 1, 1])))
off2 = [R.me_kernel_offsets([2, 2, 2, 1], [1 << L, 1 << L, 1 << L, 1]) for L in range(3)]
f_dn = [group_fill(R.me_nbr(lv[L + 1][0], lv[L][1], off2[L], +1)) for L in range(3)]
t_up = [R.me_nbr(lv[L][0], lv[L + 1][1], off2[L], -1) for L in range(3)]
t_last = [int(c[:, 3].max()) for c, _ in lv]


def live(L, depth):
    """rows of level L within `depth` scans of the newest one (rows are time-major)"""
    return lv[L][0][:, 3] >= t_last[L] - depth


ALL = 99
add("motionnet", "conv0p1s1", 125, 1, 8, n[0], n[0], int(v125.sum()))
for name, ci, co, L, K, depth in (("conv1p1s2", 8, 8, 1, 8, ALL), ("conv2p2s2", 8, 8, 2, 8, ALL), ("conv3p4s2", 16, 16, 3, 8, ALL)):
    add("motionnet", name, K, ci, co, n[L - 1], n[L], n[L - 1], fill=f_dn[L - 1])  # k2s2: every fine voxel has one parent
blocks = (("block1.0", 8, 8, 1, ALL), ("block2.0", 8, 16, 2, ALL), ("block3.0", 16, 32, 3, 6), ("block6.0", 48, 32, 2, 4),
          ("block7.0", 24, 16, 1, 2), ("block8.0", 16, 8, 0, 0))
for name, ci, co, L, depth in blocks:
    m2, m1 = live(L, depth), live(L, depth + 1)
    add("motionnet", name + ".conv1", 81, ci, co, n[L], n[L], int(v81[L].sum()), int(v81[L][m1].sum()), group_fill(t81[L], m1))
    add("motionnet", name + ".conv2", 81, co, co, n[L], n[L], int(v81[L].sum()), int(v81[L][m2].sum()), group_fill(t81[L], m2))
    if ci != co:
        add("motionnet", name + ".downsample", 1, ci, co, n[L], n[L], n[L], int(m2.sum()))
for name, ci, co, L, depth in (("convtr5p8s2", 32, 32, 2, 6), ("convtr6p4s2", 32, 16, 1, 4), ("convtr7p2s2", 16, 8, 0, 2)):
    add("motionnet", name, 8, ci, co, n[L + 1], n[L], n[L], int(live(L, depth).sum()), group_fill(t_up[L], live(L, depth)))
add("motionnet", "final", 1, 8, 3, n[0], n[0], n[0], int(live(0, 0).sum()))

# ---------------- UNetV2 (spconv_unet.py:267-416); the kernel maps depend on the current points' xyz only
cur = np.zeros((int((pts4[:, 3] / np.float32(0.1) == 0).sum()), 7), np.float32)
cur[:, :4] = w[(pts4[:, 3] / np.float32(0.1)) == 0, :4]
vs, rng = cfg["DATA"]["VOXEL_SIZE"], cfg["DATA"]["POINT_CLOUD_RANGE"]
grid = np.round((np.array(rng[3:6], float) - np.array(rng[0:3], float)) / np.array(vs)).astype(np.int64)
shape1 = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
_, coords1, _, _ = R.voxelize_with_id(cur, vs, rng, 100000, 5)
k1, p1 = R.sorted_index(R.key3(coords1, shape1))
S = {1: (coords1, k1, p1, shape1)}
for l in (2, 3, 4):
    oc, ok, osz = R.spconv_down_coords(S[l - 1][0], S[l - 1][3], (3, 3, 3), (2, 2, 2), (1, 1, 1))
    S[l] = (oc, ok, None, osz)
c5, k5, s5 = R.spconv_down_coords(S[4][0], S[4][3], (3, 1, 1), (2, 1, 1), (0, 0, 0))
V = {l: len(S[l][0]) for l in S}
V[5] = len(c5)
T_subm = {l: R.spconv_nbr_subm(S[l][0], S[l][1], S[l][2], S[l][3]) for l in (1, 2, 3, 4)}
T_down = {l: R.spconv_nbr_down(S[l][0], S[l - 1][1], S[l - 1][2], S[l - 1][3], (3, 3, 3), (2, 2, 2), (1, 1, 1)) for l in (2, 3, 4)}
T_inv = {l: R.spconv_nbr_inverse(S[l - 1][0], S[l][1], S[l][2], S[l][3], (3, 3, 3), (2, 2, 2), (1, 1, 1)) for l in (2, 3, 4)}
T_down5 = R.spconv_nbr_down(c5, S[4][1], S[4][2], S[4][3], (3, 1, 1), (2, 1, 1), (0, 0, 0))
T_inv5 = R.spconv_nbr_inverse(S[4][0], k5, None, s5, (3, 1, 1), (2, 1, 1), (0, 0, 0))
p_subm = {l: int((T_subm[l] >= 0).sum()) for l in T_subm}
p_down = {l: int((T_down[l] >= 0).sum()) for l in T_down}
p_down5 = int((T_down5 >= 0).sum())
f_subm = {l: group_fill(T_subm[l]) for l in T_subm}
f_down = {l: group_fill(T_down[l]) for l in T_down}
f_inv = {l: group_fill(T_inv[l]) for l in T_inv}
C = {1: 16, 2: 32, 3: 64, 4: 128}
add("unet", "conv_input", 27, 7, 16, V[1], V[1], p_subm[1], fill=f_subm[1])
add("unet", "conv1", 27, 16, 16, V[1], V[1], p_subm[1], fill=f_subm[1])
for l in (2, 3, 4):
    add("unet", f"conv{l}.0 (spconv{l})", 27, C[l - 1], C[l], V[l - 1], V[l], p_down[l], fill=f_down[l])
    add("unet", f"conv{l}.1", 27, C[l], C[l], V[l], V[l], p_subm[l], fill=f_subm[l])
    add("unet", f"conv{l}.2", 27, C[l], C[l], V[l], V[l], p_subm[l], fill=f_subm[l])
add("unet", "conv_out (spconv_down2)", 3, 128, 128, V[4], V[5], p_down5, fill=group_fill(T_down5))
H, W = int(grid[1]) // 8, int(grid[0]) // 8
dense_pairs = sum((H - abs(ky - 1)) * (W - abs(kx - 1)) for ky in range(3) for kx in range(3))
add("bev", "blocks.0.1", 9, 256, 128, H * W, H * W, dense_pairs)
for k in range(5):
    add("bev", f"blocks.0.{4 + 3 * k}", 9, 128, 128, H * W, H * W, dense_pairs)
add("bev", "deblocks.0.0 (convT 2x2)", 4, 128, 256, H * W, 4 * H * W, 4 * H * W)
add("bev", "center_head (cls + box)", 1, 256, 11, 4 * H * W, 4 * H * W, 4 * H * W)
add("unet", "inv_conv_out", 3, 128, 128, V[5], V[4], p_down5, fill=group_fill(T_inv5))
ncls = 3
for l, Cc in ((4, 128), (3, 64), (2, 32), (1, 16)):
    add("unet", f"conv_up_instance_block (level {l})", 27, Cc + ncls, Cc, V[l], V[l], p_subm[l], fill=f_subm[l])
    add("unet", f"conv_up_t{l}.conv1", 27, Cc, Cc, V[l], V[l], p_subm[l], fill=f_subm[l])
    add("unet", f"conv_up_t{l}.conv2", 27, Cc, Cc, V[l], V[l], p_subm[l], fill=f_subm[l])
    add("unet", f"conv_up_m{l}", 27, 2 * Cc, Cc, V[l], V[l], p_subm[l], fill=f_subm[l])
    if l > 1:
        add("unet", f"inv_conv{l}", 27, Cc, Cc // 2, V[l], V[l - 1], p_down[l], fill=f_inv[l])
add("unet", "conv_up_out", 27, 16, 16, V[1], V[1], p_subm[1], fill=f_subm[1])
add("unet", "conv_up_instance_block_up1", 27, 16 + ncls, 16, V[1], V[1], p_subm[1], fill=f_subm[1])
add("unet", "mos_seg_layer", 1, 16, 3, V[1], V[1], V[1])

print("branch,layer,K,cin,cout,n_in,n_out,pairs,pairs_executed,flops,flops_executed,compulsory_bytes,gather_bytes,group_fill")
for r in rows:
    print(",".join(str(v) for v in r))
tot = lambda i: sum(r[i] for r in rows)
print("# window: %d points, %d current; 4D voxels %s; 3D voxels %s" % (len(w), len(cur), n, [V[l] for l in (1, 2, 3, 4, 5)]), file=sys.stderr)
print("# total: %.2f GFLOP as the reference computes it, %.2f GFLOP executed after dead-row elimination; compulsory %.3f GB, "
      "no-reuse gather %.3f GB" % (tot(9) / 1e9, tot(10) / 1e9, tot(11) / 1e9, tot(12) / 1e9), file=sys.stderr)
issued = sum(r[10] / max(r[13], 1e-9) for r in rows if r[1] != "conv0p1s1")
print("# MFMA work issued for absent rows of active 16-row groups: useful fraction %.3f (flop-weighted, channel padding not "
      "counted)" % (sum(r[10] for r in rows if r[1] != "conv0p1s1") / issued), file=sys.stderr)
for b in ("motionnet", "unet", "bev"):
    print("#   %-9s %.2f GFLOP (%.2f executed)" % (b, sum(r[9] for r in rows if r[0] == b) / 1e9,
                                                  sum(r[10] for r in rows if r[0] == b) / 1e9), file=sys.stderr)
