"""Semantic anchors for the [dep-knowledge] parts of the oracle (ME / spconv restatements), which the
reference itself cannot pin (no tests, dependencies absent):
  * sparse conv == dense torch conv3d / conv_transpose3d masked to the active sites
  * C helpers == their pure-numpy twins
  * known voxel / pair counts of the survey's synthetic scene S0
CPU-only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_ops as R


def _rand_sparse3d(rng, shape, n):
    c = np.unique(rng.integers(0, shape, size=(n, 3)), axis=0).astype(np.int32)
    return c[rng.permutation(len(c))]


def _dense(feat, coords, shape):
    d = np.zeros((feat.shape[1],) + tuple(shape), np.float32)
    d[:, coords[:, 0], coords[:, 1], coords[:, 2]] = feat.T
    return torch.from_numpy(d)[None]


def _taps_to_dense_w(taps, ks):
    K, ci, co = taps.shape
    return torch.from_numpy(taps.reshape(ks[0], ks[1], ks[2], ci, co).transpose(4, 3, 0, 1, 2).copy())


def test_subm_equals_dense_conv3d():
    rng = np.random.default_rng(0)
    shape = [9, 14, 16]
    c = _rand_sparse3d(rng, shape, 500)
    x = rng.normal(size=(len(c), 5)).astype(np.float32)
    taps = rng.normal(size=(27, 5, 7)).astype(np.float32)
    ks, perm = R.sorted_index(R.key3(c, shape))
    nbr = R.spconv_nbr_subm(c, ks, perm, shape)
    y = R.sparse_conv(x, nbr, taps)
    yd = F.conv3d(_dense(x, c, shape), _taps_to_dense_w(taps, (3, 3, 3)), padding=1)[0].numpy()
    np.testing.assert_allclose(y, yd[:, c[:, 0], c[:, 1], c[:, 2]].T, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(nbr, R.nbr_lookup_numpy(
        np.stack([R.key3(c.astype(np.int64) + k - 1, shape) for k in R.spconv_kernel_offsets((3, 3, 3))]), ks, perm))
    np.testing.assert_allclose(y, R.sparse_conv_numpy(x, nbr, taps), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ksize,stride,pad", [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))])
def test_strided_and_inverse_equal_dense(ksize, stride, pad):
    rng = np.random.default_rng(1)
    shape = [9, 14, 16]
    c = _rand_sparse3d(rng, shape, 400)
    x = rng.normal(size=(len(c), 4)).astype(np.float32)
    taps = rng.normal(size=(int(np.prod(ksize)), 4, 6)).astype(np.float32)
    ks, perm = R.sorted_index(R.key3(c, shape))
    oc, ok, oshape = R.spconv_down_coords(c, shape, ksize, stride, pad)
    nbr = R.spconv_nbr_down(oc, ks, perm, shape, ksize, stride, pad)
    y = R.sparse_conv(x, nbr, taps)
    yd = F.conv3d(_dense(x, c, shape), _taps_to_dense_w(taps, ksize), stride=stride, padding=pad)[0].numpy()
    assert list(yd.shape[1:]) == oshape
    # output set = every site whose receptive field holds an active input (ascending linear order)
    occ = F.conv3d(_dense(np.ones((len(c), 1), np.float32), c, shape), torch.ones(1, 1, *ksize), stride=stride,
                   padding=pad)[0, 0].numpy() > 0
    exp = np.argwhere(occ).astype(np.int32)
    np.testing.assert_array_equal(oc, exp)
    np.testing.assert_allclose(y, yd[:, oc[:, 0], oc[:, 1], oc[:, 2]].T, rtol=1e-4, atol=1e-4)
    # inverse conv: same pairs reversed == conv_transpose3d evaluated at the fine active sites
    z = rng.normal(size=(len(oc), 6)).astype(np.float32)
    taps_i = rng.normal(size=(int(np.prod(ksize)), 6, 3)).astype(np.float32)
    nbr_i = R.spconv_nbr_inverse(c, ok, None, oshape, ksize, stride, pad)
    assert (nbr_i >= 0).sum() == (nbr >= 0).sum()
    u = R.sparse_conv(z, nbr_i, taps_i)
    K, ci, co = taps_i.shape
    wT = torch.from_numpy(taps_i.reshape(ksize[0], ksize[1], ksize[2], ci, co).transpose(3, 4, 0, 1, 2).copy())
    opad = [shape[d] - ((oshape[d] - 1) * stride[d] - 2 * pad[d] + ksize[d]) for d in range(3)]
    ud = F.conv_transpose3d(_dense(z, oc, oshape), wT, stride=stride, padding=pad, output_padding=opad)[0].numpy()
    np.testing.assert_allclose(u, ud[:, c[:, 0], c[:, 1], c[:, 2]].T, rtol=1e-4, atol=1e-4)


def test_me_4d_conv_equals_sum_of_dense_3d():
    """3x3x3x3 ME conv at tensor stride s == sum over dt of dense conv3d on the (x/s) grid."""
    rng = np.random.default_rng(2)
    s = 2
    T = 4
    xyz = np.unique(rng.integers(-6, 6, size=(600, 3)), axis=0) * s
    t = rng.integers(-T + 1, 1, size=(len(xyz), 1))
    coords = np.unique(np.concatenate([xyz, t], 1), axis=0).astype(np.int32)
    keys = R.key4(coords)
    order = np.argsort(keys)
    coords, keys = coords[order], keys[order]
    x = rng.normal(size=(len(coords), 3)).astype(np.float32)
    taps = rng.normal(size=(81, 3, 5)).astype(np.float32)
    nbr = R.me_nbr(coords, keys, R.me_kernel_offsets([3, 3, 3, 3], [s, s, s, 1]))
    y = R.sparse_conv(x, nbr, taps)
    g = coords.copy()
    g[:, :3] = g[:, :3] // s + 6
    g[:, 3] += T - 1
    dense = np.zeros((T, 3, 12, 12, 12), np.float32)  # [t][c][z][y][x]
    dense[g[:, 3], :, g[:, 2], g[:, 1], g[:, 0]] = x
    w = taps.reshape(3, 3, 3, 3, 3, 5)  # [it][iz][iy][ix][ci][co]
    out = np.zeros((T, 5, 12, 12, 12), np.float32)
    for to in range(T):
        for it in range(3):
            ti = to + it - 1
            if 0 <= ti < T:
                wd = torch.from_numpy(w[it].transpose(4, 3, 0, 1, 2).copy())
                out[to] += F.conv3d(torch.from_numpy(dense[ti])[None], wd, padding=1)[0].numpy()
    np.testing.assert_allclose(y, out[g[:, 3], :, g[:, 2], g[:, 1], g[:, 0]], rtol=1e-4, atol=1e-4)


def test_me_k2s2_down_and_transpose_equal_dense():
    rng = np.random.default_rng(3)
    xyz = np.unique(rng.integers(-8, 8, size=(500, 3)), axis=0)
    coords = np.concatenate([xyz, np.zeros((len(xyz), 1), np.int64)], 1).astype(np.int32)
    keys = R.key4(coords)
    order = np.argsort(keys)
    coords, keys = coords[order], keys[order]
    pc, pk, parent = R.me_stride_down(coords, keys, 1)
    np.testing.assert_array_equal(pc[:, :3], np.unique((coords[:, :3] >> 1) << 1, axis=0)[np.argsort(
        R.key4(np.concatenate([np.unique((coords[:, :3] >> 1) << 1, axis=0), np.zeros((len(pc), 1), np.int64)], 1)))])
    off = R.me_kernel_offsets([2, 2, 2, 1], [1, 1, 1, 1])
    x = rng.normal(size=(len(coords), 4)).astype(np.float32)
    taps = rng.normal(size=(8, 4, 6)).astype(np.float32)
    nbr = R.me_nbr(pc, keys, off, +1)
    assert (nbr >= 0).sum() == len(coords)  # every fine voxel has exactly one parent
    y = R.sparse_conv(x, nbr, taps)
    g = coords[:, :3] + 8
    dense = np.zeros((4, 16, 16, 16), np.float32)
    dense[:, g[:, 2], g[:, 1], g[:, 0]] = x.T
    wd = torch.from_numpy(taps.reshape(2, 2, 2, 4, 6).transpose(4, 3, 0, 1, 2).copy())  # [co][ci][kz][ky][kx]
    yd = F.conv3d(torch.from_numpy(dense)[None], wd, stride=2)[0].numpy()
    gp = (pc[:, :3] + 8) // 2
    np.testing.assert_allclose(y, yd[:, gp[:, 2], gp[:, 1], gp[:, 0]].T, rtol=1e-4, atol=1e-4)
    # transposed conv back onto the cached fine map
    z = rng.normal(size=(len(pc), 6)).astype(np.float32)
    tt = rng.normal(size=(8, 6, 3)).astype(np.float32)
    nbr_t = R.me_nbr(coords, pk, off, -1)
    assert np.all((nbr_t >= 0).sum(0) == 1)
    u = R.sparse_conv(z, nbr_t, tt)
    dz = np.zeros((6, 8, 8, 8), np.float32)
    dz[:, gp[:, 2], gp[:, 1], gp[:, 0]] = z.T
    wT = torch.from_numpy(tt.reshape(2, 2, 2, 6, 3).transpose(3, 4, 0, 1, 2).copy())  # [ci][co][kz][ky][kx]
    ud = F.conv_transpose3d(torch.from_numpy(dz)[None], wT, stride=2)[0].numpy()
    np.testing.assert_allclose(u, ud[:, g[:, 2], g[:, 1], g[:, 0]].T, rtol=1e-4, atol=1e-4)


def test_voxelize_first_come_and_caps():
    pts = np.array([[0.05, 0.05, -2.95, 1, 0, 0, 0], [5.0, 5.0, 0.0, 2, 0, 0, 0], [0.06, 0.04, -2.96, 3, 0, 0, 0],
                    [100.0, 0, 0, 4, 0, 0, 0], [0.01, 0.02, -2.99, 5, 0, 0, 0], [5.01, 5.01, 0.01, 6, 0, 0, 0],
                    [-59.99, -49.99, -2.99, 7, 0, 0, 0], [0.07, 0.07, -2.91, 8, 0, 0, 0]], np.float32)
    vox, co, num, pid = R.voxelize_with_id(pts, [0.1, 0.1, 0.1], [-60, -50, -3, 60, 50, 1], 2, 2)
    np.testing.assert_array_equal(pid, [0, 1, 0, -1, 0, 1, -1, 0])  # 3rd voxel is over the cap of 2
    np.testing.assert_array_equal(num, [2, 2])  # only the first 2 points of each voxel are stored
    np.testing.assert_array_equal(co, [[0, 500, 600], [30, 550, 650]])
    np.testing.assert_allclose(R.mean_vfe(vox, num)[:, 3], [2.0, 4.0])


def test_me_5x5x5x1_tap_order_equals_dense_conv3d_with_an_asymmetric_kernel():
    """The first MotionNet layer (kernel [5,5,5,1], motionnet.py / minkunet.py:55-58): ME's kernel-region order is x fastest,
    then y, z -- with a kernel that has no symmetry any other enumeration (z fastest, mirrored offsets) fails this."""
    rng = np.random.default_rng(5)
    xyz = np.unique(rng.integers(-7, 7, size=(700, 3)), axis=0)
    coords = np.concatenate([xyz, np.zeros((len(xyz), 1), np.int64)], 1).astype(np.int32)
    keys = R.key4(coords)
    order = np.argsort(keys)
    coords, keys = coords[order], keys[order]
    x = rng.normal(size=(len(coords), 2)).astype(np.float32)
    taps = rng.normal(size=(125, 2, 3)).astype(np.float32)
    taps[0] += 5.0            # tap 0 = offset (-2, -2, -2): a strongly marked corner
    nbr = R.me_nbr(coords, keys, R.me_kernel_offsets([5, 5, 5, 1], [1, 1, 1, 1]))
    y = R.sparse_conv(x, nbr, taps)
    g = coords[:, :3] + 7
    dense = np.zeros((2, 14, 14, 14), np.float32)      # [c][z][y][x]
    dense[:, g[:, 2], g[:, 1], g[:, 0]] = x.T
    wd = torch.from_numpy(taps.reshape(5, 5, 5, 2, 3).transpose(4, 3, 0, 1, 2).copy())   # tap = ix + 5*iy + 25*iz -> [co][ci][kz][ky][kx]
    yd = F.conv3d(torch.from_numpy(dense)[None], wd, padding=2)[0].numpy()
    np.testing.assert_allclose(y, yd[:, g[:, 2], g[:, 1], g[:, 0]].T, rtol=1e-4, atol=1e-4)
    # and the enumeration with z fastest is NOT the same function (the test has teeth)
    wz = torch.from_numpy(taps.reshape(5, 5, 5, 2, 3).transpose(4, 3, 2, 1, 0).copy())
    yz = F.conv3d(torch.from_numpy(dense)[None], wz, padding=2)[0].numpy()
    assert np.abs(y - yz[:, g[:, 2], g[:, 1], g[:, 0]].T).max() > 1.0


def test_me_4d_conv_does_not_reach_across_a_missing_time_slice():
    """A 3^4 kernel spans t-1 .. t+1 in COORDINATE units: with the scan at t = -1 missing, the voxels at t = 0 must not see
    the scan at t = -2 (a table built on slice RANKS instead of time coordinates would).  Dense equivalent: an empty slice."""
    rng = np.random.default_rng(6)
    T = 4
    xyz = np.unique(rng.integers(-5, 5, size=(400, 3)), axis=0)
    parts = []
    for t in (-3, -2, 0):                                  # t = -1 is absent
        sel = xyz[rng.random(len(xyz)) < 0.7]
        parts.append(np.concatenate([sel, np.full((len(sel), 1), t)], 1))
    coords = np.concatenate(parts).astype(np.int32)
    keys = R.key4(coords)
    order = np.argsort(keys)
    coords, keys = coords[order], keys[order]
    x = rng.normal(size=(len(coords), 3)).astype(np.float32)
    taps = rng.normal(size=(81, 3, 4)).astype(np.float32)
    nbr = R.me_nbr(coords, keys, R.me_kernel_offsets([3, 3, 3, 3], [1, 1, 1, 1]))
    y = R.sparse_conv(x, nbr, taps)
    g = coords.copy()
    g[:, :3] += 5
    g[:, 3] += T - 1
    dense = np.zeros((T, 3, 10, 10, 10), np.float32)
    dense[g[:, 3], :, g[:, 2], g[:, 1], g[:, 0]] = x
    w = taps.reshape(3, 3, 3, 3, 3, 4)
    out = np.zeros((T, 4, 10, 10, 10), np.float32)
    for to in range(T):
        for it in range(3):
            ti = to + it - 1
            if 0 <= ti < T:
                out[to] += F.conv3d(torch.from_numpy(dense[ti])[None], torch.from_numpy(w[it].transpose(4, 3, 0, 1, 2).copy()),
                                    padding=1)[0].numpy()
    np.testing.assert_allclose(y, out[g[:, 3], :, g[:, 2], g[:, 1], g[:, 0]], rtol=1e-4, atol=1e-4)
    cur = coords[:, 3] == 0
    past_taps = nbr[:27][:, cur]                           # it = 0 <-> t - 1: nothing there for the current scan
    assert (past_taps < 0).all() and (nbr[27:54][:, cur] >= 0).any()


def test_voxel_cap_and_point_cap_interplay():
    """spconv's CPU PointToVoxel walk (voxel_generate.py:19-28 -> generate_voxel_with_id): points are visited in order; a point
    whose cell is new opens a voxel only while fewer than max_voxels exist -- afterwards the cell is dropped for good (-1) --
    while points of cells opened EARLIER keep their voxel id even beyond max_num_points, and only the first max_num_points of
    a voxel enter its mean."""
    A, B, C, D = [0.05, 0.05, -2.95], [5.0, 5.0, 0.0], [-3.0, 2.0, 0.5], [7.0, -7.0, -1.0]
    seq = [A, B, C, A, A, A, B, C, D, A]                   # cap 2 voxels, 3 points: C and D never get a voxel
    pts = np.array([p + [float(i + 1), 0, 0, 0] for i, p in enumerate(seq)], np.float32)
    vox, co, num, pid = R.voxelize_with_id(pts, [0.1, 0.1, 0.1], [-60, -50, -3, 60, 50, 1], 2, 3)
    np.testing.assert_array_equal(pid, [0, 1, -1, 0, 0, 0, 1, -1, -1, 0])
    np.testing.assert_array_equal(num, [3, 2])
    np.testing.assert_allclose(R.mean_vfe(vox, num)[:, 3], [(1 + 4 + 5) / 3.0, (2 + 7) / 2.0])   # A's 4th and 5th point are not stored
    # with room for three voxels C is kept, D still is not; A's id does not depend on the cap
    _, co3, num3, pid3 = R.voxelize_with_id(pts, [0.1, 0.1, 0.1], [-60, -50, -3, 60, 50, 1], 3, 3)
    np.testing.assert_array_equal(pid3, [0, 1, 2, 0, 0, 0, 1, 2, -1, 0])
    np.testing.assert_array_equal(num3, [3, 2, 2])
    np.testing.assert_array_equal(co3[:2], co)


@pytest.fixture(scope="module")
def s0():
    from insmos_amd.synth import make_window
    return make_window(0, 10, 1886)


def test_s0_known_answers(s0):
    """voxel counts per tensor stride / per spconv level, and rulebook sizes."""
    assert s0.shape == (1199606, 5)
    pts4 = np.concatenate([s0[:, :3], s0[:, 4:5]], 1)
    c, k, inv = R.me_quantize(pts4, [0.1, 0.1, 0.1, 0.1])
    counts = [len(c)] + [len(R.me_stride_down(c, k, L)[0]) for L in (1, 2, 3)]
    assert counts == [468007, 211916, 83805, 30183]
    pc, pk, _ = R.me_stride_down(c, k, 3)
    n = R.me_nbr(pc, pk, R.me_kernel_offsets([3, 3, 3, 3], [8, 8, 8, 1]))
    assert int((n >= 0).sum()) == 691903
    cur = s0[s0[:, 4] == 0]
    assert len(cur) == 119817
    cur7 = np.concatenate([cur[:, :4], np.zeros((len(cur), 3), np.float32)], 1)
    vox, co, num, pid = R.voxelize_with_id(cur7, [0.1, 0.1, 0.1], [-60, -50, -3, 60, 50, 1], 100000, 5)
    assert len(co) == 42280 and int((pid >= 0).sum()) == 118333
    shape = [41, 1000, 1200]
    ks, perm = R.sorted_index(R.key3(co, shape))
    assert int((R.spconv_nbr_subm(co, ks, perm, shape) >= 0).sum()) == 263860
    lv = [(co, ks, perm, shape)]
    for _ in range(3):
        ci, ki, pi, si = lv[-1]
        oc, ok, osz = R.spconv_down_coords(ci, si, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        lv.append((oc, ok, None, osz))
    assert [len(l[0]) for l in lv] == [42280, 30867, 12564, 6717]
    assert [int((R.spconv_nbr_subm(l[0], l[1], l[2], l[3]) >= 0).sum()) for l in lv[1:]] == [350793, 151384, 102743]
    c5, k5, s5 = R.spconv_down_coords(lv[3][0], lv[3][3], (3, 1, 1), (2, 1, 1), (0, 0, 0))
    assert len(c5) == 4810 and s5 == [2, 125, 150]
