"""-m gpu: coordinate kernels (quantise / levels / voxelise / strided sets / neighbour tables) against
the oracle, bit-exact (integer work)."""
import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def window():
    from insmos_amd.synth import make_window
    return make_window(seed=3, n_scans=10, n_az=256)


@pytest.fixture(scope="module")
def engine():
    from insmos_amd import params as P
    from insmos_amd.engine import Engine
    cfg = P.default_cfg()
    return Engine(cfg, P.random_state_dict(cfg, 0))


def test_quantize_levels_and_tables(window, engine):
    from gpu_util import dev, u64
    pts = dev(window)
    engine.const_input = True
    engine.motionnet(pts)
    torch.cuda.synchronize()
    out_p1_const = engine._me_debug["cat8"][:, 8:16].clone()
    r1 = engine.last_counts["me_row_starts"][0][1] & ~15  # dead-row elimination: level-0 table rows of the last 2 scans
    assert r1 > 0
    tab0_pruned = engine._me_tables["nbr81"][0].nbr[:, r1:].clone()
    mask0_pruned = engine._me_tables["nbr81"][0].mask16[r1 // 16:].clone()
    engine.const_input = False  # generic path: materialises the 125-tap table and gathers the 0.5 features
    engine.prune_dead_rows = False  # ... and every row of every table, as the reference's libraries do
    try:
        cur = engine.motionnet(pts)
        torch.cuda.synchronize()
    finally:
        engine.const_input = True
        engine.prune_dead_rows = True
    assert torch.equal(tab0_pruned, engine._me_tables["nbr81"][0].nbr[:, r1:])
    assert torch.equal(mask0_pruned, engine._me_tables["nbr81"][0].mask16[r1 // 16:])
    # the constant-input first layer is bitwise the generic MFMA path (same tap order, 0.5*w exact)
    assert torch.equal(out_p1_const, engine._me_debug["cat8"][:, 8:16])
    T = engine._me_tables
    pts4 = np.concatenate([window[:, :3], window[:, 4:5]], 1)
    c, k, inv = R.me_quantize(pts4, [0.1, 0.1, 0.1, 0.1])
    np.testing.assert_array_equal(T["coords"][0].cpu().numpy(), c)
    np.testing.assert_array_equal(u64(T["keys"][0]), k)
    np.testing.assert_array_equal(T["inverse"].cpu().numpy(), inv)
    lv = [(c, k)]
    for L in (1, 2, 3):
        pc, pk, _ = R.me_stride_down(c, k, L)
        np.testing.assert_array_equal(T["coords"][L].cpu().numpy(), pc)
        np.testing.assert_array_equal(u64(T["keys"][L]), pk)
        lv.append((pc, pk))
    n125 = R.me_nbr(c, k, R.me_kernel_offsets([5, 5, 5, 1], [1] * 4))
    np.testing.assert_array_equal(T["nbr125"].cpu().numpy(), n125)
    from gpu_util import tap_masks
    np.testing.assert_array_equal(T["nbr125"].mask16.cpu().numpy().view(np.uint32), tap_masks(n125))
    np.testing.assert_array_equal(T["nbr81"][2].mask16.cpu().numpy().view(np.uint32),
                                  tap_masks(T["nbr81"][2].cpu().numpy()))
    for L in range(4):
        s = 1 << L
        np.testing.assert_array_equal(T["nbr81"][L].cpu().numpy(),
                                      R.me_nbr(lv[L][0], lv[L][1], R.me_kernel_offsets([3, 3, 3, 3], [s, s, s, 1])))
    for L in range(3):
        s = 1 << L
        off = R.me_kernel_offsets([2, 2, 2, 1], [s, s, s, 1])
        np.testing.assert_array_equal(T["dn"][L].cpu().numpy(), R.me_nbr(lv[L + 1][0], lv[L][1], off, +1))
        np.testing.assert_array_equal(T["up"][L].cpu().numpy(), R.me_nbr(lv[L][0], lv[L + 1][1], off, -1))
    ncur = int((window[:, 4] == 0).sum())
    assert cur.shape == (ncur, 8)
    np.testing.assert_array_equal(cur[:, :4].cpu().numpy(), window[window[:, 4] == 0][:, :4])


def test_quantize_compact_and_full_width_keys(window, engine):
    """The 40-bit sort keys order exactly like the canonical 64-bit ones; windows beyond +-2048 voxels / 16 time steps
    fall back to the full-width sort.  Both against the oracle."""
    from gpu_util import dev, u64
    import ctypes
    from insmos_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2)
    base = window[rng.permutation(len(window))[:60000]].copy()
    wide = base.copy()
    wide[:3, 0] = [250.0, -230.5, 1000.0]   # 2500 / -2305 / 10000 voxels
    wide[3:6, 4] = [-1.7, -2.0, -3.1]         # t = -17, -20, -31
    corner = base.copy()
    corner[:4, :3] = [[204.79, 204.79, 204.79], [-204.8, -204.8, -204.8], [204.79, -204.8, 0.0], [-0.05, 0.05, -0.05]]
    corner[4:6, 4] = [-1.5, -1.45]            # t = -15 (inside the compact box)
    for name, w, expect_fallback in (("base", base, False), ("corner", corner, False), ("wide", wide, True)):
        pts = dev(w)
        N = len(w)
        c_ref, k_ref, inv_ref = R.me_quantize(np.concatenate([w[:, :3], w[:, 4:5]], 1), [0.1, 0.1, 0.1, 0.1])
        ws = torch.empty(int(lib.insmos_quantize4d_ws_bytes(N)), dtype=torch.uint8, device="cuda")
        quant = np.array([0.1, 0.1, 0.1, 0.1], np.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for compact in (1, 0):
            keys = torch.empty(N, dtype=torch.int64, device="cuda")
            coords = torch.empty((N, 4), dtype=torch.int32, device="cuda")
            inverse = torch.empty(N, dtype=torch.int32, device="cuda")
            cur = torch.empty(N, dtype=torch.int32, device="cuda")
            counts = torch.empty(4, dtype=torch.int32, device="cuda")
            _lib.check(lib.insmos_quantize4d_ex(pts.data_ptr(), N, 5, quant.ctypes.data_as(ctypes.c_void_p), keys.data_ptr(),
                                                coords.data_ptr(), inverse.data_ptr(), cur.data_ptr(), counts.data_ptr(),
                                                ws.data_ptr(), ws.numel(), compact, st), "quantize")
            c = counts.cpu().numpy()
            assert c[2] == 0
            if compact and expect_fallback:
                assert c[3] == 6, (name, c)
                continue
            assert c[3] == 0 and c[0] == len(k_ref), (name, compact, c)
            np.testing.assert_array_equal(u64(keys[:c[0]]), k_ref, err_msg=f"{name} compact={compact}")
            np.testing.assert_array_equal(coords[:c[0]].cpu().numpy(), c_ref)
            np.testing.assert_array_equal(inverse.cpu().numpy(), inv_ref)
    # and through the engine: the wide window takes the fallback transparently
    engine.motionnet(dev(wide))
    c_ref, k_ref, _ = R.me_quantize(np.concatenate([wide[:, :3], wide[:, 4:5]], 1), [0.1, 0.1, 0.1, 0.1])
    np.testing.assert_array_equal(u64(engine._me_tables["keys"][0]), k_ref)


@pytest.mark.parametrize("max_voxels", [100000, 1500])
def test_voxelize_and_3d_tables(window, engine, max_voxels):
    from gpu_util import dev
    rng = np.random.default_rng(0)
    curp = window[window[:, 4] == 0]
    cur8 = np.concatenate([curp[:, :4], rng.normal(size=(len(curp), 3)).astype(np.float32),
                           np.zeros((len(curp), 1), np.float32)], 1)
    old = engine.max_voxels
    engine.max_voxels = max_voxels
    try:
        engine._conv_log = []
        engine.unet(dev(cur8))
        torch.cuda.synchronize()
    finally:
        engine.max_voxels = old
    T = engine._un_tables
    vox, co, num, pid = R.voxelize_with_id(cur8[:, :7], [0.1] * 3, [-60, -50, -3, 60, 50, 1], max_voxels, 5)
    np.testing.assert_array_equal(T["coords"][1].cpu().numpy()[:, 1:], co)
    assert int(T["coords"][1][:, 0].abs().sum()) == 0
    np.testing.assert_array_equal(T["pcid"].cpu().numpy(), pid)
    np.testing.assert_array_equal(T["num_points"].cpu().numpy(), num)
    np.testing.assert_allclose(T["feat"].cpu().numpy()[:, :7], R.mean_vfe(vox, num), rtol=0, atol=1e-6)
    assert float(T["feat"][:, 7].abs().sum()) == 0.0
    shape = engine.shape
    ks, perm = R.sorted_index(R.key3(co, shape[1]))
    S = {1: (co, ks, perm, shape[1])}
    for l in (2, 3, 4):
        oc, ok, osz = R.spconv_down_coords(S[l - 1][0], S[l - 1][3], (3, 3, 3), (2, 2, 2), (1, 1, 1))
        assert osz == shape[l]
        S[l] = (oc, ok, None, osz)
        np.testing.assert_array_equal(T["coords"][l].cpu().numpy()[:, 1:], oc)
    c5, k5, s5 = R.spconv_down_coords(S[4][0], S[4][3], (3, 1, 1), (2, 1, 1), (0, 0, 0))
    np.testing.assert_array_equal(T["coords"][5].cpu().numpy()[:, 1:], c5)
    for l in (1, 2, 3, 4):
        np.testing.assert_array_equal(T["subm"][l].cpu().numpy(), R.spconv_nbr_subm(*S[l]))
    for l in (2, 3, 4):
        np.testing.assert_array_equal(T["down"][l].cpu().numpy(),
                                      R.spconv_nbr_down(S[l][0], S[l - 1][1], S[l - 1][2], S[l - 1][3], (3, 3, 3),
                                                        (2, 2, 2), (1, 1, 1)))
        np.testing.assert_array_equal(T["inv"][l].cpu().numpy(),
                                      R.spconv_nbr_inverse(S[l - 1][0], S[l][1], None, S[l][3], (3, 3, 3), (2, 2, 2),
                                                           (1, 1, 1)))
    np.testing.assert_array_equal(T["down5"].cpu().numpy(),
                                  R.spconv_nbr_down(c5, S[4][1], None, S[4][3], (3, 1, 1), (2, 1, 1), (0, 0, 0)))
    np.testing.assert_array_equal(T["inv5"].cpu().numpy(),
                                  R.spconv_nbr_inverse(S[4][0], k5, None, s5, (3, 1, 1), (2, 1, 1), (0, 0, 0)))


def test_quantize_edge_cases():
    """negative coordinates, exact voxel boundaries, duplicate points, t rounding (-0.3/0.1 etc.)."""
    from gpu_util import dev, lib, stream, ws, hp, u64
    from insmos_amd import _lib
    pts = np.array([[0.0, 0.0, 0.0, 0.5, 0.0], [-0.0, -0.05, 0.1, 0.5, -0.3], [0.1, 0.2, 0.3, 0.1, -0.3],
                    [0.1, 0.2, 0.3, 0.9, -0.3], [-12.34, 56.78, -1.0, 0.2, -0.9], [79.99, -79.99, 3.3, 0.3, -0.7],
                    [0.30000001, 0.6, 0.7, 0.2, -0.1], [0.29999998, 0.6, 0.7, 0.2, -0.1], [1e-8, -1e-8, 0, 0, 0.0]],
                   np.float32)
    n = len(pts)
    d = dev(pts)
    keys = torch.empty(n, dtype=torch.int64, device="cuda:0")
    coords = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
    inv = torch.empty(n, dtype=torch.int32, device="cuda:0")
    cur = torch.empty(n, dtype=torch.int32, device="cuda:0")
    counts = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    w = ws(lib().insmos_quantize4d_ws_bytes(n))
    q = np.array([0.1, 0.1, 0.1, 0.1], np.float32)
    _lib.check(lib().insmos_quantize4d(d.data_ptr(), n, 5, hp(q), keys.data_ptr(), coords.data_ptr(), inv.data_ptr(),
                                       cur.data_ptr(), counts.data_ptr(), w.data_ptr(), w.numel(), stream()), "quant")
    c, k, i = R.me_quantize(np.concatenate([pts[:, :3], pts[:, 4:5]], 1), q)
    V, ncur = int(counts[0]), int(counts[1])
    assert V == len(c) and int(counts[2]) == 0
    np.testing.assert_array_equal(coords[:V].cpu().numpy(), c)
    np.testing.assert_array_equal(inv.cpu().numpy(), i)
    np.testing.assert_array_equal(cur[:ncur].cpu().numpy(), np.nonzero((pts[:, 4] / q[3]) == 0)[0])


def test_rank_map_tables_match_the_oracle_and_the_searched_builder():
    """The search-free 3D kernel maps (insmos_rankmap_from_keys / insmos_down_coords3d_rank / insmos_build_nbr_rank) on a batch
    of two windows stacked along the batch column: the strided coordinate set, the submanifold / strided / inverse tables and
    their active-tap masks equal the oracle's and the searched builder's (insmos_build_nbr), including a level-1 permutation
    with dropped voxels (-1)."""
    from gpu_util import dev, hp, i32, lib, stream, tap_masks, u64, ws
    L = lib()
    rng = np.random.default_rng(21)
    shape, B = (9, 40, 52), 2
    cells = shape[0] * shape[1] * shape[2]
    per = [np.sort(rng.choice(cells, size=n, replace=False)) for n in (1500, 900)]
    coords = np.concatenate([np.concatenate([np.full((len(c), 1), b), np.stack(np.unravel_index(c, shape), 1)], 1)
                             for b, c in enumerate(per)], 0).astype(np.int32)          # [b, z, y, x], ascending keys
    keys = np.concatenate([b * cells + c for b, c in enumerate(per)]).astype(np.uint64)
    n = len(coords)
    # level-1 style permutation: sorted position -> row in "first seen" order, some voxels dropped by a cap
    perm = rng.permutation(n).astype(np.int32)
    perm[rng.choice(n, size=60, replace=False)] = -1
    kept = perm >= 0
    rows = np.full(n, -1, np.int64)
    rows[perm[kept]] = np.flatnonzero(kept)
    # the kept voxels' coordinates in ROW order (rows are a permutation of 0..n-1 with holes where voxels were dropped)
    order = np.argsort(perm[kept])
    row_coords = coords[kept][order]
    perm_c = np.full(n, -1, np.int32)
    perm_c[np.flatnonzero(kept)[order]] = np.arange(kept.sum(), dtype=np.int32)       # compact rows 0..n_kept-1
    shp = i32(shape)
    nw = int(L.insmos_rankmap_words(hp(shp), B))
    bits = torch.zeros(nw, dtype=torch.int64, device="cuda")
    incl = torch.zeros(nw // 4, dtype=torch.int32, device="cuda")
    w = ws(L.insmos_rankmap_ws_bytes(hp(shp), B))
    keys_d = dev(keys.view(np.int64))
    assert L.insmos_rankmap_from_keys(keys_d.data_ptr(), n, hp(shp), B, bits.data_ptr(), incl.data_ptr(), w.data_ptr(), w.numel(),
                                      stream()) == 0
    torch.cuda.synchronize()
    assert int(incl[-1]) == n
    one = i32([1, 1, 1, 1])
    d_subm = i32([[0, kz - 1, ky - 1, kx - 1] for kz in range(3) for ky in range(3) for kx in range(3)])

    def both(out_coords, in_perm, in_shape, delta, mul, div, bits_t, incl_t, in_keys):
        no = len(out_coords)
        oc = dev(out_coords)
        res = []
        for use_rank in (True, False):
            nbr = torch.full((len(delta), no), -7, dtype=torch.int32, device="cuda")
            mask = torch.full(((no + 15) // 16, 4), -1, dtype=torch.int32, device="cuda")   # garbage: the rank builder overwrites
            pm = dev(in_perm) if in_perm is not None else None
            if use_rank:
                rc = L.insmos_build_nbr_rank(oc.data_ptr(), no, bits_t.data_ptr(), incl_t.data_ptr(), pm.data_ptr() if pm is not None else None,
                                             hp(i32(in_shape)), hp(delta), len(delta), hp(mul), hp(div), nbr.data_ptr(), mask.data_ptr(), stream())
            else:
                kd = dev(in_keys.view(np.int64))
                rc = L.insmos_build_nbr(oc.data_ptr(), no, kd.data_ptr(), pm.data_ptr() if pm is not None else None, len(in_keys), 1,
                                        hp(i32(in_shape)), hp(delta), len(delta), hp(mul), hp(div), nbr.data_ptr(), mask.data_ptr(), stream())
            assert rc == 0
            torch.cuda.synchronize()
            res.append((nbr.cpu().numpy(), mask.cpu().numpy().view(np.uint32)))
        np.testing.assert_array_equal(res[0][0], res[1][0])
        np.testing.assert_array_equal(res[0][1], res[1][1])
        np.testing.assert_array_equal(res[0][1], tap_masks(res[0][0]))
        # sparse stores: the same masks, and the same entries wherever a (16-row group, tap) is in its group's mask (the others stay
        # as they were -- here the prefill -- and are never read by the 16-row-tile convolution kernels)
        nbr_s = torch.full((len(delta), no), -7, dtype=torch.int32, device="cuda")
        mask_s = torch.zeros(((no + 15) // 16, 4), dtype=torch.int32, device="cuda")
        pm = dev(in_perm) if in_perm is not None else None
        assert L.insmos_build_nbr_rank_sparse(oc.data_ptr(), no, bits_t.data_ptr(), incl_t.data_ptr(), pm.data_ptr() if pm is not None else None,
                                              hp(i32(in_shape)), hp(delta), len(delta), hp(mul), hp(div), nbr_s.data_ptr(), mask_s.data_ptr(),
                                              stream()) == 0
        torch.cuda.synchronize()
        np.testing.assert_array_equal(mask_s.cpu().numpy().view(np.uint32), res[0][1])
        ns_, full = nbr_s.cpu().numpy(), res[0][0]
        act = np.zeros((len(delta), (no + 15) // 16), bool)
        for k in range(len(delta)):
            act[k] = (res[0][1][:, k >> 5] >> np.uint32(k & 31)) & 1
        act_rows = np.repeat(act, 16, axis=1)[:, :no]
        np.testing.assert_array_equal(ns_[act_rows], full[act_rows])
        assert (ns_[~act_rows] == -7).all() and (~act_rows).any()
        return res[0][0]

    # submanifold table over the permuted level (rows = first-seen order, some cells dropped)
    sub = both(row_coords, perm_c, shape, d_subm, one, one, bits, incl, keys)
    for b in range(B):   # per window against the oracle (its tables are single-window)
        sel = row_coords[:, 0] == b
        kb = (keys[(keys // cells) == b] - b * cells).astype(np.uint64)
        pb = perm_c[(keys // cells) == b]
        ref = R.spconv_nbr_subm(row_coords[sel][:, 1:], kb, pb, shape)
        np.testing.assert_array_equal(sub[:, sel], ref)
    # strided k3 s2 p1 output set + its rank map, then the strided and inverse tables (identity permutation on both sides)
    oshape = tuple(R.spconv_out_shape(shape, (3, 3, 3), (2, 2, 2), (1, 1, 1)))
    osh = i32(oshape)
    nwo = int(L.insmos_rankmap_words(hp(osh), B))
    obits = torch.zeros(nwo, dtype=torch.int64, device="cuda")
    oincl = torch.zeros(nwo // 4, dtype=torch.int32, device="cuda")
    cap = min(n * 27, int(np.prod(oshape)) * B)
    okeys = torch.zeros(cap, dtype=torch.int64, device="cuda")
    ocoords = torch.zeros((cap, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(8, dtype=torch.int32, device="cuda")
    w2 = ws(L.insmos_rankmap_ws_bytes(hp(osh), B))
    cd = dev(coords)
    assert L.insmos_down_coords3d_rank(cd.data_ptr(), n, hp(i32([3, 3, 3])), hp(i32([2, 2, 2])), hp(i32([1, 1, 1])), hp(osh), B,
                                       okeys.data_ptr(), ocoords.data_ptr(), counts.data_ptr(), obits.data_ptr(), oincl.data_ptr(),
                                       w2.data_ptr(), w2.numel(), stream()) == 0
    torch.cuda.synchronize()
    no = int(counts[0])
    ocells = int(np.prod(oshape))
    ref_c, ref_k = [], []
    for b in range(B):
        c_b, k_b, _ = R.spconv_down_coords(coords[coords[:, 0] == b][:, 1:], shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        ref_c.append(np.concatenate([np.full((len(c_b), 1), b, np.int32), c_b], 1))
        ref_k.append(k_b.astype(np.uint64) + np.uint64(b * ocells))
    ref_c, ref_k = np.concatenate(ref_c), np.concatenate(ref_k)
    assert no == len(ref_c)
    np.testing.assert_array_equal(ocoords[:no].cpu().numpy(), ref_c)
    np.testing.assert_array_equal(u64(okeys[:no]), ref_k)
    two = i32([1, 2, 2, 2])
    d_inv = i32([[0, 1 - kz, 1 - ky, 1 - kx] for kz in range(3) for ky in range(3) for kx in range(3)])
    down = both(ref_c, None, shape, d_subm, two, one, bits, incl, keys)          # coarse rows read fine rows
    inv = both(coords, None, oshape, d_inv, one, two, obits, oincl, ref_k)       # fine rows read coarse rows
    for b in range(B):
        fine_b = coords[coords[:, 0] == b][:, 1:]
        off_f, off_c = int((coords[:, 0] < b).sum()), int((ref_c[:, 0] < b).sum())
        kf = (keys[(keys // cells) == b] - b * cells).astype(np.uint64)
        sel_c = ref_c[:, 0] == b
        rd = R.spconv_nbr_down(ref_c[sel_c][:, 1:], kf, None, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        got = down[:, sel_c]
        np.testing.assert_array_equal(np.where(got >= 0, got - off_f, -1), rd)
        ri = R.spconv_nbr_inverse(fine_b, (ref_k[sel_c] - np.uint64(b * ocells)), None, oshape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        got = inv[:, coords[:, 0] == b]
        np.testing.assert_array_equal(np.where(got >= 0, got - off_c, -1), ri)
    # ---- the same three tables (+ a 3-tap k311 one with an empty job in between) from ONE launch (insmos_build_nbr_rank_multi, what
    # the native runner uses): entries and masks equal the per-table builder's, dense and with sparse stores
    import ctypes
    from insmos_amd import _lib
    d_k3 = i32([[0, kz, 0, 0] for kz in range(3)])
    two1 = i32([1, 2, 1, 1])
    jobs_np = [(row_coords, perm_c, shape, d_subm, one, one, bits, incl), (ref_c, None, shape, d_subm, two, one, bits, incl),
               (coords[:0], None, shape, d_subm, one, one, bits, incl),          # an empty level: skipped
               (coords, None, oshape, d_inv, one, two, obits, oincl), (ref_c, None, shape, d_k3, two1, None, bits, incl)]
    for sparse in (0, 1):
        keep, outs = [], []
        arr = (_lib.RankJob * len(jobs_np))()
        for i, (oc_np, pm_np, shp_in, delta, mul, div, b_t, i_t) in enumerate(jobs_np):
            no_ = len(oc_np)
            oc_t = dev(oc_np) if no_ else torch.zeros((1, 4), dtype=torch.int32, device="cuda")
            pm_t = dev(pm_np) if pm_np is not None else None
            nbr_t = torch.full((len(delta), max(no_, 1)), -7, dtype=torch.int32, device="cuda")
            mk_t = torch.full(((max(no_, 1) + 15) // 16, 4), -1, dtype=torch.int32, device="cuda")
            shp_h, keepalive = i32(shp_in), (oc_t, pm_t, delta, mul, div)
            keep.append((keepalive, shp_h))
            j = arr[i]
            j.out_coords, j.bits, j.blk_incl = oc_t.data_ptr(), b_t.data_ptr(), i_t.data_ptr()
            j.in_perm = pm_t.data_ptr() if pm_t is not None else None
            j.nbr, j.mask16 = nbr_t.data_ptr(), mk_t.data_ptr()
            j.in_shape, j.delta = hp(shp_h), hp(delta)
            j.mul = hp(mul) if mul is not None else None
            j.div = hp(div) if div is not None else None
            j.n_out, j.K, j.reserved = no_, len(delta), 0
            outs.append((nbr_t, mk_t, no_))
        assert L.insmos_build_nbr_rank_multi(ctypes.byref(arr), len(jobs_np), sparse, stream()) == 0
        torch.cuda.synchronize()
        for (nbr_t, mk_t, no_), (oc_np, pm_np, shp_in, delta, mul, div, b_t, i_t) in zip(outs, jobs_np):
            if no_ == 0:
                assert (nbr_t.cpu().numpy() == -7).all()
                continue
            ref_n = torch.full((len(delta), no_), -7, dtype=torch.int32, device="cuda")
            ref_m = torch.full(((no_ + 15) // 16, 4), -1, dtype=torch.int32, device="cuda")
            oc_t = dev(oc_np)
            pm_t = dev(pm_np) if pm_np is not None else None
            fn = L.insmos_build_nbr_rank_sparse if sparse else L.insmos_build_nbr_rank
            assert fn(oc_t.data_ptr(), no_, b_t.data_ptr(), i_t.data_ptr(), pm_t.data_ptr() if pm_t is not None else None, hp(i32(shp_in)),
                      hp(delta), len(delta), hp(mul) if mul is not None else None, hp(div) if div is not None else None,
                      ref_n.data_ptr(), ref_m.data_ptr(), stream()) == 0
            torch.cuda.synchronize()
            assert torch.equal(nbr_t, ref_n) and torch.equal(mk_t, ref_m)
    # limits are refused, not truncated
    bad = (_lib.RankJob * 1)()
    bad[0] = arr[0]
    bad[0].K = 33
    assert L.insmos_build_nbr_rank_multi(ctypes.byref(bad), 1, 0, stream()) != 0
    assert L.insmos_build_nbr_rank_multi(ctypes.byref(arr), 17, 0, stream()) != 0


@pytest.mark.parametrize("n_per,B", [(700, 1), (40000, 2), (160000, 1), (260000, 3), (1300000, 1), (600000, 8)])
def test_packed_sort_keys_give_the_pair_sort_results(n_per, B):
    """insmos_quantize4d_windows: the packed-key sort (mode 2: 40-bit key + 24-bit point index in one u64, keys-only radix sort)
    yields exactly what the pair sorts (modes 1 and 0) yield -- voxel keys, coordinates, point -> voxel map, current-point list
    and counts -- from a few hundred points (rocPRIM's small-size paths) to a full launch set; a point outside the packed box
    (z beyond +-256 voxels) is reported in counts[3] so that the caller falls back."""
    import ctypes
    from gpu_util import hp, lib, stream, ws
    L = lib()
    rng = np.random.default_rng(n_per + B)
    wins = []
    for b in range(B):
        n = n_per + 17 * b
        p = np.zeros((n, 5), np.float32)
        p[:, 0] = rng.uniform(-80, 80, n); p[:, 1] = rng.uniform(-60, 60, n); p[:, 2] = rng.uniform(-3, 12, n)
        p[: n // 3, :3] = np.round(p[: n // 3, :3], 1)                     # many points per voxel + exact cell borders
        p[:, 3] = rng.uniform(0, 1, n)
        p[:, 4] = -np.round(rng.integers(0, 10, n) * 0.1, 3)
        wins.append(torch.from_numpy(p).cuda())
    N = sum(int(w.shape[0]) for w in wins)
    win_ptr = (ctypes.c_void_p * B)(*[w.data_ptr() for w in wins])
    win_n = (ctypes.c_int64 * B)(*[int(w.shape[0]) for w in wins])
    quant = np.array([0.1, 0.1, 0.1, 0.1], np.float32)
    w_ = ws(L.insmos_quantize4d_ws_bytes(N))
    res = {}
    for mode in (2, 1, 0):
        keys = torch.zeros(N, dtype=torch.int64, device="cuda")
        coords = torch.zeros((N, 4), dtype=torch.int32, device="cuda")
        inverse = torch.full((N,), -5, dtype=torch.int32, device="cuda")
        cur = torch.full((N,), -5, dtype=torch.int32, device="cuda")
        counts = torch.zeros(8 + B, dtype=torch.int32, device="cuda")
        assert L.insmos_quantize4d_windows(win_ptr, win_n, B, 5, hp(quant), keys.data_ptr(), coords.data_ptr(), inverse.data_ptr(),
                                           cur.data_ptr(), counts.data_ptr(), w_.data_ptr(), w_.numel(), mode, stream()) == 0
        torch.cuda.synchronize()
        c = counts.cpu().numpy()
        assert c[2] == 0 and c[3] == 0, (mode, c)
        nv, nc = int(c[0]), int(c[1])
        res[mode] = (c.copy(), keys[:nv].cpu().numpy(), coords[:nv].cpu().numpy(), inverse.cpu().numpy(), cur[:nc].cpu().numpy())
    for mode in (1, 0):
        for a, b_ in zip(res[2], res[mode]):
            np.testing.assert_array_equal(a, b_)
    ku = res[2][1].view(np.uint64)
    assert (ku[1:] > ku[:-1]).all()                                                     # ascending unique keys
    # one point outside the packed box: mode 2 reports it, mode 1 handles it
    wins[0][5, 2] = 30.0
    counts = torch.zeros(8 + B, dtype=torch.int32, device="cuda")
    scratch = [torch.zeros(N * 4, dtype=torch.int64, device="cuda") for _ in range(4)]
    assert L.insmos_quantize4d_windows(win_ptr, win_n, B, 5, hp(quant), scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(),
                                       scratch[3].data_ptr(), counts.data_ptr(), w_.data_ptr(), w_.numel(), 2, stream()) == 0
    torch.cuda.synchronize()
    assert int(counts[3]) == 1


@pytest.mark.parametrize("block", [256, 1024, 4096, 1 << 30])
def test_regroup_rows_is_a_blockwise_signature_sort(block):
    """insmos_regroup_rows3d: inside every block of `block` consecutive rows the rows come out sorted by (window, 27-bit
    submanifold tap signature, coordinate parity class, old row); new_of_old is a permutation that never leaves its block; new_coords follows it.  And the point of it:
    the submanifold map built on the new order has fewer active (16-row group, tap) slots than on the old one.
    insmos_regroup_apply_voxels renames the voxeliser's arrays accordingly (-1 entries stay)."""
    from gpu_util import dev, hp, i32, lib, stream, ws
    L = lib()
    rng = np.random.default_rng(5)
    shape, B = (7, 64, 80), 2
    cells = shape[0] * shape[1] * shape[2]
    # surface-like occupancy: a few noisy sheets, so that neighbourhoods are neither empty nor full
    per = []
    for b in range(B):
        zz, yy, xx = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
        sheet = np.abs(zz - (2 + b + 2 * np.sin(yy / 9.0) + np.cos(xx / 7.0))) < 0.8
        sheet &= rng.random(sheet.shape) < 0.8
        per.append(np.flatnonzero(sheet.ravel()))
    coords = np.concatenate([np.concatenate([np.full((len(c), 1), b), np.stack(np.unravel_index(c, shape), 1)], 1)
                             for b, c in enumerate(per)], 0).astype(np.int32)
    keys = np.concatenate([b * cells + c for b, c in enumerate(per)]).astype(np.uint64)
    n = len(coords)
    assert n > 2 * block or block >= 4096
    shp = i32(shape)
    nw = int(L.insmos_rankmap_words(hp(shp), B))
    bits = torch.zeros(nw, dtype=torch.int64, device="cuda")
    incl = torch.zeros(nw // 4, dtype=torch.int32, device="cuda")
    w = ws(L.insmos_rankmap_ws_bytes(hp(shp), B))
    assert L.insmos_rankmap_from_keys(dev(keys.view(np.int64)).data_ptr(), n, hp(shp), B, bits.data_ptr(), incl.data_ptr(), w.data_ptr(),
                                      w.numel(), stream()) == 0
    cd = dev(coords)
    newc = torch.full((n, 4), -1, dtype=torch.int32, device="cuda")
    n2o = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    o2n = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    w2 = ws(L.insmos_regroup_ws_bytes(n))
    assert L.insmos_regroup_rows3d(cd.data_ptr(), n, bits.data_ptr(), hp(shp), 512, newc.data_ptr(), n2o.data_ptr(), None, w2.data_ptr(),
                                   w2.numel(), stream()) != 0
    if block == 1 << 30:   # whole windows: one stable sort
        assert L.insmos_regroup_rows3d_global(cd.data_ptr(), n, bits.data_ptr(), hp(shp), newc.data_ptr(), n2o.data_ptr(), o2n.data_ptr(),
                                              w2.data_ptr(), w2.numel(), stream()) == 0
    else:
        assert L.insmos_regroup_rows3d(cd.data_ptr(), n, bits.data_ptr(), hp(shp), block, newc.data_ptr(), n2o.data_ptr(), o2n.data_ptr(),
                                       w2.data_ptr(), w2.numel(), stream()) == 0
    torch.cuda.synchronize()
    n2o_h, newc_h = n2o.cpu().numpy().astype(np.int64), newc.cpu().numpy()
    np.testing.assert_array_equal(n2o_h[o2n.cpu().numpy()], np.arange(n))
    assert np.all(np.diff(newc_h[:, 0]) >= 0)          # window-major rows stay window-major
    np.testing.assert_array_equal(np.sort(n2o_h), np.arange(n))
    np.testing.assert_array_equal(n2o_h // block, np.arange(n) // block)
    np.testing.assert_array_equal(newc_h[n2o_h], coords)
    occ = set(keys.tolist())
    sig = np.zeros(n, np.int64)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = coords[:, 1:].astype(np.int64) + np.array([dz, dy, dx])
                ok = np.all((q >= 0) & (q < np.array(shape)), axis=1)
                kk = coords[:, 0].astype(np.int64) * cells + (q[:, 0] * shape[1] + q[:, 1]) * shape[2] + q[:, 2]
                hit = np.array([o and (int(v) in occ) for o, v in zip(ok, kk)])
                sig |= hit.astype(np.int64) << k
                k += 1
    old_of_new = np.empty(n, np.int64)
    old_of_new[n2o_h] = np.arange(n)
    for s0 in range(0, n, block):
        seg = old_of_new[s0:s0 + block]
        cb = coords[s0:s0 + block]
        par = (cb[:, 1] & 1) * 4 + (cb[:, 2] & 1) * 2 + (cb[:, 3] & 1)
        want = s0 + np.lexsort((np.arange(len(seg)), par, sig[s0:s0 + block], cb[:, 0]))   # window, signature, parity, old row
        np.testing.assert_array_equal(seg, want)

    if block == 4096:   # parity class above the signature (block_rows < 0)
        n2o_p = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        assert L.insmos_regroup_rows3d(cd.data_ptr(), n, bits.data_ptr(), hp(shp), -4096, newc.data_ptr(), n2o_p.data_ptr(), None,
                                       w2.data_ptr(), w2.numel(), stream()) == 0
        torch.cuda.synchronize()
        oon = np.empty(n, np.int64)
        oon[n2o_p.cpu().numpy().astype(np.int64)] = np.arange(n)
        for s0 in range(0, n, block):
            cb = coords[s0:s0 + block]
            par = (cb[:, 1] & 1) * 4 + (cb[:, 2] & 1) * 2 + (cb[:, 3] & 1)
            np.testing.assert_array_equal(oon[s0:s0 + block], s0 + np.lexsort((np.arange(len(cb)), sig[s0:s0 + block], par, cb[:, 0])))

    def active_slots(order):
        pres = np.stack([(sig[order] >> t) & 1 for t in range(27)]).astype(bool)
        pad = (-n) % 16
        pres = np.concatenate([pres, np.zeros((27, pad), bool)], 1).reshape(27, -1, 16)
        return int(pres.any(2).sum())
    assert active_slots(old_of_new) < 0.97 * active_slots(np.arange(n))
    # the voxeliser's arrays under the renaming
    num = rng.integers(1, 6, n).astype(np.int32)
    uperm = rng.permutation(n).astype(np.int32)
    uperm[rng.choice(n, 17, replace=False)] = -1
    pcid = rng.integers(-1, n, 3 * n).astype(np.int64)
    num_new = torch.zeros(n, dtype=torch.int32, device="cuda")
    up_d, pc_d = dev(uperm), dev(pcid)
    assert L.insmos_regroup_apply_voxels(n2o.data_ptr(), n, dev(num).data_ptr(), num_new.data_ptr(), up_d.data_ptr(), n, pc_d.data_ptr(),
                                         3 * n, stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(num_new.cpu().numpy()[n2o_h], num)
    np.testing.assert_array_equal(up_d.cpu().numpy(), np.where(uperm >= 0, n2o_h[np.maximum(uperm, 0)], -1))
    np.testing.assert_array_equal(pc_d.cpu().numpy(), np.where(pcid >= 0, n2o_h[np.maximum(pcid, 0)], -1))


def test_tables_derived_from_sparsely_stored_coarse_tables(window, engine):
    """A 4D table written with sparse stores (entries outside a 16-row group's active-tap mask stay unwritten) is a valid INPUT of
    the next finer level's derivation and of the first layer's occupancy cubes when the readers get its mask array
    (insmos_nbr81_from_coarse_rows_masked, insmos_const_conv125_cubes): same fine table / same masks / same first-layer output as
    from the fully written table.  The unwritten entries are prefilled with garbage row numbers here."""
    from gpu_util import dev, lib, stream
    L = lib()
    engine.const_input = True
    engine.prune_dead_rows = False
    try:
        engine.motionnet(dev(window))
    finally:
        engine.prune_dead_rows = True
    torch.cuda.synchronize()
    T = engine._me_tables
    n = [int(c.shape[0]) for c in T["coords"]]
    garbage = 0x3FFFFFF0
    for lv in (2, 1, 0):   # derive level lv from level lv + 1
        dense_c = T["nbr81"][lv + 1]
        nc, nf = n[lv + 1], n[lv]
        # the coarse table again, sparsely stored over garbage, from ITS coarse level (or as a masked copy at the top)
        cs = torch.full((81, nc), garbage, dtype=torch.int32, device="cuda")
        grp_mask = dense_c.mask16.view(torch.int32)   # (groups, 4)
        tap = torch.arange(81, device="cuda")
        bit = (grp_mask[:, (tap // 32).long()] >> (tap % 32).int()) & 1       # (groups, 81)
        live = bit.bool().repeat_interleave(16, dim=0)[:nc].t()               # (81, nc)
        cs[live] = dense_c.nbr[live]
        fine = torch.full((81, nf), -9, dtype=torch.int32, device="cuda")
        fmask = torch.zeros(((nf + 15) // 16, 4), dtype=torch.int32, device="cuda")
        args = (T["coords"][lv].data_ptr(), nf, 0, T["parent"][lv].data_ptr(), lv, cs.data_ptr(), dense_c.mask16.data_ptr(), nc,
                T["child_start"][lv].data_ptr(), T["child_mask"][lv].data_ptr(), fine.data_ptr(), fmask.data_ptr())
        assert L.insmos_nbr81_from_coarse_rows_masked(*args, 0, stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(fine, T["nbr81"][lv].nbr)
        assert torch.equal(fmask, T["nbr81"][lv].mask16.view(torch.int32))
        # ... and written sparsely itself: the same wherever its own mask has the tap, untouched elsewhere
        fine_s = torch.full((81, nf), -9, dtype=torch.int32, device="cuda")
        fmask_s = torch.zeros_like(fmask)
        args_s = args[:10] + (fine_s.data_ptr(), fmask_s.data_ptr())
        assert L.insmos_nbr81_from_coarse_rows_masked(*args_s, 1, stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(fmask_s, fmask)
        bit_f = (fmask[:, (tap // 32).long()] >> (tap % 32).int()) & 1
        live_f = bit_f.bool().repeat_interleave(16, dim=0)[:nf].t()
        assert torch.equal(fine_s[live_f], fine[live_f]) and bool((fine_s[~live_f] == -9).all())
        if lv == 0:   # the first layer from the sparsely stored level-1 table
            out_ref = engine._me_debug["cat8"][:, 8:16]
            out = torch.zeros((nf, 16), dtype=torch.float32, device="cuda")
            cubes = torch.empty(nc * 12, dtype=torch.int32, device="cuda")
            assert L.insmos_const_conv125_cubes(T["coords"][0].data_ptr(), nf, T["parent"][0].data_ptr(), 0, cs.data_ptr(),
                                                dense_c.mask16.data_ptr(), nc, T["child_start"][0].data_ptr(),
                                                T["child_mask"][0].data_ptr(), engine.w0_const.data_ptr(), engine.b0_const.data_ptr(),
                                                out.data_ptr() + 32, 16, 1, cubes.data_ptr(), stream()) == 0
            torch.cuda.synchronize()
            assert torch.equal(out[:, 8:16], out_ref)


@pytest.mark.parametrize("B", [1, 3])
def test_level_down_chain_equals_the_per_level_calls(window, engine, B):
    """insmos_level_down4d_chain (levels 1..3 and the time-slice starts in one chain of launches, counts on the device: one host wait
    for a single window's forward instead of four) against three insmos_level_down4d + four insmos_tslice_starts_batched calls:
    counts, keys, coordinates, parent / child_start / child_mask and the starts, every array below its level's count, bit for bit."""
    import ctypes
    from gpu_util import dev, lib, stream, ws
    from insmos_amd import _lib
    pts = dev(window)
    engine.const_input = True
    engine.motionnet(pts)
    torch.cuda.synchronize()
    keys0 = engine._me_tables["keys"][0].clone()
    if B > 1:   # a launch set's time field is t * B + b: spread the rows over B windows (order kept: b ascends with the row)
        k = keys0.cpu().numpy().view(np.uint64)
        t = (k >> np.uint64(48)).astype(np.int64) - 32768
        b = (np.arange(len(k)) * B // len(k)).astype(np.int64)
        tp = t * B + b
        k2 = (k & np.uint64((1 << 48) - 1)) | ((tp + 32768).astype(np.uint64) << np.uint64(48))
        k2.sort()
        keys0 = torch.from_numpy(k2.view(np.int64)).cuda()
    n0 = int(keys0.shape[0])
    L, st = lib(), stream()
    w = ws(L.insmos_level_down4d_ws_bytes(n0))
    counts = torch.zeros(8, dtype=torch.int32, device="cuda")
    ref, n, kin = [], [n0], keys0
    for l in (1, 2, 3):
        arrs = [torch.empty(n[-1], dtype=torch.int64, device="cuda"), torch.empty((n[-1], 4), dtype=torch.int32, device="cuda")] + \
               [torch.empty(n[-1], dtype=torch.int32, device="cuda") for _ in range(3)]
        _lib.check(L.insmos_level_down4d(kin.data_ptr(), n[-1], l, *[a.data_ptr() for a in arrs], counts.data_ptr(), w.data_ptr(),
                                         w.numel(), st), "insmos_level_down4d")
        nl = int(counts[0].item())
        ref.append(arrs)
        n.append(nl)
        kin = arrs[0][:nl]
    starts = torch.zeros((4, 16), dtype=torch.int32, device="cuda")
    level_keys = [keys0] + [r[0] for r in ref]
    for l in range(4):
        _lib.check(L.insmos_tslice_starts_batched(level_keys[l].data_ptr(), n[l], 16, B, starts[l].data_ptr(), st), "insmos_tslice_starts_batched")
    got = [[torch.full((n0,), -7, dtype=torch.int64, device="cuda"), torch.full((n0, 4), -7, dtype=torch.int32, device="cuda")] +
           [torch.full((n0,), -7, dtype=torch.int32, device="cuda") for _ in range(3)] for _ in range(3)]
    chain = torch.full((4 + 64,), -1, dtype=torch.int32, device="cuda")
    ptrs = [(ctypes.c_void_p * 3)(*[g[i].data_ptr() for g in got]) for i in range(5)]
    _lib.check(L.insmos_level_down4d_chain(keys0.data_ptr(), n0, 3, B, *ptrs, chain.data_ptr(), w.data_ptr(), w.numel(), st),
               "insmos_level_down4d_chain")
    torch.cuda.synchronize()
    ch = chain.cpu().numpy()
    assert ch[:3].tolist() == n[1:], (ch[:3], n)
    np.testing.assert_array_equal(ch[4:].reshape(4, 16), starts.cpu().numpy())
    for l in range(3):
        for i, rows in enumerate((n[l + 1], n[l + 1], n[l], n[l + 1], n[l + 1])):   # keys, coords, parent (per fine row), child_start, child_mask
            assert torch.equal(got[l][i][:rows], ref[l][i][:rows]), (l, i)
        assert int((got[l][0][n[l + 1]:] != -7).sum()) == 0                        # nothing written past the count
    assert L.insmos_level_down4d_chain(keys0.data_ptr(), 1 << 24, 3, B, *ptrs, chain.data_ptr(), w.data_ptr(), w.numel(), st) != 0
