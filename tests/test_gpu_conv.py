"""-m gpu: the MFMA sparse-conv kernel against the oracle's C conv (same fp32 inputs; tolerance covers
summation order only) and against torch conv2d / conv_transpose2d for the dense BEV use."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
TOL = dict(rtol=2e-5, atol=2e-5)


def _rand_nbr(rng, K, n_out, n_in, density):
    nbr = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    nbr[rng.uniform(size=(K, n_out)) > density] = -1
    return nbr


@pytest.mark.parametrize("cin,cout,K,n_out,density", [
    (8, 8, 81, 1000, 0.2), (1, 8, 125, 777, 0.15), (16, 8, 81, 513, 0.2), (24, 16, 81, 300, 0.3),
    (48, 32, 81, 260, 0.3), (32, 32, 8, 4000, 0.13), (7, 16, 27, 2000, 0.3), (131, 128, 27, 500, 0.4),
    (67, 64, 27, 300, 0.4), (35, 32, 27, 300, 0.4), (19, 16, 27, 129, 0.4), (256, 128, 27, 200, 0.5),
    (128, 128, 3, 64, 0.7), (16, 3, 1, 1000, 1.0), (8, 3, 1, 70, 1.0), (64, 64, 27, 70000, 0.25),
    # >= 2048 row groups, one or two channel tiles
    (8, 8, 81, 40000, 0.2), (16, 8, 81, 36000, 0.2), (16, 16, 27, 40000, 0.3), (8, 16, 27, 33000, 0.3),
    (32, 16, 27, 34000, 0.3), (16, 32, 27, 35011, 0.3), (8, 8, 8, 33000, 0.13)])
def test_sparse_conv_matches_oracle(cin, cout, K, n_out, density):
    from gpu_util import dev, pack_layer, run_conv
    rng = np.random.default_rng(cin * 1000 + cout)
    identity = (K == 1)
    n_in = n_out if identity else max(n_out // 2, 50)
    cin_pad = 4 if cin <= 4 else 8 if cin <= 8 else (cin + 15) // 16 * 16
    x = np.zeros((n_in, cin_pad), np.float32)
    x[:, :cin] = rng.normal(size=(n_in, cin))
    x[:, cin:] = 7.0  # padded columns must be neutral (their taps are zero)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * density + 1)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    nbr = None if identity else _rand_nbr(rng, K, n_out, n_in, density)
    layer = pack_layer(taps, bias, cin_pad, cout)
    y = run_conv(layer, dev(x), None if identity else dev(nbr), n_out).cpu().numpy()
    ref = R.sparse_conv(x[:, :cin], nbr, taps) + bias
    np.testing.assert_allclose(y[:, :cout], ref, **TOL)
    if not identity:  # same result when the active-tap bitmasks drive the tap loop
        from gpu_util import tap_masks
        ym = run_conv(layer, dev(x), dev(nbr), n_out, mask=dev(tap_masks(nbr).view(np.int32))).cpu().numpy()
        # not bitwise: tap-split tiles hand taps to waves by rank in the ACTIVE set, so the summation grouping differs
        np.testing.assert_allclose(ym, y, rtol=1e-5, atol=1e-5)


def test_clustered_occupancy_with_masks():
    """Spatially coherent tables (whole 16-row groups / whole tiles without a tap) exercise the skip paths."""
    from gpu_util import dev, pack_layer, run_conv, tap_masks
    rng = np.random.default_rng(11)
    for cin, cout, K, n in ((8, 8, 81, 5000), (16, 16, 27, 3000), (48, 32, 81, 2000), (128, 64, 27, 1500), (32, 16, 27, 900)):
        n_in = n
        nbr = rng.integers(0, n_in, size=(K, n)).astype(np.int32)
        grp = rng.uniform(size=(K, (n + 15) // 16)) < 0.45     # whole groups empty
        tile = rng.uniform(size=(K, (n + 63) // 64)) < 0.3     # whole 64-row tiles empty
        keep = np.repeat(grp, 16, axis=1)[:, :n] & ~np.repeat(tile, 64, axis=1)[:, :n] & (rng.uniform(size=(K, n)) < 0.5)
        nbr[~keep] = -1
        x = rng.normal(size=(n_in, cin)).astype(np.float32)
        taps = (rng.normal(size=(K, cin, cout)) * 0.1).astype(np.float32)
        bias = rng.normal(size=cout).astype(np.float32)
        layer = pack_layer(taps, bias, cin, cout)
        y = run_conv(layer, dev(x), dev(nbr), n, mask=dev(tap_masks(nbr).view(np.int32))).cpu().numpy()
        np.testing.assert_allclose(y, R.sparse_conv(x, nbr, taps) + bias, **TOL)


def test_epilogue_variants_and_column_slices():
    from gpu_util import dev, pack_layer, run_conv
    rng = np.random.default_rng(5)
    n, cin, cout, K = 700, 32, 16, 27
    x = rng.normal(size=(n, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    nbr = _rand_nbr(rng, K, n, n, 0.3)
    layer = pack_layer(taps, bias, cin, cout)
    base = R.sparse_conv(x, nbr, taps) + bias
    res = rng.normal(size=(n, cout)).astype(np.float32)
    y = run_conv(layer, dev(x), dev(nbr), n, res=dev(res), res_mode=1, relu_post=1).cpu().numpy()
    np.testing.assert_allclose(y, np.maximum(base + res, 0), **TOL)
    res2 = rng.normal(size=(n, 2 * cout)).astype(np.float32)
    y = run_conv(layer, dev(x), dev(nbr), n, res=dev(res2), res_mode=2, relu_pre=1).cpu().numpy()
    np.testing.assert_allclose(y, np.maximum(base, 0) + res2.reshape(n, cout, 2).sum(2), **TOL)
    # write into a column slice of a wider row; the other columns must stay untouched
    out = torch.full((n, 40), -5.0, device="cuda:0")
    run_conv(layer, dev(x), dev(nbr), n, ld_out=40, col_out=20, relu_post=1, out=out)
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[:, 20:36], np.maximum(base, 0), **TOL)
    assert np.all(o[:, :20] == -5.0) and np.all(o[:, 36:] == -5.0)


def test_dense_bev_convs_match_torch():
    """3x3 pad-1 Conv2d, ConvTranspose2d(2,2) and the 1x1 heads through the sparse-conv kernel (NHWC)."""
    from gpu_util import dev, pack_layer, run_conv, lib, stream
    from insmos_amd import params as P, _lib
    rng = np.random.default_rng(9)
    H, W, ci, co = 13, 17, 32, 48
    x = rng.normal(size=(1, ci, H, W)).astype(np.float32)
    w = (rng.normal(size=(co, ci, 3, 3)) * 0.1).astype(np.float32)
    nbr = torch.empty((9, H * W), dtype=torch.int32, device="cuda:0")
    _lib.check(lib().insmos_dense_nbr2d(H, W, nbr.data_ptr(), stream()), "dense_nbr2d")
    layer = pack_layer(P.conv2d_weight_to_taps(w), None, ci, co)
    xs = np.ascontiguousarray(x[0].transpose(1, 2, 0).reshape(H * W, ci))
    y = run_conv(layer, dev(xs), nbr, H * W).cpu().numpy()
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1)[0].permute(1, 2, 0).reshape(H * W, co).numpy()
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)
    wt = (rng.normal(size=(ci, co, 2, 2)) * 0.1).astype(np.float32)
    tt = P.convT2d_weight_to_taps(wt)
    wide = np.ascontiguousarray(tt.transpose(1, 0, 2).reshape(1, ci, 4 * co))
    lt = pack_layer(wide, None, ci, 4 * co)
    yt = run_conv(lt, dev(xs), None, H * W).cpu().numpy().reshape(H, W, 2, 2, co)
    reft = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wt), stride=2)[0].permute(1, 2, 0).numpy()
    got = yt.transpose(0, 2, 1, 3, 4).reshape(2 * H, 2 * W, co)
    np.testing.assert_allclose(got, reft, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,H,W,ci,co", [(1, 13, 17, 32, 128), (2, 19, 35, 64, 64), (3, 125, 150, 128, 128), (1, 8, 16, 256, 128)])
def test_bev_conv3x3_kernel_matches_torch_conv2d(B, H, W, ci, co):
    """insmos_bev_conv3x3 (csrc/bev.hip: LDS-tiled implicit GEMM, base_bev_backbone.py:33-61) against torch conv2d + bias +
    ReLU on B stacked NHWC images -- patch borders, image borders (zero padding), partial patches, both patch heights --
    and against the generic kernel over the dense 9-tap table (same function, another summation order)."""
    from gpu_util import dev, pack_layer, run_conv, lib, stream
    from insmos_amd import params as P, _lib
    rng = np.random.default_rng(B * 1000 + H)
    x = rng.normal(size=(B, ci, H, W)).astype(np.float32)
    w = (rng.normal(size=(co, ci, 3, 3)) * (1.0 / np.sqrt(9 * ci))).astype(np.float32)
    bias = rng.normal(size=co).astype(np.float32)
    layer = pack_layer(P.conv2d_weight_to_taps(w), bias, ci, co)
    ld = ci + 16                                                      # a row pitch wider than the channel count
    xs = torch.zeros((B * H * W, ld), device="cuda:0")
    xs[:, :ci] = dev(np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(B * H * W, ci)))
    ref = F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(bias), padding=1))
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, co).numpy()
    for relu in (1, 0):
        out = torch.full((B * H * W, co + 4), 9.0, device="cuda:0")
        _lib.check(lib().insmos_bev_conv3x3(xs.data_ptr(), B, H, W, ld, ci, layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(),
                                            co + 4, co, relu, stream()), "insmos_bev_conv3x3")
        torch.cuda.synchronize()
        y = out.cpu().numpy()
        assert (y[:, co:] == 9.0).all()                                # nothing written beyond the channel count
        if relu:
            np.testing.assert_allclose(y[:, :co], ref, rtol=1e-4, atol=1e-4)
        else:
            lin = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(bias), padding=1)
            np.testing.assert_allclose(y[:, :co], lin.permute(0, 2, 3, 1).reshape(B * H * W, co).numpy(), rtol=1e-4, atol=1e-4)
            assert (y[:, :co] < 0).any()
    nbr = torch.empty((9, B * H * W), dtype=torch.int32, device="cuda:0")
    _lib.check(lib().insmos_dense_nbr2d_b(H, W, B, nbr.data_ptr(), stream()), "dense_nbr2d_b")
    yt = run_conv(layer, xs, nbr, B * H * W, relu_post=1).cpu().numpy()
    np.testing.assert_allclose(yt, ref, rtol=1e-4, atol=1e-4)
    out = torch.zeros((B * H * W, co), device="cuda:0")
    _lib.check(lib().insmos_bev_conv3x3(xs.data_ptr(), B, H, W, ld, ci, layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(),
                                        co, co, 1, stream()), "insmos_bev_conv3x3")
    assert float((out.cpu() - torch.from_numpy(yt)).abs().max()) < 1e-4
    # the patch a site falls into does not matter: image 0 alone == image 0 of the stack
    if B > 1:
        one = torch.zeros((H * W, co), device="cuda:0")
        _lib.check(lib().insmos_bev_conv3x3(xs.data_ptr(), 1, H, W, ld, ci, layer.w.data_ptr(), layer.b.data_ptr(), one.data_ptr(),
                                            co, co, 1, stream()), "insmos_bev_conv3x3")
        assert torch.equal(one, out[:H * W])


@pytest.mark.parametrize("cin,cout,K,res_mode", [(128, 128, 27, 0), (64, 64, 27, 1), (256, 128, 27, 2), (64, 128, 27, 0), (128, 64, 81, 1)])
def test_wide_masked_layers_vs_oracle_and_layout_independence(cin, cout, K, res_mode):
    """The wide masked layers (split tiles: four waves share a 16-row group) against the oracle's conv with every epilogue,
    and -- the property batching rests on -- a row's bits depend neither on the rows sharing its tile (taps are handed to the
    waves by tap index, not by rank in the tile's active set), nor on the launch size, nor on where in the row list it sits."""
    from gpu_util import dev, lib, pack_layer, stream, tap_masks
    from insmos_amd import _lib
    rng = np.random.default_rng(K * 1000 + cin + cout)
    n_out, n_in = 33000, 9000
    nbr = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    grp = rng.uniform(size=(K, (n_out + 15) // 16)) < 0.6            # spatially coherent occupancy, ~70 % fill inside a group
    nbr[~(np.repeat(grp, 16, axis=1)[:, :n_out] & (rng.uniform(size=(K, n_out)) < 0.7))] = -1
    nbr[:, 5000:5100] = -1                                            # rows without any neighbour: bias only
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.4)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    layer = pack_layer(taps, bias, cin, cout)
    ld_res = cout if res_mode == 1 else 2 * cout
    res = rng.normal(size=(n_out, ld_res)).astype(np.float32) if res_mode else None
    xd, resd = dev(x), (dev(res) if res_mode else None)

    def run(table, row0=0, res_t=resd):
        n = table.shape[1]
        nd, md = dev(table), dev(tap_masks(table).view(np.int32))
        out = torch.zeros((n, cout), device="cuda:0")
        _lib.check(lib().insmos_sparse_conv_rows(xd.data_ptr(), n_in, cin, layer.cin, nd.data_ptr(), md.data_ptr(), K, n, row0,
                                                 layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(), cout, layer.cout,
                                                 res_t.data_ptr() if res_t is not None else None, ld_res if res_mode else 0, res_mode,
                                                 1 if res_mode == 2 else 0, 1, stream()), "insmos_sparse_conv_rows")
        torch.cuda.synchronize()
        return out

    full = run(nbr)
    ref = R.sparse_conv(x, nbr, taps) + bias
    if res_mode == 2:
        ref = np.maximum(ref, 0.0) + res[:, 0::2] + res[:, 1::2]      # relu_pre, then the channel-pair residual
    elif res_mode == 1:
        ref = ref + res
    ref = np.maximum(ref, 0.0)
    np.testing.assert_allclose(full.cpu().numpy(), ref, **TOL)
    # half-width tiles (two blocks per tile, half of the channel tiles each: what the launcher picks for few row groups -- single
    # windows, and since round 5 the C = 128 level-4 layers of a launch set) against full-width ones: the same bits
    if cout >= 64 and cin % 64 == 0:
        try:
            assert lib().insmos_debug_conv_split_half(0, 0) == 0
            never = run(nbr)
            assert lib().insmos_debug_conv_split_half(1 << 30, 1 << 30) == 0
            always = run(nbr)
        finally:
            lib().insmos_debug_conv_split_half(-1, -1)
        assert torch.equal(never, always) and torch.equal(never, full)
    tail_a = run(nbr, row0=16 * 1000)                                 # a row suffix: 17000 rows, then 2600
    tail_b = run(nbr, row0=16 * 1900)
    assert torch.equal(full[16000:], tail_a[16000:]) and torch.equal(full[30400:], tail_b[30400:])
    # the same rows, 5 places further down a longer list: every row gets other tile mates and other active-tap sets
    shifted = np.concatenate([nbr[:, 777:782], nbr], axis=1)
    res_s = dev(np.concatenate([res[777:782], res])) if res_mode else None
    out_s = run(shifted, res_t=res_s)
    assert torch.equal(out_s[5:], full)


@pytest.mark.parametrize("cin,cout,K,n_out,res_mode", [
    (128, 128, 27, 3000, 1), (64, 64, 27, 5001, 0), (256, 128, 27, 2000, 2), (128, 64, 27, 1000, 0), (64, 128, 27, 333, 0),
    (128, 128, 3, 64, 0), (256, 64, 27, 40, 1), (64, 64, 27, 70000, 1), (128, 128, 9, 17, 0)])
def test_wide_staged_kernel_is_bitwise_the_split_tiles(cin, cout, K, n_out, res_mode):
    """csrc/spconv_wide.hip (32-row tiles, whole-row gathers staged once per tap in LDS, waves split by output-channel tile, four
    chunk-class accumulators per wave) against the chunk-split 16-row tiles of spconv.hip: the SAME bits -- with and without
    active-tap masks, a table with garbage outside its masks (sparse stores), every epilogue, a row suffix, row counts that are not
    multiples of 32, an input view that is a column slice of wider rows, tap lists of every length."""
    from gpu_util import dev, lib, pack_layer, stream, tap_masks
    from insmos_amd import _lib
    rng = np.random.default_rng(K * 100 + cin + cout + 5)
    n_in = max(n_out // 2, 40)
    nbr = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    ngrp = (n_out + 15) // 16
    n_act = rng.integers(0, K + 1, size=ngrp)                         # active taps per 16-row group: every count 0..K
    grp = np.zeros((K, ngrp), bool)
    for gi in range(ngrp):
        grp[rng.permutation(K)[:n_act[gi]], gi] = True
    slot_on = np.repeat(grp, 16, axis=1)[:, :n_out]
    nbr[~(slot_on & (rng.uniform(size=(K, n_out)) < 0.7))] = -1
    masks = tap_masks(nbr)
    sparse = nbr.copy()
    sparse[~slot_on] = rng.integers(-5, n_in + 1000, size=int((~slot_on).sum()))
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.3)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    layer = pack_layer(taps, bias, cin, cout)
    ld_res = cout if res_mode == 1 else 2 * cout
    res = dev(rng.normal(size=(n_out, ld_res)).astype(np.float32)) if res_mode else None
    xd, nd, sd, md = dev(x), dev(nbr), dev(sparse), dev(masks.view(np.int32))
    wide = torch.zeros((n_in, cin + 32), device="cuda:0")             # the same rows as a column slice of wider rows
    wide[:, 16:16 + cin] = xd
    L = lib()
    assert L.insmos_conv_tap_classes(K, layer.cin, layer.cout, 1) == (0 if K >= 3 else 1)   # chunk-split shapes (K >= 3)

    def run(on, table, masked, row0=0, xin=xd, ld=cin, col=0):
        assert L.insmos_debug_conv_wide(on) == 0
        try:
            out = torch.full((n_out, cout), -7.0, device="cuda:0")
            _lib.check(L.insmos_sparse_conv_rows(xin.data_ptr() + 4 * col, n_in, ld, layer.cin, table.data_ptr(),
                                                 md.data_ptr() if masked else None, K, n_out, row0, layer.w.data_ptr(),
                                                 layer.b.data_ptr(), out.data_ptr(), cout, layer.cout,
                                                 res.data_ptr() if res is not None else None, ld_res if res_mode else 0, res_mode,
                                                 1 if res_mode == 2 else 0, 1, stream()), "insmos_sparse_conv_rows")
            torch.cuda.synchronize()
            return out
        finally:
            L.insmos_debug_conv_wide(-1)

    ref = R.sparse_conv(x, nbr, taps) + bias
    if res_mode == 2:
        ref = np.maximum(ref, 0.0) + res.cpu().numpy()[:, 0::2] + res.cpu().numpy()[:, 1::2]
    elif res_mode == 1:
        ref = ref + res.cpu().numpy()
    ref = np.maximum(ref, 0.0)
    for masked in (True, False):
        a, b = run(1, nd, masked), run(0, nd, masked)
        assert torch.equal(a, b)
        np.testing.assert_allclose(a.cpu().numpy(), ref, **TOL)
    assert torch.equal(run(1, sd, True), run(0, nd, True))            # entries outside the masks are never used
    r0 = 16 * (n_out // 48)
    assert torch.equal(run(1, nd, True, r0)[r0:], run(0, nd, True, r0)[r0:])
    assert torch.equal(run(1, nd, True, xin=wide, ld=cin + 32, col=16), run(0, nd, True))


@pytest.mark.parametrize("cin,cout,K,n_out,res_mode", [
    (48, 32, 81, 1000, 1), (32, 32, 81, 3001, 0), (16, 32, 81, 777, 2), (32, 16, 81, 2049, 1), (16, 16, 81, 5000, 0),
    (8, 8, 81, 4000, 1), (8, 16, 81, 300, 0), (16, 8, 81, 513, 0), (16, 16, 27, 700, 2), (32, 32, 27, 1500, 0), (48, 32, 81, 100, 0),
    (32, 32, 81, 40000, 1), (32, 16, 81, 130, 0), (32, 8, 81, 5000, 2)])
def test_tap_compacted_kernel_is_bitwise_the_tile_kernels(cin, cout, K, n_out, res_mode):
    """csrc/spconv_tapc.hip: the rows of a 128-row block that HAVE a tap packed into dense groups of 16 per tap, accumulators parked
    in LDS between taps -- against the 16-row tiles of insmos_sparse_conv_rows: the SAME bits for the tap-split shapes (four class
    chains) and the one-chain shapes, every epilogue, a row suffix, row counts that are not multiples of 128, blocks in which a class
    has no pair at all; the item table is what the oracle's compaction of the same table gives; a class count that is not the
    layer's summation order is refused."""
    from gpu_util import dev, lib, pack_layer, stream, tap_masks
    from insmos_amd import _lib
    rng = np.random.default_rng(K * 100 + cin + cout + 11)
    n_in = max(n_out // 2, 40)
    nbr = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    # clustered occupancy: per 64-row stretch a random set of live taps at a random density; a few stretches with no tap at all
    nstr = (n_out + 63) // 64
    live = rng.uniform(size=(K, nstr)) < rng.uniform(0.05, 0.9, size=(1, nstr))
    live[:, rng.integers(0, nstr, size=max(1, nstr // 9))] = False
    dens = rng.uniform(0.05, 1.0, size=(K, nstr))
    on = np.repeat(live, 64, axis=1)[:, :n_out] & (rng.uniform(size=(K, n_out)) < np.repeat(dens, 64, axis=1)[:, :n_out])
    nbr[~on] = -1
    masks = tap_masks(nbr)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.3)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    layer = pack_layer(taps, bias, cin, cout)
    ld_res = cout if res_mode == 1 else 2 * cout
    res = dev(rng.normal(size=(n_out, ld_res)).astype(np.float32)) if res_mode else None
    xd, nd, md = dev(x), dev(nbr), dev(masks.view(np.int32))
    L = lib()
    ncls = L.insmos_conv_tap_classes(K, layer.cin, layer.cout, 1)
    assert ncls == (4 if K * max(layer.cin // 16, 1) * ((layer.cout + 15) // 16) >= 100 else 1)
    nblk = L.insmos_tapc_blocks(n_out)
    assert nblk == (n_out + 127) // 128
    words = L.insmos_tapc_words(K, n_out, ncls)
    tc = torch.zeros(words, dtype=torch.int32, device="cuda:0")
    ni = torch.zeros(nblk * ncls, dtype=torch.int32, device="cuda:0")
    _lib.check(L.insmos_tapc_build(nd.data_ptr(), K, n_out, 0, ncls, tc.data_ptr(), ni.data_ptr(), stream()), "insmos_tapc_build")
    torch.cuda.synchronize()
    # the item table against a numpy compaction of the same table
    cap = words // (nblk * ncls * 16)
    tcn = tc.cpu().numpy().view(np.uint32).reshape(nblk, ncls, cap, 16)
    nin = ni.cpu().numpy().reshape(nblk, ncls)
    for b in rng.choice(nblk, size=min(nblk, 6), replace=False):
        for c in range(ncls):
            items = []
            for k in range(c, K, ncls):
                rows = np.nonzero(nbr[k, b * 128:(b + 1) * 128] >= 0)[0]
                for g0 in range(0, len(rows), 16):
                    grp = rows[g0:g0 + 16]
                    ent = [(int(nbr[k, b * 128 + r]), int(r)) for r in grp] + [(0x7FFFFF, 128)] * (16 - len(grp))
                    items.append([(e | (r << 23) | ((((k >> j) & 1) << 31) if j < 7 else 0)) for j, (e, r) in enumerate(ent)])
            while len(items) % 3:
                items.append([0x7FFFFF | (128 << 23)] * 16)
            assert nin[b, c] == len(items)
            assert np.array_equal(tcn[b, c, :len(items)], np.array(items, np.uint32).reshape(len(items), 16))

    def run(tapc, row0=0):
        out = torch.full((n_out, cout), -7.0, device="cuda:0")
        common = (layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(), cout, layer.cout, res.data_ptr() if res is not None else None,
                  ld_res if res_mode else 0, res_mode, 1 if res_mode == 2 else 0, 1, stream())
        if tapc:
            _lib.check(L.insmos_sparse_conv_tapc_rows(xd.data_ptr(), n_in, cin, layer.cin, tc.data_ptr(), ni.data_ptr(), ncls, K, n_out, row0,
                                                      *common), "insmos_sparse_conv_tapc_rows")
        else:
            _lib.check(L.insmos_sparse_conv_rows(xd.data_ptr(), n_in, cin, layer.cin, nd.data_ptr(), md.data_ptr(), K, n_out, row0, *common),
                       "insmos_sparse_conv_rows")
        torch.cuda.synchronize()
        return out

    ref = R.sparse_conv(x, nbr, taps) + bias
    if res_mode == 2:
        ref = np.maximum(ref, 0.0) + res.cpu().numpy()[:, 0::2] + res.cpu().numpy()[:, 1::2]
    elif res_mode == 1:
        ref = ref + res.cpu().numpy()
    ref = np.maximum(ref, 0.0)
    a, b = run(True), run(False)
    assert torch.equal(a, b)
    np.testing.assert_allclose(a.cpu().numpy(), ref, **TOL)
    # a SPARSE table (entries of (16-row group, tap) pairs outside the group's mask are unwritten: garbage here) through the masked
    # builder: the same item table
    group_has = ((masks.reshape(-1, 4)[:, np.arange(K) >> 5] >> (np.arange(K) & 31)) & 1).astype(bool).T    # (K, groups)
    garbage = nbr.copy()
    off_mask = ~np.repeat(group_has, 16, axis=1)[:, :n_out]
    assert not (off_mask & (nbr >= 0)).any()
    garbage[off_mask] = rng.integers(0, n_in, size=int(off_mask.sum()))
    tc2 = torch.zeros(words, dtype=torch.int32, device="cuda:0")
    ni2 = torch.zeros(nblk * ncls, dtype=torch.int32, device="cuda:0")
    gd = dev(garbage)
    _lib.check(L.insmos_tapc_build_masked(gd.data_ptr(), md.data_ptr(), K, n_out, 0, ncls, tc2.data_ptr(), ni2.data_ptr(), stream()),
               "insmos_tapc_build_masked")
    torch.cuda.synchronize()
    assert torch.equal(ni2, ni)
    t1 = tc.cpu().numpy().reshape(nblk, ncls, cap, 16)
    t2 = tc2.cpu().numpy().reshape(nblk, ncls, cap, 16)
    for b_ in range(nblk):
        for c in range(ncls):
            assert np.array_equal(t1[b_, c, :nin[b_, c]], t2[b_, c, :nin[b_, c]])
    if cin == 32:                                                      # both gather forms of the Cin = 32 kernels (read per call)
        import os
        for form in ("0", "2"):
            os.environ["INSMOS_TAPC_ROW32"] = form
            try:
                assert torch.equal(run(True), b), form
            finally:
                os.environ.pop("INSMOS_TAPC_ROW32", None)
    r0 = 16 * (n_out // 37)                                            # a row suffix that starts inside a 128-row block
    ar, br = run(True, r0), run(False, r0)
    assert torch.equal(ar[r0:], br[r0:]) and bool((ar[:r0] == -7.0).all())   # rows below the start stay untouched
    tc2, ni2 = torch.full_like(tc, -1), torch.full_like(ni, -1)        # an item table built from that row on serves the suffix
    _lib.check(L.insmos_tapc_build(nd.data_ptr(), K, n_out, r0, ncls, tc2.data_ptr(), ni2.data_ptr(), stream()), "insmos_tapc_build")
    b0 = r0 // 128
    assert torch.equal(ni2[b0 * ncls:], ni[b0 * ncls:]) and bool((ni2[:b0 * ncls] == -1).all())
    tc, ni = tc2, ni2
    assert torch.equal(run(True, r0)[r0:], br[r0:])
    wrong = 1 if ncls == 4 else 4                                      # the class count is the summation order: not the caller's choice
    out = torch.zeros((n_out, cout), device="cuda:0")
    assert L.insmos_sparse_conv_tapc_rows(xd.data_ptr(), n_in, cin, layer.cin, tc.data_ptr(), ni.data_ptr(), wrong, K, n_out, 0,
                                          layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(), cout, layer.cout, None, 0, 0, 0, 1,
                                          stream()) == -1


@pytest.mark.parametrize("cin,cout,K,n_out,res_mode", [
    (8, 8, 81, 40003, 1), (8, 16, 81, 5000, 0), (16, 16, 81, 33333, 1), (16, 8, 81, 2049, 0), (16, 16, 27, 40000, 2),
    (8, 16, 27, 777, 0), (16, 32, 27, 3000, 1), (8, 8, 8, 33000, 0), (16, 16, 8, 1000, 0), (16, 16, 3, 17, 0), (8, 32, 27, 4100, 0)])
def test_quad_index_kernel_is_bitwise_the_generic_one(cin, cout, K, n_out, res_mode):
    """Single-chunk layers run on k_sparse_conv_q (index loads of four taps at a time, ds_bpermute to the lane groups): the
    same bits as the generic kernel with and without active-tap masks, for every epilogue, for a row suffix, and for tap lists
    of every length modulo 8 (the quad pipeline's tails)."""
    from gpu_util import dev, lib, pack_layer, stream, tap_masks
    from insmos_amd import _lib
    rng = np.random.default_rng(K * 100 + cin + cout)
    n_in = max(n_out // 2, 40)
    nbr = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    ngrp = (n_out + 15) // 16
    n_act = rng.integers(0, K + 1, size=ngrp)                         # active taps per 16-row group: every count 0..K
    grp = np.zeros((K, ngrp), bool)
    for gi in range(ngrp):
        grp[rng.permutation(K)[:n_act[gi]], gi] = True
    nbr[~(np.repeat(grp, 16, axis=1)[:, :n_out] & (rng.uniform(size=(K, n_out)) < 0.7))] = -1
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.3)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    layer = pack_layer(taps, bias, cin, cout)
    ld_res = cout if res_mode == 1 else 2 * cout
    res = dev(rng.normal(size=(n_out, ld_res)).astype(np.float32)) if res_mode else None
    xd, nd, md = dev(x), dev(nbr), dev(tap_masks(nbr).view(np.int32))

    def run(quad, masked, row0=0):
        lib().insmos_debug_conv_quad(quad)
        lib().insmos_debug_conv_rowlane(0, 0)   # (these shapes would otherwise take the row-per-lane kernel: tested below)
        try:
            out = torch.full((n_out, cout), -7.0, device="cuda:0")
            _lib.check(lib().insmos_sparse_conv_rows(xd.data_ptr(), n_in, cin, layer.cin, nd.data_ptr(), md.data_ptr() if masked else None,
                                                     K, n_out, row0, layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(), cout,
                                                     layer.cout, res.data_ptr() if res is not None else None, ld_res if res_mode else 0,
                                                     res_mode, 1 if res_mode == 2 else 0, 1, stream()), "insmos_sparse_conv_rows")
            torch.cuda.synchronize()
            return out
        finally:
            lib().insmos_debug_conv_quad(1)
            lib().insmos_debug_conv_rowlane(-1, 0)

    ref = R.sparse_conv(x, nbr, taps) + bias
    if res_mode == 2:
        ref = np.maximum(ref, 0.0) + res.cpu().numpy()[:, 0::2] + res.cpu().numpy()[:, 1::2]
    elif res_mode == 1:
        ref = ref + res.cpu().numpy()
    ref = np.maximum(ref, 0.0)
    for masked in (True, False):
        a, b = run(1, masked), run(0, masked)
        assert torch.equal(a, b)
        np.testing.assert_allclose(a.cpu().numpy(), ref, **TOL)
    r0 = 16 * (n_out // 48)
    assert torch.equal(run(1, True, r0)[r0:], run(0, True, r0)[r0:])


@pytest.mark.parametrize("cout,K,n_out,res_mode", [
    (16, 27, 40003, 0), (32, 27, 33333, 1), (32, 8, 9000, 0), (16, 81, 20011, 1), (32, 81, 12000, 2), (64, 27, 5000, 0), (16, 8, 777, 2),
    (32, 27, 17, 0), (16, 1, 100, 0), (32, 3, 4100, 1), (64, 81, 3000, 0)])
def test_whole_row_gather_kernel_is_bitwise_the_generic_one(cout, K, n_out, res_mode):
    """The Cin = 32 layers whose input rows are whole 128-byte lines run on csrc/spconv_row32.hip (two whole-row gathers per tap, rows
    turned into B fragments through a wave-private LDS buffer, three-stage pipeline over taps): the SAME bits as the generic tiles --
    unsplit and tap-split, Cout 16 / 32 / 64, with and without active-tap masks, every epilogue, a row suffix, tap lists of every
    length (the pipeline's prologue / tail), a table with garbage outside its masks (sparse stores) -- and the generic kernel is what
    runs when the rows are NOT lines (a column slice of wider rows, a base off the 128-byte grid)."""
    from gpu_util import dev, lib, pack_layer, stream, tap_masks
    from insmos_amd import _lib
    cin = 32
    rng = np.random.default_rng(K * 100 + cout + 31)
    n_in = max(n_out // 2, 40)
    nbr = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    ngrp = (n_out + 15) // 16
    n_act = rng.integers(0, K + 1, size=ngrp)                         # active taps per 16-row group: every count 0..K
    grp = np.zeros((K, ngrp), bool)
    for gi in range(ngrp):
        grp[rng.permutation(K)[:n_act[gi]], gi] = True
    slot_on = np.repeat(grp, 16, axis=1)[:, :n_out]
    nbr[~(slot_on & (rng.uniform(size=(K, n_out)) < 0.7))] = -1
    masks = tap_masks(nbr)
    sparse = nbr.copy()                                                # what a sparse-storing table builder leaves outside the masks
    sparse[~slot_on] = rng.integers(-5, n_in + 1000, size=int((~slot_on).sum()))
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.3)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    layer = pack_layer(taps, bias, cin, cout)
    ld_res = cout if res_mode == 1 else 2 * cout
    res = dev(rng.normal(size=(n_out, ld_res)).astype(np.float32)) if res_mode else None
    xd, nd, sd, md = dev(x), dev(nbr), dev(sparse), dev(masks.view(np.int32))
    wide = torch.zeros((n_in, 48), device="cuda:0")                    # the same rows as a column slice of 192-byte rows: not lines
    wide[:, 16:48] = xd

    def run(on, table, masked, row0=0, xin=xd, ld=cin, col=0):
        assert lib().insmos_debug_conv_row32(on) == 0
        try:
            out = torch.full((n_out, cout), -7.0, device="cuda:0")
            _lib.check(lib().insmos_sparse_conv_rows(xin.data_ptr() + 4 * col, n_in, ld, layer.cin, table.data_ptr(),
                                                     md.data_ptr() if masked else None, K, n_out, row0, layer.w.data_ptr(),
                                                     layer.b.data_ptr(), out.data_ptr(), cout, layer.cout,
                                                     res.data_ptr() if res is not None else None, ld_res if res_mode else 0, res_mode,
                                                     1 if res_mode == 2 else 0, 1, stream()), "insmos_sparse_conv_rows")
            torch.cuda.synchronize()
            return out
        finally:
            lib().insmos_debug_conv_row32(-1)

    ref = R.sparse_conv(x, nbr, taps) + bias
    if res_mode == 2:
        ref = np.maximum(ref, 0.0) + res.cpu().numpy()[:, 0::2] + res.cpu().numpy()[:, 1::2]
    elif res_mode == 1:
        ref = ref + res.cpu().numpy()
    ref = np.maximum(ref, 0.0)
    for masked in (True, False):
        a, b = run(2, nd, masked), run(0, nd, masked)                  # (2: every shape the kernel is built for; 1 = the product rule)
        assert torch.equal(a, b)
        assert torch.equal(run(1, nd, masked), b)
        assert torch.equal(run(3, nd, masked), b)                      # (3: the half-swizzled form of the same kernel, every shape)
        np.testing.assert_allclose(a.cpu().numpy(), ref, **TOL)
    assert torch.equal(run(2, sd, True), run(0, nd, True))             # entries outside the masks are never read
    r0 = 16 * (n_out // 48)
    assert torch.equal(run(2, nd, True, r0)[r0:], run(0, nd, True, r0)[r0:])
    assert torch.equal(run(3, nd, True, r0)[r0:], run(0, nd, True, r0)[r0:]) and torch.equal(run(3, sd, True), run(0, nd, True))
    assert torch.equal(run(2, nd, True, xin=wide, ld=48, col=16), run(0, nd, True))   # rows that are not lines: the generic tiles


@pytest.mark.parametrize("cin,cout,K,n_out,res_mode", [
    (8, 8, 81, 40003, 1), (8, 16, 81, 5000, 0), (16, 16, 81, 33333, 1), (16, 8, 81, 2049, 0), (16, 16, 27, 40000, 2),
    (8, 16, 27, 777, 0), (8, 8, 8, 33000, 0), (16, 8, 8, 1000, 0), (16, 16, 3, 17, 0), (8, 8, 81, 63, 2), (8, 8, 99, 1300, 0)])
def test_rowlane_kernel_is_bitwise_the_mfma_one(cin, cout, K, n_out, res_mode):
    """The small-channel layers on the row-per-lane VALU kernel (csrc/spconv_rowlane.hip: one lane per output row, the tap's
    weights from SGPRs, v_pk_fma_f32) against the MFMA tiles: the SAME bits -- per row and channel both are one fmaf chain over
    (tap, MFMA step, lane group) -- for one and two rows per lane, with and without active-tap masks, every epilogue, a row
    suffix, tap lists of every length (the pipeline's clamped tail) and a table whose slots outside the group masks hold
    garbage (sparse table stores: insmos_build_nbr_rank_sparse, k_resolve_taps<.., 2>)."""
    from gpu_util import dev, lib, pack_layer, stream, tap_masks
    from insmos_amd import _lib
    rng = np.random.default_rng(K * 100 + cin + cout + 7)
    n_in = max(n_out // 2, 40)
    nbr = rng.integers(0, n_in, size=(K, n_out)).astype(np.int32)
    ngrp = (n_out + 15) // 16
    n_act = rng.integers(0, K + 1, size=ngrp)
    grp = np.zeros((K, ngrp), bool)
    for gi in range(ngrp):
        grp[rng.permutation(K)[:n_act[gi]], gi] = True
    slot_on = np.repeat(grp, 16, axis=1)[:, :n_out]
    nbr[~(slot_on & (rng.uniform(size=(K, n_out)) < 0.7))] = -1
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.3)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    layer = pack_layer(taps, bias, cin, cout)
    ld_res = cout if res_mode == 1 else 2 * cout
    res = dev(rng.normal(size=(n_out, ld_res)).astype(np.float32)) if res_mode else None
    masks = tap_masks(nbr).view(np.int32)
    garbage = nbr.copy()                                   # what a sparsely stored table looks like outside its masks
    off = ~slot_on
    garbage[off] = rng.integers(-5, n_in + 5, size=int(off.sum())).astype(np.int32)
    xd, nd, gd, md = dev(x), dev(nbr), dev(garbage), dev(masks)

    def run(mode, rpl, table, masked, row0=0):
        assert lib().insmos_debug_conv_rowlane(mode, rpl) == 0
        try:
            out = torch.full((n_out, cout), -7.0, device="cuda:0")
            _lib.check(lib().insmos_sparse_conv_rows(xd.data_ptr(), n_in, cin, layer.cin, table.data_ptr(), md.data_ptr() if masked else None,
                                                     K, n_out, row0, layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(), cout,
                                                     layer.cout, res.data_ptr() if res is not None else None, ld_res if res_mode else 0,
                                                     res_mode, 1 if res_mode == 2 else 0, 1, stream()), "insmos_sparse_conv_rows")
            torch.cuda.synchronize()
            return out
        finally:
            lib().insmos_debug_conv_rowlane(-1, 0)

    ref = R.sparse_conv(x, nbr, taps) + bias
    if res_mode == 2:
        ref = np.maximum(ref, 0.0) + res.cpu().numpy()[:, 0::2] + res.cpu().numpy()[:, 1::2]
    elif res_mode == 1:
        ref = ref + res.cpu().numpy()
    ref = np.maximum(ref, 0.0)
    base = {m: run(0, 0, nd, m) for m in (True, False)}    # the MFMA tiles
    np.testing.assert_allclose(base[True].cpu().numpy(), ref, **TOL)
    for rpl in (1, 2):
        for masked in (True, False):
            assert torch.equal(run(15, rpl, nd, masked), base[masked]), (rpl, masked)
        assert torch.equal(run(15, rpl, gd, True), base[True]), ("garbage outside the masks", rpl)
        r0 = 16 * (n_out // 48)
        assert torch.equal(run(15, rpl, nd, True, r0)[r0:], base[True][r0:])


def _with_precision(mode, layers, fn):
    """Run fn() in conv precision `mode` (3: with the layers' split weights registered); always back to exact fp32."""
    from gpu_util import lib, stream
    from insmos_amd import _lib
    bufs = []
    try:
        if mode == 3:
            for l in layers:
                buf = torch.empty_like(l.w)
                _lib.check(lib().insmos_split_weights_bf16(l.w.data_ptr(), l.w.numel(), buf.data_ptr(), stream()), "split")
                _lib.check(lib().insmos_register_split_weights(l.w.data_ptr(), buf.data_ptr()), "register")
                bufs.append(buf)
            torch.cuda.synchronize()
        _lib.check(lib().insmos_conv_precision(mode), "insmos_conv_precision")
        return fn()
    finally:
        lib().insmos_conv_precision(0)
        for l in layers:
            lib().insmos_register_split_weights(l.w.data_ptr(), None)


@pytest.mark.parametrize("cin,cout,K,n_out", [(128, 128, 27, 20000), (64, 64, 27, 9000), (256, 128, 27, 3000), (32, 32, 81, 5000),
                                              (48, 32, 81, 4000), (32, 64, 27, 4000), (32, 16, 27, 900), (64, 128, 27, 700)])
def test_reduced_precision_modes_are_opt_in_and_accurate(cin, cout, K, n_out):
    """insmos_conv_precision: mode 3 (split-bf16 x 3, the experiment) stays within ~1e-4 of the exact fp32 kernel relative to
    the output scale, mode 1 (bf16 operands, training opt-in) within bf16's 2^-8 per product; a mode-3 layer WITHOUT registered
    split weights, and everything after the mode is reset, is bit-identical fp32."""
    from gpu_util import dev, pack_layer, run_conv, tap_masks, lib
    rng = np.random.default_rng(cin * 7 + cout + K)
    n_in = max(n_out // 2, 100)
    nbr = _rand_nbr(rng, K, n_out, n_in, 0.35)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.35)).astype(np.float32)
    layer = pack_layer(taps, rng.normal(size=cout).astype(np.float32), cin, cout)
    xd, nd, md = dev(x), dev(nbr), dev(tap_masks(nbr).view(np.int32))
    run = lambda: run_conv(layer, xd, nd, n_out, mask=md)
    exact = run()
    scale = float(exact.abs().mean())
    y3 = _with_precision(3, [layer], run)
    assert not torch.equal(y3, exact)                                  # the variant really ran
    assert float((y3 - exact).abs().max()) < 2e-4 * scale
    y1 = _with_precision(1, [layer], run)
    e1 = float((y1 - exact).abs().max())
    assert 2e-4 * scale < e1 < 5e-2 * scale
    assert torch.equal(_with_precision(3, [], run), exact)             # mode 3 without split weights: the fp32 kernel
    assert torch.equal(run(), exact) and lib().insmos_conv_precision(2) != 0   # back to the default; unknown modes are refused


def test_single_chunk_layers_ignore_the_precision_modes():
    """Cin = 8 / 16 layers are vector-memory bound: they stay on the exact fp32 kernels in every mode."""
    from gpu_util import dev, pack_layer, run_conv, tap_masks
    rng = np.random.default_rng(3)
    for cin, cout in ((16, 16), (8, 16)):
        nbr = _rand_nbr(rng, 27, 3000, 1500, 0.4)
        layer = pack_layer((rng.normal(size=(27, cin, cout)) * 0.1).astype(np.float32), np.zeros(cout, np.float32), cin, cout)
        xd, nd, md = dev(rng.normal(size=(1500, cin)).astype(np.float32)), dev(nbr), dev(tap_masks(nbr).view(np.int32))
        run = lambda: run_conv(layer, xd, nd, 3000, mask=md)
        exact = run()
        assert torch.equal(_with_precision(3, [layer], run), exact) and torch.equal(_with_precision(1, [layer], run), exact)


def test_training_convs_bf16_opt_in():
    """autograd.set_train_conv_precision(1): forward and d/dx of the training conv round their operands to bf16 (fp32
    accumulate), d/dW stays fp32; the library is back in exact fp32 after every launch."""
    from gpu_util import dev, run_conv, pack_layer, lib
    from insmos_amd import autograd as A
    rng = np.random.default_rng(9)
    K, cin, cout, n = 27, 64, 32, 3000
    nbr = dev(_rand_nbr(rng, K, n, n, 0.4))
    x0 = dev(rng.normal(size=(n, cin)).astype(np.float32))
    t0 = dev((rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.4)).astype(np.float32))
    dy = dev(rng.normal(size=(n, cout)).astype(np.float32))
    res = {}
    for mode in (0, 1):
        A.set_train_conv_precision(mode)
        try:
            x, t = x0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
            y = A.sparse_conv(x, t, None, nbr)
            y.backward(dy)
            res[mode] = (y.detach(), x.grad.clone(), t.grad.clone())
        finally:
            A.set_train_conv_precision(0)
    for a, b, lo, hi in zip(res[0][:2], res[1][:2], (2e-4, 2e-4), (5e-2, 5e-2)):
        e = float((a - b).abs().max()) / float(a.abs().mean())
        assert lo < e < hi, e
    assert torch.equal(res[0][2], res[1][2])                           # d/dW: fp32 kernel on the saved fp32 input and dy
    # and the library is in exact fp32 again
    layer = pack_layer(t0.cpu().numpy(), np.zeros(cout, np.float32), cin, cout)
    a = run_conv(layer, x0, nbr, n)
    assert torch.equal(a, run_conv(layer, x0, nbr, n)) and float((a - res[0][0]).abs().max()) < 1e-4


def test_split_weights_bf16_is_round_to_nearest_even_hi_plus_lo():
    """insmos_split_weights_bf16: every lane's 4 floats become (hi4 | lo4) with hi = bf16_rne(x), lo = bf16_rne(x - hi) -- checked
    bit for bit against a numpy restatement (including ties, denormal-range residuals, zeros and negative values)."""
    from gpu_util import dev, lib, stream
    from insmos_amd import _lib
    rng = np.random.default_rng(1)
    x = (rng.normal(size=4096) * np.exp(rng.uniform(-20, 20, size=4096))).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 1.00390625, 1.01171875, 3.0e-39, -65504.0]   # exact, ties (odd / even), tiny, large
    xd = dev(x)
    out = torch.empty(4096, dtype=torch.float32, device="cuda:0")
    _lib.check(lib().insmos_split_weights_bf16(xd.data_ptr(), 4096, out.data_ptr(), stream()), "insmos_split_weights_bf16")
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16).reshape(-1, 8)              # per float4: hi0..hi3, lo0..lo3

    def bf16_rne(v):
        b = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint16)
        return r

    def widen(h):
        return (h.astype(np.uint32) << 16).view(np.float32)

    hi = bf16_rne(x)
    lo = bf16_rne(x - widen(hi))
    np.testing.assert_array_equal(got[:, :4].reshape(-1), hi)
    np.testing.assert_array_equal(got[:, 4:].reshape(-1), lo)
    rec = widen(hi).astype(np.float64) + widen(lo).astype(np.float64)
    big = np.abs(x) > 1e-30
    assert np.max(np.abs(rec[big] - x[big]) / np.abs(x[big])) < 2.0 ** -16


def test_bev_kernel_split_bf16x3_experiment():
    from gpu_util import dev, pack_layer, lib, stream
    from insmos_amd import params as P, _lib
    rng = np.random.default_rng(5)
    B, H, W, ci, co = 2, 47, 61, 128, 128
    x = rng.normal(size=(B * H * W, ci)).astype(np.float32)
    w = (rng.normal(size=(co, ci, 3, 3)) * (1.0 / np.sqrt(9 * ci))).astype(np.float32)
    layer = pack_layer(P.conv2d_weight_to_taps(w), rng.normal(size=co).astype(np.float32), ci, co)
    xd = dev(x)

    def run():
        out = torch.zeros((B * H * W, co), device="cuda:0")
        _lib.check(lib().insmos_bev_conv3x3(xd.data_ptr(), B, H, W, ci, ci, layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(),
                                            co, co, 1, stream()), "insmos_bev_conv3x3")
        torch.cuda.synchronize()
        return out

    exact = run()
    y3 = _with_precision(3, [layer], run)
    assert not torch.equal(y3, exact)
    assert float((y3 - exact).abs().max()) < 2e-4 * float(exact.abs().mean() + 1.0)
    assert torch.equal(run(), exact)


@pytest.mark.parametrize("n_site", [1000, 18750 * 2 + 5])
def test_fused_deconv_head_is_bitwise_the_two_launches(n_site):
    """insmos_deconv_head (ConvTranspose2d(2,2)+BN+ReLU and the 1x1 heads in one kernel, base_bev_backbone.py:104-115,
    center_head.py:65-72) == the deconv as a 1x1 layer followed by the head layer, bit for bit."""
    from gpu_util import dev, lib, pack_layer, run_conv, stream
    from insmos_amd import _lib
    rng = np.random.default_rng(n_site)
    cin, cup, hc = 128, 256, 12
    x = dev(rng.normal(size=(n_site, cin)).astype(np.float32))
    wd = (rng.normal(size=(1, cin, 4 * cup)) / np.sqrt(cin)).astype(np.float32)
    wh = (rng.normal(size=(1, cup, hc)) / np.sqrt(cup)).astype(np.float32)
    ld = pack_layer(wd, rng.normal(size=4 * cup).astype(np.float32), cin, 4 * cup)
    lh = pack_layer(wh, rng.normal(size=hc).astype(np.float32), cup, hc)
    up = run_conv(ld, x, None, n_site, relu_post=1)                                  # (n_site, 4*cup) == (4*n_site, cup) sub-site rows
    ref = run_conv(lh, up.view(4 * n_site, cup), None, 4 * n_site)
    head = torch.full((4 * n_site, hc), 5.0, device="cuda:0")
    _lib.check(lib().insmos_deconv_head(x.data_ptr(), n_site, cin, cin, ld.w.data_ptr(), ld.b.data_ptr(), cup, lh.w.data_ptr(),
                                        lh.b.data_ptr(), hc, head.data_ptr(), hc, stream()), "insmos_deconv_head")
    torch.cuda.synchronize()
    assert torch.equal(head, ref)


@pytest.mark.parametrize("cin,cout,res_mode", [(8, 8, 0), (8, 8, 1), (8, 16, 0), (16, 16, 1), (16, 32, 0), (16, 8, 2)])
def test_lds_staged_81_tap_kernel_is_bitwise_the_generic_one(cin, cout, res_mode):
    """The 81-tap single-chunk layers can run on k_conv_ldsw (csrc/spconv_lds.hip: a window of input rows + an overflow area staged
    in LDS per 256-row workgroup and time offset, B fragments read from LDS).  Same bits as the unsplit generic kernels, on tables
    built to visit every mode of the kernel: blocks whose neighbours all fall in the window, blocks with a few far neighbours
    (overflow area), blocks with more far neighbours than the overflow area holds (gathers from memory), empty time offsets,
    a missing centre tap, a row suffix (dead-row elimination) and a ragged last block; against the oracle too."""
    from gpu_util import dev, lib, pack_layer, stream, tap_masks
    from insmos_amd import _lib
    rng = np.random.default_rng(1000 + cin * 10 + cout + res_mode)
    K, n_out, n_in = 81, 64 * 37 + 23, 9000
    nbr = np.full((K, n_out), -1, np.int32)
    blocks = (n_out + 63) // 64
    for b in range(blocks):
        r = np.arange(b * 64, min(n_out, b * 64 + 64))
        kind = b % 6
        for dg in range(3):
            if kind == 5 and dg == 1:
                continue                                              # an empty time offset
            base = int(rng.integers(300, n_in - 600))                # where this block's neighbours sit at this time offset
            for k in range(27):
                kt = dg * 27 + k
                if rng.uniform() < 0.35:
                    continue                                          # tap unused by the whole block
                if kind == 4 and k == 13:
                    continue                                          # no centre tap: the window centres on the smallest index
                present = rng.uniform(size=len(r)) < 0.45
                near = base + (r - r[0]) + rng.integers(-90, 90, size=len(r))
                far = rng.integers(0, n_in, size=len(r))
                p_far = {0: 0.0, 1: 0.05, 2: 0.97, 3: 0.12, 4: 0.03, 5: 0.1}[kind]
                v = np.where(rng.uniform(size=len(r)) < p_far, far, near)
                nbr[kt, r] = np.where(present, np.clip(v, 0, n_in - 1), -1)
    nbr[:, 64 * 3:64 * 3 + 16] = -1                                    # a 16-row tile without any tap inside an active block
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin * K * 0.15)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    layer = pack_layer(taps, bias, cin, cout)
    ld_res = cout if res_mode == 1 else 2 * cout
    res = dev(rng.normal(size=(n_out, ld_res)).astype(np.float32)) if res_mode else None
    xd, nd, md = dev(x), dev(nbr), dev(tap_masks(nbr).view(np.int32))
    ntile = (cout + 15) // 16

    def run(lds, row0=0):
        lib().insmos_debug_conv_lds(lds)
        if not lds and cin == 16:
            lib().insmos_debug_conv_force(ntile, 1, 3)                # the unsplit generic kernel (a tap-split tile sums in another order)
        try:
            out = torch.full((n_out, cout), -7.0, device="cuda:0")
            _lib.check(lib().insmos_sparse_conv_rows(xd.data_ptr(), n_in, cin, layer.cin, nd.data_ptr(), md.data_ptr(), K, n_out, row0,
                                                     layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr(), cout, layer.cout,
                                                     res.data_ptr() if res is not None else None, ld_res if res_mode else 0, res_mode,
                                                     1 if res_mode == 2 else 0, 1, stream()), "insmos_sparse_conv_rows")
            torch.cuda.synchronize()
            return out
        finally:
            lib().insmos_debug_conv_lds(0)                              # (the default: off)
            lib().insmos_debug_conv_force(0, 0, 0)

    ref = R.sparse_conv(x, nbr, taps) + bias
    if res_mode == 2:
        ref = np.maximum(ref, 0.0) + res.cpu().numpy()[:, 0::2] + res.cpu().numpy()[:, 1::2]
    elif res_mode == 1:
        ref = ref + res.cpu().numpy()
    ref = np.maximum(ref, 0.0)
    a, b = run(1), run(0)
    np.testing.assert_allclose(a.cpu().numpy(), ref, **TOL)
    assert torch.equal(a, b), float((a - b).abs().max())
    assert torch.equal(run(1), a)                                       # deterministic
    for r0 in (16, 64 * 5 + 32, 64 * 30):
        assert torch.equal(run(1, r0)[r0:], b[r0:])                     # a row suffix: other block boundaries, the same rows' bits


@pytest.mark.parametrize("B,H,W,c0", [(1, 125, 150, 256), (3, 37, 53, 32), (2, 16, 20, 64)])
def test_bev_constant_region_skipping_is_bitwise_the_dense_kernel(B, H, W, c0):
    """insmos_bev_conv3x3_skip against insmos_bev_conv3x3 over a stack of three 3x3 layers on a mostly EMPTY map (the BEV map of a
    LiDAR window: 14 % of the sites occupied): the empty region stays one constant vector per layer (insmos_bev_constant, evaluated
    by the kernel itself), row groups of constant sites skip their matrix work -- every output bit equal, layer after layer, with
    the constant really reached far from the occupied sites, zero padding honoured at the image border, an all-empty image and an
    all-occupied one in the batch."""
    from gpu_util import dev, lib, pack_layer, stream
    from insmos_amd import params as P, _lib
    rng = np.random.default_rng(B * 100 + H)
    occ = np.zeros((B, H, W), bool)
    for b in range(B):
        for _ in range(max(1, H * W // 400)):                      # a few blobs per image
            y, x = rng.integers(0, H), rng.integers(0, W)
            occ[b, max(0, y - 1):y + 2, max(0, x - 2):x + 2] |= rng.uniform(size=occ[b, max(0, y - 1):y + 2, max(0, x - 2):x + 2].shape) < 0.6
    if B >= 2:
        occ[1] = False                                              # an image without a single voxel
    if B >= 3:
        occ[2] = True                                               # and a full one
    x0 = np.zeros((B, H, W, c0), np.float32)
    x0[occ] = np.abs(rng.normal(size=(int(occ.sum()), c0))).astype(np.float32)
    ys, xs = np.nonzero(occ.reshape(B * H, W))
    coords = np.stack([ys // H, np.zeros_like(ys), ys % H, xs], 1).astype(np.int32)   # [b, z, y, x]
    chans = [c0, 128, 128, 128]
    layers = []
    for l in range(3):
        w = (rng.normal(size=(chans[l + 1], chans[l], 3, 3)) * (1.0 / np.sqrt(9 * chans[l]))).astype(np.float32)
        layers.append(pack_layer(P.conv2d_weight_to_taps(w), (rng.normal(size=chans[l + 1]) * 0.3).astype(np.float32), chans[l], chans[l + 1]))
    L = lib()
    st = stream()
    dist = torch.empty(B * H * W, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(int(L.insmos_bev_distance_map_ws_bytes(B, H, W)), dtype=torch.uint8, device="cuda:0")
    cd = dev(coords) if len(coords) else None
    _lib.check(L.insmos_bev_distance_map(cd.data_ptr() if cd is not None else None, len(coords), B, H, W, 4, dist.data_ptr(), ws.data_ptr(),
                                         ws.numel(), st), "insmos_bev_distance_map")
    # the distance map against a brute-force numpy one (capped at cap + 1 = 5)
    want = np.full((B, H, W), 5, np.int64)
    for b in range(B):
        oy, ox = np.nonzero(occ[b])
        if len(oy):
            yy, xx = np.mgrid[0:H, 0:W]
            d = np.maximum(np.abs(yy[..., None] - oy), np.abs(xx[..., None] - ox)).min(-1)
            want[b] = np.minimum(d, 5)
    np.testing.assert_array_equal(dist.cpu().numpy().reshape(B, H, W).astype(np.int64), want)
    consts, prev = [], None
    for l in range(3):
        cv = torch.empty(128, device="cuda:0")
        wsf = torch.empty(int(L.insmos_bev_constant_ws_floats(chans[l], 128)), device="cuda:0")
        _lib.check(L.insmos_bev_constant(layers[l].w.data_ptr(), layers[l].b.data_ptr(), chans[l], 128, 1,
                                         prev.data_ptr() if prev is not None else None, cv.data_ptr(), wsf.data_ptr(), st), "insmos_bev_constant")
        consts.append(cv)
        prev = cv
    xa = dev(x0.reshape(B * H * W, c0))
    xb = xa.clone()
    for l in range(3):
        oa = torch.empty((B * H * W, 128), device="cuda:0")
        ob = torch.full((B * H * W, 128), -3.0, device="cuda:0")
        _lib.check(L.insmos_bev_conv3x3(xa.data_ptr(), B, H, W, chans[l], chans[l], layers[l].w.data_ptr(), layers[l].b.data_ptr(),
                                        oa.data_ptr(), 128, 128, 1, st), "insmos_bev_conv3x3")
        _lib.check(L.insmos_bev_conv3x3_skip(xb.data_ptr(), B, H, W, chans[l], chans[l], layers[l].w.data_ptr(), layers[l].b.data_ptr(),
                                             ob.data_ptr(), 128, 128, 1, dist.data_ptr(), l, consts[l].data_ptr(), st), "insmos_bev_conv3x3_skip")
        torch.cuda.synchronize()
        assert torch.equal(oa, ob), (l, float((oa - ob).abs().max()))
        # launch shape: these small launches ran as two 64-channel workgroups per patch (insmos_bev_cosplit, default 1024 workgroups);
        # the one-workgroup-per-patch launch gives the same bits, dense and skipping
        try:
            assert L.insmos_bev_cosplit(0) == 0 and L.insmos_bev_cosplit(-2) != 0
            oc = torch.full((B * H * W, 128), -5.0, device="cuda:0")
            od = torch.full((B * H * W, 128), -7.0, device="cuda:0")
            _lib.check(L.insmos_bev_conv3x3(xa.data_ptr(), B, H, W, chans[l], chans[l], layers[l].w.data_ptr(), layers[l].b.data_ptr(),
                                            oc.data_ptr(), 128, 128, 1, st), "insmos_bev_conv3x3")
            _lib.check(L.insmos_bev_conv3x3_skip(xb.data_ptr(), B, H, W, chans[l], chans[l], layers[l].w.data_ptr(), layers[l].b.data_ptr(),
                                                 od.data_ptr(), 128, 128, 1, dist.data_ptr(), l, consts[l].data_ptr(), st), "insmos_bev_conv3x3_skip")
            torch.cuda.synchronize()
        finally:
            L.insmos_bev_cosplit(-1)
        assert torch.equal(oa, oc) and torch.equal(oa, od), (l, float((oa - oc).abs().max()), float((oa - od).abs().max()))
        # ... and so does the skipping layer over compacted row-group lists (insmos_bev_conv3x3_skip_ws), split and unsplit
        wsl = torch.empty(int(L.insmos_bev_skip_ws_bytes(B, H, W)), dtype=torch.uint8, device="cuda:0")
        for split in (-1, 0):
            try:
                L.insmos_bev_cosplit(split)
                oe = torch.full((B * H * W, 128), -9.0, device="cuda:0")
                _lib.check(L.insmos_bev_conv3x3_skip_ws(xb.data_ptr(), B, H, W, chans[l], chans[l], layers[l].w.data_ptr(), layers[l].b.data_ptr(),
                                                        oe.data_ptr(), 128, 128, 1, dist.data_ptr(), l, consts[l].data_ptr(), wsl.data_ptr(),
                                                        wsl.numel(), st), "insmos_bev_conv3x3_skip_ws")
                torch.cuda.synchronize()
            finally:
                L.insmos_bev_cosplit(-1)
            assert torch.equal(oa, oe), (l, split, float((oa - oe).abs().max()))
        assert L.insmos_bev_conv3x3_skip_ws(xb.data_ptr(), B, H, W, chans[l], chans[l], layers[l].w.data_ptr(), layers[l].b.data_ptr(), oe.data_ptr(),
                                            128, 128, 1, dist.data_ptr(), l, consts[l].data_ptr(), wsl.data_ptr(), 16, st) == -3   # EWORKSPACE
        # the constant is what the kernel computes far from every voxel and from the border (when the map has such a site)
        far = (want > l + 1)
        far[:, :max(l, 0), :] = False; far[:, H - max(l, 0):, :] = False; far[:, :, :max(l, 0)] = False; far[:, :, W - max(l, 0):] = False
        if far.any():
            rows = oa.reshape(B, H, W, 128)[torch.from_numpy(far).cuda()]
            assert torch.equal(rows, consts[l].expand_as(rows)), l
        xa, xb = oa, ob
    # ---- behind the stack: the fused deblock + heads with the constant sites skipped (insmos_deconv_head_skip) == insmos_deconv_head
    cup, hc = 256, 12
    wd = (rng.normal(size=(1, 128, 4 * cup)) / np.sqrt(128)).astype(np.float32)
    wh = (rng.normal(size=(1, cup, hc)) / np.sqrt(cup)).astype(np.float32)
    ldl = pack_layer(wd, rng.normal(size=4 * cup).astype(np.float32), 128, 4 * cup)
    lhl = pack_layer(wh, rng.normal(size=hc).astype(np.float32), cup, hc)
    n_site = B * H * W
    chead = torch.full((64,), 7.0, device="cuda:0")
    wsc = torch.empty(int(L.insmos_deconv_head_constant_ws_floats(128)), device="cuda:0")
    _lib.check(L.insmos_deconv_head_constant(ldl.w.data_ptr(), ldl.b.data_ptr(), 128, cup, lhl.w.data_ptr(), lhl.b.data_ptr(), hc,
                                             consts[2].data_ptr(), chead.data_ptr(), wsc.data_ptr(), st), "insmos_deconv_head_constant")
    ha = torch.full((4 * n_site, 16), 5.0, device="cuda:0")
    hb = torch.full((4 * n_site, 16), 5.0, device="cuda:0")
    _lib.check(L.insmos_deconv_head(xa.data_ptr(), n_site, 128, 128, ldl.w.data_ptr(), ldl.b.data_ptr(), cup, lhl.w.data_ptr(),
                                    lhl.b.data_ptr(), hc, ha.data_ptr(), 16, st), "insmos_deconv_head")
    _lib.check(L.insmos_deconv_head_skip(xa.data_ptr(), n_site, 128, 128, ldl.w.data_ptr(), ldl.b.data_ptr(), cup, lhl.w.data_ptr(),
                                         lhl.b.data_ptr(), hc, hb.data_ptr(), 16, dist.data_ptr(), H, W, 3, chead.data_ptr(), st),
               "insmos_deconv_head_skip")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    _lib.check(L.insmos_deconv_head_skip_active_sites(dist.data_ptr(), n_site, H, W, 3, cnt.data_ptr(), st), "insmos_deconv_head_skip_active_sites")
    torch.cuda.synchronize()
    assert torch.equal(ha, hb), float((ha - hb).abs().max())
    # the count: sites of the 16-site (linear) groups that hold a site within 3 of a voxel or within 1 of the border
    bd = np.minimum(np.minimum(np.arange(H)[:, None], H - 1 - np.arange(H)[:, None]), np.minimum(np.arange(W)[None, :], W - 1 - np.arange(W)[None, :]))
    on = ((want <= 3) | (bd[None] <= 1)).reshape(-1)
    pad = (-len(on)) % 16
    grp = np.concatenate([on, np.zeros(pad, bool)]).reshape(-1, 16).any(1)
    valid = np.concatenate([np.ones(len(on), bool), np.zeros(pad, bool)]).reshape(-1, 16)
    assert int(cnt.item()) == int((valid & grp[:, None]).sum())
    if not grp.all():   # (a skipped group really holds the constant head rows)
        g0 = int(np.nonzero(~grp)[0][0])
        rows = hb[4 * 16 * g0:4 * 16 * (g0 + 1), :hc].reshape(16, 4, hc)
        assert torch.equal(rows, chead.reshape(4, 16)[:, :hc].expand(16, 4, hc))
