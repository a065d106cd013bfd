"""Refine stage ("next" row 3, scripts/refine.py): the oracle against golden vectors produced by the reference's own
code (compiled Array_Index.cpp; refine.py main() run as-is), then the HIP kernels / InstanceRefiner against both."""
import os

import numpy as np
import pytest

from oracle import ref_ops as R
from oracle import ref_refine as RR


SEQS = ("s0_", "s1_")  # high / low dynamic scene


def _poses(g, tag="s0_"):
    P = np.array([np.vstack([np.array(l.split(), float).reshape(3, 4), [0, 0, 0, 1]])
                  for l in str(g[tag + "poses_txt"]).strip().split("\n")])
    T = np.vstack([np.array(str(g[tag + "calib_txt"]).replace("Tr:", "").split(), float).reshape(3, 4), [0, 0, 0, 1]])
    return RR.to_lidar_poses(P, T)


@pytest.mark.parametrize("order", ["a", "b", "sorted"])
def test_oracle_instance_index_vs_compiled_reference(golden_dir, order):
    g = np.load(os.path.join(golden_dir, "instance_index.npz"))
    for og, tag in ((0.03, ""), (0.0, "_g0")):
        np.testing.assert_array_equal(R.points_in_instance_boxes(g["points_" + order], g["boxes"], 3, og),
                                      g["index_" + order + tag])
    assert not np.array_equal(np.sort(g["index_a"][:, 0]), np.sort(g["index_sorted"][:, 0]))  # the order dependence is real


def test_oracle_refine_vs_reference_script(golden_dir):
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    changed, both = 0, set()
    for tag in SEQS:
        r = RR.RefRefiner(_poses(g, tag))
        for i in range(int(g[tag + "n_frames"])):
            k = tag + "f%02d_" % i
            out = r.frame(g[k + "scan"], g[k + "boxes"], g[k + "labels"], g[k + "mos"], g[k + "conf"])
            np.testing.assert_array_equal(out, g[k + "refined"], err_msg=k)
            a = g[k + "mos"].astype(np.int32)
            changed += int((out != a).sum())
            both |= {(int(x), int(y)) for x, y in zip(a[a != out], out[a != out])}
    assert changed > 1500  # the sequences really exercise the refinement ...
    assert (9, 251) in both and (251, 9) in both  # ... in both directions


def test_host_decisions_vs_reference_script(golden_dir):
    """InstanceRefiner.decide (the vectorised host part of the product) fed with oracle-computed instance statistics
    reproduces the reference script's labels -- the GPU kernels only supply index / stats / relabel around it."""
    from insmos_amd.refine import InstanceRefiner
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    for tag in SEQS:
        ref = InstanceRefiner(_poses(g, tag), device="cpu")
        for i in range(int(g[tag + "n_frames"])):
            k = tag + "f%02d_" % i
            boxes, labels = g[k + "boxes"], g[k + "labels"]
            sem = (g[k + "mos"] & 0xFFFF).astype(np.int32)
            mos = np.where(sem == 251, 2, np.where(sem == 9, 1, sem)).astype(np.int32)
            conf = g[k + "conf"] if i >= 9 else np.zeros((len(mos), 2), np.float32)
            index = R.points_in_instance_boxes(g[k + "scan"], np.hstack([boxes, labels[:, None].astype(np.float32)]), 3, 0.03)
            stats = np.zeros((len(labels), 3), np.int32)
            for b in range(len(labels)):
                sel = index[:, 0] == b + 1
                stats[b] = [sel.sum(), (mos[sel] == 2).sum(), (conf[sel, 1] >= 0.00001).sum()]
            dec = ref.decide(boxes, labels, stats)
            ref.frame_idx += 1
            for b in np.nonzero(dec)[0]:
                mos[index[:, 0] == b + 1] = dec[b]
            out = mos.copy()
            for kk, v in {0: 0, 1: 9, 2: 251}.items():
                out[mos == kk] = v
            np.testing.assert_array_equal(out, g[k + "refined"], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("quirk", [1, 0])
def test_points_in_instance_boxes_kernel(golden_dir, quirk):
    import ctypes
    import torch
    from insmos_amd import _lib
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "instance_index.npz"))
    rng = np.random.default_rng(1)
    cases = [(g["points_" + o], g["boxes"], og) for o in ("a", "b", "sorted") for og in (0.03, 0.0)]
    # overlapping same-class boxes, 150 boxes (three 64-box chunks), 50k points: the largest box number must win
    M = 150
    bx = np.zeros((M, 8), np.float32)
    bx[:, 0:2] = rng.uniform(-30, 30, (M, 2)); bx[:, 2] = rng.uniform(-1.5, 0, M)
    bx[:, 3:6] = rng.uniform([2, 1, 1], [6, 3, 2.5], (M, 3)); bx[:, 6] = rng.uniform(-3.2, 3.2, M)
    bx[:, 7] = rng.integers(0, 4, M)
    pts = rng.uniform([-35, -35, -3, 0], [35, 35, 2, 1], (50000, 4)).astype(np.float32)
    cases.append((pts, bx, 0.03))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for p, b, og in cases:
        ref = R.points_in_instance_boxes(p, b, 3, og, quirk=bool(quirk))
        pd = torch.from_numpy(p).cuda()
        bd = torch.from_numpy(np.ascontiguousarray(b[:, :7])).cuda()
        ld = torch.from_numpy(b[:, 7].astype(np.int64)).cuda()
        idx = torch.full((len(p), 3), -7, dtype=torch.int32, device="cuda")
        scratch = torch.empty(20 * len(b) + 16, dtype=torch.int32, device="cuda")
        _lib.check(lib.insmos_points_in_instance_boxes(pd.data_ptr(), len(p), 4, bd.data_ptr(), ld.data_ptr(), len(b), og, 3,
                                                       quirk, idx.data_ptr(), scratch.data_ptr(), st), "inst")
        np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    assert int((ref > 0).sum()) > 500


@pytest.mark.gpu
def test_instance_refiner_matches_reference_script(golden_dir):
    """Whole refine stage on the device + vectorised host decisions == the labels the reference script wrote."""
    import torch
    from insmos_amd.refine import InstanceRefiner
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    for tag in SEQS:
        ref = InstanceRefiner(_poses(g, tag))
        for i in range(int(g[tag + "n_frames"])):
            k = tag + "f%02d_" % i
            out = ref.frame(torch.from_numpy(g[k + "scan"]).cuda(), g[k + "boxes"], g[k + "labels"],
                            torch.from_numpy(g[k + "mos"].astype(np.int64)).cuda(), torch.from_numpy(g[k + "conf"]).cuda())
            np.testing.assert_array_equal(out.cpu().numpy(), g[k + "refined"], err_msg=k)


@pytest.mark.gpu
def test_refine_cli_files_and_random_sequences(golden_dir, tmp_path):
    """refine_sequence on the reference's directory layout == oracle; plus random sequences (other seeds, ego motion with
    rotation) against the oracle restatement."""
    import torch
    from insmos_amd.refine import InstanceRefiner, refine_sequence
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    seq = tmp_path / "data" / "08"
    os.makedirs(seq / "velodyne")
    (seq / "poses.txt").write_text(str(g["s1_poses_txt"]))
    (seq / "calib.txt").write_text(str(g["s1_calib_txt"]))
    pred = tmp_path / "preb_out" / "InsMOS"
    for sub in ("bbox_preb", "mos_preb", "confidence"):
        os.makedirs(pred / sub / "sequences" / "08" / "predictions")
    n = int(g["s1_n_frames"])
    for i in range(n):
        k, stem = "s1_f%02d_" % i, "%06d" % i
        g[k + "scan"].tofile(seq / "velodyne" / (stem + ".bin"))
        np.save(pred / "bbox_preb" / "sequences" / "08" / "predictions" / (stem + ".npy"),
                {"pred_boxes": g[k + "boxes"], "pred_scores": np.ones(len(g[k + "boxes"]), np.float32), "pred_labels": g[k + "labels"]})
        g[k + "mos"].astype(np.int32).tofile(pred / "mos_preb" / "sequences" / "08" / "predictions" / (stem + ".label"))
        np.save(pred / "confidence" / "sequences" / "08" / "predictions" / (stem + ".npy"), g[k + "conf"])
    assert refine_sequence(str(seq), str(pred), "08", str(tmp_path / "out"), {0: 0, 1: 9, 2: 251}) == n
    for i in range(n):
        got = np.fromfile(tmp_path / "out" / "mos_preb" / "sequences" / "08" / "predictions" / ("%06d.label" % i), dtype=np.int32)
        np.testing.assert_array_equal(got, g["s1_f%02d_refined" % i])
    # random sequences vs the oracle: curved ego trajectory, different box jitter
    rng = np.random.default_rng(9)
    for trial in range(3):
        poses = []
        for f in range(10):
            a = 0.05 * f * (trial + 1)
            T = np.eye(4); T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]; T[:3, 3] = [0.9 * f, 0.1 * f * trial, 0]
            poses.append(T)
        poses = np.array(poses)
        base = rng.uniform([-20, -10, -1.0], [20, 10, -0.8], (8, 3))
        dims = rng.uniform([3.8, 1.6, 1.4], [4.6, 1.9, 1.7], (8, 3))
        gpu, cpu = InstanceRefiner(poses), RR.RefRefiner(poses)
        for f in range(10):
            inv = np.linalg.inv(poses[f])
            ctr = (inv[:3, :3] @ (base + [0.4 * f, 0, 0] * (np.arange(8)[:, None] < 4)).T).T + inv[:3, 3] + rng.normal(0, 0.05, (8, 3))
            boxes = np.hstack([ctr, dims + rng.normal(0, 0.03, (8, 3)), rng.uniform(-3, 3, (8, 1))]).astype(np.float32)
            labels = np.where(np.arange(8) == 7, 3, 1).astype(np.int64)
            pts = (ctr[:, None, :] + rng.uniform(-0.8, 0.8, (8, 50, 3))).reshape(-1, 3)
            pts = np.vstack([pts, rng.uniform([-30, -20, -2.5], [30, 20, 0.5], (600, 3))]).astype(np.float32)
            scan = np.hstack([pts, np.zeros((len(pts), 1), np.float32)])
            mos = np.where(rng.uniform(size=len(pts)) < (0.5 if trial else 0.25), 251, 9).astype(np.uint32)
            mos[rng.uniform(size=len(pts)) < 0.03] = 0
            cf = rng.uniform(0, 1, len(pts)).astype(np.float32) * (rng.uniform(size=len(pts)) < 0.6)
            conf = np.stack([1 - cf, cf], 1).astype(np.float32)
            want = cpu.frame(scan, boxes, labels, mos, conf)
            got = gpu.frame(torch.from_numpy(scan).cuda(), boxes, labels, torch.from_numpy(mos.astype(np.int64)).cuda(),
                            torch.from_numpy(conf).cuda())
            np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"trial {trial} frame {f}")
