#!/usr/bin/env python3
"""MI355X counterpart of the reference's instance-level refinement (scripts/refine.py:133-302), the step that follows
the forward in the published pipeline (README.md:162-168): cars whose points are mostly predicted moving pull their
whole instance to "moving", a 5-frame instance track (pose-aligned box matching) stabilises the decision, and in
crowded scenes untracked cars are pulled back to "static".

What runs where: the O(points x boxes) work -- the point -> instance map (Array_Index.find_point_in_instance_bbox_
with_yaw), the per-instance point statistics and the relabelling -- are HIP kernels (insmos_points_in_instance_boxes,
insmos_instance_stats, insmos_instance_relabel); the per-frame decisions touch <= 500 instances and stay on the host,
vectorised (one (cars x previous cars) match matrix per look-back frame instead of the script's nested loops).  Per
frame one 12-byte-per-box read-back.  Same inputs / outputs / file layout as the reference script.
"""
import argparse
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .data import load_calib, load_poses
from .models import load_semantic_config

INSTANCE_WINDOW = 5  # refine.py:168


def lidar_poses(pose_file, calib_file):
    """refine.py:87-101: camera-frame KITTI poses -> LiDAR frame, relative to the first pose."""
    poses = np.asarray(load_poses(pose_file))
    T_cam_velo = np.asarray(load_calib(calib_file)).reshape(4, 4)
    inv0, T_velo_cam = np.linalg.inv(poses[0]), np.linalg.inv(T_cam_velo)
    return np.array([T_velo_cam.dot(inv0).dot(p).dot(T_cam_velo) for p in poses])


class InstanceRefiner:
    """One sequence; call frame() in scan order."""

    def __init__(self, poses_lidar, learning_map_inv=None, device="cuda:0", ground_offset=0.03, quirk_exact=True):
        self.lib = None  # loaded on first frame(): the decision logic alone (decide()) needs no GPU
        self.poses = np.asarray(poses_lidar, dtype=np.float64)
        self.inv = dict(learning_map_inv or {0: 0, 1: 9, 2: 251})
        self.device = torch.device(device)
        self.ground = float(ground_offset)
        self.quirk = 1 if quirk_exact else 0
        self.window = []  # per frame (cars, 7) float32 [x, y, z, dx, dy, dz, moving flag]
        self.frame_idx = 0

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def decide(self, boxes, labels, stats):
        """Host part of refine.py:205-293 for one frame -> per-box label override (0 = keep, 1 = static, 2 = moving)."""
        f = self.frame_idx
        car = (labels == 1) & (stats[:, 0] > 0)
        ids = np.nonzero(car)[0]  # the script's car numbering: box order, cars with points only
        npts, nmov, nconf = (stats[ids, k].astype(np.int64) for k in range(3))
        ratio = nmov / np.maximum(npts, 1)
        attrs = np.array(boxes[ids], dtype=np.float32, copy=True)
        attrs[:, 6] = ratio > 0.6  # the heading slot carries the moving flag from here on (refine.py:224-228)
        moving_cars = int((ratio > 0.3).sum())
        some_moving = ratio > 0.001
        confident = (nconf / np.maximum(npts, 1)) > 0.5
        dec = np.zeros(len(ids), dtype=np.int32)
        if f != 0:  # refine.py:242-253
            for thresh, sel in ((3, some_moving), (5, confident)):
                if moving_cars >= thresh:
                    if f < INSTANCE_WINDOW:
                        dec[sel] = 2
                    attrs[sel, 6] = 1
        elif moving_cars >= 5:  # refine.py:254-259
            dec[some_moving | confident] = 2
        self.window.append(attrs)
        if f >= INSTANCE_WINDOW:  # refine.py:263-295
            found = np.zeros(len(ids), dtype=np.int64)
            moving = np.zeros(len(ids), dtype=np.int64)
            xyz1 = np.hstack([attrs[:, :3].astype(np.float64), np.ones((len(ids), 1))])
            for back in range(INSTANCE_WINDOW):
                prev = self.window[INSTANCE_WINDOW - 1 - back]
                if len(prev) == 0 or len(ids) == 0:
                    continue
                T = np.linalg.inv(self.poses[f - back - 1]) @ self.poses[f]
                c = (T @ xyz1.T).T[:, :3]  # this frame's centres in the past frame (float64, as numpy promotes there)
                m = (np.abs(c[:, None, 0] - prev[None, :, 0]) < 1) & (np.abs(c[:, None, 1] - prev[None, :, 1]) < 1) & \
                    (np.abs(c[:, None, 2] - prev[None, :, 2]) < 0.5)
                for k in (3, 4, 5):
                    m &= np.abs(attrs[:, None, k] - prev[None, :, k]) < 0.3
                hit = m.any(1)
                first = m.argmax(1)  # the script stops at the first matching past box
                found += hit
                moving += hit & (prev[first, 6] == 1)
            full = found == INSTANCE_WINDOW
            attrs[full & (moving > 3), 6] = 1
            attrs[~full & ((moving > 1) | ((moving > 0) & (moving_cars >= 3))), 6] = 1
            dec[attrs[:, 6] == 1] = 2  # top-down
            if len(ids) > 6:
                dec[attrs[:, 6] == 0] = 1
            self.window.pop(0)
        out = np.zeros(len(labels), dtype=np.int32)
        out[ids] = dec
        return out

    def frame(self, scan, boxes, labels, mos_raw, conf):
        """scan (N, >=3) fp32 device; boxes (K,7), labels (K,) host or device; mos_raw (N,) predicted .label values
        (device, any integer dtype); conf (N,2) device or None.  Returns refined labels (N,) int32 on the device."""
        if self.lib is None:
            self.lib = _lib.load()
        dev, lib, st = self.device, self.lib, self._stream()
        scan = scan.to(dev, torch.float32)
        if scan.stride(1) != 1:
            scan = scan.contiguous()
        N = int(scan.shape[0])
        boxes_h = np.ascontiguousarray(torch.as_tensor(boxes).cpu().numpy(), dtype=np.float32).reshape(-1, 7)
        labels_h = np.ascontiguousarray(torch.as_tensor(labels).cpu().numpy(), dtype=np.int64).reshape(-1)
        K = len(labels_h)
        sem = torch.bitwise_and(torch.as_tensor(mos_raw).to(dev).to(torch.int64), 0xFFFF).to(torch.int32)  # refine.py:28
        mos = sem.clone()
        mos[sem == 251] = 2  # refine.py:181-182
        mos[sem == 9] = 1
        if K > 0 and N > 0:
            boxes_d = torch.from_numpy(boxes_h).to(dev)
            labels_d = torch.from_numpy(labels_h).to(dev)
            index = torch.empty((N, 3), dtype=torch.int32, device=dev)
            scratch = torch.empty((20 * K + 16,), dtype=torch.int32, device=dev)
            _lib.check(lib.insmos_points_in_instance_boxes(scan.data_ptr(), N, scan.stride(0), boxes_d.data_ptr(),
                                                           labels_d.data_ptr(), K, self.ground, 3, self.quirk,
                                                           index.data_ptr(), scratch.data_ptr(), st),
                       "insmos_points_in_instance_boxes")
            use_conf = conf is not None and self.frame_idx >= 9  # refine.py:177-178
            conf_d = conf.to(dev, torch.float32).reshape(-1, 2).contiguous() if use_conf else None
            stats = torch.empty((K, 3), dtype=torch.int32, device=dev)
            _lib.check(lib.insmos_instance_stats(index.data_ptr(), 3, 0, mos.data_ptr(),
                                                 conf_d.data_ptr() if conf_d is not None else None, N, K,
                                                 stats.data_ptr(), st), "insmos_instance_stats")
            decision = self.decide(boxes_h, labels_h, stats.cpu().numpy())
            if decision.any():
                dec_d = torch.from_numpy(decision).to(dev)
                _lib.check(lib.insmos_instance_relabel(index.data_ptr(), 3, 0, dec_d.data_ptr(), N, K, mos.data_ptr(), st),
                           "insmos_instance_relabel")
        else:
            self.decide(boxes_h, labels_h, np.zeros((K, 3), np.int32))
        self.frame_idx += 1
        out = mos.clone()  # refine.py:129-133
        for k, v in self.inv.items():
            out[mos == int(k)] = int(v)
        return out


def _walk_sorted(folder):
    paths = [os.path.join(dp, f) for dp, dn, fn in os.walk(os.path.expanduser(folder)) for f in fn]
    paths.sort()
    return paths


def refine_sequence(data_dir, pred_root, seq, out_root, learning_map_inv, device="cuda:0"):
    """refine.py:143-299 for one sequence; returns the number of frames written."""
    scans = _walk_sorted(os.path.join(data_dir, "velodyne"))
    sub = lambda name: os.path.join(pred_root, name, "sequences", seq, "predictions")  # noqa: E731
    boxes_p, mos_p, conf_p = _walk_sorted(sub("bbox_preb")), _walk_sorted(sub("mos_preb")), _walk_sorted(sub("confidence"))
    out_dir = os.path.join(out_root, "mos_preb", "sequences", seq, "predictions")
    os.makedirs(out_dir, exist_ok=True)
    ref = InstanceRefiner(lidar_poses(os.path.join(data_dir, "poses.txt"), os.path.join(data_dir, "calib.txt")),
                          learning_map_inv, device)
    for i in range(len(scans)):
        scan = torch.from_numpy(np.fromfile(scans[i], dtype=np.float32).reshape(-1, 4)).to(device)
        pb = np.load(boxes_p[i], allow_pickle=True).item()
        raw = torch.from_numpy(np.fromfile(mos_p[i], dtype=np.uint32).astype(np.int64)).to(device)
        conf = torch.from_numpy(np.load(conf_p[i]).reshape(-1, 2).astype(np.float32)).to(device)
        out = ref.frame(scan, pb["pred_boxes"], pb["pred_labels"], raw, conf)
        out.cpu().numpy().astype(np.int32).tofile(os.path.join(out_dir, str(mos_p[i])[-12:-6] + ".label"))
    return len(scans)


def main():
    ap = argparse.ArgumentParser(description="instance-level refinement on MI355X (counterpart of scripts/refine.py)")
    ap.add_argument("--split", type=str, default="valid", help="valid or test")
    ap.add_argument("--data_path", type=str, default="demo_data")
    ap.add_argument("--pred_root", type=str, default=os.path.join("preb_out", "InsMOS"))
    ap.add_argument("--out", type=str, default="preb_out_refine")
    args = ap.parse_args()
    sem = load_semantic_config({})
    seqs = [8] if args.split == "valid" else list(range(11, 22))
    for s in seqs:
        seq = str(s).zfill(2)
        n = refine_sequence(os.path.join(args.data_path, seq), args.pred_root, seq, args.out, sem["learning_map_inv"])
        print(f"sequence {seq}: {n} frames refined")


if __name__ == "__main__":
    main()
