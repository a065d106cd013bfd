"""Input stage of the reference's inference driver, on the GPU ("next" row 1 of SURVEY.md section 8f).

Restates (does not import) the reference's SemanticKITTI plumbing:
  * `load_poses` / `load_calib` / `load_files`      dataloader/utils.py:10-68
  * KITTI camera poses -> LiDAR-frame poses        scripts/predict_mos.py:181-197  (T_velo_cam . inv(P0) . P . T_cam_velo)
  * window indexing                                scripts/predict_mos.py:97-107,120-129
  * pose alignment of every past scan into the current scan's frame (float64 transform, stored as float32),
    timestamps round((i - N + 1) * dt, 3), concatenation oldest -> current   scripts/predict_mos.py:131-166
What changes is WHERE it runs: raw scans are uploaded once and kept in an LRU cache on the device (each scan is part of
N consecutive windows), and transform + timestamp + concat is one HIP kernel per scan (`insmos_stack_scan`) writing
straight into the (sum N_i, 5) window tensor the model consumes.
"""
import collections
import ctypes
import os

import numpy as np
import torch

from . import _lib


def load_poses(pose_path):
    """dataloader/utils.py:10-38: one pose per line, 12 (3x4) or 16 (4x4) floats; `.npz`-style files via np.load."""
    poses = []
    if ".txt" in pose_path:
        with open(pose_path, "r") as f:
            for line in f.readlines():
                T = np.array([float(v) for v in line.split()], dtype=float)
                if len(T) == 12:
                    T = np.vstack((T.reshape(3, 4), [0, 0, 0, 1]))
                elif len(T) == 16:
                    T = T.reshape(4, 4)
                poses.append(T)
    else:
        poses = np.load(pose_path)["arr_0"]
    return np.array(poses)


def load_calib(calib_path):
    """dataloader/utils.py:41-59: the `Tr:` line (T_cam_velo, 3x4) as a 4x4 matrix."""
    T_cam_velo = []
    with open(calib_path, "r") as f:
        for line in f.readlines():
            if "Tr:" in line:
                v = np.array([float(x) for x in line.replace("Tr:", "").split()], dtype=float)
                T_cam_velo = np.vstack((v.reshape(3, 4), [0, 0, 0, 1]))
    return np.array(T_cam_velo)


def load_files(folder):
    """dataloader/utils.py:62-68."""
    paths = [os.path.join(dp, f) for dp, dn, fn in os.walk(os.path.expanduser(folder)) for f in fn]
    paths.sort()
    return paths


def read_lidar_poses(path_to_seq, filename_poses="poses.txt"):
    """scripts/predict_mos.py:181-197."""
    poses = np.array(load_poses(os.path.join(path_to_seq, filename_poses)))
    inv_frame0 = np.linalg.inv(poses[0])
    T_cam_velo = np.asarray(load_calib(os.path.join(path_to_seq, "calib.txt"))).reshape((4, 4))
    T_velo_cam = np.linalg.inv(T_cam_velo)
    return np.array([T_velo_cam.dot(inv_frame0).dot(p).dot(T_cam_velo) for p in poses])


class SequenceWindows:
    """Windows of one sequence with the reference's indexing (predict_mos.py:97-107): window j covers scans
    scan_idx - skip*(N-1) ... scan_idx (step skip), scan_idx = skip*(N-1) + j."""

    def __init__(self, cfg, seq_dir, n_past_steps=None, device="cuda:0", cache_scans=32):
        self.seq_dir = seq_dir
        self.n = int(n_past_steps if n_past_steps is not None else cfg["MODEL"]["N_PAST_STEPS"])
        self.dt_pred = float(cfg["MODEL"]["DELTA_T_PREDICTION"])
        self.dt_data = float(cfg["DATA"].get("DELTA_T_DATA", self.dt_pred))
        assert self.dt_pred >= self.dt_data - 1e-9, "DELTA_T_PREDICTION needs to be larger than DELTA_T_DATA!"
        self.skip = int(round(self.dt_pred / self.dt_data))
        self.transform = bool(cfg["DATA"]["TRANSFORM"])
        self.files = load_files(os.path.join(seq_dir, "velodyne"))
        self.poses = read_lidar_poses(seq_dir, cfg["DATA"]["POSES"]) if self.transform else None
        if self.transform:
            assert len(self.poses) == len(self.files)
        self.device = torch.device(device)
        self._cache = collections.OrderedDict()
        self._cache_cap = max(cache_scans, self.n + 1)

    def __len__(self):
        return max(0, len(self.files) - self.skip * (self.n - 1))

    def indices(self, j):
        scan_idx = self.skip * (self.n - 1) + j
        return list(range(scan_idx - self.skip * (self.n - 1), scan_idx + 1, self.skip))

    def _scan(self, i):
        if i in self._cache:
            self._cache.move_to_end(i)
            return self._cache[i]
        pts = np.fromfile(self.files[i], dtype=np.float32).reshape((-1, 4))  # predict_mos.py:199-203
        t = torch.from_numpy(pts).to(self.device, non_blocking=True)
        self._cache[i] = t
        while len(self._cache) > self._cache_cap:
            self._cache.popitem(last=False)
        return t

    def transforms(self, idx_list):
        """float64 4x4 per scan: inv(to_pose) @ from_pose (predict_mos.py:161-166); identity without poses."""
        if not self.transform:
            return [np.eye(4) for _ in idx_list]
        to_pose = self.poses[idx_list[-1]]
        return [np.linalg.inv(to_pose) @ self.poses[i] for i in idx_list]

    def window(self, j):
        """-> (past_point_clouds (sum N_i, 5) float32 device tensor, meta) exactly as DemoDataset.__getitem__."""
        lib = _lib.load()
        idx = self.indices(j)
        scans = [self._scan(i) for i in idx]
        Ts = self.transforms(idx)
        total = sum(int(s.shape[0]) for s in scans)
        out = torch.empty((total, 5), dtype=torch.float32, device=self.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        row = 0
        for k, (s, T) in enumerate(zip(scans, Ts)):
            timestamp = round((k - self.n + 1) * self.dt_pred, 3)
            Th = np.ascontiguousarray(T, dtype=np.float64)
            _lib.check(lib.insmos_stack_scan(s.data_ptr(), int(s.shape[0]), Th.ctypes.data_as(ctypes.c_void_p),
                                             float(np.float32(timestamp)), out.data_ptr() + row * 20, 5, st),
                       "insmos_stack_scan")
            row += int(s.shape[0])
        meta = (os.path.basename(os.path.normpath(self.seq_dir)), idx[-1], [self.files[i] for i in idx])
        return out, meta
