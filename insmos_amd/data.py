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
        # read-ahead: files are read and uploaded by one background thread on its own copy stream, so the disk read and
        # the H2D copy of the NEXT windows' scans overlap with the forward of the current ones (the forward takes ~2 ms per
        # window; reading 1.9 MB and copying it does not fit in front of it)
        self._pending = {}  # scan index -> Future of (device tensor, ready event)
        self._pool = None
        self._copy_stream = None

    def __len__(self):
        return max(0, len(self.files) - self.skip * (self.n - 1))

    def indices(self, j):
        scan_idx = self.skip * (self.n - 1) + j
        return list(range(scan_idx - self.skip * (self.n - 1), scan_idx + 1, self.skip))

    def _load(self, i):
        """Background thread: file -> pinned staging buffer (a small ring, allocated once: page-locking memory per scan
        costs more than the copy) -> device on the copy stream."""
        torch.cuda.set_device(self.device)
        pts = np.fromfile(self.files[i], dtype=np.float32).reshape((-1, 4))  # predict_mos.py:199-203
        n = pts.shape[0]
        slot = self._stage_next % len(self._stage)
        self._stage_next += 1
        buf, ev_prev = self._stage[slot]
        if ev_prev is not None:
            ev_prev.synchronize()  # the previous copy out of this buffer has finished
        if buf is None or buf.shape[0] < n:
            buf = torch.empty((max(n, 150000) * 5 // 4, 4), dtype=torch.float32, pin_memory=True)
        np.copyto(buf.numpy()[:n], pts)  # plain memcpy (a torch CPU op here would wake the whole intra-op thread pool)
        with torch.cuda.stream(self._copy_stream):
            t = buf[:n].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._stage[slot] = (buf, ev)
        return t, ev

    def prefetch(self, j_list):
        """Start reading / uploading the scans of windows j_list that are not resident yet."""
        if self.device.type != "cuda":
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="insmos-scan-reader")
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._stage, self._stage_next = [(None, None)] * 6, 0
        for j in j_list:
            if 0 <= j < len(self):
                for i in self.indices(j):
                    if i not in self._cache and i not in self._pending:
                        self._pending[i] = self._pool.submit(self._load, i)

    def _scan(self, i):
        if i in self._cache:
            self._cache.move_to_end(i)
            return self._cache[i]
        if i in self._pending:
            t, ev = self._pending.pop(i).result()
            torch.cuda.current_stream(self.device).wait_event(ev)  # the consumer stream waits, the host does not
            t.record_stream(torch.cuda.current_stream(self.device))
        else:
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
