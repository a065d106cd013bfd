"""oracle/ref_refine.py -- CPU restatement of the instance-level refinement that follows the forward in the published
pipeline (scripts/refine.py:167-302).  TEST INFRASTRUCTURE ONLY (see oracle/ref_ops.py).

Pinned: tests/golden/refine.npz holds the labels the reference script itself wrote for a synthetic 12-frame sequence
(tests/golden/make_golden.py:refine_golden runs its main() as-is); the point -> instance map underneath is pinned to
the compiled Array_Index.cpp by tests/golden/instance_index.npz.
"""
import numpy as np

from . import ref_ops as R

INSTANCE_WINDOW = 5  # refine.py:168


def to_lidar_poses(poses, T_cam_velo):
    """refine.py:87-101 (get_lidar_pose)."""
    inv0 = np.linalg.inv(poses[0])
    T_velo_cam = np.linalg.inv(T_cam_velo)
    return np.array([T_velo_cam.dot(inv0).dot(p).dot(T_cam_velo) for p in poses])


def _move_point(xyz, from_pose, to_pose):
    """refine.py:122-127 (transform_point_cloud) for one point, float64 like numpy's promotion there."""
    T = np.linalg.inv(to_pose) @ from_pose
    v = np.hstack([np.asarray(xyz).reshape(1, 3), np.ones((1, 1))]).T
    return (T @ v).T[0, :3]


class RefRefiner:
    """One sequence; call frame() in order.  Mirrors the loop body of refine.py:169-299 statement by statement."""

    def __init__(self, lidar_poses, learning_map_inv=None):
        self.poses = lidar_poses
        self.window = []  # per frame: list of 7-vectors [x, y, z, dx, dy, dz, moving flag] (refine.py:167)
        self.frame_idx = 0
        self.inv = learning_map_inv or {0: 0, 1: 9, 2: 251}

    def frame(self, scan, boxes, labels, mos_raw, conf):
        f = self.frame_idx
        boxes = np.array(boxes, dtype=np.float32, copy=True)
        sem = (np.asarray(mos_raw).astype(np.uint32) & 0xFFFF).astype(np.int32)  # refine.py:28 (load_labels)
        mos = sem.copy()
        mos[sem == 251] = 2  # refine.py:181-182
        mos[sem == 9] = 1
        conf = np.zeros((len(mos), 2)) if f < 9 else np.asarray(conf).reshape(-1, 2)  # refine.py:177-178
        b8 = np.concatenate([boxes, np.asarray(labels, np.float32).reshape(-1, 1)], 1)  # refine.py:184
        index = R.points_in_instance_boxes(scan, b8, 3, 0.03)  # refine.py:196
        moving_cars = 0
        some_moving, confident, members, attrs = [], [], [], []
        for i in range(len(labels)):  # refine.py:211-240, cars only
            if labels[i] != 1:
                continue
            idx = np.where(index[:, 0] == i + 1)[0]
            if len(idx) == 0:
                continue
            n_mov = int((mos[idx] == 2).sum())
            n_conf = int((conf[idx][:, 1] >= 0.00001).sum())
            car = len(members)
            members.append(idx)
            a = boxes[i]  # a view: the script overwrites the heading slot with the moving flag (refine.py:224-228)
            a[-1] = 1 if n_mov / len(idx) > 0.6 else 0
            attrs.append(a)
            if n_mov / len(idx) > 0.3:
                moving_cars += 1
            if n_mov / len(idx) > 0.001:
                some_moving.append(car)
            if n_conf / len(idx) > 0.5:
                confident.append(car)
        if f != 0:  # refine.py:242-253
            if moving_cars >= 3:
                for c in some_moving:
                    if f < INSTANCE_WINDOW:
                        mos[members[c]] = 2
                    attrs[c][-1] = 1
            if moving_cars >= 5:
                for c in confident:
                    if f < INSTANCE_WINDOW:
                        mos[members[c]] = 2
                    attrs[c][-1] = 1
        elif moving_cars >= 5:  # refine.py:254-259
            for c in some_moving:
                mos[members[c]] = 2
            for c in confident:
                mos[members[c]] = 2
        self.window.append(attrs)  # refine.py:262
        if f >= INSTANCE_WINDOW:
            assert len(self.window) == INSTANCE_WINDOW + 1
            for a in attrs:  # refine.py:266-285: look the car up in each of the 5 previous frames
                found = moving = 0
                for back in range(INSTANCE_WINDOW):
                    c = _move_point(a[0:3], self.poses[f], self.poses[f - back - 1])
                    for q in self.window[INSTANCE_WINDOW - 1 - back]:
                        if (abs(c[0] - q[0]) < 1 and abs(c[1] - q[1]) < 1 and abs(c[2] - q[2]) < 0.5 and abs(a[3] - q[3]) < 0.3
                                and abs(a[4] - q[4]) < 0.3 and abs(a[5] - q[5]) < 0.3):
                            found += 1
                            if q[-1] == 1:
                                moving += 1
                            break
                if found == 5:
                    if moving > 3:
                        a[-1] = 1
                elif moving > 1 or (moving > 0 and moving_cars >= 3):
                    a[-1] = 1
            for j, a in enumerate(attrs):  # refine.py:288-293 (top-down)
                if a[-1] == 1:
                    mos[members[j]] = 2
                if a[-1] == 0 and len(attrs) > 6:
                    mos[members[j]] = 1
            self.window.pop(0)
        self.frame_idx += 1
        out = mos.copy()  # refine.py:129-133 (to_original_labels)
        for k, v in self.inv.items():
            out[mos == k] = v
        return out.astype(np.int32)
