"""Synthetic scene S0 -- deterministic HDL-64E-like ray-cast scans.

Used by bench.py, __graft_entry__.smoke() and the tests; there is no SemanticKITTI data in
this environment.  This is survey tooling written for this build, not reference code.
Layout of a window matches what the reference's DemoDataset hands to the model
(scripts/predict_mos.py:114-159): (sum N_i, 5) float32 rows [x, y, z, intensity, t], scans
concatenated oldest -> current, t = round((i - N + 1) * dt, 3), all scans expressed in the
frame of the last scan.
"""
import numpy as np


def make_world(rng, n_build=60, n_cars=40):
    boxes = []
    for _ in range(n_build):
        cx = rng.uniform(-80, 120)
        side = rng.choice([-1, 1])
        cy = side * rng.uniform(8, 45)
        sx, sy, sz = rng.uniform(6, 25), rng.uniform(6, 20), rng.uniform(3, 12)
        boxes.append((np.array([cx - sx / 2, cy - sy / 2, -1.73]),
                      np.array([cx + sx / 2, cy + sy / 2, -1.73 + sz])))
    for _ in range(n_cars):
        cx = rng.uniform(-60, 100)
        cy = rng.choice([-1, 1]) * rng.uniform(2.5, 7)
        boxes.append((np.array([cx - 2.1, cy - 0.9, -1.73]), np.array([cx + 2.1, cy + 0.9, -0.2])))
    return boxes


def make_scan(rng, ego_x, n_az, boxes, n_beams=64, sensor_h=1.73):
    el = np.deg2rad(np.linspace(2.0, -24.8, n_beams))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    EL, AZ = np.meshgrid(el, az, indexing='ij')
    d = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], -1).reshape(-1, 3)
    o = np.array([ego_x, 0.0, 0.0])
    t = np.full(len(d), np.inf)
    down = d[:, 2] < -1e-6
    t = np.minimum(t, np.where(down, -sensor_h / np.where(down, d[:, 2], -1), np.inf))
    inv = 1.0 / np.where(np.abs(d) < 1e-9, 1e-9, d)  # (loop-invariant)
    for lo, hi in boxes:
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
        tmin = np.minimum(t0, t1).max(1)
        tmax = np.maximum(t0, t1).min(1)
        ok = (tmax >= np.maximum(tmin, 0)) & (tmin > 0)
        t = np.where(ok, np.minimum(t, tmin), t)
    keep = t < 80.0
    r = t[keep] + rng.normal(0, 0.02, keep.sum())
    return (d[keep] * r[:, None]).astype(np.float32)


def make_window(seed=0, n_scans=10, n_az=1886, dt=0.1, n_build=60, n_cars=40):
    """Returns (sum N_i, 5) float32 [x,y,z,intensity,t]; S0 = make_window(0, 10, 1886)."""
    rng = np.random.default_rng(seed)
    world = make_world(rng, n_build, n_cars)
    scans = [make_scan(rng, 1.0 * i, n_az, world) for i in range(n_scans)]
    for i, p in enumerate(scans):
        p[:, 0] += 1.0 * i - (n_scans - 1.0)
    out = []
    for i, p in enumerate(scans):
        inten = rng.uniform(0, 1, len(p)).astype(np.float32)
        ts = np.full(len(p), round((i - n_scans + 1) * dt, 3), dtype=np.float32)
        out.append(np.concatenate([p, inten[:, None], ts[:, None]], 1))
    return np.concatenate(out, 0).astype(np.float32)


def make_labels(points_cur, seed=0):
    """Pseudo ground-truth MOS labels (0 unlabeled, 1 static, 2 moving) for IoU plumbing tests."""
    rng = np.random.default_rng(seed + 1000)
    lab = np.ones(len(points_cur), dtype=np.int64)
    lab[rng.uniform(size=len(lab)) < 0.05] = 2
    lab[rng.uniform(size=len(lab)) < 0.02] = 0
    return lab
