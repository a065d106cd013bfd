"""-m gpu: the multi-rank code paths on the ONE GPU a test box has -- two processes, both on cuda:0, rendezvous over gloo
(RCCL refuses two ranks on one device; the collectives of the dry run move host copies, see metrics.all_gather_confusion and
ddp._Staged -- on an 8-GPU node the same calls take the device tensors over RCCL).  What runs is the real thing otherwise:

  * bench.timed_steps with the real model (InsMOSNet on the HIP library): barrier + device sync on both sides, the per-rank
    confusion counters all-gathered inside the region -> equals the sum of the two ranks' single-rank matrices;
  * one InsMOSTrainer step per rank on DEVICE gradients through BucketedGradReducer(overlap=True): buckets leave during
    backward, and the reduced gradient of every parameter equals the mean of the two ranks' gradients, bit for bit (the
    kernels are deterministic, and a two-term sum has one order).

Reference: scripts/predict_mos.py:103-106 (independent windows), scripts/train.py:74-83 (DDP).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench
from insmos_amd import params as P
from insmos_amd.metrics import ClassificationMetrics
from insmos_amd.models import InsMOSNet
from insmos_amd.synth import make_labels, make_window
from insmos_amd.train_unet import InsMOSTrainer
rank, world = int(sys.argv[1]), 2
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = "cuda:0"
cfg = P.default_cfg()
sd = P.random_state_dict(cfg, seed=0)

# ---- (1) the timed region of bench.py with the real model; ranks hold different windows
W, steps, warm = 3, 2, 1
def windows_of(r):
    ws = [make_window(seed=10 * r + i, n_scans=4, n_az=160) for i in range(W)]
    gts = [torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=10 * r + i)).to(dev) for i, w in enumerate(ws)]
    return [{"past_point_clouds": torch.from_numpy(w).to(dev)} for w in ws], gts
model = InsMOSNet(cfg, state_dict=sd).cuda(0).eval()
metrics = ClassificationMetrics(3, [0])
batch, gts = windows_of(rank)
dt, value, cm_all = bench.timed_steps(model.forward, batch, gts, metrics, steps, warm, world, dev, torch.cuda.synchronize)
exp = torch.zeros((3, 3), dtype=torch.int64, device=dev)
for r in range(world):                       # both ranks' single-rank matrices, computed locally
    b_r, g_r = windows_of(r)
    _, _, logits = model.forward(b_r, "test")
    for lg, gt in zip(logits, g_r):
        metrics.compute_confusion_matrix(lg, gt, out=exp)
exp = exp * steps
assert cm_all.device.type == "cuda" and torch.equal(cm_all, exp), (cm_all, exp)
assert int(cm_all.sum()) == steps * sum(int(g.numel()) for r in range(world) for g in windows_of(r)[1])
assert abs(value - world * steps * W / dt) < 1e-9

# ---- (2) one training step per rank, overlapped bucketed gradient exchange on device gradients
def train_batch(r):
    rng = np.random.default_rng(100 + r)
    w = make_window(seed=20 + r, n_scans=3, n_az=96)
    m = 5
    gt = np.zeros((1, m, 8), np.float32)
    gt[0, :, 0] = rng.uniform(-30, 30, m); gt[0, :, 1] = rng.uniform(-20, 20, m); gt[0, :, 2] = rng.uniform(-1.5, -0.5, m)
    gt[0, :, 3] = rng.uniform(1.5, 4.5, m); gt[0, :, 4] = rng.uniform(0.6, 2.0, m); gt[0, :, 5] = rng.uniform(1.2, 1.8, m)
    gt[0, :, 6] = rng.uniform(-3.1, 3.1, m); gt[0, :, 7] = rng.integers(1, 4, m)
    return [{"past_point_clouds": torch.from_numpy(w).to(dev),
             "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=r)).to(dev)],
             "gt_boxes": torch.from_numpy(gt).to(dev)}]
sd_t = P.random_state_dict(cfg, 2, cls_bias=-1.0, box_w_std=0.05)
tr = InsMOSTrainer(cfg, sd_t, device=dev)
grads = []
for r in range(world):                       # every rank's gradient, computed locally BEFORE the reducer hooks exist
    for v in tr.params.values():
        v.grad = None
    loss, _, _, _ = tr.forward(train_batch(r), "train")
    loss.backward()
    grads.append({k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in tr.params.items()})
for v in tr.params.values():
    v.grad = None
red = tr.make_reducer(bucket_bytes=2 << 20, overlap=True)
assert len(red.buckets) >= 4
loss, _, _, _ = tr.forward(train_batch(rank), "train")
loss.backward()
assert red._launched >= 1, "no bucket left during backward"
n = red.reduce(average=True)
assert n == len(red.buckets) and red.launched_in_backward >= 1
bad = []
for k, v in tr.params.items():
    want = (grads[0][k] + grads[1][k]) / world
    if not torch.equal(v.grad, want):
        bad.append((k, float((v.grad - want).abs().max())))
assert not bad, bad[:5]
own_differs = sum(1 for k in tr.params if not torch.equal(grads[0][k], grads[1][k]))
assert own_differs > 100, own_differs       # the two ranks really had different gradients
red.close()
torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group()
print("OK", rank, "timed_region_s=%.3f" % dt, "buckets=%d launched_in_backward=%d" % (n, red.launched_in_backward))
"""


def test_two_ranks_on_one_gpu_timed_region_and_overlapped_grad_exchange(tmp_path):
    script = tmp_path / "two_rank_worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, ROOT], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT\n" + p.communicate()[0].decode())
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o[-3000:]
    print("\n".join(o.strip().splitlines()[-1] for o in outs))


def test_bench_gpus_2_dry_run_on_one_gpu():
    """The command the driver uses, with --gpus 2: bench.py starts both ranks itself (gloo, both on GPU 0), rank 0 prints ONE line
    with n_gpus = 2, the whole-job rate of both ranks' windows, and a confusion sum that holds both ranks' points."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", INSMOS_WINDOWS_PER_LAUNCH="2", INSMOS_WINDOWS_IN_FLIGHT="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device-index", "0",
                        "--steps", "2", "--warmup", "1", "--n-az", "320", "--windows-per-step", "4", "--sustain-seconds", "0",
                        "--no-cpu-baseline", "--candidates", "200"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=400)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    lines = [json.loads(l) for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    j = lines[0]
    assert j["n_gpus"] == 2 and j["config"]["parallelism"].startswith("dp2") and j["config"]["backend"] == "gloo"
    # both ranks' counters are in the sum, exactly: 2 steps x 4 slots per rank, every slot holding the rank's own window
    # (seed = rank), each contributing its current-scan points
    sys.path.insert(0, ROOT)
    import bench
    ncur = [int((bench.load_window(r_, 320)[:, 4] == 0).sum()) for r_ in range(2)]
    assert ncur[0] == j["config"]["current_points"]
    assert j["confusion_points"] == 2 * 4 * (ncur[0] + ncur[1]), (j["confusion_points"], ncur)
    assert abs(j["value"] - 2 * 2 * 4 / j["timed_region_s"]) / j["value"] < 2e-2


def test_bench_cfg5_training_step_two_rank_dry_run():
    """`bench.py --config cfg5 --gpus 2` (BASELINE.json configs[4], the training step): both ranks on GPU 0 over gloo, bucketed
    gradient exchange during backward, ONE line from rank 0 with the DDP world in it; and the single-rank line carries the
    roofline of the convolution kernels (forward + d/dx + d/dW)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--config", "cfg5", "--steps", "2", "--warmup", "1", "--n-az", "160", "--windows-per-step", "2", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device-index", "0"] + common,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    lines = [json.loads(l) for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    j = lines[0]
    assert j["metric"] == "train_scans_per_sec" and j["n_gpus"] == 2 and j["config"]["parallelism"].startswith("ddp2")
    assert j["config"]["grad_buckets"] >= 1 and j["config"]["backend"] == "gloo" and j["loss"] > 0
    assert abs(j["value"] - 2 * 2 * 2 / j["timed_region_s"]) / j["value"] < 2e-2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=400)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    j1 = [json.loads(l) for l in out.splitlines() if l.startswith('{"metric"')][0]
    rf = j1["roofline"]
    assert j1["n_gpus"] == 1 and rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and rf["algorithmic_gflop_per_step"] > 0
    assert len(rf["gflop_forward_dx_dw"]) == 3 and all(v > 0 for v in rf["gflop_forward_dx_dw"])
