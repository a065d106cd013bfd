"""-m gpu: the RCCL code of the N-rank job, EXECUTED on the one GPU a test box has.  RCCL refuses two ranks on one device, so
the two-rank tests (tests/test_zz_gpu_two_ranks.py) run over gloo with host staging and never enter the `nccl` branches; here
the process group has ONE rank on cuda:0 over backend "nccl" (= RCCL on ROCm) and every collective of the N-rank code is
forced, so first contact with RCCL does not happen on an 8-GPU node:

  * bench.timed_steps(force_collectives=True) with the real model: barrier, all_gather of the DEVICE confusion counters
    (metrics.all_gather_confusion(force=True)), MAX all-reduce of the time -- gathered == local counters, still on the device;
  * one InsMOSTrainer step through BucketedGradReducer(overlap=True, force_collective=True): the buckets leave as async RCCL
    all-reduces during backward; a sum over one rank is the gradient itself, bit for bit;
  * `bench.py --gpus 1 --rccl-selfcheck`: the bench's own timed region inside that group, and the `rccl_world1` leg of a
    plain run.

Each case runs in its own process (a process group is process-wide state).
Reference: scripts/predict_mos.py:100-106 (independent windows), models/metrics.py:16-45 (the counters), scripts/train.py:74-83.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import datetime, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[2])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench
from insmos_amd import params as P
from insmos_amd.metrics import ClassificationMetrics, all_gather_confusion
from insmos_amd.models import InsMOSNet
from insmos_amd.synth import make_labels, make_window
from insmos_amd.train_unet import InsMOSTrainer
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[1]
dev = "cuda:0"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=180), device_id=torch.device(dev))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
cfg = P.default_cfg()
sd = P.random_state_dict(cfg, seed=0)

# ---- (1) the timed region with every collective forced
W, steps = 3, 2
ws = [make_window(seed=i, n_scans=4, n_az=160) for i in range(W)]
gts = [torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=i)).to(dev) for i, w in enumerate(ws)]
batch = [{"past_point_clouds": torch.from_numpy(w).to(dev)} for w in ws]
model = InsMOSNet(cfg, state_dict=sd).cuda(0).eval()
metrics = ClassificationMetrics(3, [0])
dt, value, cm_all = bench.timed_steps(model.forward, batch, gts, metrics, steps, 1, 1, dev, torch.cuda.synchronize,
                                      force_collectives=True)
exp = torch.zeros((3, 3), dtype=torch.int64, device=dev)
_, _, logits = model.forward(batch, "test")
for lg, gt in zip(logits, gts):
    metrics.compute_confusion_matrix(lg, gt, out=exp)
assert cm_all.is_cuda and torch.equal(cm_all, exp * steps), (cm_all, exp)
assert int(cm_all.sum()) == steps * sum(int(g.numel()) for g in gts)
assert abs(value - steps * W / dt) < 1e-9
# without force a one-rank world exchanges nothing and hands the SAME tensor back
assert all_gather_confusion(exp) is exp
got = all_gather_confusion(exp, force=True)
assert got is not exp and got.is_cuda and torch.equal(got, exp)

# ---- (2) one training step, buckets through RCCL during backward
rng = np.random.default_rng(100)
w = make_window(seed=20, n_scans=3, n_az=96)
m = 5
gt = np.zeros((1, m, 8), np.float32)
gt[0, :, 0] = rng.uniform(-30, 30, m); gt[0, :, 1] = rng.uniform(-20, 20, m); gt[0, :, 2] = rng.uniform(-1.5, -0.5, m)
gt[0, :, 3] = rng.uniform(1.5, 4.5, m); gt[0, :, 4] = rng.uniform(0.6, 2.0, m); gt[0, :, 5] = rng.uniform(1.2, 1.8, m)
gt[0, :, 6] = rng.uniform(-3.1, 3.1, m); gt[0, :, 7] = rng.integers(1, 4, m)
tb = [{"past_point_clouds": torch.from_numpy(w).to(dev),
       "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=0)).to(dev)],
       "gt_boxes": torch.from_numpy(gt).to(dev)}]
tr = InsMOSTrainer(cfg, P.random_state_dict(cfg, 2, cls_bias=-1.0, box_w_std=0.05), device=dev)
loss, _, _, _ = tr.forward(tb, "train")
loss.backward()
want = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in tr.params.items()}
for v in tr.params.values():
    v.grad = None
red = tr.make_reducer(bucket_bytes=2 << 20, overlap=True, force_collective=True)
assert len(red.buckets) >= 4
loss, _, _, _ = tr.forward(tb, "train")
loss.backward()
assert red._launched >= 1, "no bucket left during backward"
n = red.reduce(average=True)
assert n == len(red.buckets) and red.collectives_issued == n, (n, red.collectives_issued)
bad = [(k, float((v.grad - want[k]).abs().max())) for k, v in tr.params.items() if not torch.equal(v.grad, want[k])]
assert not bad, bad[:5]
red.close()
torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group()
print("OK rccl world 1: timed_region_s=%.3f buckets=%d launched_in_backward=%d" % (dt, n, red.launched_in_backward))
"""


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_rccl_world1_timed_region_and_bucketed_reducer(tmp_path):
    script = tmp_path / "rccl_world1_worker.py"
    script.write_text(_WORKER)
    port = str(29800 + os.getpid() % 150)
    r = subprocess.run([sys.executable, str(script), port, ROOT], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "OK rccl world 1" in out, out[-3000:]
    print(out.strip().splitlines()[-1])


_SMALL = ["--steps", "2", "--warmup", "1", "--n-az", "320", "--windows-per-step", "4", "--sustain-seconds", "0", "--no-cpu-baseline",
          "--candidates", "200", "--no-extras"]


def _bench_line(extra):
    env = _env()
    env.update(INSMOS_WINDOWS_PER_LAUNCH="2", INSMOS_WINDOWS_IN_FLIGHT="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra + _SMALL, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=500)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    lines = [json.loads(l) for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    return lines[0], out


def test_bench_rccl_selfcheck_runs_the_headline_inside_a_one_rank_nccl_group():
    j, out = _bench_line(["--rccl-selfcheck"])
    assert "backend nccl" in out                       # the backend the process group reported, printed at start-up
    assert j["n_gpus"] == 1 and j["config"]["backend"] == "nccl"
    assert j["rccl_world1_ok"] is True
    assert j["confusion_points"] == 2 * 4 * j["config"]["current_points"]


def test_bench_default_line_carries_the_rccl_world1_leg():
    j, _ = _bench_line([])
    leg = j["rccl_world1"]
    assert j["config"]["backend"] is None              # the headline itself: one rank, no process group
    assert leg["ok"] is True and j["rccl_world1_ok"] is True, leg
    assert leg["backend"] == "nccl" and leg["world_size"] == 1
    assert leg["all_gather_equals_local_counters"] and leg["all_reduce_equals_input"]
