#!/usr/bin/env python3
"""bench.py -- scans/sec of the InsMOS inference hot path on MI355X.

A "step" is one forward() over a batch of `--windows-per-step` windows (each N=10 pose-aligned
scans, ~1.2 M points in, per-point MOS logits + boxes out) with the inputs already resident in HBM:
BASELINE.json configs[1] (synthetic S0); the model runs them as launch sets of
INSMOS_WINDOWS_PER_LAUNCH windows (one set of kernel launches per group), INSMOS_WINDOWS_IN_FLIGHT sets at a time.
One process per GPU; rank r holds the S0-style window of seed r and there is no data-path collective -- the only exchange is the all_gather of the 3x3 confusion counters at the
end of the timed region.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3     # no launcher around it: starts the 8 ranks itself (maybe_spawn_ranks)
    python bench.py --gpus 2 --backend gloo --device-index 0   # dry run of the multi-rank line on a one-GPU box
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import glob
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # a hardware queue per launch set in flight (see insmos_amd/__init__.py)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0
SUSTAINED_FP32_MFMA_TFLOPS = 139.4   # measured, profiles/r05_mfma_rate_probe.txt (not the roofline's `peak`)


def load_window(seed, n_az, n_scans=10):
    from insmos_amd.synth import make_window
    cache = f"/tmp/insmos_s0_seed{seed}_az{n_az}_n{n_scans}.npy"
    if os.path.exists(cache):
        return np.load(cache)
    w = make_window(seed=seed, n_scans=n_scans, n_az=n_az)
    try:
        np.save(cache, w)
    except OSError:
        pass
    return w


def load_windows(seeds, n_az, n_scans=10):
    """The step's windows; the ones not cached on this box yet are ray-cast in parallel worker processes (a window takes
    ~10 s of numpy on one core)."""
    missing = [sd for sd in seeds if not os.path.exists(f"/tmp/insmos_s0_seed{sd}_az{n_az}_n{n_scans}.npy")]
    if len(missing) > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=min(len(missing), max(1, (os.cpu_count() or 2) // 2)),
                                 mp_context=mp.get_context("spawn")) as ex:
            list(ex.map(load_window, missing, [n_az] * len(missing), [n_scans] * len(missing)))
    return [load_window(sd, n_az, n_scans) for sd in seeds]


def calibrate_head(model, pts, target, cache=None, tag="", load_only=False):
    """Synthetic weights never fire the CenterHead (bias -log 99).  Shift the class bias so that about
    `target` cells pass SCORE_THRESH, giving the NMS / instance-feature stages a realistic load."""
    from insmos_amd import params as P
    eng = model.model.engine
    shift = None
    if cache and os.path.exists(cache):
        with open(cache) as f:
            shift = json.load(f).get(tag)
    if shift is None:
        if load_only:
            raise SystemExit(f"--timed-only needs the calibration cache {cache} (run bench.py once without it)")
        eng.forward_window(pts, native=False)  # the step-by-step path keeps the head map
        head = eng._head_debug["head"][:, :eng.ncls]
        best = head.max(dim=1).values
        kth = torch.topk(best, min(target, best.numel())).values[-1].item()
        shift = math.log(0.1 / 0.9) - kth
        if cache:
            try:
                old = {}
                if os.path.exists(cache):
                    with open(cache) as f:
                        old = json.load(f)
                old[tag] = shift
                with open(cache, "w") as f:
                    json.dump(old, f)
            except OSError:
                pass
    key = P.UNET_PREFIX + "center_head.conv_cls.bias"
    sd = model.state_dict()
    sd[key] = (np.asarray(sd[key], np.float32) + np.float32(shift)).astype(np.float32)
    eng._load_weights(sd)


def lib_source_hash():
    """The content hash __graft_entry__.build_product() leaves next to the library it built (sources + headers + flags), 12 hex
    digits: what ties a committed profile to the binary it was taken from."""
    try:
        with open(os.path.join(ROOT, "insmos_amd", "libinsmos_hip.so.srchash")) as f:
            return f.read().strip()[:12]
    except OSError:
        return None


def pmc_traffic(args, wpl, cfg="cfg2", profiles_dir=None, lib_hash=None):
    """HBM bytes of the conv launches from the rocprofv3 PMC passes of `bench.py --timed-only` at this configuration
    (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE in separate passes, units and gfx950 corrections as the microarch guide
    prescribes), committed under profiles/.  Returns (record, stale): record = None when there is no pass for this workload /
    launch-set size; stale = True when the latest pass was taken from ANOTHER binary than the one loaded now (its `lib_source_hash`
    differs from insmos_amd/libinsmos_hip.so.srchash, or it carries none) -- the line then prints traffic: null, traffic_stale: true."""
    suffix = "" if cfg == "cfg2" else "_" + cfg
    cands = sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "r0*_pmc_traffic%s.json" % suffix)))
    if not cands or (cfg == "cfg2" and args.n_az != 1886):
        return None, False
    path = cands[-1]   # the latest round's passes
    with open(path) as f:
        j = json.load(f)
    if int(j.get("windows_per_launch", -1)) != int(wpl):
        return None, False
    j["_path"] = path
    have = lib_hash if lib_hash is not None else lib_source_hash()
    stale = not j.get("lib_source_hash") or j.get("lib_source_hash") != have
    return j, stale


def pin_host_threads(local_rank, local_world):
    """N ranks on one host: give every rank its own slice of the cores and a small torch intra-op pool.  A rank drives its GPU
    from a handful of host threads (launch-set workers, HIP queue threads); torch's default pool (one thread per core, spinning)
    times N ranks starves exactly those threads -- the effect DESIGN.md section 5 records for a single rank."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    if local_world > 1 and len(cores) >= 2 * local_world:
        per = len(cores) // local_world
        mine = cores[local_rank * per:(local_rank + 1) * per]
        try:
            os.sched_setaffinity(0, mine)
        except (AttributeError, OSError):
            mine = cores
        torch.set_num_threads(max(1, min(8, len(mine) // 2)))
    else:
        mine = cores   # a single rank keeps the host as it is (the CPU-baseline leg uses every core)
    return len(mine)


def timed_steps(forward, batch, gts, metrics, steps, warmup, world, dev, sync, force_collectives=False):
    """The timed region of the contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync
    on both sides, the per-rank confusion counters gathered inside the region (the path's only exchange), time = MAX over
    ranks.  Device-agnostic (tests run it on CPU over gloo with a stub model): `sync` is torch.cuda.synchronize or a no-op."""
    import torch.distributed as dist
    from insmos_amd.metrics import all_gather_confusion
    grouped = world > 1 or (force_collectives and dist.is_initialized())   # force: the one-rank RCCL self-check runs every
    for _ in range(warmup):                                                # collective of the N-rank region (barrier, gather, MAX)
        forward(batch, "test")
    sync()
    if grouped:
        dist.barrier()
    cm = torch.zeros((3, 3), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, logits = forward(batch, "test")
        for lg, gt in zip(logits, gts):
            metrics.compute_confusion_matrix(lg, gt, out=cm)
    if grouped:
        sync()                          # (this rank's own steps done, before the gather makes it wait for the others)
    dt_own = time.perf_counter() - t0   # ... a straggler shows in the spread of these (rank_ms_min / rank_ms_max)
    cm_all = all_gather_confusion(cm, force=force_collectives)
    sync()
    if not grouped:
        dt_own = time.perf_counter() - t0
    if grouped:
        dist.barrier()
    dt = time.perf_counter() - t0
    timed_steps.last_rank_ms = (1000.0 * dt_own / steps, 1000.0 * dt_own / steps)
    if grouped:
        cdev = "cpu" if dist.get_backend() == "gloo" else dev
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        own = torch.tensor([dt_own, -dt_own], dtype=torch.float64, device=cdev)   # one MAX all-reduce gives max and -min
        dist.all_reduce(own, op=dist.ReduceOp.MAX)
        timed_steps.last_rank_ms = (1000.0 * -float(own[1].item()) / steps, 1000.0 * float(own[0].item()) / steps)
    value = world * steps * len(batch) / dt   # every rank runs the same number of windows per step (weak scaling)
    return dt, value, cm_all


timed_steps.last_rank_ms = (None, None)   # (min, max) over the ranks of a rank's OWN ms per step in the last timed region


def emit_result_line(out):
    """Print the ONE JSON line as the LAST line of stdout.  RCCL writes a version banner through C stdio when a communicator is
    created; on a pipe it sits in libc's buffer until exit and would land BEHIND the line (seen on the GPU box: `tail -1` of a run
    with the rccl_world1 leg was "Librccl path : ..."): flush libc first, write the line, then point fd 1 at /dev/null so that
    nothing a library flushes at teardown follows it."""
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.write(1, (json.dumps(out) + "\n").encode())
    try:
        dn = os.open(os.devnull, os.O_WRONLY)
        os.dup2(dn, 1)
        os.close(dn)
    except OSError:
        pass


def rccl_world1_selfcheck(forward, batch, gts, metrics, dev, steps=2):
    """The RCCL code of the N-rank job executed on the ONE GPU a bench box has: a process group of one rank over the `nccl`
    backend (= RCCL on ROCm), the timed region with every collective forced (barrier, all_gather of the confusion counters on
    the device tensor, MAX all-reduce of the time) and one 8 MB bucket through an async all-reduce, as insmos_amd/ddp.py sends
    it.  A sum / gather over one rank is the identity, so both results are checked bit for bit.  Never raises: the line reports
    ok / error (scripts/predict_mos.py:100-106, models/metrics.py:16-45 are what the exchange serves)."""
    import datetime
    import torch.distributed as dist
    res = {"ok": False, "backend": None}
    own = False
    try:
        if not dist.is_initialized():
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(free_port())
            dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120),
                                    device_id=torch.device(dev))
            own = True
        res["backend"] = dist.get_backend()
        res["world_size"] = dist.get_world_size()
        try:
            res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        dt, value, cm_all = timed_steps(forward, batch, gts, metrics, steps, 1, 1, dev, torch.cuda.synchronize,
                                        force_collectives=True)
        exp = torch.zeros((3, 3), dtype=torch.int64, device=dev)
        _, _, logits = forward(batch, "test")
        for lg, gt in zip(logits, gts):
            metrics.compute_confusion_matrix(lg, gt, out=exp)
        res["all_gather_equals_local_counters"] = bool(cm_all.is_cuda and torch.equal(cm_all, exp * steps))
        flat = torch.randn(2 << 20, dtype=torch.float32, device=dev)
        want = flat.clone()
        t1 = time.perf_counter()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True).wait()
        torch.cuda.synchronize()
        res["bucket_8mb_all_reduce_ms"] = round(1000.0 * (time.perf_counter() - t1), 3)
        res["all_reduce_equals_input"] = bool(torch.equal(flat, want))
        res["windows_per_s_inside_the_group"] = round(value, 2)
        res["ok"] = bool(res["backend"] == "nccl" and res["all_gather_equals_local_counters"] and res["all_reduce_equals_input"])
    except Exception as e:   # reported, not fatal: the headline above does not depend on it
        res["error"] = repr(e)[:400]
    finally:
        if own:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
    return res


def run_extras(dev_index):
    """BASELINE.json configs[3] and configs[4] inside the DEFAULT line, so that the driver's one `bench.py --gpus 1` run carries
    numbers for them that the driver itself timed: each leg is this script run as its own process (own HIP context, own
    environment defaults) with its own bracketed timed region, after the headline -- nothing of it is inside `value`.
      cfg4: 4 steps of 4 windows of the dense stress scene (launch sets of 2), parity of one quarter-size window against the oracle;
      cfg5: 4 training steps of B = 4 windows in fp32 (the reference's precision), and the bf16-operand opt-in as a second number."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extras", "--sustain-seconds", "0", "--device-index", str(dev_index)]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}

    def leg(argv, timeout):
        t0 = time.perf_counter()
        try:
            r = subprocess.run(base + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{"metric"')]
            if r.returncode != 0 or len(lines) != 1:
                return None, {"error": "rc %d: %s" % (r.returncode, r.stderr.decode()[-300:])}
            return json.loads(lines[0]), {"leg_wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:
            return None, {"error": repr(e)[:300]}

    out = {}
    # (timeouts: a few times the legs' observed wall times -- 25 / 12 / 12 s on the round-5 driver run)
    j, meta = leg(["--config", "cfg4", "--steps", "4", "--warmup", "2"], 150)
    out["cfg4"] = meta if j is None else dict(meta, **{
        "value": j["value"], "unit": "scans/s", "steps": j["steps"], "windows_per_step": j["config"]["windows_per_step"],
        "ms_per_window": j["ms_per_window"], "points_per_window": j["config"]["points_per_window"],
        "frac": j.get("roofline", {}).get("frac"), "parity_on_sample": j.get("parity_on_sample"),
        "traffic": j.get("roofline", {}).get("traffic"), "traffic_stale": j.get("roofline", {}).get("traffic_stale"),
        "workload": "BASELINE.json configs[3]: 300k pts/scan, N=10, voxel 0.05 m, 1 GPU"})
    j, meta = leg(["--config", "cfg5", "--steps", "4", "--no-cpu-baseline"], 120)
    out["cfg5"] = meta if j is None else dict(meta, **{
        "ms_per_step": j["ms_per_step"], "value": j["value"], "unit": "windows trained / s", "steps": j["steps"],
        "windows_per_step": j["config"]["windows_per_step_per_rank"], "dtype": j["dtype"], "frac": j.get("roofline", {}).get("frac"),
        "loss": j.get("loss"),
        "workload": "BASELINE.json configs[4] on ONE GPU: forward (train mode) + four losses + backward + Adam, B = 4, fp32"})
    if j is not None:
        jb, mb = leg(["--config", "cfg5", "--steps", "4", "--no-cpu-baseline", "--train-bf16"], 120)
        out["cfg5"]["bf16_operands"] = mb if jb is None else dict(mb, ms_per_step=jb["ms_per_step"], dtype=jb["dtype"], loss=jb.get("loss"),
                                                                  note="opt-in: bf16 operands in the convolutions' forward and d/dx, "
                                                                       "fp32 accumulate; d/dW, BatchNorm and losses fp32")
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_command(n, argv, port=None):
    """`python bench.py --gpus N ...` outside a launcher: the command that starts N ranks of this script on this node
    (one process per GPU, rendezvous on 127.0.0.1), exactly the form the bench contract names."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def check_enough_gpus(args, device_count=None):
    """First contact with a node that has fewer GPUs than --gpus asks for must end at once with ONE clear line -- not in N ranks
    that die on `invalid device ordinal`, and not in a rendezvous that waits for its timeout.  (A multi-rank dry run on one GPU
    names the device itself: --device-index; gloo runs need no GPU.)"""
    if args.gpus <= 1 or args.backend != "nccl" or args.device_index is not None:
        return
    if device_count is None:
        device_count = torch.cuda.device_count()
    if device_count < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {device_count} GPU(s) (torch.cuda.device_count()); "
                         f"one rank per GPU over RCCL needs {args.gpus}.  Run --gpus {max(device_count, 1)}, or a dry run of the "
                         "N-rank code on one GPU: --gpus N --device-index 0")


def maybe_spawn_ranks(args, argv):
    """--gpus N > 1 without a launcher's environment: start the N ranks here and pass their output through.  Inside a launcher
    (WORLD_SIZE set) --gpus must agree with it: an 8-GPU line must never be a 1-GPU measurement in disguise."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None:
        check_enough_gpus(args)
        if args.gpus > 1:
            import subprocess
            sys.stdout.flush()
            raise SystemExit(subprocess.call(spawn_command(args.gpus, argv)))
        return
    if int(world_env) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks; pass --gpus {world_env} "
                         "(or run `python bench.py --gpus N` without a launcher: it starts the N ranks itself)")
    check_enough_gpus(args)


def read_profile(lib):
    ids = (ctypes.c_int * 64)()
    ms = (ctypes.c_double * 64)()
    cnt = (ctypes.c_int64 * 64)()
    n = lib.insmos_prof_read(64, ids, ms, cnt)
    return {lib.insmos_prof_name(ids[i]).decode(): (ms[i], cnt[i]) for i in range(n)}


def synthetic_gt_boxes(rng, m=40):
    """(1, M, 8) ground-truth boxes [x, y, z, dx, dy, dz, yaw, class 1..3] inside the point-cloud range (what
    dataloader/datasets.py hands CenterHead.assign_targets; synthetic: there is no dataset here)."""
    g = np.zeros((1, m, 8), np.float32)
    g[0, :, 0] = rng.uniform(-55, 55, m)
    g[0, :, 1] = rng.uniform(-45, 45, m)
    g[0, :, 2] = rng.uniform(-1.5, -0.5, m)
    g[0, :, 3:6] = rng.uniform([1.5, 0.6, 1.2], [4.5, 2.0, 1.8], (m, 3))
    g[0, :, 6] = rng.uniform(-3.1, 3.1, m)
    g[0, :, 7] = rng.integers(1, 4, m)
    return g


def main_cfg5(args, rank, world, gpu, dev, host_cores):
    """BASELINE.json configs[4]: one TRAINING step = InsMOS_Model.forward(list, 'train') (models/models.py:313-345: MotionNet and
    the 3D branch in train mode, CenterHead targets + loss, MOS losses) + backward + Adam (models/models.py:188-193) over a batch
    of `--windows-per-step` windows per rank; N ranks = DDP (scripts/train.py:74-83): every rank its own windows, gradients
    all-reduced in 8 MB buckets that leave during backward (insmos_amd/ddp.py).  `value` = windows trained per second, whole job."""
    import torch.distributed as dist
    from insmos_amd import _lib, autograd, params as P
    from insmos_amd.synth import make_labels
    from insmos_amd.train_unet import InsMOSTrainer
    if args.train_bf16:
        os.environ["INSMOS_TRAIN_BF16"] = "1"
    cfg = P.default_cfg()
    B = max(1, args.windows_per_step)
    seeds = [rank * B + i for i in range(B)]
    wins = load_windows(seeds, args.n_az)
    rng = np.random.default_rng(1000 + rank)
    batch = [{"past_point_clouds": torch.from_numpy(w).to(dev),
              "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=sd_)).to(dev)],
              "gt_boxes": torch.from_numpy(synthetic_gt_boxes(rng)).to(dev)} for sd_, w in zip(seeds, wins)]
    tr = InsMOSTrainer(cfg, P.random_state_dict(cfg, 0, cls_bias=-2.0, box_w_std=0.05), device=dev)
    opt = torch.optim.Adam(list(tr.params.values()), lr=float(cfg["TRAIN"]["LR"]),
                           weight_decay=float(cfg["TRAIN"].get("WEIGHT_DECAY", 0.0)))
    red = tr.make_reducer(bucket_bytes=8 << 20, overlap=True) if world > 1 else None
    lib = _lib.load()
    last = {}

    def step():
        opt.zero_grad(set_to_none=True)
        out = tr.forward(batch, "train")
        out[0].backward()
        if red is not None:
            red.reduce(average=True)
        opt.step()
        last["loss"], last["tb"] = out[0], out[1]

    warmup = args.warmup
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * args.steps * B / dt
    loss = float(last["loss"].detach())
    out = {
        "metric": "train_scans_per_sec", "value": round(value, 3),
        "unit": "windows (N=10 scans, ~120k pts/scan) trained per second: forward + losses + backward + Adam",
        "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": warmup,
        "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 conv operands (forward, d/dx), fp32 accumulate / d/dW / BatchNorm / losses" if args.train_bf16 else "f32",
        "data": "synthetic",
        "config": {"workload": "cfg-5 (BASELINE.json configs[4], NOT the headline): training step of the whole model on synthetic "
                               "S0-style windows (seeds rank*B ..), synthetic point labels and 40 ground-truth boxes per window, "
                               "seeded random weights",
                   "windows_per_step_per_rank": B, "n_az": args.n_az, "points_per_window": int(len(wins[0])),
                   "optimizer": "Adam(lr=%g, weight_decay=%g)" % (float(cfg["TRAIN"]["LR"]), float(cfg["TRAIN"].get("WEIGHT_DECAY", 0.0))),
                   "parallelism": f"ddp{world} (per-rank batch of {B} windows; gradients all-reduced in 8 MB buckets during backward)",
                   "backend": (dist.get_backend() if world > 1 else None), "grad_buckets": (len(red.buckets) if red else 0),
                   "host_cores_per_rank": host_cores},
        "ms_per_window": round(1000.0 * dt / (args.steps * B), 3), "timed_region_s": round(dt, 3),
        "loss": round(loss, 4), "loss_terms": {k: round(float(v), 4) for k, v in dict(last["tb"][0] if isinstance(last["tb"], (list, tuple)) else last["tb"]).items()},
    }
    # ---- roofline of the convolution kernels of the step (forward + d/dx + d/dW, all on the fp32 MFMA path): executed flops
    # counted at the autograd nodes / their HIP-event time in a second, profiled pass.  EVERY rank runs the pass (its steps hold
    # the gradient collectives); rank 0 alone profiles and counts.
    nprof = 2
    if rank == 0:
        autograd.WORK_COUNTER = {}
        lib.insmos_prof_reset()
        lib.insmos_prof_enable(1)
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        prof = read_profile(lib)
        lib.insmos_prof_enable(0)
        lib.insmos_prof_reset()
        wc, autograd.WORK_COUNTER = autograd.WORK_COUNTER, None
        flops = sum(wc.get(k, 0) for k in ("forward", "dx", "dw")) / nprof
        conv_ms, conv_launches = prof.get("sparse_conv_mfma", (0.0, 0))
        conv_ms /= nprof
        ach = flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out["roofline"] = {
            "kernel": "k_sparse_conv* (forward and d/dx) + k_conv_dw_rows (d/dW) of one training step",
            "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
            "algorithmic_gflop_per_step": round(flops / 1e9, 3),
            "gflop_forward_dx_dw": [round(wc.get(k, 0) / nprof / 1e9, 3) for k in ("forward", "dx", "dw")],
            "kernel_ms_per_step": round(conv_ms, 3), "launches_per_step": conv_launches // nprof,
            "method": "executed flops = 2 * pairs * Cin * Cout per conv node, for its forward, its d/dx (where the input needs a "
                      "gradient) and its d/dW launch / HIP-event time of those launches (library profiler, second pass)"}
        out["kernel_ms_per_step"] = {k: round(v[0] / nprof, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        out["device_ms_per_step_sum"] = round(sum(v[0] for v in prof.values()) / nprof, 3)
        if world == 1 and not args.no_cpu_baseline:
            # ---- CPU baseline (a port, bounded sample): the oracle's forward of ONE smaller window + the float64 numpy backward
            # (oracle/ref_ops.sparse_conv_backward) of the 24 table convolutions of MotionNet on the oracle's own kernel maps.
            # The 3D branch's backward, BatchNorm and the losses are NOT in it: an UPPER bound of the CPU rate.
            from oracle import ref_model as M
            from oracle import ref_ops as R
            az = args.cpu_sample_az if args.cpu_sample_az != 1886 else 236
            sw = load_window(0, az)
            sd = tr.export_state_dict() if hasattr(tr, "export_state_dict") else P.random_state_dict(cfg, 0, cls_bias=-2.0, box_w_std=0.05)
            t1 = time.perf_counter()
            M.forward_window(sd, cfg, sw)
            t_fwd = time.perf_counter() - t1
            _, dbg = M.motionnet_forward(sd, sw, want_debug=True)
            t1 = time.perf_counter()
            rng2 = np.random.default_rng(0)
            nb = 0
            for name, ci, co in P.ME_BLOCKS:
                lvl = {"block1.0": 1, "block2.0": 2, "block3.0": 3, "block6.0": 2, "block7.0": 1, "block8.0": 0}[name]
                nbr = dbg["nbr81"][lvl]
                n = nbr.shape[1]
                for cin_, cout_ in ((ci, co), (co, co)):
                    taps = rng2.normal(size=(81, cin_, cout_))
                    R.sparse_conv_backward(rng2.normal(size=(n, cin_)), nbr, taps, rng2.normal(size=(n, cout_)))
                    nb += 1
            t_bwd = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": round(1.0 / (t_fwd + t_bwd), 4), "unit": "windows/s", "cores": R.num_threads(), "kind": "port",
                "sample": f"ONE window of n_az={az} ({len(sw)} points = {len(sw) / len(wins[0]):.3f} of a bench window): oracle forward "
                          f"{t_fwd:.1f} s + float64 numpy backward of MotionNet's {nb} 81-tap convolutions {t_bwd:.1f} s; 3D-branch "
                          "backward, BatchNorm and losses not included (an upper bound of the CPU rate); CPU restatement, NOT the "
                          "reference's libraries",
                "seconds": round(t_fwd + t_bwd, 2)}
        emit_result_line(out)
    if red is not None:
        red.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before the timed ones (default 3; cfg5: 4 -- the caching "
                    "allocator needs a few training steps to settle, with fewer the timed steps still hit hipMalloc)")
    ap.add_argument("--config", type=str, default="cfg2", choices=["cfg2", "cfg4", "cfg5"],
                    help="cfg2 (default, the headline): BASELINE.json configs[1], S0 windows, voxel 0.1 m.  cfg4: configs[3], the dense "
                         "stress scene -- 300k pts/scan (n_az 4710), voxel 0.05 m, BEV 250 x 300 x 640, the 100 000-voxel cap hit "
                         "(models/models.py:287); launch sets of 2 (a set's level-0 table must stay below 2 GiB), 4 windows per step.  "
                         "cfg5: configs[4], the TRAINING step (forward in train mode, the four losses, backward, Adam; fp32, "
                         "--train-bf16 for the bf16-operand opt-in), 4 windows per step and rank, gradients all-reduced in buckets "
                         "over the process group")
    ap.add_argument("--train-bf16", action="store_true", help="cfg5 only: bf16 operands in the convolutions' forward and d/dx "
                    "(fp32 accumulate; d/dW, BatchNorm and losses stay fp32) -- a labelled extra, not the fp32 line")
    ap.add_argument("--n-az", type=int, default=None, help="azimuth steps of the synthetic scan (default: 1886 = S0, 120k pts; cfg4: 4710)")
    ap.add_argument("--candidates", type=int, default=1500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-az", type=int, default=None,
                    help="azimuth steps of the CPU-baseline window (default 1886 = the bench window itself: ~10-30 s of oracle time; "
                         "cfg4: 1178 = a quarter of the azimuth steps, the full window takes the oracle minutes)")
    ap.add_argument("--calibration", type=str, default="/tmp/insmos_bench_calibration.json",
                    help="where the head-bias calibration of this workload is cached (so that a profiled run can skip it)")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling mode (rocprofv3 --kernel-trace --stats / --pmc): load the cached calibration, run the "
                         "warm-up and the timed steps and nothing else -- every kernel launch in the trace belongs to a step")
    ap.add_argument("--sustain-seconds", type=float, default=5.0, help="extra sustained loop after the timed steps")
    ap.add_argument("--layer-times", type=str, default=None, help="write per-conv-launch timings (CSV) here")
    ap.add_argument("--mixed-seeds", action="store_true",
                    help="the step's slots hold DIFFERENT windows (seeds rank*W .. rank*W+W-1: ~10 % less executed work on average "
                         "than S0) instead of W device copies of the rank's own S0-style window; reported as value_mixed_seeds by "
                         "the default run")
    ap.add_argument("--windows-per-step", type=int, default=None, help="batch items of one forward() = one step (default 32 = four launch sets of 8 in flight; cfg4: 4)")
    ap.add_argument("--conv-precision", type=int, default=0, choices=[0, 3],
                    help="EXPERIMENT ONLY (the line is then labelled as such and is not the benchmark): 3 = split-bf16 x 3 "
                         "convolutions (include/insmos_hip.h: insmos_conv_precision); 0 = exact fp32, the product path")
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for a multi-rank "
                                                               "dry run of this script on a one-GPU box)")
    ap.add_argument("--device-index", type=int, default=None, help="GPU of this rank (default LOCAL_RANK; the dry run puts "
                                                                    "every rank on GPU 0)")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg4 / cfg5 legs (`extras`) of the default cfg2 line")
    ap.add_argument("--rccl-selfcheck", action="store_true",
                    help="one GPU: run the HEADLINE's timed region inside a one-rank `nccl` (RCCL) process group with every collective "
                         "of the N-rank region forced (config.backend then says nccl); without the flag the default line still "
                         "carries a short `rccl_world1` leg")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch check (no GPU needed with --backend gloo): start / join the ranks, all-reduce a counter, print the "
                         "world the backend saw and exit -- what tests/test_host_logic.py runs through `bench.py --gpus 2` here")
    args = ap.parse_args()
    maybe_spawn_ranks(args, sys.argv[1:])
    cfg4 = args.config == "cfg4"
    if args.n_az is None:
        args.n_az = 4710 if cfg4 else 1886
    if args.cpu_sample_az is None:
        args.cpu_sample_az = 1178 if cfg4 else 1886
    if args.windows_per_step is None:
        args.windows_per_step = 4 if (cfg4 or args.config == "cfg5") else 32
    if args.warmup is None:
        args.warmup = 4 if args.config == "cfg5" else 3
    if cfg4:
        os.environ.setdefault("INSMOS_WINDOWS_PER_LAUNCH", "2")
        os.environ.setdefault("INSMOS_WINDOWS_IN_FLIGHT", "2")
        args.calibration = args.calibration + ".cfg4"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    gpu = local_rank if args.device_index is None else args.device_index
    if rank != 0:   # only rank 0 prints the result line; a library banner of another rank must not follow it on the shared pipe
        try:
            dn = os.open(os.devnull, os.O_WRONLY)
            os.dup2(dn, 1)
            os.close(dn)
        except OSError:
            pass
    if args.rendezvous_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        seen = 1
        if world > 1:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            one = torch.ones(1, dtype=torch.int64)
            if args.backend != "gloo":
                torch.cuda.set_device(gpu)
                one = one.cuda(gpu)
            dist.all_reduce(one)
            seen = int(one.item())
            assert seen == dist.get_world_size() == args.gpus, (seen, dist.get_world_size(), args.gpus)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": seen, "backend": args.backend if world > 1 else None}), flush=True)
        return
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(gpu)
        # backend nccl == RCCL on ROCm; device_id binds the communicator to this rank's GPU at init (barriers then never guess)
        kw = {"device_id": torch.device(f"cuda:{gpu}")} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    dev = f"cuda:{gpu}"
    torch.cuda.set_device(gpu)
    forced = bool(args.rccl_selfcheck and world == 1)
    if forced:
        import datetime
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device(dev))
        print(f"[bench] --rccl-selfcheck: backend {dist.get_backend()}, world {dist.get_world_size()}, device {dev}", file=sys.stderr, flush=True)
    host_cores = pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build_product()   # the HIP library only: the product build does not touch oracle/
    if world > 1:
        dist.barrier()
    from insmos_amd import _lib, params as P
    from insmos_amd.metrics import ClassificationMetrics, all_gather_confusion
    from insmos_amd.models import InsMOSNet
    from insmos_amd.synth import make_labels

    if args.config == "cfg5":
        return main_cfg5(args, rank, world, gpu, dev, host_cores)
    cfg = P.default_cfg()
    if cfg4:
        import copy
        cfg = copy.deepcopy(cfg)
        cfg["DATA"]["VOXEL_SIZE"] = [0.05, 0.05, 0.05]
        cfg["MODEL"]["MAP_TO_BEV"]["NUM_BEV_FEATURES"] = 640
        cfg["MODEL"]["DENSE_HEAD"]["TARGET_ASSIGNER_CONFIG"]["VOXEL_SIZE"] = [0.05, 0.05, 0.05]
    sd = P.random_state_dict(cfg, seed=4 if cfg4 else 0)
    # one step = one forward() over a batch of W slots.  Headline workload: every slot holds the rank's S0-style
    # window -- seed = rank, so N = 1 is the survey's scene S0 itself (cfg-2) and N ranks hold seeds 0 .. N-1 (cfg-3) -- as W
    # SEPARATE device buffers (nothing is shared between slots; the model runs each like any other window).  --mixed-seeds: slots
    # hold different windows.  InsMOS_Model keeps up to INSMOS_WINDOWS_IN_FLIGHT launch sets in flight (threads + streams).
    W = max(1, args.windows_per_step)
    seeds = [rank * W + i for i in range(W)] if args.mixed_seeds else [rank] * W
    uniq = load_windows(sorted(set(seeds)), args.n_az)
    by_seed = dict(zip(sorted(set(seeds)), uniq))
    windows = [by_seed[sd_] for sd_ in seeds]
    window = windows[0]
    pts_list = [torch.from_numpy(w).to(dev).clone() for w in windows]
    pts = pts_list[0]
    model = InsMOSNet(cfg, state_dict=sd).cuda(gpu).eval()
    if world > 1 and args.calibration:   # one cache file per rank: ranks calibrate on their own first window, concurrently
        args.calibration = f"{args.calibration}.rank{rank}"
    calibrate_head(model, pts, args.candidates, cache=args.calibration, tag=f"rank{rank}_az{args.n_az}_c{args.candidates}",
                   load_only=args.timed_only)
    eng = model.model.engine
    if args.conv_precision:
        eng.set_conv_precision(args.conv_precision)
    wpl = min(W, model.model.windows_per_launch)
    in_flight = min((W + wpl - 1) // wpl, model.model.windows_in_flight)
    batch = [{"past_point_clouds": p} for p in pts_list]
    ncur = int((window[:, 4] == 0).sum())
    gts = [torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=sd_)).to(dev) for sd_, w in zip(seeds, windows)]
    metrics = ClassificationMetrics(3, [0])

    dt, value, cm_all = timed_steps(model.forward, batch, gts, metrics, args.steps, args.warmup, world, dev,
                                    torch.cuda.synchronize, force_collectives=forced)
    if args.timed_only:
        if rank == 0:
            print(json.dumps({"timed_only": True, "value": round(value, 3), "steps": args.steps, "warmup": args.warmup,
                              "windows_per_step": W, "windows_per_launch": wpl, "launch_sets_in_flight": in_flight,
                              "windows_total": (args.steps + args.warmup) * W}), flush=True)
        if world > 1 or forced:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the same loop sustained for several seconds (visible to an SMI sampler; not the headline)
    sustained = None
    if args.sustain_seconds > 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        k = 0
        while time.perf_counter() - t1 < args.sustain_seconds:
            model.forward(batch, "test")
            k += 1
        torch.cuda.synchronize()
        sustained = k * W / (time.perf_counter() - t1)
        if world > 1:
            t = torch.tensor([sustained], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            sustained = float(t.item())

    out = {
        "metric": "scans_per_sec", "value": round(value, 3), "unit": "scans/s (windows of N=10 scans, ~120k pts/scan)",
        "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "rank_ms_min": None if timed_steps.last_rank_ms[0] is None else round(timed_steps.last_rank_ms[0], 3),
        "rank_ms_max": None if timed_steps.last_rank_ms[1] is None else round(timed_steps.last_rank_ms[1], 3),
        "dtype": "f32" if not args.conv_precision else "EXPERIMENT split-bf16x3 (NOT the product path, NOT a benchmark line)",
        "data": "synthetic",
        "config": {"workload": ("cfg-4 (BASELINE.json configs[3], NOT the headline): dense stress scene, 300k pts/scan, N=10 scans, voxel "
                                "0.05 m, the 100 000-voxel cap hit, BEV 250 x 300 x 640, full InsMOS forward" if cfg4 else
                                ("cfg-2: synthetic windows (seeds rank*W..), N=10 scans, voxel 0.1 m, full InsMOS forward; MIXED seeds, not the "
                                 "headline workload" if args.mixed_seeds else
                                 "cfg-2 (SURVEY.md 8d): every slot of the step holds the synthetic scene S0 (seed = rank: seed 0 at N=1; "
                                 "cfg-3 = seeds 0..N-1, one per rank) in its own device buffer, N=10 scans, voxel 0.1 m, full InsMOS "
                                 "forward (MotionNet 4D UNet + voxelise + UNetV2 + BEV CenterHead + NMS + instance fusion)")),
                   "window_seeds": sorted(set(seeds)),
                   "windows_per_step": W, "windows_per_launch": wpl, "launch_sets_in_flight": in_flight,
                   "n_az": args.n_az, "points_per_window": int(len(window)), "current_points": ncur,
                   "host_cores_per_rank": host_cores, "torch_threads": torch.get_num_threads(),
                   "weights": "seeded random (He-normal, occupancy-corrected), head bias calibrated to "
                              f"~{args.candidates} candidates", "parallelism": f"dp{world} (windows sharded by rank)",
                   "backend": (dist.get_backend() if (world > 1 or forced) else None)},
        "ms_per_window": round(1000.0 * dt / (args.steps * W), 3),
        "timed_region_s": round(dt, 3),
        "sustained_scans_per_sec": round(sustained, 3) if sustained is not None else None,
        # every rank's counters went through the gather: the sum holds all ranks' current points (weights are random and the
        # labels synthetic, so an IoU of these counters would mean nothing and is not printed)
        "confusion_points": int(cm_all.sum().item()),
        # what `value` is measured on, so that cross-round tooling never compares unlike figures: 1 = rounds 1-3 (slots hold
        # seeds rank*W .. rank*W+W-1, today's value_mixed_seeds), 2 = round 4 on (every slot holds S0, seed = rank: SURVEY.md 8d)
        "workload_version": 1 if args.mixed_seeds else 2,
    }

    if rank == 0:
        # ---- the reference's own calling pattern, measured while the GPU is still warm from the timed region
        # latency of ONE window, nothing else in flight
        for _ in range(5):
            eng.forward_window(pts)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            eng.forward_window(pts)
        torch.cuda.synchronize()
        out["single_window_latency_ms"] = round((time.perf_counter() - t1) * 50.0, 3)
        # the reference's own caller hands forward() ONE window (scripts/predict_mos.py:290 forces BATCH_SIZE = 1): the rate of
        # the unmodified drop-in loop, through InsMOS_Model.forward (python boundary included), nothing batched
        one = [batch[0]]
        for _ in range(3):
            model.forward(one, "test")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nb1 = 30
        for _ in range(nb1):
            model.forward(one, "test")
        torch.cuda.synchronize()
        out["value_b1"] = round(nb1 / (time.perf_counter() - t1), 3)
        # the same step with DIFFERENT windows in the slots (seeds 0 .. W-1: ~10 % less executed work on average than S0) -- an
        # extra, un-bracketed; the headline `value` is the S0 workload of SURVEY.md 8d
        if world == 1 and not args.mixed_seeds and not cfg4:
            mixed = [{"past_point_clouds": torch.from_numpy(w).to(dev)} for w in load_windows(list(range(W)), args.n_az)]
            model.forward(mixed, "test")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nmx = max(3, min(args.steps, 10))
            for _ in range(nmx):
                model.forward(mixed, "test")
            torch.cuda.synchronize()
            out["value_mixed_seeds"] = round(nmx * W / (time.perf_counter() - t1), 3)
            del mixed
        # ---- the RCCL path of the N-rank job, executed on this one GPU (one launch set of windows, every collective forced)
        if world == 1 and not forced:
            out["rccl_world1"] = rccl_world1_selfcheck(model.forward, batch[:wpl], gts[:wpl], metrics, dev)
        elif forced:
            out["rccl_world1"] = {"ok": True, "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "note": "the headline's own timed region ran inside this group with every collective forced"}
        out["rccl_world1_ok"] = bool(out.get("rccl_world1", {}).get("ok", False)) if world == 1 else None
        # ---- roofline of the dominant kernel (k_sparse_conv), HIP events on the launch stream, same workload, ONE launch
        # set at a time (nothing else on the GPU): plain per-launch durations -- what `rocprofv3 --kernel-trace --stats` of
        # `INSMOS_WINDOWS_IN_FLIGHT=1 bench.py --timed-only` shows (profiles/, tools/roofline_from_rocprof.py)
        lib = _lib.load()
        lib.insmos_forward_streams(0)   # per-kernel durations: ONE stream (the second stream would overlap the spans it measures)
        flops = flops_ref = gather = launches = comp = 0
        counts0 = None
        seen_work = {}
        for sd_, p in zip(seeds, pts_list):     # algorithmic work of every window of the step (step path, once per distinct window)
            if sd_ in seen_work:
                f_ref, wk = seen_work[sd_]
                flops_ref += f_ref
                flops += wk["flops"]
                gather += wk["gather_bytes"]
                comp += wk["compulsory_bytes"]
                continue
            eng.prune_dead_rows = False
            eng.forward_window(p, native=False)  # every row of every layer, as the reference computes them
            f_ref = eng.algorithmic_work()["flops"]
            flops_ref += f_ref
            eng.prune_dead_rows = True
            eng.bev_skip_accounting = os.environ.get("INSMOS_BEV_SKIP", "1") != "0"   # count what the runner's BEV kernels execute
            eng.forward_window(p, native=False)  # fills the launch log the EXECUTED work is counted from
            wk = eng.algorithmic_work()
            flops += wk["flops"]
            gather += wk["gather_bytes"]
            comp += wk["compulsory_bytes"]
            launches = wk["launches"]
            eng.bev_skip_accounting = False
            seen_work[sd_] = (f_ref, wk)
            if counts0 is None:
                counts0 = dict(eng.last_counts)
                eng.forward_window(p)
                counts0.update({k: eng.last_counts.get(k) for k in ("n_candidates", "n_boxes")})
        out["config"].update({"me_voxels": counts0.get("me_voxels"), "unet_voxels": counts0.get("unet_voxels"),
                              "nms_candidates": counts0.get("n_candidates"), "boxes": counts0.get("n_boxes")})
        groups = [pts_list[i:i + wpl] for i in range(0, W, wpl)]
        nprof = 2
        for g_ in groups:
            eng.forward_windows(g_)
        lib.insmos_prof_reset()
        lib.insmos_prof_enable(1)
        for _ in range(nprof):
            for g_ in groups:
                eng.forward_windows(g_)
        prof = read_profile(lib)
        lib.insmos_prof_enable(0)
        lib.insmos_prof_reset()
        lib.insmos_forward_streams(-1)
        n_win = nprof * W
        conv_ms, conv_launches = prof.get("sparse_conv_mfma", (0.0, 0))
        conv_ms_per_window = conv_ms / n_win
        total_ms = sum(v[0] for v in prof.values()) / n_win
        flops_w, flops_ref_w, gather_w = flops / W, flops_ref / W, gather / W
        ach = flops_w / (conv_ms_per_window * 1e-3) / 1e12 if conv_ms_per_window > 0 else 0.0
        traffic, traffic_stale = pmc_traffic(args, wpl, "cfg4" if cfg4 else "cfg2")
        stale_path = os.path.relpath(traffic["_path"], ROOT) if traffic else None
        if traffic_stale:
            traffic = None
        out["roofline"] = {
            "kernel": "k_sparse_conv* / k_conv_rowlane / k_conv_row32 / k_conv_tapc* + k_bev_conv3x3(_list) + k_deconv_head + the constant-input first layer (the %d "
                      "convolution launches of a launch set)" % launches,
            "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
            # against what the fp32 matrix pipe SUSTAINS on this part (pure v_mfma_f32_16x16x4_f32 issue loops hold 139.4 TFLOP/s =
            # 0.886 of nominal, an MFMA clock of ~2.13 GHz: tools/probes/mfma_rate_probe.hip, profiles/r05_mfma_rate_probe.txt) --
            # an extra; `frac` above stays on the nominal peak of the guide
            "frac_of_sustained_mfma_rate": round(ach / SUSTAINED_FP32_MFMA_TFLOPS, 4),
            # the same kernel time against the flops the REFERENCE computes for the window (every MotionNet row, every BEV site):
            # what skipping constant / unused work buys shows up here, not in `frac`
            "frac_reference_work": round(flops_ref_w / (conv_ms_per_window * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if conv_ms_per_window > 0 else 0.0,
            "traffic": round(traffic["hbm_bytes_per_launch"]) if traffic else None,
            "traffic_bytes_per_window": round(traffic["hbm_bytes_per_window"]) if traffic else None,
            "traffic_stale": bool(traffic_stale),
            "traffic_source": ("NOT measured in this run: read from the committed %s (rocprofv3 --pmc passes of `bench.py "
                               "--timed-only` at this configuration, tools/pmc_traffic.sh), taken from the binary with source hash %s "
                               "= the one loaded now" % (os.path.relpath(traffic["_path"], ROOT), traffic.get("lib_source_hash")))
                              if traffic else
                              ("%s was taken from another binary (its lib_source_hash != %s): re-run tools/pmc_traffic.sh"
                               % (stale_path, lib_source_hash())) if traffic_stale else None,
            "algorithmic_gflop_per_window": round(flops_w / 1e9, 3),
            "reference_gflop_per_window": round(flops_ref_w / 1e9, 3),
            "work_note": "achieved uses the EXECUTED flops (2*pairs*Cin*Cout of the rows actually computed): MotionNet rows nothing "
                         "consumes are skipped (DESIGN.md 3.3) and so are the BEV site groups -- 3x3 stack and fused deblock + heads -- "
                         "that hold only the layer's constant (DESIGN.md 3.10); the reference computes reference_gflop_per_window",
            "method": f"one launch set of {wpl} windows at a time on one stream: achieved = algorithmic FLOP of the conv "
                      "launches / sum of their HIP-event durations (= rocprofv3 kernel stats of INSMOS_WINDOWS_IN_FLIGHT=1 "
                      "bench.py --timed-only, tools/roofline_from_rocprof.py)",
            "kernel_ms_per_window": round(conv_ms_per_window, 4),
            "avg_launch_us": round(1000.0 * conv_ms / max(conv_launches, 1), 2),
            "launches_per_window": round(conv_launches / n_win, 2),
            # a MODEL figure, not a bandwidth measurement: the bytes a per-offset gather->GEMM->scatter with zero cache reuse
            # would move (SURVEY.md 8d, 4*pairs*(Cin+Cout) + 8*pairs) over the measured kernel time
            "gather_model_gbs": round(gather_w / (conv_ms_per_window * 1e-3) / 1e9, 1) if conv_ms_per_window else 0,
        }
        # the north star's framing ("scans/sec ... as fraction of HBM roofline"): compulsory bytes of one window (every
        # layer reads its input and its table rows and writes its output once; profiles/r01_layer_work_s0.csv) x windows/s
        # over the HBM peak.  A few per cent: the path is not HBM-bound, which is why `bound` above is "mfma".
        comp = comp / W
        out["roofline"]["hbm_view"] = {"compulsory_gb_per_window": round(comp / 1e9, 3),
                                       "achieved_gbs": round(comp / 1e9 * value / max(world, 1), 1),
                                       "frac_of_hbm_peak": round(comp / 1e9 * value / max(world, 1) / PEAK_HBM_GBS, 4),
                                       "measured_hbm_gb_per_window": round(traffic["hbm_bytes_per_window"] / 1e9, 3) if traffic else None,
                                       "note": "per GPU; compulsory = sum over the conv launches of 4*(rows_in*Cin + rows_out*Cout) "
                                               "+ 4*K*rows_out table bytes (mean over the step's windows); HBM peak 8 TB/s"}
        out["kernel_ms_per_window"] = {k: round(v[0] / n_win, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        out["launches_per_window"] = round(sum(v[1] for v in prof.values()) / n_win, 1)
        out["device_ms_per_window_sum"] = round(total_ms, 3)
        if args.layer_times:
            eng.layer_timing = []
            eng.forward_window(pts, native=False)
            torch.cuda.synchronize()
            rows = [(nm, K, ci, co, n, e0.elapsed_time(e1) * 1000.0) for nm, K, ci, co, n, e0, e1 in eng.layer_timing]
            eng.layer_timing = None
            with open(args.layer_times, "w") as f:
                f.write("layer,K,cin,cout,n_out,us\n")
                for r in rows:
                    f.write("%s,%d,%d,%d,%d,%.1f\n" % r)

        # ---- CPU baseline: the oracle (a port, not MinkowskiEngine) on the bench window itself (one window of the step)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import ref_model as M
            from oracle import ref_ops as R
            sw = windows[0] if args.cpu_sample_az == args.n_az else load_window(0, args.cpu_sample_az)
            t1 = time.perf_counter()
            ref_logits, ref_pred = M.forward_window(model.state_dict(), cfg, sw)
            cpu_s = time.perf_counter() - t1
            pr, _, lg = model.forward([{"past_point_clouds": torch.from_numpy(sw).to(dev)}], "test")
            got = lg[0].cpu().numpy()
            lab, _ = R.output_stage(got)
            lab_ref, _ = R.output_stage(ref_logits)
            out["cpu_baseline"] = {
                "value": round(1.0 / cpu_s, 4), "unit": "scans/s", "cores": R.num_threads(), "kind": "port",
                "sample": f"ONE window of the bench step (seed 0, n_az={args.cpu_sample_az}, {len(sw)} points = "
                          f"{len(sw) / len(window):.3f} of a bench window), oracle/ref_model.forward_window "
                          "(numpy + OpenMP C, torch-CPU for the dense BEV convs); CPU restatement, NOT MinkowskiEngine",
                "seconds": round(cpu_s, 2),
            }
            out["parity_on_sample"] = {"max_abs_logit_diff": float(np.abs(got - ref_logits).max()),
                                       "labels_equal": bool((lab == lab_ref).all()),
                                       "label_mismatches": int((lab != lab_ref).sum()), "points": int(len(lab)),
                                       "boxes_oracle_gpu": [int(len(ref_pred["pred_boxes"])), int(len(pr[0][0]["pred_boxes"]))]}
        if world == 1 and not (args.no_extras or cfg4 or args.mixed_seeds or args.conv_precision or args.n_az != 1886):
            torch.cuda.empty_cache()
            # (the headline is complete here: should a leg below hang past the caller's patience, it is already on stderr and in a file)
            stash = json.dumps(out)
            print("[bench] headline before the extras legs: " + stash, file=sys.stderr, flush=True)
            try:
                with open("/tmp/insmos_bench_headline.json", "w") as f:
                    f.write(stash + "\n")
            except OSError:
                pass
            out["extras"] = run_extras(gpu)
        emit_result_line(out)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()



if __name__ == "__main__":
    main()
