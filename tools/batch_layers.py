#!/usr/bin/env python3
"""tools/batch_layers.py -- per-layer time and TFLOP/s of the convolution launches of ONE launch set of B windows.

The native runner launches the convolutions in the order the step path (Engine.forward_window(native=False)) does, so the
per-launch HIP-event durations of a launch set line up with the step path's layer list; the algorithmic flops of a launch
are the sum over the set's windows of 2 * pairs * Cin * Cout of that layer (rows actually computed).

    python tools/batch_layers.py [B=4] [out.csv]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import _lib, params as P  # noqa: E402
from insmos_amd.engine import NbrTable  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402


def layer_work(eng):
    """[(name, K, cin, cout, rows, flops)] of the last step-path window."""
    rows = []
    cache = {}
    for nbr, n_out, layer, row0 in eng._conv_log:
        if nbr is None:
            pairs = n_out - row0
        else:
            key = (nbr.data_ptr(), row0)
            if key not in cache:
                tab = nbr.nbr if isinstance(nbr, NbrTable) else nbr
                cache[key] = int((tab[:, row0:] >= 0).sum().item())
            pairs = cache[key]
        if layer.name in getattr(eng, "_bev_exec_pairs", {}):      # what the runner's constant-region skipping executes
            pairs = eng._bev_exec_pairs[layer.name]
        if layer.name == "head" and rows and rows[-1][0] == "deconv":   # one fused launch (k_deconv_head)
            nm, K, ci, co, r, fl = rows[-1]
            rows[-1] = ("deconv+head", K, ci, co, r, fl + pairs * layer.flops_per_pair)
            continue
        rows.append((layer.name, layer.K, layer.cin, layer.cout, n_out - row0, pairs * layer.flops_per_pair))
    return rows


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    out_csv = sys.argv[2] if len(sys.argv) > 2 else None
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    wins = [torch.from_numpy(w).cuda() for w in bench.load_windows(list(range(B)), 1886)]
    bench.calibrate_head(model, wins[0], 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    eng = model.model.engine
    if os.environ.get("INSMOS_CONV_PRECISION"):   # per-layer times of the split-bf16 x 3 EXPERIMENT (tools/bf16x3_experiment.py)
        eng.set_conv_precision(int(os.environ["INSMOS_CONV_PRECISION"]))
    lib = eng.lib
    per_win = []
    eng.bev_skip_accounting = os.environ.get("INSMOS_BEV_SKIP", "1") != "0"
    for w in wins:
        eng.forward_window(w, native=False)
        per_win.append(layer_work(eng))
    eng.bev_skip_accounting = False
    n_layers = len(per_win[0])
    kk = next(k for k in range(64) if lib.insmos_prof_name(k) == b"sparse_conv_mfma")

    def measure(reps=3):
        eng.forward_windows(wins)
        lib.insmos_prof_reset()
        lib.insmos_prof_enable(1)
        for _ in range(reps):
            eng.forward_windows(wins)
        cap = 4096
        ms = (ctypes.c_double * cap)()
        meta = (ctypes.c_int64 * (4 * cap))()
        n = lib.insmos_prof_read_spans(kk, cap, ms, meta)
        lib.insmos_prof_enable(0)
        lib.insmos_prof_reset()
        assert n == reps * n_layers, (n, reps, n_layers)
        t = np.array(ms[:n]).reshape(reps, n_layers).mean(0) * 1e3  # us
        m = np.array(meta[:4 * n]).reshape(reps, n_layers, 4)[0]
        return t, m

    # BATCH_LAYERS_ROWLANE="0:1,1:1,3:1,7:2": the same launch set under several settings of the row-per-lane kernel
    # (insmos_debug_conv_rowlane(mode, rows_per_lane)) in ONE process, interleaved twice -- one column per setting
    variants = os.environ.get("BATCH_LAYERS_ROWLANE")
    if variants:
        # (mode | dbg << 4 selects a probe build; an optional third field = INSMOS_ROWLANE_LDS bytes under INSMOS_ROWLANE_PROBE=1)
        vs = [tuple(int(v) for v in item.split(":")) for item in variants.split(",")]
        cols = {v: [] for v in vs}
        for _ in range(2):
            for v in vs:
                assert lib.insmos_debug_conv_rowlane(v[0], v[1]) == 0
                os.environ["INSMOS_ROWLANE_LDS"] = str(v[2]) if len(v) > 2 else "0"
                cols[v].append(measure(2)[0])
        lib.insmos_debug_conv_rowlane(-1, 0)
        tv = {v: np.minimum(*cols[v]) for v in vs}
        hdr = "layer,K,cin,cout,rows," + ",".join("us_mode%d_dbg%d_rpl%d%s" % (v[0] & 15, v[0] >> 4, v[1], "_lds%d" % v[2] if len(v) > 2 else "")
                                                   for v in vs)
        lines = [hdr]
        for i in range(n_layers):
            name, K, cin, cout, _, _ = per_win[0][i]
            rows = sum(pw[i][4] for pw in per_win)
            lines.append("%s,%d,%d,%d,%d," % (name, K, cin, cout, rows) + ",".join("%.1f" % tv[v][i] for v in vs))
        lines.append("TOTAL,,,,," + ",".join("%.1f" % float(tv[v].sum()) for v in vs))
        txt = "\n".join(lines)
        print(txt)
        if out_csv:
            with open(out_csv, "w") as f:
                f.write(txt + "\n")
        return

    # BATCH_LAYERS_ENV="A=0;A=1,B=2;...": the same launch set under several ENVIRONMENT settings the library reads per call (e.g.
    # INSMOS_CONV_TAPC, INSMOS_TAPC_DBG), in ONE process, interleaved BATCH_LAYERS_ROUNDS (3) times; per layer the minimum over the
    # rounds -- box-to-box and run-to-run differences of +-20 % per layer make separate processes useless for an A/B
    env_variants = os.environ.get("BATCH_LAYERS_ENV")
    if env_variants:
        vs = [v for v in env_variants.split(";") if v]
        rounds = int(os.environ.get("BATCH_LAYERS_ROUNDS", "3"))
        cols = {v: [] for v in vs}
        keys = sorted({kv.split("=")[0] for v in vs for kv in v.split(",")})
        for _ in range(rounds):
            for v in vs:
                for k in keys:
                    os.environ.pop(k, None)
                for kv in v.split(","):
                    k, val = kv.split("=")
                    os.environ[k] = val
                cols[v].append(measure(2)[0])
        for k in keys:
            os.environ.pop(k, None)
        tv = {v: np.min(np.stack(cols[v]), axis=0) for v in vs}
        lines = ["layer,K,cin,cout,rows," + ",".join("us[%s]" % v.replace(",", "&") for v in vs)]
        for i in range(n_layers):
            name, K, cin, cout, _, _ = per_win[0][i]
            rows = sum(pw[i][4] for pw in per_win)
            lines.append("%s,%d,%d,%d,%d," % (name, K, cin, cout, rows) + ",".join("%.1f" % tv[v][i] for v in vs))
        lines.append("TOTAL,,,,," + ",".join("%.1f" % float(tv[v].sum()) for v in vs))
        txt = "\n".join(lines)
        only = os.environ.get("BATCH_LAYERS_ONLY_K")
        print("\n".join(l for l in lines if not only or l.split(",")[1] in ("K", only, "")))
        if out_csv:
            with open(out_csv, "w") as f:
                f.write(txt + "\n")
        return

    t, m = measure()
    lines = ["layer,K,cin,cout,rows,us,gflop,tflops,pct_time"]
    tot_us = float(t.sum())
    tot_fl = 0
    for i in range(n_layers):
        name, K, cin, cout, _, _ = per_win[0][i]
        fl = sum(pw[i][5] for pw in per_win)
        rows = sum(pw[i][4] for pw in per_win)
        assert (int(m[i][0]), int(m[i][2])) == (K, cout) or name == "deconv+head", (name, m[i], K, cin, cout)
        tot_fl += fl
        lines.append("%s,%d,%d,%d,%d,%.1f,%.3f,%.1f,%.1f" % (name, K, cin, cout, int(m[i][3]), t[i], fl / 1e9, fl / t[i] / 1e6,
                                                            100.0 * t[i] / tot_us))
    lines.append("TOTAL,,,,,%.1f,%.3f,%.1f,100.0" % (tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6))
    txt = "\n".join(lines)
    print(txt)
    print("# B = %d windows per launch set: %.3f ms conv per window, %.1f TFLOP/s" % (B, tot_us / 1e3 / B, tot_fl / tot_us / 1e6))
    if out_csv:
        with open(out_csv, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
