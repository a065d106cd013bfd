#!/usr/bin/env python3
"""tools/roofline_from_rocprof.py -- the roofline fraction of the convolution kernels from a rocprofv3 kernel-stats CSV.

    INSMOS_WINDOWS_IN_FLIGHT=1 rocprofv3 --kernel-trace --stats -d DIR -o prof --output-format csv -- \
        python bench.py --timed-only --steps K --warmup W          # prints {"windows_total": ...}
    python tools/roofline_from_rocprof.py DIR/.../prof_kernel_stats.csv --windows (K+W)*windows_per_step \
        [--gflop-per-window G | --bench-json bench.json] [--layer-work profiles/r01_layer_work_s0.csv]

With one launch set in flight the kernels of the trace run one after the other, so the summed duration of the convolution
kernels IS the time the GPU spent on them: achieved = algorithmic FLOP per window / (conv time per window), frac = achieved /
157.3 TFLOP/s (dense fp32 MFMA peak of the MI355X, MI355X_MICROARCH.md).  The algorithmic work comes from bench.py's own
line (`roofline.algorithmic_gflop_per_window`, counted from the GPU's kernel maps) or, independently, from the oracle's
kernel maps (tools/layer_work.py -> profiles/r01_layer_work_s0.csv, column flops_executed, window seed 0)."""
import argparse
import csv
import json

PEAK = 157.3
CONV = ("k_sparse_conv", "k_conv_rowlane", "k_conv_row32", "k_conv_tapc", "k_conv_wide", "k_deconv_head", "k_resolve_taps<2, 1, 1>", "k_parent_cubes", "k_const_conv125", "k_bev_conv", "k_bev_group_lists")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats_csv")
    ap.add_argument("--windows", type=int, required=True, help="windows processed during the trace: (steps + warmup) * windows_per_step")
    ap.add_argument("--gflop-per-window", type=float, default=None)
    ap.add_argument("--bench-json", type=str, default=None)
    ap.add_argument("--layer-work", type=str, default=None)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.stats_csv)))
    conv_ns = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in CONV))
    conv_calls = sum(int(r["Calls"]) for r in rows if any(k in r["Name"] for k in CONV))
    all_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    all_calls = sum(int(r["Calls"]) for r in rows)
    g = a.gflop_per_window
    src = "--gflop-per-window"
    if g is None and a.bench_json:
        line = [l for l in open(a.bench_json) if l.lstrip().startswith("{")][-1]
        g = json.loads(line)["roofline"]["algorithmic_gflop_per_window"]
        src = "bench.py roofline.algorithmic_gflop_per_window"
    if g is None and a.layer_work:
        g = sum(float(r["flops_executed"]) for r in csv.DictReader(open(a.layer_work))) / 1e9
        src = "layer_work flops_executed (oracle kernel maps, window seed 0)"
    if g is None:
        raise SystemExit("need --gflop-per-window, --bench-json or --layer-work")
    ms = conv_ns / 1e6 / a.windows
    ach = g / ms  # GFLOP / ms = TFLOP/s
    out = {"conv_ms_per_window": round(ms, 4), "conv_launches_per_window": round(conv_calls / a.windows, 2),
           "conv_avg_launch_us": round(conv_ns / 1e3 / max(conv_calls, 1), 2), "gflop_per_window": round(g, 3), "work_from": src,
           "achieved_tflops": round(ach, 3), "peak_tflops": PEAK, "frac": round(ach / PEAK, 4),
           "all_kernels_ms_per_window": round(all_ns / 1e6 / a.windows, 4),
           "all_launches_per_window": round(all_calls / a.windows, 1),
           "non_conv_ms_per_window": round((all_ns - conv_ns) / 1e6 / a.windows, 4), "windows": a.windows}
    print(json.dumps(out) if a.json else json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
