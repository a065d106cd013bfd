#!/usr/bin/env python3
"""tools/lds_probe.py -- where a workgroup of the LDS-staged 81-tap kernel spends its cycles (probe build, INSMOS_CONV_LDS=2)."""
import ctypes, os, sys
os.environ["INSMOS_CONV_LDS"] = "2"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from insmos_amd import params as P
from insmos_amd.models import InsMOSNet

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    wins = [torch.from_numpy(w).cuda() for w in bench.load_windows(list(range(B)), 1886)]
    eng = model.model.engine
    lib = eng.lib
    eng.forward_windows(wins)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.insmos_debug_conv_lds_stats(out, 1)
    eng.forward_windows(wins)
    torch.cuda.synchronize()
    lib.insmos_debug_conv_lds_stats(out, 1)
    v = list(out)
    wg, groups = max(v[12], 1), max(v[8], 1)
    names = ["A idx loads + reductions", "barrier 1", "C classify", "D staging", "barrier 4", "E contraction"]
    tot = sum(v[:6])
    print("workgroups %d, (workgroup, dt) groups %d, raw groups %d (%.1f %%), overflow rows per group %.1f, active taps per wave-group %.1f" %
          (v[12], v[8], v[9], 100.0 * v[9] / groups, v[10] / groups, v[11] / (4.0 * groups)))
    for n, c in zip(names, v[:6]):
        print("  %-26s %9.0f cycles per workgroup (%.1f %%)" % (n, c / wg, 100.0 * c / max(tot, 1)))
    print("  total %.0f cycles per workgroup of wave 0" % (tot / wg))

if __name__ == "__main__":
    main()
