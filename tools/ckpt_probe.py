#!/usr/bin/env python3
"""tools/ckpt_probe.py -- first contact with a REAL InsMOS Lightning checkpoint (README.md:146 of the reference; none ships
with this repository, so every layout conversion in insmos_amd/params.py is pinned to the reference's module definitions and
to dependency knowledge only -- SURVEY.md 8c, "parity unpinned" for the MinkowskiEngine / spconv primitives).

    python tools/ckpt_probe.py path/to/InsMOS.ckpt        # CPU only, no GPU needed

What it checks, tensor by tensor, against insmos_amd.params.param_spec(cfg):
  * presence and SHAPE of all tensors of the inference path (MinkowskiEngine kernels (K_vol, Cin, Cout) -- (Cin, Cout) for the
    1x1 convs --, spconv 2.3.6 weights (Cout, kz, ky, kx, Cin), torch Conv2d / ConvTranspose2d / BatchNorm / Linear);
  * shapes that are consistent with ANOTHER layout of the same layer (e.g. an spconv 1.x / 2.0 checkpoint stores
    (kz, ky, kx, Cin, Cout)) are named as such -- loading them as they are would silently permute the taps;
  * what it CANNOT check: the order of MinkowskiEngine's K_vol axis (params.me_kernel_to_taps assumes the x-fastest
    kernel-region enumeration with even-kernel offsets {0, 1}; two orders of the same 81 rows have the same shape).  The probe
    therefore prints the symmetric-kernel heuristic below: a trained 3x3x3x3 kernel is usually NOT symmetric under
    reversing the tap axis, so this only reports numbers for a human to look at next to a reference run.
Exit status 0 = every tensor present with the expected shape.
"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import params as P  # noqa: E402


def probe(ckpt, cfg=None, out=print):
    """Returns (n_ok, problems) for a loaded checkpoint dict ({'state_dict', 'hyper_parameters'})."""
    sd = ckpt["state_dict"]
    cfg = cfg or ckpt.get("hyper_parameters") or P.default_cfg()
    spec = P.param_spec(cfg)
    problems, n_ok = [], 0
    for name, (shape, kind) in spec.items():
        if name not in sd:
            problems.append((name, "MISSING", None, shape))
            continue
        got = tuple(int(d) for d in np.shape(sd[name]))
        if got == tuple(shape):
            n_ok += 1
            continue
        why = "shape"
        if kind == "spconv" and len(got) == 5 and len(shape) == 5:
            co, kz, ky, kx, ci = shape
            if got == (kz, ky, kx, ci, co):
                why = "spconv 1.x / 2.0 layout (kz, ky, kx, Cin, Cout): needs permute(4, 0, 1, 2, 3) before loading"
            elif got == (co, ci, kz, ky, kx):
                why = "torch Conv3d layout (Cout, Cin, kz, ky, kx): needs permute(0, 2, 3, 4, 1) before loading"
        if kind == "me" and len(got) == 3 and len(shape) == 3 and got == (shape[0], shape[2], shape[1]) and shape[1] != shape[2]:
            why = "MinkowskiEngine kernel with Cin / Cout swapped (a transposed-conv kernel stored the other way round?)"
        problems.append((name, why, got, tuple(shape)))
    extra = [k for k in sd if k not in spec and not k.endswith("num_batches_tracked")]
    out(f"{n_ok} / {len(spec)} tensors of the inference path present with the expected shape; "
        f"{len(problems)} problems; {len(extra)} tensors in the checkpoint the path does not read")
    for name, why, got, want in problems[:40]:
        out(f"  {name}: {why}; checkpoint {got}, expected {want}")
    # tap-order heuristic for the 81-tap MinkowskiEngine kernels (see the module docstring)
    for name, (shape, kind) in spec.items():
        if kind == "me" and len(shape) == 3 and shape[0] == 81 and name in sd:
            w = np.asarray(sd[name], np.float32)
            if w.shape != tuple(shape):
                continue
            centre = float(np.abs(w[40]).mean())
            rest = float(np.abs(np.delete(w, 40, axis=0)).mean())
            out(f"  [tap order, unverifiable here] {name}: |centre tap 40| / |other taps| = {centre / max(rest, 1e-12):.2f} "
                "(trained kernels usually weight the centre tap highest: a ratio < 1 on most layers would hint at another "
                "K_vol order than the x-fastest one params.me_kernel_to_taps assumes)")
            break
    return n_ok, problems


def main():
    import torch
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    ckpt = torch.load(sys.argv[1], map_location="cpu", weights_only=False)
    n_ok, problems = probe(ckpt)
    return 0 if not problems else 1


if __name__ == "__main__":
    sys.exit(main())
