"""-m gpu: MotionNet of several windows in one set of launches (Engine.motionnet_windows; DESIGN.md section 2
step 1) must give every window the bits it gets alone."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [2, 3, 8])
def test_batched_motionnet_matches_single_windows(B):
    from insmos_amd import params as P
    from insmos_amd.engine import Engine
    from insmos_amd.synth import make_window
    cfg = P.default_cfg()
    eng = Engine(cfg, P.random_state_dict(cfg, 3), "cuda:0")
    wins = [torch.from_numpy(make_window(seed=40 + i, n_scans=10 if i % 2 == 0 else 4, n_az=96 + 16 * i)).cuda() for i in range(B)]
    single = [eng.motionnet(w).clone() for w in wins]
    for prune in (True, False):
        eng.prune_dead_rows = prune
        batched = eng.motionnet_windows(wins)
        assert len(batched) == B
        for a, b in zip(single, batched):
            # per-row arithmetic is the same; the conv kernel's tile / split variant is picked per launch, and a split
            # variant sums its partial results in a different (still fixed) order, so equality is to fp32 round-off
            # unless every layer happens to pick the same variant at both sizes
            assert a.shape == b.shape
            assert torch.equal(a[:, :4], b[:, :4])
            assert float((a - b).abs().max()) < 2e-5, float((a - b).abs().max())
            print("B", B, "prune", prune, "bitwise", bool(torch.equal(a, b)))
    # the batched coordinate set is the union of the windows' sets, window index folded into t
    n_single = []
    for w in wins:
        eng.motionnet(w)
        n_single.append(list(eng.last_counts["me_voxels"]))
    eng.motionnet_windows(wins)
    assert list(eng.last_counts["me_voxels"]) == [int(v) for v in np.sum(n_single, 0)]
