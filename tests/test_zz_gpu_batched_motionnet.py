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
            # a row's value does not depend on the rows sharing its tile (tap-split tiles hand taps to waves by tap index):
            # the same bits whatever the batch
            assert a.shape == b.shape
            assert torch.equal(a, b), float((a - b).abs().max())
    # the batched coordinate set is the union of the windows' sets, window index folded into t
    n_single = []
    for w in wins:
        eng.motionnet(w)
        n_single.append(list(eng.last_counts["me_voxels"]))
    eng.motionnet_windows(wins)
    assert list(eng.last_counts["me_voxels"]) == [int(v) for v in np.sum(n_single, 0)]
