"""-m gpu: the HIP path against what the REFERENCE's OWN code produced (tests/golden/wiring.npz, train_wiring.npz: the
reference's model classes run as written over the oracle-backed stand-ins, see oracle/shims/README.md) -- directly, without
the oracle in between.  (File name: runs last; these cross-checks were added after round 1's GPU budget was spent, the
inference one follows from two checks that are already green -- HIP == oracle to 3e-5, oracle == golden to 5e-6.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hip_forward_vs_reference_model_code(golden_dir):
    from insmos_amd import params as P
    from insmos_amd.models import InsMOSNet
    from insmos_amd.synth import make_window
    from model_util import detecting_state_dict
    from oracle import ref_ops as R
    g = np.load(os.path.join(golden_dir, "wiring.npz"))
    cfg = P.default_cfg()
    window = make_window(seed=21, n_scans=3, n_az=160)
    sd = detecting_state_dict(cfg, window, seed=4, target=(60, 200))
    model = InsMOSNet(cfg, state_dict=sd).cuda().eval()
    with torch.no_grad():
        preds, _, logits = model.forward([{"past_point_clouds": torch.from_numpy(window).cuda()}], "test")
    lg = logits[0].cpu().numpy()
    ref = g["logits"]
    assert lg.shape == ref.shape
    print("max |logit - reference-code golden| = %.2e" % float(np.abs(lg - ref).max()))
    np.testing.assert_allclose(lg, ref, atol=1e-3, rtol=0)                        # the north star's logit bar
    lab, _ = R.output_stage(lg)
    lab_ref, _ = R.output_stage(ref)
    top2 = np.sort(ref[:, 1:], axis=1)
    decided = (top2[:, -1] - top2[:, -2]) > 2e-3                                   # argmax is not a coin toss there
    assert decided.mean() > 0.99
    np.testing.assert_array_equal(lab[decided], lab_ref[decided])
    pb = preds[0][0]["pred_boxes"].cpu().numpy()
    gb = g["pred_boxes"]
    assert len(pb) == len(gb) >= 5
    dist = np.abs(pb[:, None, :] - gb[None, :, :]).max(2)
    assert (dist.min(1) < 1e-3).all() and len(set(dist.argmin(1).tolist())) == len(gb)   # the same boxes
    np.testing.assert_array_equal(preds[0][0]["pred_labels"].cpu().numpy(), g["pred_labels"][dist.argmin(1)])


def test_hip_training_step_vs_reference_training_code(golden_dir):
    import zlib
    from insmos_amd import params as P
    from insmos_amd.synth import make_window
    from insmos_amd.train_unet import InsMOSTrainer
    g = np.load(os.path.join(golden_dir, "train_wiring.npz"))
    cfg = P.default_cfg()
    window = make_window(seed=21, n_scans=3, n_az=96)
    sd = P.random_state_dict(cfg, 9, cls_bias=-1.0, box_w_std=0.05)
    tr = InsMOSTrainer(cfg, sd)
    batch = [{"past_point_clouds": torch.from_numpy(window).cuda(), "past_labels": [torch.from_numpy(g["gt_labels"]).cuda()],
              "gt_boxes": torch.from_numpy(g["gt_boxes"]).cuda()}]
    loss, tb, _, _ = tr.forward(batch, "train")
    loss.backward()
    ref_cls, ref_loc, ref_mos, ref_motion, ref_total = (float(v) for v in g["losses"])
    assert abs(tb[0]["rpn_loss_cls"] - ref_cls) < 1e-3 * ref_cls and abs(tb[0]["rpn_loss_loc"] - ref_loc) < 1e-3 * ref_loc
    assert abs(tb[0]["loss_mos"] - ref_mos) < 1e-3 * ref_mos and abs(tb[0]["loss_motion_encoder"] - ref_motion) < 1e-3 * ref_motion
    grads = {str(n): (float(g["grad_norms"][i]), g["grad_samples"][i]) for i, n in enumerate(g["grad_names"])}
    bad = []
    for stem, v in tr.unet.params.items():
        name = P.UNET_PREFIX + stem
        gr = tr.unet.to_reference_layout(stem, v.grad).astype(np.float64).reshape(-1)
        idx = np.random.default_rng(zlib.crc32(name.encode())).integers(0, gr.size, 12)
        scale = max(float(np.abs(gr).max()), 1e-12)
        if abs(np.sqrt((gr * gr).sum()) - grads[name][0]) > 1e-2 * grads[name][0] or np.abs(gr[idx] - grads[name][1]).max() > 1e-2 * scale:
            bad.append(name)
    assert not bad, bad[:8]
