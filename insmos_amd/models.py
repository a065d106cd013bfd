"""Drop-in surface of the reference's `models.models` for the inference path.

`InsMOSNet` mirrors the Lightning wrapper the reference's driver uses (models/models.py:27-59,
scripts/predict_mos.py:326-328,405-407,434): `InsMOSNet.load_from_checkpoint(ckpt, hparams=cfg)`,
`.cuda()`, `.eval()`, `.forward(list_of_dicts, 'test')`.  `InsMOS_Model` mirrors
models/models.py:269-376: batch items are processed one after another and three lists are returned:
`(pred_dicts_list, recall_dicts_list, point_logits_list)` with
  pred_dicts_list[i][0] = {"pred_boxes" (K,7) fp32, "pred_scores" (K,) fp32, "pred_labels" (K,) int64}
  recall_dicts_list[i]  = {}           (no gt_boxes on the test path, post_process.py:68-69)
  point_logits_list[i]  = (Ncur, 3) fp32 raw MOS logits, rows in the order of the t == 0 input rows.
Only Model_mode == 'test' is implemented (the north-star path); 'train'/'eval' need ground truth and
the training harness, which are out of scope.  No pytorch_lightning is needed: a Lightning .ckpt is a
torch-pickled dict with "hyper_parameters" and "state_dict" (models/models.py:30,52).
"""
import os

import torch
import yaml

from .engine import Engine
from . import params as P

_DEFAULT_SEMANTIC = {
    # config/semantic-kitti-mos.yaml:115-160 (learning_map_inv / learning_ignore) of the reference
    "learning_map_inv": {0: 0, 1: 9, 2: 251},
    "learning_ignore": {0: True, 1: False, 2: False},
}


def load_semantic_config(cfg):
    path = cfg.get("DATA", {}).get("SEMANTIC_CONFIG_FILE", None)
    if path and os.path.exists(path):
        with open(path) as f:
            return yaml.safe_load(f)
    return dict(_DEFAULT_SEMANTIC)


class InsMOS_Model:
    """models/models.py:269-376 (test mode)."""

    def __init__(self, cfg, n_mos_classes, ignore_index, state_dict, device="cuda:0", quirk_exact=True):
        self.cfg = cfg
        self.mos_class = n_mos_classes
        self.ignore_index = ignore_index
        self.state_dict_ref = state_dict
        self.device = device
        self.quirk_exact = quirk_exact
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            if not torch.cuda.is_available():
                raise RuntimeError("insmos_amd needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
            self._engine = Engine(self.cfg, self.state_dict_ref, self.device, quirk_exact=self.quirk_exact)
        return self._engine

    def forward(self, list_batch_dict, Model_mode):
        if Model_mode != "test":
            raise NotImplementedError("insmos_amd implements the inference path: Model_mode == 'test'")
        preb_dict_list, recall_dict_list, preb_mos_lable_list = [], [], []
        for batch_dict in list_batch_dict:  # sequential, as models/models.py:313
            logits, pred = self.engine.forward_window(batch_dict["past_point_clouds"])
            preb_dict_list.append([pred])
            recall_dict_list.append({})
            preb_mos_lable_list.append(logits)
        return preb_dict_list, recall_dict_list, preb_mos_lable_list

    __call__ = forward


class InsMOSNet:
    """models/models.py:27-59 without Lightning."""

    def __init__(self, hparams, state_dict=None, seed=0):
        self.hparams = hparams
        self.cfg = hparams
        self.id = hparams["EXPERIMENT"]["ID"]
        self.dt_prediction = hparams["MODEL"]["DELTA_T_PREDICTION"]
        self.n_past_steps = hparams["MODEL"]["N_PAST_STEPS"]
        self.semantic_config = load_semantic_config(hparams)
        self.n_mos_classes = len(self.semantic_config["learning_map_inv"])
        self.ignore_index = [k for k, ig in self.semantic_config["learning_ignore"].items() if ig]
        if state_dict is None:
            state_dict = P.random_state_dict(hparams, seed)
        missing = [k for k in P.param_spec(hparams) if k not in state_dict]
        if missing:
            raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
        self._state_dict = state_dict
        self._device = "cuda:0"
        self.model = InsMOS_Model(hparams, self.n_mos_classes, self.ignore_index, state_dict, self._device)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, hparams=None, map_location="cpu", **kw):
        ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        cfg = hparams if hparams is not None else ckpt["hyper_parameters"]
        return cls(cfg, state_dict=ckpt["state_dict"])

    def state_dict(self):
        return self._state_dict

    def cuda(self, device=None):
        if device is not None:
            self._device = f"cuda:{device}" if isinstance(device, int) else str(device)
            self.model.device = self._device
            self.model._engine = None
        _ = self.model.engine  # build now: weights are packed and uploaded once
        return self

    def to(self, device):
        return self.cuda(device)

    def eval(self):
        return self

    def forward(self, batch_data, Model_mode):
        return self.model(batch_data, Model_mode)

    __call__ = forward


def save_checkpoint(path, cfg, state_dict):
    """Write a Lightning-shaped checkpoint ({'hyper_parameters', 'state_dict'}) for tests/tools."""
    torch.save({"hyper_parameters": cfg,
                "state_dict": {k: torch.as_tensor(v) for k, v in state_dict.items()}}, path)
