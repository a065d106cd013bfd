"""Drop-in surface of the reference's `models.models` for the inference path.

`InsMOSNet` mirrors the Lightning wrapper the reference's driver uses (models/models.py:27-59,
scripts/predict_mos.py:326-328,405-407,434): `InsMOSNet.load_from_checkpoint(ckpt, hparams=cfg)`,
`.cuda()`, `.eval()`, `.forward(list_of_dicts, 'test')`.  `InsMOS_Model` mirrors
models/models.py:269-376: three lists are returned (batch items may be in flight concurrently, see the class):
`(pred_dicts_list, recall_dicts_list, point_logits_list)` with
  pred_dicts_list[i][0] = {"pred_boxes" (K,7) fp32, "pred_scores" (K,) fp32, "pred_labels" (K,) int64}
  recall_dicts_list[i]  = {}           (no gt_boxes on the test path, post_process.py:68-69)
  point_logits_list[i]  = (Ncur, 3) fp32 raw MOS logits, rows in the order of the t == 0 input rows.
Model_mode 'test' (the north-star path) and 'eval' (the validation step's forward: the same path plus the two MOS losses,
models/models.py:347-353) run on the inference engine; 'train' (models/models.py:313-345) is delegated to the training twin
(insmos_amd/train_unet.py: InsMOSTrainer) and returns the reference's four values.  No pytorch_lightning is needed: a Lightning .ckpt is a
torch-pickled dict with "hyper_parameters" and "state_dict" (models/models.py:30,52).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import torch
import yaml

from .engine import Engine, MAX_WINDOWS_PER_LAUNCH
from .metrics import generate_recall_record
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
    """models/models.py:269-376 (test mode).

    The reference walks the batch list sequentially (models/models.py:313).  Here the list is cut into groups of
    `windows_per_launch` items and every group runs as ONE set of launches (Engine.forward_windows: the windows of a group
    share every kernel launch and every count read-back), because one window leaves a large part of an MI355X idle (few
    tiles per SIMD in the deep layers, host syncs).  Up to `windows_in_flight` groups are processed concurrently -- one
    host thread, HIP stream and arena each, all sharing the device weights -- so that the read-backs of one group hide
    behind the kernels of another.  Every item gets the same bits as in the sequential walk; the caller's current stream
    waits for all of them."""

    def __init__(self, cfg, n_mos_classes, ignore_index, state_dict, device="cuda:0", quirk_exact=True,
                 windows_in_flight=None, windows_per_launch=None):
        self.cfg = cfg
        self.mos_class = n_mos_classes
        self.ignore_index = ignore_index
        self.state_dict_ref = state_dict
        self.device = device
        self.quirk_exact = quirk_exact
        if windows_in_flight is None:
            windows_in_flight = int(os.environ.get("INSMOS_WINDOWS_IN_FLIGHT", "4"))
        if windows_per_launch is None:
            windows_per_launch = int(os.environ.get("INSMOS_WINDOWS_PER_LAUNCH", "8"))
        self.windows_in_flight = max(1, int(windows_in_flight))         # launch sets (groups) in flight
        self.windows_per_launch = max(1, min(int(windows_per_launch), MAX_WINDOWS_PER_LAUNCH))
        self._engine = None
        self._trainer = None
        self._workers = None  # (engines, streams, executor)

    @property
    def trainer(self):
        """The training-side twin (insmos_amd/train_unet.py), built on first use from the same checkpoint.  Its parameters
        are torch leaf tensors a caller hands to an optimiser; `trainer.unet.export_state_dict()` / tools/train_synthetic.py
        turn them back into a checkpoint the inference engine loads."""
        if self._trainer is None:
            if not torch.cuda.is_available():
                raise RuntimeError("insmos_amd needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
            from .train_unet import InsMOSTrainer
            self._trainer = InsMOSTrainer(self.cfg, self.state_dict_ref, self.device)
        return self._trainer

    @property
    def engine(self):
        if self._engine is None:
            if not torch.cuda.is_available():
                raise RuntimeError("insmos_amd needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
            self._engine = Engine(self.cfg, self.state_dict_ref, self.device, quirk_exact=self.quirk_exact, native=True)
            self._drop_workers()
        return self._engine

    def _drop_workers(self):
        if self._workers is not None:
            engines, _, pool = self._workers[:3]
            # every worker thread hands back its second stream and events (csrc/forward.hip keeps them per host thread); the
            # barrier makes each of the pool's threads take exactly one of the tasks
            import threading
            bar = threading.Barrier(len(engines))

            def release():
                try:
                    bar.wait(timeout=5.0)
                except threading.BrokenBarrierError:
                    pass
                engines[0].lib.insmos_forward_thread_release()

            for f in [pool.submit(release) for _ in engines]:
                f.result()
            pool.shutdown(wait=True)
            self._workers = None

    def _get_workers(self, w):
        eng = self.engine
        if self._workers is None or len(self._workers[0]) < w or self._workers[3] is not eng.L:
            self._drop_workers()
            dev = torch.device(self.device)
            engines = [eng] + [eng.clone_shared() for _ in range(w - 1)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(w)]
            self._workers = (engines, streams, ThreadPoolExecutor(max_workers=w, thread_name_prefix="insmos-window"), eng.L)
        return self._workers[:3]

    def forward(self, list_batch_dict, Model_mode):
        """'test' (models/models.py:355-359,375-376) and 'eval' (:347-353,369-373, the validation step: the same forward
        plus MOSLoss on the point logits and on the motion features, needs batch_dict["past_labels"])."""
        if Model_mode == "train":
            # models/models.py:313-345,365-367: the same module serves the training step.  The differentiable graph lives in
            # InsMOSTrainer (fp32 HIP forward + backward, its own leaf tensors in `self.trainer.params`); the same four
            # values come back: (loss, train_loss_dict, gt_mos_label_list, preb_mos_lable_list).
            return self.trainer.forward(list_batch_dict, "train")
        if Model_mode not in ("test", "eval"):
            raise ValueError(f"unknown Model_mode {Model_mode!r}")
        keep = Model_mode == "eval"
        n = len(list_batch_dict)
        wpl = self.windows_per_launch
        groups = [list(range(i, min(i + wpl, n))) for i in range(0, n, wpl)]
        w = min(len(groups), self.windows_in_flight)
        results = [None] * n

        def one(engine, idxs):
            # one launch set at a time: its non-convolution work goes to a second stream (latency); several in flight: they
            # overlap each other already and the extra stream only adds contention (measured: 669 vs 648 scans/s)
            engine.lib.insmos_forward_streams(15 if w <= 1 else 0)
            engine.keep_current_points = keep
            res = engine.forward_windows([list_batch_dict[i]["past_point_clouds"] for i in idxs])
            engine.last_current_points = None
            for i, r in zip(idxs, res):
                results[i] = r

        if w <= 1:
            for idxs in groups:
                one(self.engine, idxs)
        else:
            engines, streams, pool = self._get_workers(w)
            dev = torch.device(self.device)
            cur = torch.cuda.current_stream(dev)

            def run(wi):
                torch.cuda.set_device(dev)  # device and current stream are per host thread
                streams[wi].wait_stream(cur)  # the inputs were produced on the caller's stream
                with torch.cuda.stream(streams[wi]):
                    for gi in range(wi, len(groups), w):
                        one(engines[wi], groups[gi])

            for f in [pool.submit(run, wi) for wi in range(w)]:
                f.result()  # re-raises worker exceptions
            for st in streams[:w]:
                cur.wait_stream(st)
            for logits, pred, cur_pts in results:
                for t in (logits, *pred.values(), *([cur_pts] if cur_pts is not None else [])):
                    t.record_stream(cur)  # allocated on a worker stream, consumed on the caller's
        preb_dict_list, recall_dict_list, preb_mos_lable_list = [], [], []
        thresh = self.cfg["MODEL"]["POST_PROCESSING"].get("RECALL_THRESH_LIST", [0.3, 0.5, 0.7])
        for b, (logits, pred, _) in zip(list_batch_dict, results):
            preb_dict_list.append([pred])
            # post_process.py:216-220: recall bookkeeping only when the batch carries ground-truth boxes (validation)
            recall_dict_list.append(generate_recall_record(pred["pred_boxes"], {}, 0, b, thresh) if "gt_boxes" in b else {})
            preb_mos_lable_list.append(logits)
        if Model_mode == "test":
            return preb_dict_list, recall_dict_list, preb_mos_lable_list
        # ---- 'eval': models/models.py:321-323 (motion loss on current_motion_feature[:, :3]), :350 (loss on the logits),
        # :369-373 (means over the batch items; val_loss as a Python float, val_motion_loss as a (1,) tensor)
        from .autograd import mos_loss
        gt_mos_label_list = []
        val_loss = torch.zeros(1, device=self.device)
        val_motion_loss = torch.zeros(1, device=self.device)
        with torch.no_grad():
            for b, (logits, _, cur_pts) in zip(list_batch_dict, results):
                gt = b["past_labels"][-1]
                if int(gt.shape[0]) != int(logits.shape[0]):
                    raise ValueError(f"past_labels[-1] has {int(gt.shape[0])} labels for {int(logits.shape[0])} current points")
                motion = cur_pts[:, 4:4 + self.mos_class]  # use_motion_loss False keeps the first 3 columns (:318-319)
                val_motion_loss = val_motion_loss + mos_loss(motion, gt, self.mos_class, self.ignore_index)
                val_loss = val_loss + mos_loss(logits, gt, self.mos_class, self.ignore_index)
                # MOSLoss.compute_loss overwrites the ignored columns of ITS INPUT with -inf (loss.py:25), and that input
                # is the tensor the reference hands back as the prediction (:353) -- so do the returned logits
                logits[:, list(self.ignore_index)] = float("-inf")
                gt_mos_label_list.append(gt)
        val_loss = val_loss / n
        val_motion_loss = val_motion_loss / n
        return (preb_dict_list, recall_dict_list, gt_mos_label_list, preb_mos_lable_list, val_loss.item(), val_motion_loss)

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
        _check_checkpoint_layouts(ckpt, cfg, checkpoint_path)
        return cls(cfg, state_dict=ckpt["state_dict"])

    def state_dict(self):
        return self._state_dict

    def cuda(self, device=None):
        if device is not None:
            self._device = f"cuda:{device}" if isinstance(device, int) else str(device)
            self.model.device = self._device
            self.model._engine = None
            self.model._drop_workers()
        _ = self.model.engine  # build now: weights are packed and uploaded once
        return self

    def to(self, device):
        return self.cuda(device)

    def eval(self):
        return self

    def forward(self, batch_data, Model_mode):
        return self.model(batch_data, Model_mode)

    __call__ = forward


def _check_checkpoint_layouts(ckpt, cfg, path):
    """Shapes of every tensor of the inference path against params.param_spec (a wrong LAYOUT of the right size -- spconv 1.x's
    (kz, ky, kx, Cin, Cout), a torch Conv3d export -- must fail here, not produce silently permuted taps), and, for a checkpoint
    this package did not write itself, one loud warning: the MinkowskiEngine tap order and the spconv 2.3.6 weight layout that
    insmos_amd/params.py converts from are pinned to the reference's module definitions and to dependency knowledge only -- no
    published checkpoint was available to verify them (tools/ckpt_probe.py prints the details)."""
    import warnings
    spec = P.param_spec(cfg)
    sd = ckpt["state_dict"]
    bad = [(k, tuple(int(d) for d in sd[k].shape), tuple(shape)) for k, (shape, _) in spec.items()
           if k in sd and tuple(int(d) for d in sd[k].shape) != tuple(shape)]
    if bad:
        k, got, want = bad[0]
        raise ValueError(f"{path}: {len(bad)} tensors have another shape than the reference's modules define, e.g. {k}: checkpoint "
                         f"{got}, expected {want} -- run `python tools/ckpt_probe.py {path}` (it names known foreign layouts)")
    if not ckpt.get("insmos_amd_synthetic", False):
        warnings.warn(f"{path}: first-contact warning -- this looks like a REAL InsMOS checkpoint.  Shapes match, but the order of "
                      "MinkowskiEngine's kernel-volume axis (x-fastest region enumeration, even kernels at offsets {0, 1}) and the "
                      "spconv 2.3.6 (Cout, kz, ky, kx, Cin) layout assumed by insmos_amd/params.py have never been checked against "
                      "published weights (none were available when this was built): compare a few scans' labels with the "
                      "reference's own output before trusting the result; `python tools/ckpt_probe.py <ckpt>` prints what can "
                      "be checked offline.", RuntimeWarning, stacklevel=3)


def save_checkpoint(path, cfg, state_dict):
    """Write a Lightning-shaped checkpoint ({'hyper_parameters', 'state_dict'}) for tests/tools; marked as written by this
    package (load_from_checkpoint warns about checkpoints that are not: their layout conversions are unverified)."""
    torch.save({"hyper_parameters": cfg, "insmos_amd_synthetic": True,
                "state_dict": {k: torch.as_tensor(v) for k, v in state_dict.items()}}, path)
