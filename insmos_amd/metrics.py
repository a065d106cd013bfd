"""ClassificationMetrics of the reference (models/metrics.py:10-52) on the device, plus the only
cross-GPU exchange of the inference path: gathering the 3x3 int64 confusion counters over RCCL."""
import ctypes

import torch

from . import _lib


class ClassificationMetrics:
    def __init__(self, n_classes, ignore_index):
        self.n_classes = n_classes
        self.ignore_index = list(ignore_index)
        self.ignore_mask = 0
        for c in self.ignore_index:
            self.ignore_mask |= 1 << int(c)

    def compute_confusion_matrix(self, pred_logits, gt_labels, out=None):
        """models/metrics.py:16-30; (n_classes, n_classes) int64 [pred, gt].  Accumulates into `out` if given.
        Unlike the reference it does not overwrite the ignored logit columns in place."""
        if pred_logits.device.type != "cuda":
            raise RuntimeError("insmos_amd.metrics runs on the GPU; there is no CPU fallback")
        lib = _lib.load()
        logits = pred_logits if pred_logits.stride(1) == 1 else pred_logits.contiguous()
        gt = gt_labels.to(device=logits.device, dtype=torch.int64).contiguous()
        cm = out if out is not None else torch.zeros((self.n_classes, self.n_classes), dtype=torch.int64,
                                                     device=logits.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        _lib.check(lib.insmos_confusion3(logits.data_ptr(), logits.stride(0), gt.data_ptr(), logits.shape[0],
                                         self.n_classes, self.ignore_mask, cm.data_ptr(), st), "insmos_confusion3")
        return cm

    def getStats(self, confusion_matrix):
        cm = confusion_matrix.clone()
        cm[:, self.ignore_index] = 0  # models/metrics.py:33-34
        tp = cm.diag()
        fp = cm.sum(dim=1) - tp
        fn = cm.sum(dim=0) - tp
        return tp, fp, fn

    def getIoU(self, confusion_matrix):
        tp, fp, fn = self.getStats(confusion_matrix)
        return tp / (tp + fp + fn + 1e-15)

    def getacc(self, confusion_matrix):
        tp, fp, fn = self.getStats(confusion_matrix)
        return tp.sum() / (tp.sum() + fp.sum() + 1e-15)


def all_gather_confusion(cm):
    """Sequence-level data parallelism: every rank evaluates its own windows; the only
    exchange is an all_gather of the per-rank confusion counters (72 B/rank over RCCL/xGMI with the
    `nccl` backend, gloo on CPU in the tests), summed locally -> global IoU on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return cm
    parts = [torch.empty_like(cm) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, cm.contiguous())
    return torch.stack(parts, 0).sum(0)


def shard_indices(n_items, rank, world_size):
    """Window i goes to rank i % world_size (predict_mos.py:103-106 windows are independent)."""
    return list(range(rank, n_items, world_size))
