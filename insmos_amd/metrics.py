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
        if gt.numel() != logits.shape[0]:
            raise ValueError(f"compute_confusion_matrix: {gt.numel()} labels for {logits.shape[0]} logit rows")
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


def all_gather_confusion(cm, force=False):
    """Sequence-level data parallelism: every rank evaluates its own windows; the only
    exchange is an all_gather of the per-rank confusion counters (72 B/rank over RCCL/xGMI with the
    `nccl` backend, gloo on CPU in the tests), summed locally -> global IoU on every rank.
    A world of one rank has nothing to exchange and returns `cm` as is -- unless `force`: the one-GPU self-check
    (`bench.py --rccl-selfcheck`, tests/test_zz_gpu_rccl_world1.py) sends the device tensor through the backend's
    all_gather anyway, so that the RCCL branch below has executed before an 8-GPU node sees it."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return cm
    if dist.get_world_size() == 1 and not force:
        return cm
    src = cm.contiguous()
    if src.is_cuda and dist.get_backend() == "gloo":
        src = src.cpu()   # the two-ranks-on-one-GPU dry run (gloo moves host memory; RCCL takes the device tensor as is)
    parts = [torch.empty_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, src)
    return torch.stack(parts, 0).sum(0).to(cm.device)


def shard_indices(n_items, rank, world_size):
    """Window i goes to rank i % world_size (predict_mos.py:103-106 windows are independent)."""
    return list(range(rank, n_items, world_size))


def boxes_iou3d(boxes_a, boxes_b):
    """iou3d_nms_utils.boxes_iou3d_gpu (models/bbox_post_process/iou3d_nms_utils.py:28-61): (N,7) x (M,7) CUDA fp32 ->
    (N, M) 3D IoU, one HIP kernel (insmos_iou3d)."""
    a = boxes_a[:, :7].contiguous().float()
    b = boxes_b[:, :7].contiguous().float()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if a.shape[0] and b.shape[0]:
        st = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        _lib.check(_lib.load().insmos_iou3d(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), st), "insmos_iou3d")
    return out


def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
    """models/post_process.py:67-110: how many ground-truth boxes have a prediction (and, if present, a ROI) above each
    3D-IoU threshold.  Needs data_dict['gt_boxes'] (B, G, >=7) with trailing all-zero padding rows; accumulates into and
    returns `recall_dict` like the reference."""
    if data_dict is None or "gt_boxes" not in data_dict:
        return recall_dict
    rois = data_dict["rois"][batch_index] if "rois" in data_dict else None
    gt = data_dict["gt_boxes"][batch_index]
    if len(recall_dict) == 0:
        recall_dict = {"gt": 0}
        for t in thresh_list:
            recall_dict["roi_%s" % str(t)] = 0
            recall_dict["rcnn_%s" % str(t)] = 0
    nz = (gt.sum(dim=1) != 0).nonzero()  # the reference walks back over all-zero padding rows (never past row 0)
    k = int(nz[-1]) if len(nz) else 0
    gt = gt[:k + 1]
    if gt.shape[0] > 0:
        iou_rcnn = boxes_iou3d(box_preds, gt) if box_preds.shape[0] > 0 else None
        iou_roi = boxes_iou3d(rois, gt) if rois is not None else None
        for t in thresh_list:
            if iou_rcnn is not None:
                recall_dict["rcnn_%s" % str(t)] += int((iou_rcnn.max(dim=0)[0] > t).sum().item())
            if iou_roi is not None:
                recall_dict["roi_%s" % str(t)] += int((iou_roi.max(dim=0)[0] > t).sum().item())
        recall_dict["gt"] += int(gt.shape[0])
    return recall_dict
