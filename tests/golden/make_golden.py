"""tests/golden/make_golden.py -- generate golden input/output vectors from the REFERENCE's own code.

Runs only in the build container (needs /root/reference and oracle/_ref built by oracle/make_ref.py);
the committed .npz files are pure data (inputs + expected outputs), never reference source.
    python tests/golden/make_golden.py
Sources exercised (all importable / compilable here):
  center_head.npz   models/backbones_2d/center_head.py:251-276  CenterHead.generate_predicted_boxes
  bev_backbone.npz  models/backbones_2d/base_bev_backbone.py:84-115 + center_head.py:65-72 (eval, seeded weights)
  metrics.npz       models/metrics.py:16-45  confusion matrix / IoU
  mean_vfe.npz      models/backbones_2d/mean_vfe.py:36-55
  array_index.npz   models/utils/src/Array_Index.cpp:14-79 (compiled), incl. yaw, label 0, permuted orders
  iou_bev.npz       models/bbox_post_process/src/iou3d_cpu.cpp:232-252 (compiled) + greedy reduce
                    iou3d_nms.cpp:116-132 restated on the reference IoU matrix -> keep lists @0.01
  post_process.npz  models/post_process.py:112-224 end-to-end (class-agnostic branch) with
                    iou3d_nms_cuda.nms_gpu stubbed by the compiled reference IoU + the greedy reduce
  height_compression.npz  models/backbones_2d/height_compression.py:24-31 view semantics (on a dense tensor)
  poses.npz         dataloader/utils.py:10-68 load_poses / load_calib / load_files on tiny hand-written files
"""
import ctypes
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

_iou = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_iou3d.so"))
_iou.ref_boxes_iou_bev.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]


def ref_iou(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.zeros((len(a), len(b)), np.float32)
    _iou.ref_boxes_iou_bev(a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data)
    return out


def greedy_keep(iou, thresh):
    """iou3d_nms_kernel.cu:298-307 mask (j > i, iou > thresh) + iou3d_nms.cpp:116-132 reduce."""
    n = len(iou)
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if not removed[i]:
            keep.append(i)
            removed[i + 1:] |= iou[i, i + 1:] > thresh
    return np.array(keep, np.int64)


def rand_boxes(rng, n, spread=20.0, big=False):
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = rng.uniform(-spread, spread, n)
    b[:, 1] = rng.uniform(-spread, spread, n)
    b[:, 2] = rng.uniform(-2, 0, n)
    b[:, 3] = rng.uniform(1.5, 5.0 if not big else 12.0, n)
    b[:, 4] = rng.uniform(0.5, 2.5 if not big else 6.0, n)
    b[:, 5] = rng.uniform(1.0, 2.0, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def main():
    rng = np.random.default_rng(1234)
    torch.manual_seed(1234)

    # ---------------- center head decode
    from models.backbones_2d.center_head import CenterHead
    cfg_head = {"TARGET_ASSIGNER_CONFIG": {"VOXEL_SIZE": [0.1, 0.1, 0.1], "OUT_SIZE_FACTOR": 4}}
    head = CenterHead(cfg_head, 16, 3, ["Car", "Pedestrian", "Cyclist"], np.array([1200, 1000, 40]),
                      np.array([-60, -50, -3, 60, 50, 1]))
    H, W = 9, 11
    cls = torch.randn(1, H, W, 3)
    box = torch.randn(1, H, W, 8)
    with torch.no_grad():
        bc, bb = head.generate_predicted_boxes(cls, box)
    np.savez(os.path.join(HERE, "center_head.npz"), cls=cls.numpy(), box=box.numpy(), out_cls=bc.numpy(),
             out_boxes=bb.numpy())

    # ---------------- BEV backbone + head convs (small spatial size, full channel widths reduced)
    from models.backbones_2d.base_bev_backbone import BaseBEVBackbone
    b2d = {"LAYER_NUMS": [5], "LAYER_STRIDES": [1], "NUM_FILTERS": [32], "UPSAMPLE_STRIDES": [2],
           "NUM_UPSAMPLE_FILTERS": [48]}
    net = BaseBEVBackbone(b2d, 24).eval()
    head2 = CenterHead(cfg_head, 48, 3, ["Car", "Pedestrian", "Cyclist"], np.array([1200, 1000, 40]),
                       np.array([-60, -50, -3, 60, 50, 1])).eval()
    with torch.no_grad():
        for m in list(net.modules()) + list(head2.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.8, 1.2)
                m.bias.normal_(0, 0.05)
        x = torch.randn(1, 24, 7, 10)
        d = net({"current_bev": x})
        f2d = d["spatial_features_2d"]
        d = head2(d, "test")
    sd = {"bev." + k: v.numpy() for k, v in net.state_dict().items() if "num_batches" not in k}
    sd.update({"head." + k: v.numpy() for k, v in head2.state_dict().items()})
    np.savez(os.path.join(HERE, "bev_backbone.npz"), x=x.numpy(), f2d=f2d.numpy(),
             cls=d["batch_cls_preds"].numpy(), boxes=d["batch_box_preds"].numpy(), **sd)

    # ---------------- metrics
    from models.metrics import ClassificationMetrics
    met = ClassificationMetrics(3, [0])
    logits = torch.randn(500, 3)
    logits[:20] = 0.0  # all-zero rows -> tie rule
    gt = torch.randint(0, 3, (500,))
    cm = met.compute_confusion_matrix(logits.clone(), gt)
    iou = met.getIoU(cm.clone())
    np.savez(os.path.join(HERE, "metrics.npz"), logits=logits.numpy(), gt=gt.numpy(), cm=cm.numpy(), iou=iou.numpy())

    # ---------------- mean vfe
    from models.backbones_2d.mean_vfe import MeanVFE
    vox = torch.randn(40, 5, 7)
    num = torch.randint(0, 6, (40,))
    for i in range(40):
        vox[i, int(num[i]):] = 0
    out = MeanVFE({}, 7)({"voxels": vox, "voxel_num_points": num})["voxel_features"]
    np.savez(os.path.join(HERE, "mean_vfe.npz"), voxels=vox.numpy(), num=num.numpy(), out=out.numpy())

    # ---------------- height compression view semantics
    from models.backbones_2d.height_compression import HeightCompression

    class _Fake:
        def __init__(self, dense):
            self._d = dense

        def dense(self):
            return self._d

    dense = torch.randn(1, 6, 2, 4, 5)
    hc = HeightCompression({"NUM_BEV_FEATURES": 12})({"encoded_spconv_tensor": _Fake(dense),
                                                       "encoded_spconv_tensor_stride": 8})["spatial_features"]
    np.savez(os.path.join(HERE, "height_compression.npz"), dense5=dense.numpy(), out=hc.numpy())

    # ---------------- Array_Index (compiled reference module)
    import Array_Index
    cases = {}
    # case A: cluster of voxels around yawed boxes, incl. label 0 box and far box
    gx, gy, gz = np.meshgrid(np.arange(0, 40), np.arange(0, 30), np.arange(0, 6), indexing="ij")
    coords = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], 1).astype(np.int32)
    coords = coords[rng.uniform(size=len(coords)) < 0.5]
    boxes = np.array([[10.2, 8.1, 2.0, 9.0, 4.0, 3.0, 0.0, 1], [20.5, 15.5, 3.0, 8.0, 3.0, 4.0, 0.7, 2],
                      [30.0, 20.0, 1.0, 6.0, 6.0, 2.0, np.pi / 2, 3], [12.0, 22.0, 2.5, 10.0, 2.5, 5.0, -1.1, 1],
                      [5.0, 5.0, 2.0, 4.0, 4.0, 4.0, 0.3, 0], [100.0, 100.0, 2.0, 4.0, 4.0, 4.0, 0.3, 2]],
                     np.float32)
    for name, perm in (("sorted", np.arange(len(coords))), ("perm1", rng.permutation(len(coords))),
                       ("perm2", rng.permutation(len(coords)))):
        c = np.ascontiguousarray(coords[perm])
        out = Array_Index.find_features_by_bbox_with_yaw(c, boxes, np.zeros((len(c), 3), dtype=int))
        cases["coords_" + name] = c
        cases["out_" + name] = np.asarray(out, np.int32)
    cases["boxes"] = boxes
    # case B: the survey's 3-point order-dependence demo
    pts3 = np.array([[0, 0, 0], [0, 3, 0], [0, 1, 0]], np.int32)
    box3 = np.array([[0.0, 1.5, 0.0, 8.0, 2.0, 2.0, np.pi / 2, 1]], np.float32)
    cases["demo_coords"] = pts3
    cases["demo_box"] = box3
    cases["demo_out"] = np.asarray(Array_Index.find_features_by_bbox_with_yaw(pts3, box3, np.zeros((3, 3), dtype=int)),
                                   np.int32)
    pts3b = np.ascontiguousarray(pts3[[2, 0, 1]])
    cases["demo_coords_b"] = pts3b
    cases["demo_out_b"] = np.asarray(
        Array_Index.find_features_by_bbox_with_yaw(pts3b, box3, np.zeros((3, 3), dtype=int)), np.int32)
    # case C: random boxes in voxel units on a random sparse set
    coordsC = np.unique(rng.integers(0, [150, 125, 6], size=(3000, 3)), axis=0).astype(np.int32)
    coordsC = coordsC[rng.permutation(len(coordsC))]
    boxesC = np.zeros((40, 8), np.float32)
    boxesC[:, 0] = rng.uniform(0, 150, 40)
    boxesC[:, 1] = rng.uniform(0, 125, 40)
    boxesC[:, 2] = rng.uniform(0, 6, 40)
    boxesC[:, 3] = rng.uniform(2, 30, 40)
    boxesC[:, 4] = rng.uniform(2, 20, 40)
    boxesC[:, 5] = rng.uniform(1, 8, 40)
    boxesC[:, 6] = rng.uniform(-np.pi, np.pi, 40)
    boxesC[:, 7] = rng.integers(1, 4, 40)
    cases["coords_C"] = coordsC
    cases["boxes_C"] = boxesC
    cases["out_C"] = np.asarray(Array_Index.find_features_by_bbox_with_yaw(coordsC, boxesC,
                                                                            np.zeros((len(coordsC), 3), dtype=int)),
                                np.int32)
    np.savez(os.path.join(HERE, "array_index.npz"), **cases)

    # ---------------- rotated IoU + NMS keep lists
    a = rand_boxes(rng, 60, spread=8.0)
    b = rand_boxes(rng, 50, spread=8.0)
    special = np.array([[0, 0, 0, 4, 2, 1.5, 0.0], [0, 0, 0, 4, 2, 1.5, 0.0], [1, 0, 0, 4, 2, 1.5, 0.3],
                        [4, 0, 0, 4, 2, 1.5, 0.0], [0, 2, 0, 4, 2, 1.5, 0.0], [10, 10, 0, 4, 2, 1.5, 1.0],
                        [0, 0, 0, 2, 4, 1.5, np.pi / 2], [0.5, 0.5, 0, 1, 1, 1, 0.78539816],
                        [0, 0, 0, 4, 2, 1.5, np.pi], [0, 0, 0, 0.5, 0.5, 1, 0.2]], np.float32)
    iou_ab = ref_iou(a, b)
    iou_sp = ref_iou(special, special)
    dense = rand_boxes(rng, 300, spread=15.0, big=True)  # already "sorted by score" = given order
    iou_dd = ref_iou(dense, dense)
    keep = greedy_keep(iou_dd, 0.01)
    keep_05 = greedy_keep(iou_dd, 0.5)
    np.savez(os.path.join(HERE, "iou_bev.npz"), a=a, b=b, iou_ab=iou_ab, special=special, iou_special=iou_sp,
             dense=dense, keep_001=keep, keep_05=keep_05, iou_dense_row0=iou_dd[0])

    # ---------------- post_processing end-to-end with the stubbed nms_gpu
    stub = types.ModuleType("models.bbox_post_process.iou3d_nms_cuda")

    def nms_gpu(boxes, keep_t, thresh):
        bnp = boxes.detach().cpu().numpy()
        k = greedy_keep(ref_iou(bnp, bnp), thresh)
        keep_t[:len(k)] = torch.from_numpy(k)
        return len(k)

    stub.nms_gpu = nms_gpu
    sys.modules["models.bbox_post_process.iou3d_nms_cuda"] = stub
    torch.Tensor.cuda = lambda self, *a, **k: self  # container-only: the reference calls .cuda() on keep
    from models.post_process import post_processing
    ncell = 6000
    cls_logits = torch.randn(1, ncell, 3) * 1.5 - 3.0
    # unique scores so torch.topk / sort tie order cannot matter
    boxes_all = torch.from_numpy(rand_boxes(rng, ncell, spread=50.0))[None]
    pp_cfg = {"RECALL_THRESH_LIST": [0.3, 0.5, 0.7], "SCORE_THRESH": 0.1, "OUTPUT_RAW_SCORE": False,
              "NMS_CONFIG": {"MULTI_CLASSES_NMS": False, "NMS_TYPE": "nms_gpu", "NMS_THRESH": 0.01,
                             "NMS_PRE_MAXSIZE": 512, "NMS_POST_MAXSIZE": 100}}
    pred, recall = post_processing({"batch_cls_preds": cls_logits, "batch_box_preds": boxes_all,
                                    "cls_preds_normalized": False}, pp_cfg, 3)
    np.savez(os.path.join(HERE, "post_process.npz"), cls=cls_logits[0].numpy(), boxes=boxes_all[0].numpy(),
             pred_boxes=pred[0]["pred_boxes"].numpy(), pred_scores=pred[0]["pred_scores"].numpy(),
             pred_labels=pred[0]["pred_labels"].numpy(), pre_max=512, post_max=100)
    print("golden vectors written to", HERE)




def poses_golden():
    """dataloader/utils.py:10-68 (load_poses / load_calib / load_files) on tiny hand-written files."""
    import tempfile
    from dataloader.utils import load_calib, load_files, load_poses
    rng = np.random.default_rng(77)
    lines = []
    for i in range(6):
        a = 0.03 * i
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        t = np.array([0.1 * i, 0.01 * i, 1.3 * i]) + rng.normal(0, 1e-3, 3)
        lines.append(" ".join("%.9e" % v for v in np.hstack([R, t[:, None]]).ravel()))
    poses_txt = "\n".join(lines) + "\n"
    calib_txt = ("P0: 7.188560e+02 0 6.071928e+02 0 0 7.188560e+02 1.852157e+02 0 0 0 1 0\n"
                 "Tr: 4.276802385584e-04 -9.999672484946e-01 -8.084491683471e-03 -1.198459927713e-02 "
                 "-7.210626507497e-03 8.081198471645e-03 -9.999413164504e-01 -5.403984729748e-02 "
                 "9.999738645903e-01 4.859485810390e-04 -7.206933692422e-03 -2.921968648686e-01\n")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "poses.txt"), "w").write(poses_txt)
        open(os.path.join(d, "calib.txt"), "w").write(calib_txt)
        os.makedirs(os.path.join(d, "velodyne"))
        for n in ("000002.bin", "000000.bin", "000001.bin"):
            open(os.path.join(d, "velodyne", n), "wb").write(b"")
        poses = np.array(load_poses(os.path.join(d, "poses.txt")))
        T_cam_velo = np.asarray(load_calib(os.path.join(d, "calib.txt"))).reshape(4, 4)
        files = [os.path.basename(f) for f in load_files(os.path.join(d, "velodyne"))]
    np.savez(os.path.join(HERE, "poses.npz"), poses_txt=np.array(poses_txt), calib_txt=np.array(calib_txt), poses=poses,
             T_cam_velo=T_cam_velo, files=np.array(files))


if __name__ == "__main__":
    if "--poses-only" not in sys.argv:
        main()
    poses_golden()
