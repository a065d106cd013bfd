"""spconv.pytorch.utils subset (voxel_generate.py:19-28, spconv_unet.py:18,410)."""
import numpy as np
import torch

from oracle import ref_ops as R


class PointToVoxel:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel,
                 device=None):
        self.vsize, self.range = vsize_xyz, coors_range_xyz
        self.nfeat, self.max_voxels, self.max_points = num_point_features, max_num_voxels, max_num_points_per_voxel

    def generate_voxel_with_id(self, pc, clear_voxels=True, empty_mean=False):
        pts = pc.detach().cpu().numpy().astype(np.float32)
        voxels, coords, num, pcid = R.voxelize_with_id(pts, self.vsize, self.range, self.max_voxels, self.max_points)
        return (torch.from_numpy(voxels), torch.from_numpy(coords.astype(np.int32)), torch.from_numpy(num.astype(np.int32)),
                torch.from_numpy(pcid.astype(np.int64)))


def gather_features_by_pc_voxel_id(seg_res_features, pc_voxel_id, invalid_value=0):
    """Voxel feature to every point; points without a voxel (id -1) get invalid_value."""
    out = torch.full((pc_voxel_id.shape[0], seg_res_features.shape[1]), float(invalid_value), dtype=seg_res_features.dtype)
    ok = pc_voxel_id >= 0
    out[ok] = seg_res_features[pc_voxel_id[ok]]
    return out
