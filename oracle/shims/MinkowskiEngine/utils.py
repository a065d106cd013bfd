"""ME.utils subset (motionnet.py:33, resnet.py:90)."""
import torch


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    """Prepends the batch index column and concatenates the items; float coordinates stay float (motionnet.py:33-34)."""
    cs, fs = [], []
    for b, (c, f) in enumerate(zip(coords, feats)):
        cs.append(torch.cat([torch.full((c.shape[0], 1), float(b), dtype=c.dtype), c], 1))
        fs.append(f)
    return torch.cat(cs, 0), torch.cat(fs, 0)


def batched_coordinates(coords, dtype=torch.int32, device=None):
    return sparse_collate(coords, coords)[0]


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """Initialisation only (every weight is overwritten by the checkpoint in the golden run)."""
    with torch.no_grad():
        return tensor.normal_(0, 0.01)
