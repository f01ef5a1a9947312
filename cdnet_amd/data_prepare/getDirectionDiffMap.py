"""`generate_dd_map` with the reference's signature (data_prepare/getDirectionDiffMap.py:44), computed on the
GPU by the HIP kernels behind cdnet_ddm_codes / cdnet_ddm_normalize."""
import numpy as np
import torch

from .. import postproc


def generate_dd_map(label_direction, direction_classes):
    """label_direction: np.ndarray [H,W] of direction classes; returns np.ndarray float32 [H,W] in {0,.5,1}
    (NaN everywhere when the map is constant - the reference's 0/0)."""
    lab = torch.from_numpy(np.ascontiguousarray(label_direction).astype(np.uint8)).cuda()[None]
    code, minmax = postproc.ddm_codes(lab, int(direction_classes))
    return postproc.ddm_normalize(code, minmax)[0].cpu().numpy()
