"""Inference pipelines on the device: model forward -> get_probmaps epilogue -> direction-difference map ->
boost/argmax -> connected-component chain, without leaving the GPU (only the final int32 label maps do).

`infer_tiles`   : a batch of independent tiles, one view each (the benchmark unit "256x256 tile incl. post-proc")
`infer_image`   : the reference's per-image procedure of test_dam.py:297-563: 8 dihedral TTA views, whole-image
                  forward (all_img_test == 1) or sliding windows (utils.split_forward_dam), per-view DDM, mean,
                  point-guided boost, CC chain.
"""
import torch

from . import postproc


@torch.no_grad()
def infer_tiles(model, x, classes=9, min_area=20, radius=2, want_stages=False):
    """x: float32 NCHW [B,3,H,W] on the GPU.  Returns dict(final int32 [B,H,W], counts, pred, ...)."""
    assert not model.training
    mask, point, direction = model(x)
    B, _, H, W = mask.shape
    prob, dcm = postproc.probmaps(mask, direction)                        # test_dam.py:984, 1011-1013
    code, minmax = postproc.ddm_codes(dcm, classes)                       # generate_dd_map per tile
    r = postproc.tta_boost_argmax(prob.reshape(B, 1, 3 * H * W), point.reshape(B, 1, H * W),
                                  code.reshape(B, 1, H * W), minmax.reshape(B, 1, 2), [0], H, W,
                                  want_stages=want_stages)
    cc = postproc.cc_chain(r['pred'], 1, min_area, radius, want_stages=want_stages)
    r.update(cc)
    r.update(prob=prob, dcm=dcm, minmax=minmax, point=point)
    return r
