"""postproc_other.process on the GPU (reference: postproc_other.py:15-99).

Same signature as the reference: `process(pred, model_mode, min_size=10, ws=True) -> labels`.  `pred` is a numpy HW
probability / binary map (or a torch CUDA tensor [H,W] / [N,H,W]); the result has the input's container type.  Modes:
every non-'dcan' mode of the reference; 'unet' forces ws=False exactly like postproc_other.py:35 - that branch (fill holes ->
4-connected label -> remove small labels, :49-52) is `cdnet_fill_label_process`, built from the kernels of the watershed
entry's marker stage.  The 'dcan' contour branch and micronet's per-instance re-dilation are outside the hot path (SURVEY 8a
row 16) and raise NotImplementedError.
All compute runs in csrc/postproc.hip; there is no CPU fallback."""
import numpy as np
import torch

from . import _lib


def watershed_process(pred_u8, min_size=10, stages=False):
    """pred_u8: torch.uint8 CUDA [N,H,W], non-zero = foreground.  Returns labels int32 [N,H,W] (and dist u8, marker i32
    when stages=True)."""
    assert pred_u8.is_cuda and pred_u8.dtype == torch.uint8 and pred_u8.dim() == 3
    pred_u8 = pred_u8.contiguous()
    N, H, W = pred_u8.shape
    lib = _lib.load()
    ws = torch.empty((lib.cdnet_watershed_workspace_bytes(N, H, W),), dtype=torch.uint8, device=pred_u8.device)
    labels = torch.empty((N, H, W), dtype=torch.int32, device=pred_u8.device)
    dist = torch.empty((N, H, W), dtype=torch.uint8, device=pred_u8.device) if stages else None
    marker = torch.empty((N, H, W), dtype=torch.int32, device=pred_u8.device) if stages else None
    _lib.call('cdnet_watershed_process', _lib.ptr(pred_u8), N, H, W, int(min_size), _lib.ptr(ws), ws.numel(), _lib.ptr(dist),
              _lib.ptr(marker), _lib.ptr(labels), _lib.stream_ptr())
    return (labels, dist, marker) if stages else labels


def fill_label_process(pred_u8, min_size=10):
    """ws=False branch on the device: pred_u8 torch.uint8 CUDA [N,H,W], non-zero = foreground -> labels int32 [N,H,W]"""
    assert pred_u8.is_cuda and pred_u8.dtype == torch.uint8 and pred_u8.dim() == 3
    pred_u8 = pred_u8.contiguous()
    N, H, W = pred_u8.shape
    lib = _lib.load()
    ws = torch.empty((lib.cdnet_watershed_workspace_bytes(N, H, W),), dtype=torch.uint8, device=pred_u8.device)
    labels = torch.empty((N, H, W), dtype=torch.int32, device=pred_u8.device)
    _lib.call('cdnet_fill_label_process', _lib.ptr(pred_u8), N, H, W, int(min_size), _lib.ptr(ws), ws.numel(), _lib.ptr(labels), _lib.stream_ptr())
    return labels


def process(pred, model_mode, min_size=10, ws=True):
    if model_mode == 'dcan':
        raise NotImplementedError("postproc_other.process: the 'dcan' contour branch is outside the accelerated path")
    if model_mode == 'micronet':
        raise NotImplementedError("postproc_other.process: micronet's per-instance re-dilation is outside the accelerated path")
    is_np = isinstance(pred, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(pred)).cuda() if is_np else pred
    assert t.dim() in (2, 3), 'Prediction shape is not HW'            # postproc_other.py:32
    squeeze = t.dim() == 2
    if squeeze:
        t = t[None]
    binary = (t > 0.5).to(torch.uint8)                                 # :33-34
    if model_mode == 'unet':
        ws = False                                                     # :35
    if ws:
        out = watershed_process(binary, min_size)
    else:
        out = fill_label_process(binary, min_size)                     # :49-52
    if squeeze:
        out = out[0]
    return out.cpu().numpy() if is_np else out
