"""Training-target generation with the reference's names (my_transforms_direction.py).

  LabelEncoding(out_c=3, radius=1, do_direction=0)      :687-885   3-class label + centre-point map + centripetal classes
  label_encoding_batch(label_ch0)                        device API for whole batches (input pipeline on the GPU)
  ToTensor / Normalize                                   :889-1012  (host-side tensor conversion helpers)

The per-nucleus Python loop of the reference (EDT + numba centre search + whole-image 11x11 convolution per nucleus,
seconds per sample) is one pass of HBM-bound kernels behind cdnet_label_encoding.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

_RAYS = (C.c_double * 16)(*[f(2 * math.pi / 8 * k) for k in range(8) for f in (math.sin, math.cos)])


def _gauss_half(sigma=2.0, radius=8):
    k = [math.exp(-0.5 * i * i / (sigma * sigma)) for i in range(-radius, radius + 1)]
    s = 0.0
    for v in k:
        s += v
    k = [v / s for v in k]
    return (C.c_double * (radius + 1))(*k[radius:])


_GAUSS = _gauss_half()
_WS = {}


def label_encoding_batch(label_ch0, max_instances=4096, want_inst=False):
    """label_ch0: uint8 cuda tensor [N,H,W] (channel 0 of the 3-class label image, > 127 = inside).
    Returns (label3 uint8 {0,127,255}, point float16, direction uint8 0..8[, inst int32, counts int32])."""
    assert label_ch0.dtype == torch.uint8 and label_ch0.is_cuda and label_ch0.dim() == 3
    x = label_ch0.contiguous()
    N, H, W = x.shape
    dev = x.device
    nbytes = _lib.load().cdnet_label_encoding_workspace_bytes(N, H, W, max_instances)
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _WS[key] = ws
    label3 = torch.empty_like(x)
    point = torch.empty((N, H, W), dtype=torch.float16, device=dev)
    direction = torch.empty_like(x)
    inst = torch.empty((N, H, W), dtype=torch.int32, device=dev) if want_inst else None
    counts = torch.empty((N,), dtype=torch.int32, device=dev)
    _lib.call('cdnet_label_encoding', _lib.ptr(x), N, H, W, max_instances, C.cast(_RAYS, C.c_void_p), C.cast(_GAUSS, C.c_void_p),
              _lib.ptr(ws), ws.numel(), _lib.ptr(label3), _lib.ptr(point), _lib.ptr(direction), _lib.ptr(inst), _lib.ptr(counts),
              _lib.stream_ptr())
    if want_inst:
        return label3, point, direction, inst, counts
    return label3, point, direction


def label_encoding_instances_batch(label_inst, max_instances=4096, want_inst=False):
    """label_inst: int32 cuda tensor [N,H,W] of instance ids (the reference's <label_dir>/train_ins layout).  Returns what
    label_encoding_batch returns (my_transforms_direction.py:752-760 + :785-871)."""
    assert label_inst.dtype == torch.int32 and label_inst.is_cuda and label_inst.dim() == 3
    x = label_inst.contiguous()
    N, H, W = x.shape
    dev = x.device
    nbytes = _lib.load().cdnet_label_encoding_instances_workspace_bytes(N, H, W, max_instances)
    key = ('inst', dev.index, torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _WS[key] = ws
    label3 = torch.empty((N, H, W), dtype=torch.uint8, device=dev)
    point = torch.empty((N, H, W), dtype=torch.float16, device=dev)
    direction = torch.empty((N, H, W), dtype=torch.uint8, device=dev)
    inst = torch.empty((N, H, W), dtype=torch.int32, device=dev) if want_inst else None
    counts = torch.empty((N,), dtype=torch.int32, device=dev)
    _lib.call('cdnet_label_encoding_instances', _lib.ptr(x), N, H, W, max_instances, C.cast(_RAYS, C.c_void_p), C.cast(_GAUSS, C.c_void_p),
              _lib.ptr(ws), ws.numel(), _lib.ptr(label3), _lib.ptr(point), _lib.ptr(direction), _lib.ptr(inst), _lib.ptr(counts),
              _lib.stream_ptr())
    if want_inst:
        return label3, point, direction, inst, counts
    return label3, point, direction


class LabelEncoding(object):
    """Encoding the label, computes boundary individually (reference class; 3-class label input)."""

    def __init__(self, out_c=3, radius=1, do_direction=0):
        self.out_c = out_c
        self.radius = 1            # the reference forces 1 (:694)
        self.do_direction = do_direction

    def __call__(self, imgs):
        from PIL import Image
        out_imgs = list(imgs)
        label = np.array(imgs[2])
        ch0 = label if label.ndim == 2 else label[:, :, 0]
        assert self.out_c == 3, 'only the 3-class encoding of the CDNet path is implemented'
        if len(np.unique(ch0)) > 2:                      # instance-level label (:752-760)
            l3, point, direction = label_encoding_instances_batch(torch.from_numpy(np.ascontiguousarray(ch0).astype(np.int32)).cuda()[None])
        else:
            l3, point, direction = label_encoding_batch(torch.from_numpy(np.ascontiguousarray(ch0, dtype=np.uint8)).cuda()[None])
        out_imgs[2] = Image.fromarray(l3[0].cpu().numpy())
        if self.do_direction == 1:
            out_imgs.append(point[0].cpu().numpy())
            out_imgs.append(direction[0].cpu().numpy().astype(np.int64))
        return tuple(out_imgs)


class ToTensor(object):
    """(img, labels...) PIL / numpy -> tensors: image float CHW / 255, PIL labels int64 [1,H,W], numpy extras unchanged (:889-983)"""

    def __init__(self, index=1):
        self.index = index

    def __call__(self, imgs):
        pics = []
        for i, im in enumerate(imgs):
            if isinstance(im, np.ndarray):
                pics.append(torch.from_numpy(im) if i >= self.index else torch.from_numpy(im.transpose((2, 0, 1))).float().div(255))
                continue
            a = np.array(im)
            if a.ndim == 2:
                a = a[:, :, None]
            t = torch.from_numpy(a.transpose((2, 0, 1)).copy())
            pics.append(t.float().div(255) if i < self.index else t.long())
        return tuple(pics)


class Normalize(object):
    """channel = (channel - mean) / std on the first tensor only (:988-1012)"""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, tensors):
        tensors = list(tensors)
        for t, m, s in zip(tensors[0], self.mean, self.std):
            t.sub_(m).div_(s)
        return tuple(tensors)
