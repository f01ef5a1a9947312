"""ORACLE (test infrastructure - imported by tests/ only): the reference's per-image inference procedure on the CPU
in fp32, assembled from the cited lines (paths relative to /root/reference).  The network is any callable returning
(mask, point, direction) logits for an NCHW float32 tensor (oracle.models.Unet in eval mode); post-processing is
oracle/postproc.py (plain C).

  sliding windows          utils.py:658-726   split_forward_dam
  per-view epilogue        test_dam.py:932-1035 get_probmaps (softmax, bg-gated direction argmax, raw point map)
  8 dihedral TTA views     test_dam.py:313-450 (PIL FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM / rotate(90, expand) and their inverses)
  DDM mean, boost, argmax  test_dam.py:455-539
  CC chain                 test_dam.py:546-563

Pinned by: tests/golden/split_fwd.npz (the reference's own split_forward_dam on a position-coding toy network,
tests/test_oracle_infer.py) and the pins of oracle/postproc.py.
"""
import numpy as np
import torch

from . import postproc as orc


def split_forward(net, x, size, overlap, classes=9, out_c=3, batch=8):
    """utils.py:658-726.  x: [1,3,h0,w0] float32.  Windows are evaluated `batch` at a time (independent in eval mode)."""
    b, c, h0, w0 = x.shape
    assert b == 1
    stride = size - overlap
    ph = stride - (h0 - size) % stride if h0 - size > 0 else 0                    # :666-670
    pw = stride - (w0 - size) % stride if w0 - size > 0 else 0                    # :672-675
    xp = torch.zeros((1, c, h0 + ph, w0 + pw), dtype=x.dtype)
    xp[:, :, :h0, :w0] = x
    h, w = xp.shape[2:]
    outs = [torch.zeros((1, k, h, w)) for k in (out_c, 1, classes)]
    jobs = []
    for i in range(0, h - overlap, stride):                                       # :683-690
        r_end = min(i + size, h)
        r0 = i + overlap // 2 if i > 0 else 0
        r1 = i + size - overlap // 2 if i + size < h else h
        for j in range(0, w - overlap, stride):
            c_end = min(j + size, w)
            c0 = j + overlap // 2 if j > 0 else 0                                 # :712-713
            c1 = j + size - overlap // 2 if j + size < w else w
            jobs.append((i, r_end, j, c_end, r0, r1, c0, c1))
    by_shape = {}
    for jb in jobs:
        by_shape.setdefault((jb[1] - jb[0], jb[3] - jb[2]), []).append(jb)
    with torch.no_grad():
        for shape, lst in by_shape.items():
            for s in range(0, len(lst), batch):
                chunk = lst[s:s + batch]
                res = net(torch.cat([xp[:, :, i:re, j:ce] for (i, re, j, ce, *_r) in chunk], 0))
                for k, (i, re, j, ce, r0, r1, c0, c1) in enumerate(chunk):
                    for o, t in zip(outs, res):                                    # :714-718
                        o[:, :, r0:r1, c0:c1] = t[k:k + 1, :, r0 - i:r1 - i, c0 - j:c1 - j]
    return tuple(o[:, :, :h0, :w0] for o in outs)                                # :722-726


def view(img, xf):
    """test_dam.py:313-385: bit 2 = rotate(90, expand) first, bit 0 = FLIP_LEFT_RIGHT, bit 1 = FLIP_TOP_BOTTOM (CHW arrays)"""
    v = img
    if xf & 4:
        v = np.rot90(v, k=1, axes=(-2, -1))
    if xf & 1:
        v = np.flip(v, -1)
    if xf & 2:
        v = np.flip(v, -2)
    return np.ascontiguousarray(v)


def unview(a, xf):
    """test_dam.py:356-372, 425-441: undo the flips, then np.rot90(k=3)"""
    if xf & 2:
        a = np.flip(a, -2)
    if xf & 1:
        a = np.flip(a, -1)
    if xf & 4:
        a = np.rot90(a, k=3, axes=(-2, -1))
    return np.ascontiguousarray(a)


TTA_XFORMS = (0, 1, 2, 3, 4, 5, 6, 7)       # id, hf, vf, hvf, r90, r90_hf, r90_vf, r90_hvf (order of test_dam.py:445-468)


def view_outputs(net, image, xf, all_img_test, patch_size, overlap, classes=9):
    """one TTA view through get_probmaps: (prob f32 [3,H,W], point f32 [H,W], dcm u8 [H,W]) already un-flipped"""
    v = torch.from_numpy(view(image, xf))[None]
    if all_img_test == 1:
        with torch.no_grad():
            mask, point, direction = net(v)                                      # test_dam.py:941-950 (size == 0)
    else:
        mask, point, direction = split_forward(net, v, patch_size, overlap, classes)
    prob, dcm = orc.probmaps(mask[0].numpy(), direction[0].numpy())                # :982-1015
    return unview(prob, xf), unview(point[0, 0].numpy(), xf), unview(dcm, xf)


def infer_image(net, image, tta=True, all_img_test=1, patch_size=256, overlap=40, classes=9, min_area=20, radius=2):
    """image: float32 [3,H,W] numpy.  Returns the dict of oracle.postproc.postprocess_views (pred, fill, small, label,
    final, count, ddms, ...) plus the per-view arrays (probs, points, dcms)."""
    xforms = TTA_XFORMS if tta else (0,)
    probs, points, dcms = [], [], []
    for xf in xforms:
        p, t, d = view_outputs(net, image, xf, all_img_test, patch_size, overlap, classes)
        probs.append(p)
        points.append(t[None])
        dcms.append(d[None])
    probs, points, dcms = np.stack(probs), np.stack(points), np.stack(dcms)
    r = orc.postprocess_views(probs, points, dcms, classes, min_area, radius)
    r.update(probs=probs, points=points, dcms=dcms)
    return r
