"""ORACLE (test infrastructure): numpy-facing wrappers over oracle/postproc_oracle.c.

Follows (paths relative to /root/reference):
  generate_dd_map         data_prepare/getDirectionDiffMap.py:44-108
  label_to_vector_mapping data_prepare/SegFix_offset_helper.py:50-89
  fuse/boost/argmax       test_dam.py:445-450, 479-491, 529-539
  CC chain                test_dam.py:546-563
  get_probmaps epilogue   test_dam.py:982-1015
"""
import ctypes as C
import numpy as np
from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# SegFix_offset_helper.py:50-89 (c4_align_axis unset)
LABEL_TO_VECTOR = {
    5: [[0, 0], [-1, -1], [-1, 1], [1, 1], [1, -1]],
    9: [[0, 0], [0, -1], [-1, -1], [-1, 0], [-1, 1], [0, 1], [1, 1], [1, 0], [1, -1]],
    17: [[0, 0], [0, -2], [-1, -2], [-2, -2], [-2, -1], [-2, 0], [-2, 1], [-2, 2], [-1, 2],
         [0, 2], [1, 2], [2, 2], [2, 1], [2, 0], [2, -1], [2, -2], [1, -2]],
}


def ddm_lut(classes):
    """round(cos) for every (centre class, neighbour class) pair, computed with the reference's arithmetic
    (getDirectionDiffMap.py:92-101: float64 cos with +1e-6, stored to a float32 array, np.around)."""
    v = np.array(LABEL_TO_VECTOR[classes], dtype=np.int64)
    lut = np.zeros((classes, classes), np.int8)
    for a in range(classes):
        for b in range(classes):
            fenzi = v[a, 0] * v[b, 0] + v[a, 1] * v[b, 1]
            fenmu = np.sqrt(pow(v[a, 0], 2) + pow(v[a, 1], 2)) * np.sqrt(pow(v[b, 0], 2) + pow(v[b, 1], 2)) + 0.000001
            lut[a, b] = np.int8(np.around(np.float32(fenzi / fenmu)))
    return lut


def generate_dd_map(label_direction, direction_classes, return_code=False):
    lab = np.ascontiguousarray(label_direction, dtype=np.uint8)
    H, W = lab.shape
    lut = ddm_lut(direction_classes)
    nbr = 4 if direction_classes - 1 == 4 else 8
    extra_zero = 1 if direction_classes - 1 == 16 else 0
    code = np.empty((H, W), np.uint8)
    out = np.empty((H, W), np.float32)
    with np.errstate(all='ignore'):
        lib().orc_ddm(_p(lab, C.c_uint8), H, W, direction_classes, _p(lut, C.c_int8), nbr, extra_zero,
                      _p(code, C.c_uint8), _p(out, C.c_float))
    return (out, code) if return_code else out


def dilate_cross_u8(x):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty_like(x)
    lib().orc_dilate_cross_u8(_p(x, C.c_uint8), x.shape[0], x.shape[1], _p(out, C.c_uint8))
    return out


def fuse_boost_argmax(probs, points, ddms):
    """probs [V,3,H,W] f32, points [V,1,H,W] or [V,H,W] f32, ddms [V,H,W] f32 (normalised per-view maps)."""
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    V, _, H, W = probs.shape
    points = np.ascontiguousarray(points, dtype=np.float32).reshape(V, H, W)
    ddms = np.ascontiguousarray(ddms, dtype=np.float32).reshape(V, H, W)
    prob_mean = np.empty((3, H, W), np.float32)
    point_mean = np.empty((H, W), np.float32)
    ddm_mean = np.empty((H, W), np.float64)
    inside3 = np.empty((H, W), np.uint8)
    pred = np.empty((H, W), np.uint8)
    lib().orc_fuse_boost_argmax(_p(probs, C.c_float), _p(points, C.c_float), _p(ddms, C.c_float), V, H, W,
                                _p(prob_mean, C.c_float), _p(point_mean, C.c_float), _p(ddm_mean, C.c_double),
                                _p(inside3, C.c_uint8), _p(pred, C.c_uint8))
    return dict(prob_mean=prob_mean, point_mean=point_mean, ddm_mean=ddm_mean, inside3=inside3, pred=pred)


def fill_holes(x):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty_like(x)
    lib().orc_fill_holes(_p(x, C.c_uint8), x.shape[0], x.shape[1], _p(out, C.c_uint8))
    return out


def remove_small(x, min_size):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty_like(x)
    lib().orc_remove_small(_p(x, C.c_uint8), x.shape[0], x.shape[1], int(min_size), _p(out, C.c_uint8))
    return out


def label8(x):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty(x.shape, np.int32)
    n = lib().orc_label8(_p(x, C.c_uint8), x.shape[0], x.shape[1], _p(out, C.c_int32))
    return out, n


def dilate_disk(x, r):
    x = np.ascontiguousarray(x, dtype=np.int32)
    out = np.empty_like(x)
    lib().orc_dilate_disk_i32(_p(x, C.c_int32), x.shape[0], x.shape[1], int(r), _p(out, C.c_int32))
    return out


def cc_chain(pred_inside, min_area=20, radius=2):
    x = np.ascontiguousarray(pred_inside, dtype=np.uint8)
    H, W = x.shape
    fill = np.empty((H, W), np.uint8)
    small = np.empty((H, W), np.uint8)
    label = np.empty((H, W), np.int32)
    final = np.empty((H, W), np.int32)
    n = lib().orc_cc_chain(_p(x, C.c_uint8), H, W, int(min_area), int(radius), _p(fill, C.c_uint8),
                           _p(small, C.c_uint8), _p(label, C.c_int32), _p(final, C.c_int32))
    return dict(fill=fill, small=small, label=label, final=final, count=n)


def probmaps(mask_logits, dir_logits):
    m = np.ascontiguousarray(mask_logits, dtype=np.float32)
    d = np.ascontiguousarray(dir_logits, dtype=np.float32)
    Cd, H, W = d.shape
    prob = np.empty((3, H, W), np.float32)
    dcm = np.empty((H, W), np.uint8)
    lib().orc_probmaps(_p(m, C.c_float), _p(d, C.c_float), Cd, H, W, _p(prob, C.c_float), _p(dcm, C.c_uint8))
    return prob, dcm


def postprocess_views(probs, points, dcms, classes=9, min_area=20, radius=2):
    """The whole reference post-processing for one image given the 8 un-flipped views
    (test_dam.py:445-563).  Raises AssertionError like the reference (:535) when a view's DDM is NaN."""
    V = probs.shape[0]
    H, W = probs.shape[-2:]
    dcms = np.asarray(dcms).reshape(V, H, W)
    ddms = np.stack([generate_dd_map(dcms[v], classes) for v in range(V)])
    r = fuse_boost_argmax(probs, points, ddms)
    assert not np.isnan(r['ddm_mean']).any(), 'assert(np.min(enhanced_boundary) >= 0) fails on NaN (test_dam.py:535)'
    r.update(cc_chain(r['pred'] == 1, min_area, radius))
    r['ddms'] = ddms
    return r
