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


# ---------------------------------------------------------------------------------------------------------
# watershed variant: postproc_other.py:15-99 (ws branch :36-48)
# ---------------------------------------------------------------------------------------------------------
def ws_dist(lab):
    """gen_inst_dst_map (postproc_other.py:16-27) restated in C (brute-force exact EDT)"""
    lab = np.ascontiguousarray(lab, dtype=np.int32)
    out = np.empty(lab.shape, np.uint8)
    lib().orc_ws_dist(_p(lab, C.c_int32), lab.shape[0], lab.shape[1], _p(out, C.c_uint8))
    return out


def watershed(image_u8, markers, mask):
    image_u8 = np.ascontiguousarray(image_u8, dtype=np.uint8)
    markers = np.ascontiguousarray(markers, dtype=np.int32)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty(markers.shape, np.int32)
    lib().orc_watershed(_p(image_u8, C.c_uint8), _p(markers, C.c_int32), _p(mask, C.c_uint8), markers.shape[0], markers.shape[1],
                        _p(out, C.c_int32))
    return out


def remove_small_labels(lab, min_size):
    """skimage.morphology.remove_small_objects on an integer label image (documented behaviour: labels whose pixel count
    is below min_size are zeroed, the other ids are kept)"""
    sizes = np.bincount(lab.ravel())
    out = lab.copy()
    small = sizes < min_size
    small[0] = False
    out[small[lab]] = 0
    return out


def watershed_process(pred, min_size=10, use_scipy=True):
    """postproc_other.process(pred, model_mode != 'dcan'/'unet'/'micronet', ws=True), steps :31-48.  With use_scipy the
    distance map and the marker come from the very scipy calls the reference makes; otherwise from the C restatement
    (tests assert both agree).  Returns dict(dist, marker, labels)."""
    from scipy import ndimage as ndi
    pred = (np.asarray(pred) > 0.5).astype(np.uint8)
    lab = ndi.label(pred)[0].astype(np.int32)                                    # measurements.label (:37)
    if use_scipy:
        canvas = np.zeros(pred.shape, np.uint8)                                  # gen_inst_dst_map (:16-27)
        for k in range(1, int(lab.max()) + 1):
            d = ndi.distance_transform_edt(lab == k)
            canvas += (255 * (d / np.amax(d))).astype('uint8')
    else:
        canvas = ws_dist(lab)
    marker = canvas > 125                                                         # (:39-41)
    marker = ndi.binary_fill_holes(marker)
    marker = ndi.binary_erosion(marker, iterations=1)
    marker = ndi.label(marker)[0].astype(np.int32)
    marker = remove_small_labels(marker, min_size)                                # (:46)
    out = watershed((-canvas.astype(np.int32) & 255).astype(np.uint8), marker, pred)   # -dist on uint8 wraps (:47)
    out = remove_small_labels(out, min_size)                                      # (:48)
    return dict(dist=canvas, marker=marker, labels=out)


def fill_label_process(pred, min_size=10):
    """postproc_other.process with ws = False (postproc_other.py:33-34, 49-52), by the very scipy calls the reference makes;
    skimage.morphology.remove_small_objects on an integer label image = zero every label whose pixel count is below min_size."""
    from scipy import ndimage as ndi
    b = np.asarray(pred) > 0.5
    lab = ndi.label(ndi.binary_fill_holes(b))[0].astype(np.int32)          # default structure: 4-connected
    sizes = np.bincount(lab.ravel())
    small = sizes < min_size
    small[0] = False
    lab[small[lab]] = 0
    return lab
