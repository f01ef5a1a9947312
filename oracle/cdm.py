"""ORACLE (test infrastructure): numpy-facing wrapper over oracle/cdm_oracle.c - the reference's LabelEncoding
(my_transforms_direction.py:687-885, 3-class-PNG branch, do_direction = 1)."""
import ctypes as C
import numpy as np
from . import build as _build

_lib = None


def _l():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
    return _lib


def label_encoding(label_ch0, want_aux=False, want_field=False):
    """label_ch0: uint8 [H,W] (channel 0 of the 3-class label PNG).
    Returns (label3 u8 {0,127,255}, point float16 [H,W], direction u8 [H,W] in 0..8[, inst i32, centers i32 [n,2]])."""
    x = np.ascontiguousarray(label_ch0, dtype=np.uint8)
    H, W = x.shape
    label3 = np.empty((H, W), np.uint8)
    point = np.empty((H, W), np.float32)
    direction = np.empty((H, W), np.uint8)
    inst = np.empty((H, W), np.int32)
    centers = np.zeros((H * W // 4 + 1, 2), np.int32)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    field = np.zeros((H, W, 2), np.float32) if want_field else None
    _l().orc_set_direction_field_out(p(field, C.c_float) if want_field else None)
    n = _l().orc_label_encoding(p(x, C.c_uint8), H, W, p(label3, C.c_uint8), p(point, C.c_float), p(direction, C.c_uint8),
                                p(inst, C.c_int32), p(centers, C.c_int32))
    _l().orc_set_direction_field_out(None)
    out = (label3, point.astype(np.float16), direction)
    if want_aux:
        out = out + (inst, centers[:n].copy())
    if want_field:
        out = out + (field,)          # [H,W,2] float32: (row gradient, column gradient) the angle is taken from (:848)
    return out


def label_encoding_instances(label_inst):
    """LabelEncoding for INSTANCE-level label input (my_transforms_direction.py:752-760 + :785-871; `label_level_len > 2`, out_c = 3):
      new_label = (label > 0); remove_small_objects(new_label, 5) on the uint8 {0,1} image (= a label image: the whole foreground
      vanishes only when it has fewer than 5 pixels); boundary = dilation(label) & ~erosion(label, disk(1)) on the integer instance
      ids (bit-wise: non-zero exactly where the cross neighbourhood's max and min differ) -> 2; instances =
      dilation(postproc_other.process((new_label == 1) * 255, 'modelName', min_size=5), disk(1)) - the WATERSHED branch (the model
      name is neither 'unet' nor 'micronet').  Then the common per-instance stage.
    label_inst: integer [H,W].  Returns (label3 u8 {0,127,255}, point f16, direction u8, inst i32)."""
    from scipy import ndimage as ndi
    from . import postproc as orc
    lab = np.ascontiguousarray(label_inst).astype(np.int64)
    H, W = lab.shape
    cross = ndi.generate_binary_structure(2, 1)
    new_label = (lab > 0).astype(np.uint8)
    if new_label.sum() < 5:
        new_label[:] = 0
    inside = new_label.copy()
    mx = ndi.grey_dilation(lab, footprint=cross, mode='nearest')
    mn = ndi.grey_erosion(lab, footprint=cross, mode='nearest')
    new_label[mx != mn] = 2
    label3 = (new_label / 2 * 255).astype(np.uint8)
    ws = orc.watershed_process((new_label == 1).astype(np.uint8) * 255, min_size=5)['labels']
    inst = ndi.grey_dilation(ws.astype(np.int32), footprint=cross, mode='nearest').astype(np.int32)
    point = np.empty((H, W), np.float32)
    direction = np.empty((H, W), np.uint8)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    inst = np.ascontiguousarray(inst)
    inside = np.ascontiguousarray(inside)
    _l().orc_direction_from_instances(p(inst, C.c_int32), p(inside, C.c_uint8), H, W, p(point, C.c_float), p(direction, C.c_uint8), None)
    return label3, point.astype(np.float16), direction, inst
