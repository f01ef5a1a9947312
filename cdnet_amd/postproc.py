"""Device-side inference post-processing (host orchestration over the C ABI; all compute in HIP kernels).

Operates on torch CUDA tensors (torch = device memory + streams only).  Mirrors, step by step, the reference's
test_dam.py post-processing:  get_probmaps epilogue (:982-1015) -> per-view DDM (:479-487) -> TTA mean (:445-450)
-> DDM fuse + point-guided boost + argmax (:490-539) -> fill holes / remove small / label / dilate (:546-563).
"""
import ctypes as C
import numpy as np
import torch

from . import _lib
from .data_prepare.SegFix_offset_helper import label_to_vector_mapping

# view transform codes: bit0 = horizontal flip, bit1 = vertical flip, bit2 = rotate 90 ccw first
# order of test_dam.py:459-467: [id, hf, vf, hvf, r90, r90_hf, r90_vf, r90_hvf]
TTA_XFORMS = (0, 1, 2, 3, 4, 5, 6, 7)

_LUT_CACHE = {}


def ddm_lut(classes):
    """int8 [classes, classes]: round(cos) between the vectors of two direction classes, with the reference's
    arithmetic (getDirectionDiffMap.py:92-101: float64 cosine with +1e-6 in the denominator, stored as float32,
    np.around)."""
    if classes not in _LUT_CACHE:
        v = np.array(label_to_vector_mapping[classes], dtype=np.float64)
        dot = v @ v.T
        nrm = np.sqrt((v * v).sum(1))
        cos = (dot / (nrm[:, None] * nrm[None, :] + 0.000001)).astype(np.float32)
        _LUT_CACHE[classes] = np.ascontiguousarray(np.around(cos).astype(np.int8))
    return _LUT_CACHE[classes]


def _nbr_extra(classes):
    if classes - 1 == 4:
        return 4, 0
    if classes - 1 == 8:
        return 8, 0
    if classes - 1 == 16:
        return 8, 1
    raise ValueError('direction_classes must be 5, 9 or 17 (getDirectionDiffMap.py:58,69)')


def ddm_codes(dcm, classes, out=None, minmax_out=None):
    """dcm uint8 [N,H,W] (cuda) -> (code uint8 [N,H,W], minmax int32 [N,2]); `out` / `minmax_out`: contiguous tensors of those element
    counts to write into (slices of a larger buffer)"""
    assert dcm.dtype == torch.uint8 and dcm.dim() == 3
    N, H, W = dcm.shape
    dcm = dcm.contiguous()
    code = torch.empty_like(dcm) if out is None else out
    minmax = torch.empty((N, 2), dtype=torch.int32, device=dcm.device) if minmax_out is None else minmax_out
    assert code.is_contiguous() and code.numel() == dcm.numel() and minmax.is_contiguous() and minmax.numel() == 2 * N
    lut = ddm_lut(classes)
    nbr, extra = _nbr_extra(classes)
    _lib.call('cdnet_ddm_codes', _lib.ptr(dcm), N, H, W, classes, lut.ctypes.data_as(C.c_void_p), nbr, extra,
              _lib.ptr(code), _lib.ptr(minmax), _lib.stream_ptr())
    return code, minmax


def ddm_normalize(code, minmax):
    N, H, W = code.shape
    out = torch.empty((N, H, W), dtype=torch.float32, device=code.device)
    _lib.call('cdnet_ddm_normalize', _lib.ptr(code), _lib.ptr(minmax), N, H, W, _lib.ptr(out), _lib.stream_ptr())
    return out


def probmaps(mask_logits, dir_logits, prob_out=None, dcm_out=None):
    """float32 [N,3,H,W], [N,C,H,W] -> prob float32 [N,3,H,W], dcm uint8 [N,H,W] (`prob_out` / `dcm_out`: contiguous tensors of those element
    counts to write into)."""
    assert mask_logits.dtype == torch.float32 and dir_logits.dtype == torch.float32
    N, _, H, W = mask_logits.shape
    Cd = dir_logits.shape[1]
    mask_logits, dir_logits = mask_logits.contiguous(), dir_logits.contiguous()
    prob = torch.empty_like(mask_logits) if prob_out is None else prob_out
    dcm = torch.empty((N, H, W), dtype=torch.uint8, device=mask_logits.device) if dcm_out is None else dcm_out
    assert prob.is_contiguous() and prob.numel() == mask_logits.numel() and prob.dtype == torch.float32
    assert dcm.is_contiguous() and dcm.numel() == N * H * W and dcm.dtype == torch.uint8
    _lib.call('cdnet_probmaps', _lib.ptr(mask_logits), _lib.ptr(dir_logits), N, Cd, H, W, _lib.ptr(prob),
              _lib.ptr(dcm), _lib.stream_ptr())
    return prob, dcm


def tta_boost_argmax(probs, points, codes, minmax, xforms, H, W, want_stages=True):
    """probs f32 [I,V,3,hv,wv] (flat per view), points f32 [I,V,hv,wv], codes u8 [I,V,hv,wv], minmax i32 [I,V,2].
    Views are stored in their own frame; xforms[v] in 0..7.  For mixed (rotated) frames pass flat tensors of
    shape [I, V, 3*H*W] etc. - only the element count per view matters."""
    I, V = probs.shape[0], probs.shape[1]
    dev = probs.device
    probs, points, codes, minmax = probs.contiguous(), points.contiguous(), codes.contiguous(), minmax.contiguous()
    assert probs.numel() == I * V * 3 * H * W and points.numel() == I * V * H * W and codes.numel() == I * V * H * W
    prob_mean = torch.empty((I, 3, H, W), dtype=torch.float32, device=dev) if want_stages else None
    # one view in its own frame: the mean over views is the view - no copy (the library takes the maximum of the point map from `points`)
    single = V == 1 and int(xforms[0]) == 0 and not want_stages and (H * W) % 4 == 0 and points.data_ptr() % 16 == 0
    point_mean = None if single else torch.empty((I, H, W), dtype=torch.float32, device=dev)
    ddm16 = torch.empty((I, H, W), dtype=torch.uint8, device=dev) if want_stages else None
    pred = torch.empty((I, H, W), dtype=torch.uint8, device=dev)
    pmax = torch.empty((I,), dtype=torch.float32, device=dev)
    xf = (C.c_int * V)(*[int(x) for x in xforms])
    _lib.call('cdnet_tta_boost_argmax', _lib.ptr(probs), _lib.ptr(points), _lib.ptr(codes), _lib.ptr(minmax),
              I, V, C.cast(xf, C.c_void_p), H, W, _lib.ptr(prob_mean), _lib.ptr(point_mean), _lib.ptr(ddm16),
              _lib.ptr(pred), _lib.ptr(pmax), _lib.stream_ptr())
    return dict(prob_mean=prob_mean, point_mean=points.reshape(I, H, W) if single else point_mean, ddm16=ddm16, pred=pred)


_WS = {}


def _workspace(nbytes, device, kind='cc'):
    key = (device.index, torch.cuda.current_stream().cuda_stream, kind)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def cc_chain(pred, fg_value=1, min_area=20, radius=2, want_stages=False):
    """pred uint8 [N,H,W] -> dict(final int32 [N,H,W], counts int32 [N] [, fill, small, label])."""
    assert pred.dtype == torch.uint8 and pred.dim() == 3
    N, H, W = pred.shape
    pred = pred.contiguous()
    dev = pred.device
    nbytes = _lib.load().cdnet_cc_workspace_bytes(N, H, W)
    ws = _workspace(nbytes, dev)
    final = torch.empty((N, H, W), dtype=torch.int32, device=dev)
    counts = torch.empty((N,), dtype=torch.int32, device=dev)
    fill = small = label = None
    if want_stages:
        fill = torch.empty((N, H, W), dtype=torch.uint8, device=dev)
        small = torch.empty((N, H, W), dtype=torch.uint8, device=dev)
        label = torch.empty((N, H, W), dtype=torch.int32, device=dev)
    _lib.call('cdnet_cc_chain', _lib.ptr(pred), int(fg_value), N, H, W, int(min_area), int(radius), _lib.ptr(ws),
              ws.numel(), _lib.ptr(fill), _lib.ptr(small), _lib.ptr(label), _lib.ptr(final), _lib.ptr(counts),
              _lib.stream_ptr())
    out = dict(final=final, counts=counts)
    if want_stages:
        out.update(fill=fill, small=small, label=label)
    return out


def tile_postproc_eligible(B, classes, H, W):
    """the fused two-launch chain takes this batch of tiles (cdnet_tile_postproc: W a multiple of 64, at most 65536 pixels per tile)"""
    return _lib.load().cdnet_tile_postproc_workspace_bytes(int(B), int(classes), int(H), int(W)) > 0


def tile_postproc(mask_logits, dir_logits, point, min_area=20, radius=2, want_stages=False, want_prob=False):
    """The whole post-processing chain of a batch of independent tiles (one view each) in TWO launches (cdnet_tile_postproc): get_probmaps
    epilogue + direction-difference codes, then boost / arg-max / fill holes / remove small / label / dilate of a tile inside one workgroup.
    mask_logits f32 [B,3,H,W], dir_logits f32 [B,C,H,W], point f32 [B,1,H,W] or [B,H,W] - what Unet.forward returns.
    Returns dict(final i32 [B,H,W], counts i32 [B], pred u8, dcm u8, minmax i32 [B,2] [, prob] [, fill, small, label]) - bit-identical to
    probmaps -> ddm_codes -> tta_boost_argmax -> cc_chain."""
    assert mask_logits.dtype == torch.float32 and dir_logits.dtype == torch.float32 and point.dtype == torch.float32
    B, _, H, W = mask_logits.shape
    Cd = dir_logits.shape[1]
    dev = mask_logits.device
    mask_logits, dir_logits, point = mask_logits.contiguous(), dir_logits.contiguous(), point.contiguous()
    assert point.numel() == B * H * W
    nbytes = _lib.load().cdnet_tile_postproc_workspace_bytes(B, Cd, H, W)
    assert nbytes > 0, 'shape not served by the fused tile chain: ask tile_postproc_eligible first'
    ws = _workspace(nbytes, dev, 'tile')
    u8 = lambda: torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    i32 = lambda: torch.empty((B, H, W), dtype=torch.int32, device=dev)
    prob = torch.empty_like(mask_logits) if (want_prob or want_stages) else None
    dcm, pred, final = u8(), u8(), i32()
    minmax = torch.empty((B, 2), dtype=torch.int32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    fill = small = label = None
    if want_stages:
        fill, small, label = u8(), u8(), i32()
    lut = ddm_lut(Cd)
    nbr, extra = _nbr_extra(Cd)
    _lib.call('cdnet_tile_postproc', _lib.ptr(mask_logits), _lib.ptr(dir_logits), _lib.ptr(point), B, Cd, H, W,
              lut.ctypes.data_as(C.c_void_p), nbr, extra, int(min_area), int(radius), _lib.ptr(ws), ws.numel(), _lib.ptr(prob), _lib.ptr(dcm),
              _lib.ptr(minmax), _lib.ptr(pred), _lib.ptr(fill), _lib.ptr(small), _lib.ptr(label), _lib.ptr(final), _lib.ptr(counts),
              _lib.stream_ptr())
    out = dict(final=final, counts=counts, pred=pred, dcm=dcm, minmax=minmax, point=point)
    if prob is not None:
        out['prob'] = prob
    if want_stages:
        out.update(fill=fill, small=small, label=label)
    return out


def postprocess_views(probs, points, dcms, xforms=None, H=None, W=None, classes=9, min_area=20, radius=2,
                      want_stages=False, check=True):
    """Everything after get_probmaps for I images with V views each (test_dam.py:445-563).
    probs f32 [I,V,3,H,W], points f32 [I,V,H,W] (or [I,V,1,H,W]), dcms u8 [I,V,H,W] (or [I,V,1,H,W]); views in
    their own frame.  Raises AssertionError (like test_dam.py:535) if a view's DDM is constant (0/0 = NaN) - `check=False` leaves that test
    to the caller (`check_views(r)` on the returned dict): the assertion reads the codes' (min, max) back, a device-to-host synchronisation a
    pipelined caller (test_dam.main: image i + 1 in flight while image i is copied out) postpones."""
    I, V = probs.shape[0], probs.shape[1]
    if H is None:
        H, W = probs.shape[-2:]
    if xforms is None:
        xforms = [0] * V
    plane = H * W
    dcm_flat = dcms.reshape(I * V, -1)
    # a rotated view is [W][H]: DDM needs its true 2-D shape
    codes = torch.empty((I, V, plane), dtype=torch.uint8, device=probs.device)
    minmax = torch.empty((I, V, 2), dtype=torch.int32, device=probs.device)
    groups = {}
    for v, xf in enumerate(xforms):
        groups.setdefault(bool(xf & 4), []).append(v)
    dflat = dcms.reshape(I, V, plane)
    if H == W or len(groups) == 1:
        # one launch over every view: a rotated view of a square image has the image's own shape
        rot = next(iter(groups)) if len(groups) == 1 else False
        hv, wv = (W, H) if rot else (H, W)
        ddm_codes(dflat.reshape(I * V, hv, wv), classes, out=codes, minmax_out=minmax)
    else:
        for rot, vs in groups.items():
            hv, wv = (W, H) if rot else (H, W)
            if I == 1 and vs == list(range(vs[0], vs[0] + len(vs))):
                # the views of the group lie next to each other (the reference's order: four plain, four rotated): written in place
                ddm_codes(dflat[0, vs[0]:vs[0] + len(vs)].reshape(len(vs), hv, wv), classes, out=codes[0, vs[0]:vs[0] + len(vs)],
                          minmax_out=minmax[0, vs[0]:vs[0] + len(vs)])
                continue
            idx = torch.tensor(vs, device=probs.device)
            c, mm = ddm_codes(dflat[:, idx].reshape(I * len(vs), hv, wv), classes)
            codes[:, idx] = c.reshape(I, len(vs), plane)
            minmax[:, idx] = mm.reshape(I, len(vs), 2)
    r = tta_boost_argmax(probs.reshape(I, V, 3 * plane), points.reshape(I, V, plane), codes, minmax, xforms, H, W,
                         want_stages=want_stages)
    cc = cc_chain(r['pred'], 1, min_area, radius, want_stages=want_stages)
    r.update(cc)
    r['codes'], r['minmax'] = codes, minmax
    if check:
        check_views(r)
    return r


def check_views(r, minmax_host=None):
    """the reference's assertion test_dam.py:535 for a result of postprocess_views (or its (min, max) codes already copied to the host)"""
    mm = r['minmax'].cpu() if minmax_host is None else minmax_host
    assert bool((mm[..., 0] != mm[..., 1]).all()), \
        'a view has a constant direction-difference map: 0/0 -> NaN; the reference asserts here (test_dam.py:535)'
