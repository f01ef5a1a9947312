"""Inference pipelines on the device: model forward -> get_probmaps epilogue -> direction-difference map ->
boost/argmax -> connected-component chain, without leaving the GPU (only the final int32 label maps do).

`infer_tiles`   : a batch of independent tiles, one view each (the benchmark unit "256x256 tile incl. post-proc")
`infer_image`   : the reference's per-image procedure of test_dam.py:297-563: 8 dihedral TTA views, whole-image
                  forward (all_img_test == 1) or sliding windows (utils.split_forward_dam), per-view DDM, mean,
                  point-guided boost, CC chain.
"""
import contextlib

import torch

from . import postproc


@torch.no_grad()
def infer_tiles(model, x, classes=9, min_area=20, radius=2, want_stages=False, post_stream=None, check=False, want_prob=False, fused=True):
    """x: float32 NCHW [B,3,H,W] on the GPU.  Returns dict(final int32 [B,H,W], counts, pred, dcm, minmax, ...; `prob` with want_prob / want_stages).
    `fused=False` takes the per-step chain (probmaps -> ddm_codes -> tta_boost_argmax -> cc_chain: ~17 launches) whatever the shape - the
    form every shape falls back to that the two-launch chain does not serve (W not a multiple of 64, more than 65536 pixels per tile).

    `post_stream` (a torch.cuda.Stream; cdnet_amd.streams.side_stream() picks one that does not share the compute stream's hardware
    queue): the post-processing chain is queued on that stream, ordered after this batch's forward, and
    the call returns at once - the small, latency-bound connected-component / direction kernels of batch i then run beside the
    convolutions of batch i + 1 queued on the caller's stream.  The returned tensors belong to `post_stream`: wait for `r['done']`
    (an event) - or synchronize - before reading them on another stream.

    The reference asserts PER IMAGE on a constant direction-difference map (0/0 -> NaN, test_dam.py:535) - `infer_image` keeps that assertion.
    A batch of independent tiles may legitimately hold a background-only tile, and a host-side assert costs a device-to-host
    synchronisation: here the test is opt-in.  `check=True` (serial form only) raises the reference's AssertionError for the batch;
    otherwise call `check_tiles(r)` when wanted - it reads the (min, max) codes the DDM kernel already leaves in `r['minmax']` (int32 [B, 2];
    no extra launch in the chain), after waiting for `r['done']` in the pipelined form."""
    assert not model.training
    mask, point, direction = model(x)
    B, _, H, W = mask.shape
    ctx = contextlib.nullcontext()
    if post_stream is not None:
        post_stream.wait_stream(torch.cuda.current_stream())
        for t in (mask, point, direction):
            t.record_stream(post_stream)                  # (allocated on the caller's stream, last read on the other one)
        ctx = torch.cuda.stream(post_stream)
    with ctx:
        if fused and postproc.tile_postproc_eligible(B, direction.shape[1], H, W):
            # two launches: probabilities / direction classes / DDM codes, then everything else of a tile in one workgroup (csrc/postproc_tile.hip);
            # the probability planes are written only when asked for (want_prob / want_stages)
            r = postproc.tile_postproc(mask, direction, point, min_area, radius, want_stages=want_stages, want_prob=want_prob)
        else:
            prob, dcm = postproc.probmaps(mask, direction)                        # test_dam.py:984, 1011-1013
            code, minmax = postproc.ddm_codes(dcm, classes)                       # generate_dd_map per tile
            r = postproc.tta_boost_argmax(prob.reshape(B, 1, 3 * H * W), point.reshape(B, 1, H * W),
                                          code.reshape(B, 1, H * W), minmax.reshape(B, 1, 2), [0], H, W,
                                          want_stages=want_stages)
            cc = postproc.cc_chain(r['pred'], 1, min_area, radius, want_stages=want_stages)
            r.update(cc)
            r.update(prob=prob, dcm=dcm, minmax=minmax, point=point)
        if post_stream is not None:
            r['done'] = torch.cuda.Event()
            r['done'].record(post_stream)
    if check:
        assert post_stream is None and not torch.cuda.is_current_stream_capturing(), 'check=True synchronises: serial form, outside a graph capture'
        check_tiles(r)
    return r


def check_tiles(r):
    """the reference's `assert(np.min(enhanced_boundary) >= 0)` (test_dam.py:535: NaN when a tile's direction-difference map is constant) for
    a result of infer_tiles; reads the device flag (synchronises with the stream that produced it)"""
    if 'done' in r:
        r['done'].synchronize()
    if 'ddm_constant' in r:                                     # (a precomputed flag tensor, bool [B])
        flags = r['ddm_constant'].cpu()
    else:
        mm = r['minmax'].reshape(-1, 2).cpu()
        flags = mm[:, 0] == mm[:, 1]
    bad = torch.nonzero(flags).flatten().tolist()
    assert not bad, ('tile(s) %s have a constant direction-difference map: 0/0 -> NaN; the reference asserts here (test_dam.py:535)' % bad)


@torch.no_grad()
def infer_image(model, image, opt=None, tta=True, all_img_test=1, patch_size=256, overlap=40, classes=9, min_area=20,
                radius=2, want_stages=False, defer=False):
    """The reference's per-image inference (test_dam.py:297-563) for one image tensor [3,H,W] float32 on the GPU
    (already ToTensor'd / normalised): the eight dihedral views (TTA) through the network - whole image
    (all_img_test == 1, options.py:35) or 256/40 sliding windows (utils.split_forward_dam) - softmax / gated direction
    argmax per view (get_probmaps), per-view direction-difference maps, their mean, the point-guided boundary boost, argmax,
    fill holes, remove small objects, label, dilate.  Returns dict(final int32 [H,W], count, pred, ...).
    `defer=True`: nothing is read back - no host synchronisation; instead of `count` the device tensor `counts` stays in the result and the
    reference's constant-DDM assertion is left to the caller (postproc.check_views on `minmax`) - the form test_dam.main pipelines images with."""
    from . import utils
    if opt is not None:
        tta, all_img_test = opt.test['tta'], opt.all_img_test
        patch_size, overlap = opt.test['patch_size'], opt.test['overlap']
        classes, min_area, radius = opt.direction_classes, opt.post['min_area'], opt.post['radius']
    assert not model.training and image.dim() == 3
    _, H, W = image.shape
    xforms = list(postproc.TTA_XFORMS) if tta else [0]
    V = len(xforms)
    plane = H * W
    dev = image.device
    # every view's logits are stitched straight into one buffer per output (a rotated view as [K][W][H]: the same element count), get_probmaps'
    # epilogue is ONE launch over all views and writes the post-processing's inputs in place (round 5: eight launches and 24 device copies)
    mask_all = torch.empty((V, 3, plane), dtype=torch.float32, device=dev)
    dir_all = torch.empty((V, classes, plane), dtype=torch.float32, device=dev)
    probs = torch.empty((1, V, 3 * plane), dtype=torch.float32, device=dev)
    points = torch.empty((1, V, plane), dtype=torch.float32, device=dev)
    dcms = torch.empty((1, V, plane), dtype=torch.uint8, device=dev)
    bufs = (mask_all, points[0].view(V, 1, plane), dir_all)
    if all_img_test == 1:
        # whole-image forward: one window as large as the view
        utils.split_forward_views(model, image, max(H, W), 0, xforms, classes, out=bufs)
    else:
        utils.split_forward_views(model, image, patch_size, overlap, xforms, classes, out=bufs)
    postproc.probmaps(mask_all.view(V, 3, 1, plane), dir_all.view(V, classes, 1, plane), prob_out=probs, dcm_out=dcms)
    r = postproc.postprocess_views(probs, points, dcms, xforms=xforms, H=H, W=W, classes=classes, min_area=min_area,
                                   radius=radius, want_stages=want_stages, check=not defer)
    out = {k: (v[0] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 1 else v) for k, v in r.items()}
    if not defer:
        out['count'] = int(r['counts'][0])
    return out
