"""GPU parity of the fused two-launch tile post-processing chain (cdnet_tile_postproc, csrc/postproc_tile.hip): bit-identical to the
per-step chain (probmaps -> ddm_codes -> tta_boost_argmax -> cc_chain, itself pinned to the reference's golden vectors in
test_gpu_postproc.py) on every output, and stage by stage against the CPU oracle (oracle/postproc_oracle.c)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _per_step(mask, point, dirs, classes, min_area, radius):
    from cdnet_amd import postproc
    B, _, H, W = mask.shape
    prob, dcm = postproc.probmaps(mask, dirs)
    code, minmax = postproc.ddm_codes(dcm, classes)
    r = postproc.tta_boost_argmax(prob.reshape(B, 1, 3 * H * W), point.reshape(B, 1, H * W), code.reshape(B, 1, H * W), minmax.reshape(B, 1, 2),
                                  [0], H, W, want_stages=False)
    r.update(postproc.cc_chain(r['pred'], 1, min_area, radius, want_stages=True))
    r.update(prob=prob, dcm=dcm, minmax=minmax, code=code)
    return r


def _nuclei_logits(B, H, W, classes, dev, seed):
    """logits synthesised from the centripetal-direction maps of rendered nuclei + noise (what tools/bench_postproc.py times)"""
    import torch
    from cdnet_amd import synth
    from cdnet_amd.my_transforms_direction import label_encoding_batch
    if min(H, W) < 16:                                         # (no room for an ellipse: noise with a foreground bias)
        return _noise_logits(B, H, W, classes, dev, seed, 0.7)
    rs = np.random.RandomState(seed)
    n = max(2, 60 * H * W // 65536)
    lab = np.stack([(synth.ellipse_instances(H, W, n, rs, 5, 12, 4) > 0).astype(np.uint8) * 255 for _ in range(B)])
    lab3, point, direction = label_encoding_batch(torch.from_numpy(lab).to(dev))
    g = torch.Generator(device=dev).manual_seed(seed)
    onehot = lambda t, C: torch.nn.functional.one_hot(t.long(), C).permute(0, 3, 1, 2).float()
    mask = 5.0 * onehot((lab3.reshape(B, H, W).long() + 1) // 128, 3) + 1.5 * torch.randn((B, 3, H, W), device=dev, generator=g)
    d8 = direction.reshape(B, H, W).long()
    if classes != 9:                                           # (any labelling with `classes` values and class 0 as background)
        d8 = torch.where(d8 > 0, (d8 - 1) * (classes - 1) // 8 + 1, d8)
    dirs = 5.0 * onehot(d8, classes) + 1.5 * torch.randn((B, classes, H, W), device=dev, generator=g)
    pt = point.reshape(B, 1, H, W).float() + 0.05 * torch.randn((B, 1, H, W), device=dev, generator=g)
    return mask.contiguous(), pt.contiguous(), dirs.contiguous()


def _noise_logits(B, H, W, classes, dev, seed, fg):
    """pure noise: thousands of tiny components, holes everywhere - the union-find's worst customer"""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    mask = 2.0 * torch.randn((B, 3, H, W), device=dev, generator=g)
    mask[:, 1] += fg
    dirs = 2.0 * torch.randn((B, classes, H, W), device=dev, generator=g)
    pt = torch.rand((B, 1, H, W), device=dev, generator=g)
    return mask.contiguous(), pt.contiguous(), dirs.contiguous()


def _assert_same(f, s, what):
    import torch
    for k in ('dcm', 'minmax', 'pred', 'fill', 'small', 'label', 'final', 'counts'):
        assert torch.equal(f[k].reshape(-1), s[k].reshape(-1)), (what, k, int((f[k].reshape(-1) != s[k].reshape(-1)).sum()))
    assert torch.equal(f['prob'].view(torch.int32), s['prob'].view(torch.int32)), (what, 'prob')


@pytest.mark.parametrize('B,H,W,classes', [(3, 256, 256, 9), (5, 64, 64, 9), (2, 128, 256, 9), (2, 256, 128, 9), (2, 512, 128, 9), (1, 16, 64, 9),
                                           (2, 128, 128, 5), (2, 128, 128, 17), (1, 1024, 64, 9), (3, 1, 64, 9), (2, 3, 192, 9), (2, 80, 320, 9)])
def test_fused_equals_per_step_chain(dev, B, H, W, classes):
    import torch
    from cdnet_amd import postproc
    assert postproc.tile_postproc_eligible(B, classes, H, W)
    cases = [('nuclei', _nuclei_logits(B, H, W, classes, dev, 11)),
             ('noise', _noise_logits(B, H, W, classes, dev, 5, 0.0)),
             ('dense noise', _noise_logits(B, H, W, classes, dev, 6, 1.5)),
             ('sparse noise', _noise_logits(B, H, W, classes, dev, 7, -2.0))]
    for what, (mask, pt, dirs) in cases:
        for min_area, radius in ((20, 2), (3, 1), (1, 0)):
            f = postproc.tile_postproc(mask, dirs, pt, min_area, radius, want_stages=True)
            s = _per_step(mask, pt, dirs, classes, min_area, radius)
            _assert_same(f, s, (what, min_area, radius))
            f2 = postproc.tile_postproc(mask, dirs, pt, min_area, radius)           # no stage outputs: same labels
            assert torch.equal(f2['final'], f['final']) and torch.equal(f2['counts'], f['counts']) and 'prob' not in f2
            torch.cuda.synchronize()


def test_fused_batch_of_64_tiles_is_deterministic(dev):
    """the benchmark unit: 64 tiles of 256x256 - twice, same bits; equal to the per-step chain"""
    import torch
    from cdnet_amd import postproc
    mask, pt, dirs = _nuclei_logits(64, 256, 256, 9, dev, 3)
    a = postproc.tile_postproc(mask, dirs, pt, want_stages=True)
    b = postproc.tile_postproc(mask, dirs, pt, want_stages=True)
    s = _per_step(mask, pt, dirs, 9, 20, 2)
    _assert_same(a, s, '64 tiles')
    _assert_same(b, s, '64 tiles, second run')
    assert int(a['counts'].min()) > 10


def _logits_of_mask(m, dev):
    """logits whose arg-max is the binary image `m` ([B,H,W] of 0 / 1): saturated mask logits, ONE direction class everywhere (the codes are 0
    inside the image and 1 on its border: a non-constant direction-difference map, the reference's assertion holds) and a constant point map
    (every pixel lies inside `inside3`: the boost is zero everywhere, test_dam.py:532)"""
    import torch
    B, H, W = m.shape
    t = torch.from_numpy(m.astype(np.float32)).to(dev)
    mask = torch.stack([12.0 * (1 - t), 12.0 * t, torch.full_like(t, -12.0)], 1).contiguous()
    dirs = torch.zeros((B, 9, H, W), device=dev)
    dirs[:, 1] = 8.0
    pt = torch.ones((B, 1, H, W), device=dev)
    return mask, pt, dirs.contiguous()


def test_fused_structured_shapes_vs_oracle(dev):
    """spirals / checkerboards / nested rings / diagonals as the network's arg-max: long union chains over 16-bit labels, holes inside holes,
    diagonal-only contacts, a full and an empty tile - the connected-component part against oracle/postproc_oracle.c"""
    import torch
    from cdnet_amd import postproc
    from oracle import postproc as orc
    H = W = 192
    yy, xx = np.mgrid[:H, :W]
    imgs = [((yy + xx) % 2).astype(np.uint8),
            (((yy // 3) % 2) & (((xx + (yy // 3) * 7) % 190) > 2)).astype(np.uint8)]
    r = np.sqrt((yy - 96.0) ** 2 + (xx - 96.0) ** 2)
    imgs.append(((r.astype(int) // 5) % 2).astype(np.uint8))
    sp = np.zeros((H, W), np.uint8)
    y = x = 2; dy, dx = 0, 1; L = W - 5
    while L > 2:
        for _ in range(L):
            sp[y, x] = 1; y += dy; x += dx
        dy, dx = dx, -dy
        for _ in range(L):
            sp[y, x] = 1; y += dy; x += dx
        dy, dx = dx, -dy
        L -= 4
    imgs.append(sp)
    imgs.append(1 - sp)                                                             # the spiral as background: a hole-free 1-px corridor
    imgs.append(np.eye(H, dtype=np.uint8) | np.fliplr(np.eye(H, dtype=np.uint8)))
    imgs.append(np.ones((H, W), np.uint8))
    imgs.append(np.zeros((H, W), np.uint8))
    ring = np.zeros((H, W), np.uint8); ring[0, :] = ring[-1, :] = ring[:, 0] = ring[:, -1] = 1     # a frame on the border: everything inside is a hole
    imgs.append(ring)
    m = np.stack(imgs)
    mask, pt, dirs = _logits_of_mask(m, dev)
    for min_area in (1, 20):
        f = postproc.tile_postproc(mask, dirs, pt, min_area, 2, want_stages=True)
        pred = f['pred'].cpu().numpy()
        for n in range(len(imgs)):
            inside = pred[n] == 1
            want_in = m[n].astype(bool).copy()
            assert np.array_equal(inside, want_in), (n, int((inside != want_in).sum()))
            w = orc.cc_chain(inside, min_area, 2)
            for k in ('fill', 'small', 'label', 'final'):
                assert np.array_equal(f[k][n].cpu().numpy(), w[k]), (n, min_area, k)
            assert int(f['counts'][n]) == w['count']


def test_fused_front_stages_vs_oracle(dev):
    """direction-difference codes and boost / arg-max of the fused chain against the oracle, fed the device's own previous stage (the
    soft-max itself is pinned by test_gpu_postproc.py::test_probmaps_epilogue on the kernel whose arithmetic this one repeats)"""
    import torch
    from cdnet_amd import postproc
    from oracle import postproc as orc
    mask, pt, dirs = _nuclei_logits(3, 256, 256, 9, dev, 21)
    f = postproc.tile_postproc(mask, dirs, pt, want_stages=True)
    for n in range(3):
        dcm = f['dcm'][n].cpu().numpy()
        ddm, code = orc.generate_dd_map(dcm, 9, return_code=True)
        assert f['minmax'][n].tolist() == [int(code.min()), int(code.max())]
        with np.errstate(all='ignore'):
            w = orc.fuse_boost_argmax(f['prob'][n].cpu().numpy()[None], pt[n].cpu().numpy()[None], ddm[None])
        assert np.array_equal(f['pred'][n].cpu().numpy(), w['pred'])
        c = orc.cc_chain(w['pred'] == 1, 20, 2)
        assert np.array_equal(f['final'][n].cpu().numpy(), c['final']) and int(f['counts'][n]) == c['count']


def test_constant_ddm_tile_keeps_the_reference_semantics(dev):
    """a tile without any direction class: 0/0 -> NaN in the boost (the reference asserts, test_dam.py:535); here (min, max) of the codes are
    equal - what pipeline.check_tiles reads - and the arithmetic is the per-step chain's (NaN wins the arg-max)"""
    import torch
    from cdnet_amd import postproc, pipeline
    mask, pt, dirs = _nuclei_logits(3, 128, 128, 9, dev, 2)
    dirs[1] = 0.0
    dirs[1, 0] = 9.0
    mask[1] = 0.0
    mask[1, 0] = 9.0                                           # (class 0 is gated by the background probability, test_dam.py:1011: background everywhere)
    f = postproc.tile_postproc(mask, dirs, pt, want_stages=True)
    s = _per_step(mask, pt, dirs, 9, 20, 2)
    _assert_same(f, s, 'constant tile')
    assert f['minmax'][1, 0] == f['minmax'][1, 1] and f['minmax'][0, 0] != f['minmax'][0, 1]
    with pytest.raises(AssertionError, match='constant direction-difference map'):
        pipeline.check_tiles(f)


def test_unsupported_shapes_are_refused(dev):
    import torch
    from cdnet_amd import postproc, _lib
    assert not postproc.tile_postproc_eligible(1, 9, 100, 100)        # W not a multiple of 64
    assert not postproc.tile_postproc_eligible(1, 9, 512, 256)        # more than 65536 pixels
    assert not postproc.tile_postproc_eligible(1, 7, 64, 64)          # direction classes
    x = torch.zeros((1, 3, 100, 100), device=dev)
    with pytest.raises(AssertionError):
        postproc.tile_postproc(x, torch.zeros((1, 9, 100, 100), device=dev), torch.zeros((1, 1, 100, 100), device=dev))


def test_infer_tiles_fused_equals_per_step(dev):
    """pipeline.infer_tiles: the fused chain (default) and the per-step chain give the same label maps behind the same network"""
    import torch
    import cdnet_amd
    from cdnet_amd import pipeline, synth
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    torch.manual_seed(5)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev).eval()
    x = torch.from_numpy(synth.tiles_u8(4, seed=3).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous().to(dev)
    for prec in ('bf16', 'fp32'):
        cdnet_amd.set_precision(prec)
        a = pipeline.infer_tiles(m, x, want_prob=True)
        b = pipeline.infer_tiles(m, x, fused=False)
        for k in ('final', 'counts', 'pred', 'dcm', 'minmax', 'prob'):
            assert torch.equal(a[k].reshape(-1), b[k].reshape(-1)), (prec, k)
    cdnet_amd.set_precision('bf16')
