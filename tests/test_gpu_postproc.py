"""GPU parity of the integer post-processing path: HIP kernels (through the C ABI) vs the CPU oracle and the
golden vectors produced by the reference.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def test_ddm_golden_bit_exact(golden, dev):
    from cdnet_amd.data_prepare.getDirectionDiffMap import generate_dd_map
    z = golden('ddm')
    for name in z['names']:
        got = generate_dd_map(z['in_' + name], int(z['cls_' + name]))
        assert got.dtype == np.float32
        assert np.array_equal(got, z['out_' + name], equal_nan=True), name


def test_ddm_batched_vs_oracle(dev):
    import torch
    from cdnet_amd import postproc
    from oracle import postproc as orc
    rs = np.random.RandomState(3)
    for (N, H, W) in [(5, 33, 47), (3, 128, 256), (2, 250, 125), (1, 1000, 1000), (9, 7, 5)]:
        x = rs.randint(0, 9, size=(N, H, W)).astype(np.uint8)
        x[rs.rand(N, H, W) < 0.5] = 0
        code, mm = postproc.ddm_codes(torch.from_numpy(x).to(dev), 9)
        out = postproc.ddm_normalize(code, mm).cpu().numpy()
        for n in range(N):
            want, wcode = orc.generate_dd_map(x[n], 9, return_code=True)
            assert np.array_equal(code[n].cpu().numpy(), wcode)
            assert np.array_equal(out[n], want, equal_nan=True)
            assert mm[n].tolist() == [int(wcode.min()), int(wcode.max())]


def test_probmaps_epilogue(golden, dev):
    import torch
    from cdnet_amd import postproc
    z = golden('probmaps')
    for name in z['names']:
        ml = torch.from_numpy(z['mask_logits_' + name])[None].to(dev)
        dl = torch.from_numpy(z['dir_logits_' + name])[None].to(dev)
        prob, dcm = postproc.probmaps(ml, dl)
        np.testing.assert_allclose(prob[0].cpu().numpy(), z['prob_' + name], rtol=0, atol=3e-7)
        safe = z['margin_' + name] > 1e-6
        assert np.array_equal(dcm[0].cpu().numpy()[safe], z['dcm_' + name][0][safe])


@pytest.mark.parametrize('name', ['a', 'b', 'c', 'd'])
def test_postproc_chain_golden_bit_exact(golden, dev, name):
    import torch
    from cdnet_amd import postproc, synth
    z = golden('postproc')
    H, W, n, seed = [int(v) for v in z['cfg_' + name]]
    probs, points, dcms = synth.postproc_case(H, W, n, seed)
    assert synth.crc(probs, points, dcms) == z['crc_' + name]
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    r = postproc.postprocess_views(t(probs), t(points), t(dcms), want_stages=True)
    g = lambda k: r[k][0].cpu().numpy()
    assert synth.crc(g('prob_mean')) == z['prob_mean_crc_' + name]
    assert synth.crc(g('point_mean').reshape(1, H, W)) == z['point_mean_crc_' + name]
    assert np.array_equal(g('ddm16'), z['ddm_mean16_' + name])
    assert np.array_equal(g('pred'), z['pred_' + name])
    assert np.array_equal(g('fill'), z['fill_' + name])
    assert np.array_equal(g('small'), z['small_' + name])
    assert np.array_equal(g('label'), z['label_' + name])
    assert np.array_equal(g('final'), z['final_' + name])
    assert int(r['counts'][0]) == int(z['label_' + name].max())


def test_tta_view_transforms_match_numpy_unflips(dev):
    """views stored in their own frame + xform codes == np.flip / np.rot90(k=3) of test_dam.py:356-441"""
    import torch
    from cdnet_amd import postproc, synth
    from oracle import postproc as orc
    H, W = 72, 104
    probs, points, dcms = synth.postproc_case(H, W, 20, 9)
    # forward transforms as test_dam.py builds the views (PIL transpose / rotate(90, expand) == np.rot90 k=1)
    def fwd(a, xf):
        if xf & 4:
            a = np.rot90(a, k=1, axes=(-2, -1))
        if xf & 1:
            a = np.flip(a, -1)
        if xf & 2:
            a = np.flip(a, -2)
        return np.ascontiguousarray(a)
    xforms = postproc.TTA_XFORMS
    pv = [fwd(probs[v], xforms[v]) for v in range(8)]
    tv = [fwd(points[v], xforms[v]) for v in range(8)]
    dv = [fwd(dcms[v], xforms[v]) for v in range(8)]
    flat = lambda lst: torch.from_numpy(np.stack([a.reshape(-1) for a in lst])).to(dev)[None]
    r = postproc.postprocess_views(flat(pv), flat(tv), flat(dv), xforms=xforms, H=H, W=W, want_stages=True)
    want = orc.postprocess_views(probs, points, dcms)
    assert np.array_equal(r['prob_mean'][0].cpu().numpy(), want['prob_mean'])
    assert np.array_equal(r['pred'][0].cpu().numpy(), want['pred'])
    assert np.array_equal(r['final'][0].cpu().numpy(), want['final'])


def test_cc_chain_edge_cases_and_batches(dev):
    import torch
    from cdnet_amd import postproc
    from oracle import postproc as orc
    rs = np.random.RandomState(0)
    cases = [(1, 1, 1, 0.5), (2, 1, 9, 0.6), (2, 7, 1, 0.6), (1, 5, 5, 0.0), (1, 5, 5, 1.0), (3, 33, 65, 0.55),
             (4, 64, 64, 0.62), (2, 130, 257, 0.58), (1, 256, 256, 0.6), (1, 500, 380, 0.593), (1, 1000, 1000, 0.59),
             (16, 256, 256, 0.45)]
    for (N, H, W, p) in cases:
        x = (rs.rand(N, H, W) < p).astype(np.uint8)
        for min_area, radius in ((20, 2), (3, 1)):
            r = postproc.cc_chain(torch.from_numpy(x).to(dev), 1, min_area, radius, want_stages=True)
            for n in range(N):
                w = orc.cc_chain(x[n], min_area, radius)
                for k in ('fill', 'small', 'label', 'final'):
                    assert np.array_equal(r[k][n].cpu().numpy(), w[k]), (N, H, W, p, min_area, k, n)
                assert int(r['counts'][n]) == w['count']
            r2 = postproc.cc_chain(torch.from_numpy(x).to(dev), 1, min_area, radius, want_stages=False)
            assert torch.equal(r2['final'], r['final'])


def test_cc_chain_structured_shapes(dev):
    """spirals / checkerboards / nested rings: long union chains, holes inside holes, diagonal-only contacts"""
    import torch
    from cdnet_amd import postproc
    from oracle import postproc as orc
    H = W = 192
    imgs = []
    yy, xx = np.mgrid[:H, :W]
    imgs.append(((yy + xx) % 2).astype(np.uint8))                                   # checkerboard: one 8-conn blob
    imgs.append((((yy // 3) % 2) & (((xx + (yy // 3) * 7) % 190) > 2)).astype(np.uint8))   # serpentine bands
    r = np.sqrt((yy - 96.0) ** 2 + (xx - 96.0) ** 2)
    imgs.append(((r.astype(int) // 5) % 2).astype(np.uint8))                        # nested rings (holes in holes)
    sp = np.zeros((H, W), np.uint8)                                                 # square spiral, 1-px wide
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
    imgs.append(np.eye(H, dtype=np.uint8) | np.fliplr(np.eye(H, dtype=np.uint8)))  # diagonals only
    x = np.stack(imgs)
    for min_area in (1, 20):
        res = postproc.cc_chain(torch.from_numpy(x).to(dev), 1, min_area, 2, want_stages=True)
        for n in range(len(imgs)):
            w = orc.cc_chain(x[n], min_area, 2)
            for k in ('fill', 'small', 'label', 'final'):
                assert np.array_equal(res[k][n].cpu().numpy(), w[k]), (n, min_area, k)


def test_constant_view_raises_like_reference(dev):
    import torch
    from cdnet_amd import postproc, synth
    probs, points, dcms = synth.postproc_case(64, 64, 6, 1)
    dcms[3] = 0
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    with pytest.raises(AssertionError):
        postproc.postprocess_views(t(probs), t(points), t(dcms))
