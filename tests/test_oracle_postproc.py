"""Pin the CPU oracle (oracle/postproc*.{py,c}) against golden vectors produced by the reference itself."""
import numpy as np
import pytest
from oracle import postproc as orc
from cdnet_amd import synth


def test_ddm_bit_exact(golden):
    z = golden('ddm')
    for name in z['names']:
        x, want, classes = z['in_' + name], z['out_' + name], int(z['cls_' + name])
        got = orc.generate_dd_map(x, classes)
        assert got.dtype == np.float32
        assert np.array_equal(got, want, equal_nan=True), name
    assert np.isnan(z['out_allbg_9']).all()          # the reference's 0/0 contract for a constant view


def test_ddm_lut_is_sign_of_dot_for_9():
    lut = orc.ddm_lut(9)
    v = np.array(orc.LABEL_TO_VECTOR[9])
    assert np.array_equal(lut, np.sign(v @ v.T))


def test_probmaps_epilogue(golden):
    z = golden('probmaps')
    for name in z['names']:
        prob, dcm = orc.probmaps(z['mask_logits_' + name], z['dir_logits_' + name])
        np.testing.assert_allclose(prob, z['prob_' + name], rtol=0, atol=2e-7)
        safe = z['margin_' + name] > 1e-6
        assert safe.mean() > 0.999
        assert np.array_equal(dcm[safe], z['dcm_' + name][0][safe])


@pytest.mark.parametrize('name', ['a', 'b', 'c', 'd'])
def test_postproc_chain_bit_exact(golden, name):
    z = golden('postproc')
    H, W, n, seed = [int(v) for v in z['cfg_' + name]]
    probs, points, dcms = synth.postproc_case(H, W, n, seed)
    assert synth.crc(probs, points, dcms) == z['crc_' + name], 'synthetic input recipe drifted'
    r = orc.postprocess_views(probs, points, dcms)
    assert synth.crc(r['prob_mean']) == z['prob_mean_crc_' + name]
    assert synth.crc(r['point_mean'].reshape(1, H, W)) == z['point_mean_crc_' + name]
    assert np.array_equal(r['ddm_mean'] * 16, z['ddm_mean16_' + name])
    assert np.array_equal(r['inside3'], z['inside3_' + name])
    assert np.array_equal(r['pred'], z['pred_' + name])
    assert np.array_equal(r['fill'], z['fill_' + name])
    assert np.array_equal(r['small'], z['small_' + name])
    assert np.array_equal(r['label'], z['label_' + name])
    assert np.array_equal(r['final'], z['final_' + name])


def test_cc_edge_cases():
    # empty, full, single pixel, ragged sizes; checked against scipy definitions directly
    from scipy import ndimage as ndi
    rs = np.random.RandomState(0)
    for H, W, p in [(1, 1, 0.5), (1, 9, 0.6), (7, 1, 0.6), (5, 5, 0.0), (5, 5, 1.0), (33, 65, 0.55), (64, 64, 0.62)]:
        x = (rs.rand(H, W) < p).astype(np.uint8)
        r = orc.cc_chain(x, 3, 2)
        f = ndi.binary_fill_holes(x)
        assert np.array_equal(r['fill'], f.astype(np.uint8))
        l4, _ = ndi.label(f)
        keep = np.bincount(l4.ravel()) >= 3
        keep[0] = False
        s = keep[l4]
        assert np.array_equal(r['small'], s.astype(np.uint8))
        l8, n8 = ndi.label(s, structure=np.ones((3, 3)))
        assert np.array_equal(r['label'], l8) and r['count'] == n8
        yy, xx = np.mgrid[-2:3, -2:3]
        fp = (yy ** 2 + xx ** 2) <= 4
        assert np.array_equal(r['final'], ndi.grey_dilation(l8, footprint=fp, mode='constant', cval=0))
