"""GPU: cdnet_watershed_process (csrc/postproc.hip) against the oracle, bit-exact at every stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blobs(H, W, n, seed, rmin=4, rmax=11):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[:H, :W]
    m = np.zeros((H, W), bool)
    for _ in range(n):
        cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(rmin, rmax)
        m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    return m.astype(np.uint8)


@pytest.mark.parametrize('case', [(64, 64, 14, 1), (96, 80, 30, 2), (50, 120, 25, 3), (128, 128, 60, 4), (33, 65, 8, 5)])
def test_stages_bit_exact(case):
    import torch
    from cdnet_amd import postproc_other
    from oracle import postproc as op
    H, W, n, seed = case
    preds = np.stack([_blobs(H, W, n, seed), _blobs(H, W, n // 2 + 1, seed + 100), np.zeros((H, W), np.uint8)])
    lab, dist, marker = postproc_other.watershed_process(torch.from_numpy(preds).cuda(), 10, stages=True)
    for i in range(preds.shape[0]):
        want = op.watershed_process(preds[i], 10, use_scipy=True)
        np.testing.assert_array_equal(dist[i].cpu().numpy(), want['dist'])
        np.testing.assert_array_equal(marker[i].cpu().numpy(), want['marker'])
        np.testing.assert_array_equal(lab[i].cpu().numpy(), want['labels'])


def test_process_signature_and_full_tile():
    """the reference's call shape (numpy HW float map in, numpy labels out) on a 256x256 tile"""
    from cdnet_amd import postproc_other
    from oracle import postproc as op
    pred = _blobs(256, 256, 220, 11).astype(np.float32) * 0.9
    got = postproc_other.process(pred.copy(), 'dam', min_size=10, ws=True)
    want = op.watershed_process(pred, 10)['labels']
    assert isinstance(got, np.ndarray) and got.shape == (256, 256)
    np.testing.assert_array_equal(got, want)


def test_touching_border_and_full_mask():
    import torch
    from cdnet_amd import postproc_other
    from oracle import postproc as op
    a = np.ones((40, 40), np.uint8)
    a[0, 0] = 0                                  # one background pixel: a huge instance, max distance far from it
    b = np.zeros((40, 40), np.uint8)
    b[:, :7] = 1                                 # strip along the border
    b[10:30, 20:40] = 1
    preds = np.stack([a, b])
    lab, dist, marker = postproc_other.watershed_process(torch.from_numpy(preds).cuda(), 10, stages=True)
    for i in range(2):
        want = op.watershed_process(preds[i], 10)
        np.testing.assert_array_equal(dist[i].cpu().numpy(), want['dist'])
        np.testing.assert_array_equal(lab[i].cpu().numpy(), want['labels'])


def _holes(H, W, n, seed):
    """blobs with punched holes (some touching the border) and specks below the size threshold"""
    rs = np.random.RandomState(seed)
    m = _blobs(H, W, n, seed, 5, 14)
    yy, xx = np.mgrid[:H, :W]
    for _ in range(n):
        cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(1, 4)
        m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0
    for _ in range(n):
        m[rs.randint(0, H), rs.randint(0, W)] = 1
    return m


@pytest.mark.parametrize('case', [(64, 64, 10, 1), (96, 80, 24, 2), (50, 120, 20, 3), (256, 256, 150, 4), (33, 65, 6, 5), (1000, 1000, 900, 6)])
def test_ws_false_branch_bit_exact(case):
    """postproc_other.process(pred, 'unet') / ws=False (postproc_other.py:35, 49-52): fill holes -> 4-connected label -> remove
    small labels, against the same scipy calls the reference makes"""
    import torch
    from cdnet_amd import postproc_other
    from oracle import postproc as op
    H, W, n, seed = case
    pred = _holes(H, W, n, seed)
    for min_size in (10, 5):
        want = op.fill_label_process(pred, min_size)
        got = postproc_other.process(pred.astype(np.float32) * 0.8, 'unet', min_size=min_size)
        assert isinstance(got, np.ndarray) and got.dtype == np.int32
        np.testing.assert_array_equal(got, want)
    got = postproc_other.process(torch.from_numpy(pred).cuda(), 'UNet2RevA1_vgg16', min_size=10, ws=False)
    np.testing.assert_array_equal(got.cpu().numpy(), op.fill_label_process(pred, 10))
    # batch form + empty / full masks
    batch = np.stack([pred, np.zeros_like(pred), np.ones_like(pred)])
    got = postproc_other.fill_label_process(torch.from_numpy(batch).cuda(), 10).cpu().numpy()
    for i in range(3):
        np.testing.assert_array_equal(got[i], op.fill_label_process(batch[i], 10))
