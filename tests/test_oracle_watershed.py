"""CPU: the watershed-branch oracle (oracle/postproc_oracle.c: orc_ws_dist, orc_watershed) against the scipy calls the
reference itself makes (postproc_other.py:16-48).  skimage (remove_small_objects, watershed) is absent from this image:
the flood is checked through its defining properties instead (parity unpinned for the equal-priority tie-break)."""
import numpy as np
import pytest


def blobs(H, W, n, seed, rmin=4, rmax=11):
    """overlapping random discs: merged nuclei are what the watershed has to split"""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[:H, :W]
    m = np.zeros((H, W), bool)
    for _ in range(n):
        cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(rmin, rmax)
        m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    return m.astype(np.uint8)


@pytest.mark.parametrize('case', [(64, 64, 14, 1), (96, 80, 30, 2), (50, 120, 25, 3), (128, 128, 60, 4)])
def test_dist_and_marker_match_scipy(case):
    from oracle import postproc as op
    H, W, n, seed = case
    pred = blobs(H, W, n, seed)
    a = op.watershed_process(pred, 10, use_scipy=True)
    b = op.watershed_process(pred, 10, use_scipy=False)
    np.testing.assert_array_equal(a['dist'], b['dist'])
    np.testing.assert_array_equal(a['marker'], b['marker'])
    np.testing.assert_array_equal(a['labels'], b['labels'])


def test_watershed_properties():
    from scipy import ndimage as ndi
    from oracle import postproc as op
    pred = blobs(128, 128, 60, 7)
    r = op.watershed_process(pred, 10)
    lab, marker = r['labels'], r['marker']
    assert lab[pred == 0].max() == 0                               # stays inside the mask
    keep = marker > 0
    kept_ids = set(np.unique(lab)) - {0}
    for k in kept_ids:                                              # every surviving basin contains its own marker
        assert (marker[lab == k] == k).any()
        assert ndi.label(lab == k)[1] == 1                         # and is 4-connected
        assert (lab == k).sum() >= 10
    # a mask component with at least one surviving marker is flooded completely (before small removal every pixel of it
    # is reachable) - check on components whose basins all survive
    comp = ndi.label(pred)[0]
    for c in range(1, comp.max() + 1):
        ids = set(np.unique(marker[(comp == c) & keep])) - {0}
        if ids and ids <= kept_ids and all((lab == k).sum() >= 10 for k in ids):
            cover = (lab > 0)[comp == c].mean()
            assert cover > 0.9


def test_more_than_one_basin_per_component():
    from scipy import ndimage as ndi
    from oracle import postproc as op
    pred = blobs(128, 128, 60, 7)
    r = op.watershed_process(pred, 10)
    comp = ndi.label(pred)[0]
    split = sum(1 for c in range(1, comp.max() + 1) if len(set(np.unique(r['labels'][comp == c])) - {0}) > 1)
    assert split >= 3                                              # the case actually exercises the splitting


def test_fill_label_process_oracle_semantics():
    """ws = False branch: ids are scipy's raster-order 4-connected labels; small labels vanish, the others keep their ids"""
    import numpy as np
    from oracle import postproc as op
    m = np.zeros((12, 14), np.uint8)
    m[1:4, 1:4] = 1; m[2, 2] = 0                 # ring with a hole: filled -> 9 px
    m[6, 6] = 1                                   # speck (1 px)
    m[8:11, 8:12] = 1                             # 12 px
    m[7, 7] = 1                                   # diagonal neighbour of the speck: 4-connectivity keeps them apart
    lab = op.fill_label_process(m, 5)
    assert lab[2, 2] == 1 and (lab[1:4, 1:4] == 1).all()
    assert lab[6, 6] == 0 and lab[7, 7] == 0
    assert set(np.unique(lab)) == {0, 1, 4} and (lab[8:11, 8:12] == 4).all()
