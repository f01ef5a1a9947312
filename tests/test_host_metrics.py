"""Host-side pieces of the metric functions that need no GPU."""
import numpy as np


def _label_equal_regions(x):
    """plain statement of skimage.measure.label on an integer image: per value, its 8-connected components; ids in raster order of
    each region's first pixel"""
    from scipy import ndimage as ndi
    regions = []
    for v in np.unique(x):
        if v == 0:
            continue
        lab, n = ndi.label(x == v, structure=np.ones((3, 3), dtype=int))
        for k in range(1, n + 1):
            m = lab == k
            regions.append((int(np.flatnonzero(m.ravel())[0]), m))
    out = np.zeros(x.shape, np.int32)
    for i, (_, m) in enumerate(sorted(regions, key=lambda r: r[0]), start=1):
        out[m] = i
    return out


def test_measure_label_semantics():
    """utils.measure_label == regions of equal value, 8-connected, raster-order ids (what the reference's metric functions get from
    skimage.measure.label, utils.py:248-249): binary 0/255 images, instance maps, one value in several separate regions, touching
    instances of different value, no background at all"""
    from cdnet_amd import utils
    rs = np.random.RandomState(0)
    cases = []
    b = (rs.rand(40, 56) > 0.6).astype(np.uint8) * 255
    cases.append(b)
    inst = rs.randint(0, 4, size=(32, 48)) * (rs.rand(32, 48) > 0.3)            # few values, many separate regions each, touching
    cases.append(inst.astype(np.int32))
    big = np.zeros((24, 24), np.int64)
    big[2:6, 2:6], big[2:6, 6:10], big[2:6, 10:14] = 7, 3, 7                      # 7 | 3 | 7: one foreground component, value 7 twice
    big[10:12, 0:24] = 5
    cases.append(big)
    cases.append(np.full((8, 8), 9, np.int32))
    cases.append(np.zeros((8, 8), np.int32))
    for x in cases:
        got = utils.measure_label(x)
        want = _label_equal_regions(np.asarray(x))
        assert got.dtype == np.int32 and np.array_equal(got, want)
