"""GPU: cdnet_amd.stats_utils (AJI / PQ / Dice / remap_label over csrc/metrics.hip) against values produced by the
reference's stats_utils.py itself (tests/golden/aji.npz, made by tests/golden/make_golden.py:gen_aji)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_matches_reference_values(golden):
    from cdnet_amd import stats_utils
    z, pp = golden('aji'), golden('postproc')
    for name in z['names']:
        pred, true = pp['final_' + str(name)].astype(np.int32), z['true_' + str(name)]
        p, t = stats_utils.remap_label(pred.copy()), stats_utils.remap_label(true.copy())
        aji = stats_utils.get_fast_aji(t, p)
        assert abs(aji[0] - float(z['aji_' + str(name)])) < 1e-12
        assert abs(stats_utils.get_dice_1(t, p) - float(z['dice_' + str(name)])) < 1e-12
        np.testing.assert_allclose(stats_utils.get_fast_pq(t, p)[0], z['pq_' + str(name)], rtol=0, atol=1e-12)


def test_remap_label_and_properties():
    import torch
    from cdnet_amd import stats_utils
    rs = np.random.RandomState(0)
    lab = np.zeros((64, 80), np.int32)
    ids = [3, 7, 8, 20, 41]
    for k, i in enumerate(ids):
        lab[5 + 10 * k: 12 + 10 * k, 4 + 12 * k: 14 + 12 * k] = i
    r = stats_utils.remap_label(lab)
    assert sorted(np.unique(r)) == [0, 1, 2, 3, 4, 5]
    for k, i in enumerate(ids):
        assert (r[lab == i] == k + 1).all()
    # identical images: AJI = PQ = Dice = 1
    assert abs(stats_utils.get_fast_aji(r, r)[0] - 1.0) < 1e-12
    assert abs(stats_utils.get_dice_1(r, r) - 1.0) < 1e-12
    np.testing.assert_allclose(stats_utils.get_fast_pq(r, r)[0], [1.0, 1.0 / (1 + 1e-6 / 5), 1.0 / (1 + 1e-6 / 5)], rtol=1e-9)
    # torch CUDA inputs are accepted as well
    t = torch.from_numpy(r).cuda()
    assert abs(stats_utils.get_fast_aji(t, t)[0] - 1.0) < 1e-12


def test_nuclei_accuracy_object_level_matches_reference(golden):
    """utils.nuclei_accuracy_object_level (greedy per-object matching, Hausdorff, AJI) vs the values of the reference's own function
    (tests/golden/aji.npz `obj_*`, made with the measure.label stand-in of tests/golden/_ref_shims.py)"""
    from cdnet_amd import utils
    z, pp = golden('aji'), golden('postproc')
    for name in z['names']:
        pred, true = pp['final_' + str(name)].astype(np.int32), z['true_' + str(name)]
        got = utils.nuclei_accuracy_object_level(pred, true)
        np.testing.assert_allclose(np.array(got, dtype=np.float64), z['obj_' + str(name)], rtol=1e-12, atol=1e-12)
