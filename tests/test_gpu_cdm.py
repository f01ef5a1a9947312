"""GPU parity of the centripetal-direction-map / point-map generation against the CPU oracle and the reference's golden."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_label_encoding_golden_and_oracle(golden):
    import torch
    from cdnet_amd.my_transforms_direction import label_encoding_batch
    from oracle import cdm
    z = golden('cdm')
    for name in z['names']:
        x = z['in_' + name]
        l3, pt, dr, inst, counts = label_encoding_batch(torch.from_numpy(x).cuda()[None], want_inst=True)
        o3, opt, odr, oinst, ocent = cdm.label_encoding(x, want_aux=True)
        assert np.array_equal(l3[0].cpu().numpy(), o3) and np.array_equal(l3[0].cpu().numpy(), z['label_' + name])
        assert np.array_equal(inst[0].cpu().numpy(), oinst) and int(counts[0]) == len(ocent)
        # bit-exact against the oracle (same float64 summation order); the golden differs only on 45-degree bin edges
        assert np.array_equal(dr[0].cpu().numpy(), odr), name
        assert (dr[0].cpu().numpy() != z['direction_' + name]).mean() <= 1e-3
        assert np.array_equal(pt[0].cpu().numpy().view(np.uint16), opt.view(np.uint16)) or \
            np.abs(pt[0].cpu().numpy().astype(np.float32) - opt.astype(np.float32)).max() <= 1e-3
        np.testing.assert_allclose(pt[0].cpu().numpy().astype(np.float32), z['point_' + name].astype(np.float32), atol=1e-3)


def test_label_encoding_batched_256_tiles():
    import torch
    from cdnet_amd import synth
    from cdnet_amd.my_transforms_direction import label_encoding_batch
    from oracle import cdm
    rs = np.random.RandomState(1)
    xs = []
    for _ in range(3):
        inst = synth.ellipse_instances(256, 256, 60, rs, 5, 12, 10)
        xs.append(((inst > 0) * 255).astype(np.uint8))
    xs.append(np.zeros((256, 256), np.uint8))                 # empty tile
    full = np.full((256, 256), 255, np.uint8)                 # one nucleus covering the tile
    xs.append(full)
    x = np.stack(xs)
    l3, pt, dr = label_encoding_batch(torch.from_numpy(x).cuda())
    for i in range(len(xs)):
        o3, opt, odr = cdm.label_encoding(x[i])
        assert np.array_equal(l3[i].cpu().numpy(), o3), i
        assert np.array_equal(dr[i].cpu().numpy(), odr), i
        assert np.abs(pt[i].cpu().numpy().astype(np.float32) - opt.astype(np.float32)).max() <= 1e-3, i


def test_instance_label_branch_bit_exact_vs_oracle(golden):
    """cdnet_label_encoding_instances (my_transforms_direction.py:752-760 on the device: boundary from the ids, watershed instances,
    common stage) against the C/numpy oracle on the golden inputs and on a larger batch: label exact, direction exact, point 1e-3;
    and against the reference's own outputs as far as the oracle is"""
    import torch
    from cdnet_amd import synth
    from cdnet_amd.my_transforms_direction import label_encoding_instances_batch, LabelEncoding
    from oracle import cdm
    z = golden('cdm_inst')
    cases = [z['in_' + str(n)].astype(np.int32) for n in z['names']]
    rs = np.random.RandomState(9)
    big = synth.ellipse_instances(256, 256, 60, rs, 5, 12, 10)
    for lab in cases + [big.astype(np.int32)]:
        l3, point, direction, inst, counts = label_encoding_instances_batch(torch.from_numpy(lab).cuda()[None], want_inst=True)
        w3, wp, wd, winst = cdm.label_encoding_instances(lab)
        assert np.array_equal(l3[0].cpu().numpy(), w3)
        assert np.array_equal(inst[0].cpu().numpy(), winst)
        assert np.array_equal(direction[0].cpu().numpy(), wd)
        np.testing.assert_allclose(point[0].float().cpu().numpy(), wp.astype(np.float32), rtol=0, atol=1e-3)
    # the reference-named transform takes the branch by itself (label_level_len > 2)
    from PIL import Image
    lab = cases[0]
    res = LabelEncoding(3, 2, 1)((Image.fromarray(np.zeros(lab.shape + (3,), np.uint8)), Image.fromarray(np.full(lab.shape, 20, np.uint8)),
                                  Image.fromarray(lab.astype(np.uint8))))
    assert np.array_equal(np.array(res[2]), z['label_' + str(z['names'][0])])
