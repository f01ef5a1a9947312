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
