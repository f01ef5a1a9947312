"""Pin the fp32 oracle HRNet18_rev1 (oracle/hrnet.py) to golden vectors produced by the reference."""
import numpy as np
import torch
from cdnet_amd import synth
from oracle import hrnet as oh
from oracle import models as om
from oracle import train as ot

torch.set_num_threads(8)


def make_oracle_hrnet(gain):
    m = om.det_fill(oh.HighResolutionNet())
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.mul_(gain)
    return m


def test_hrnet_eval_matches_reference(golden):
    z = golden('hrnet_fwd')
    m = make_oracle_hrnet(float(z['gain'])).eval()
    assert len(m.state_dict()) == int(z['n_keys']) and sum(p.numel() for p in m.parameters()) == int(z['param_count'])
    with torch.no_grad():
        for tag in 'ab':
            n, c, h, w, seed = [int(v) for v in z['x_cfg_' + tag]]
            x = torch.from_numpy(synth.det_input((n, c, h, w), seed, bf16_exact=True))
            for name, o in zip(('mask', 'point', 'direction'), m(x)):
                ref = z['%s_%s' % (name, tag)]
                np.testing.assert_allclose(o.numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())))


def test_hrnet_train_iteration_matches_reference(golden):
    """two iterations of train_util_dam.train on the reference's HRNet18_rev1 == oracle train_iteration on oracle/hrnet.py"""
    z = golden('hrnet_train')
    B, _, H, W, xseed = [int(v) for v in z['x_cfg']]
    lab, dirn, point, weight = synth.train_targets(B, H, W, int(z['tgt_cfg'][3]))
    x = torch.from_numpy(synth.det_input((B, 3, H, W), xseed, bf16_exact=True))
    m = make_oracle_hrnet(float(z['gain']))
    sd = dict(m.named_parameters())
    for k in z['pick']:
        np.testing.assert_array_equal(sd[str(k)].detach().reshape(-1)[:96].numpy(), z['p_init_' + str(k)])
    opt = ot.make_adam(m)
    for it in range(2):
        L = ot.train_iteration(m, opt, x, torch.from_numpy(lab), torch.from_numpy(dirn), torch.from_numpy(point), torch.from_numpy(weight))
        r = z['results'][it]
        assert abs(L['total'] - r[0]) < 2e-4 and abs(L['dce'] - r[1]) < 1e-4 and abs(L['wdice'] - r[2]) < 2e-5
        assert abs(L['mse'] - r[3]) < 1e-4 and abs(L['ce'] - r[4]) < 1e-4
        np.testing.assert_allclose(L['metrics'], r[6:11], rtol=1e-9, atol=1e-12)
        for k in z['pick']:
            got = sd[str(k)].detach().reshape(-1)[:96].numpy()
            np.testing.assert_allclose(got, z['p%d_%s' % (it, k)], rtol=2e-3, atol=2e-5, err_msg=str(k))
    np.testing.assert_allclose(m.state_dict()['bn1.running_mean'].numpy(), z['rm_bn1.running_mean'], rtol=1e-4, atol=1e-6)
