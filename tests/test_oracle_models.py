"""Pin the PyTorch-fp32 oracle networks / losses / train step to golden vectors produced by the reference."""
import numpy as np
import pytest
import torch
from cdnet_amd import synth
from oracle import models as om
from oracle import train as ot

torch.set_num_threads(8)


def _x(cfg, f16=False):
    n, c, h, w, seed = [int(v) for v in cfg]
    return torch.from_numpy(synth.det_input((n, c, h, w), seed, f16_exact=f16))


def _loss_sum(outs):
    if isinstance(outs, torch.Tensor):
        outs = (outs,)
    tot = 0
    for k, o in enumerate(outs):
        c = torch.cos(torch.arange(o.numel(), dtype=torch.float32) * 0.013 * (k + 1)).view_as(o)
        tot = tot + (o * c).mean()
    return tot


def test_unet_matches_reference(golden):
    z = golden('unet_fwd')
    m = om.det_fill(om.UNet(3))
    x = _x(z['x_cfg'])
    m.train()
    y = m(x)
    np.testing.assert_allclose(y.detach().numpy(), z['y_train'], rtol=1e-4, atol=1e-5)
    loss = _loss_sum(y)
    loss.backward()
    assert abs(loss.item() - float(z['loss'])) < 1e-6
    for n, p in m.named_parameters():
        if p.grad is None:
            assert ('gn_' + n) not in z.files
            continue
        assert abs(p.grad.double().norm().item() - float(z['gn_' + n])) <= 1e-3 * float(z['gn_' + n]) + 1e-7, n
    m2 = om.det_fill(om.UNet(3)).eval()
    with torch.no_grad():
        np.testing.assert_allclose(m2(x).numpy(), z['y_eval'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(m2(_x(z['x_ragged_cfg'])).numpy(), z['y_eval_ragged'], rtol=1e-4, atol=1e-5)


def test_dam_unet_matches_reference(golden):
    z = golden('dam_fwd')
    m = om.det_fill(om.Unet())
    assert [k for k in m.state_dict().keys()] == list(z['param_names'])
    assert sum(p.numel() for p in m.parameters()) == int(z["param_count"])
    x = _x(z['x_cfg'])
    m.train()
    outs = m(x)
    for n, o in zip(('mask', 'point', 'direction'), outs):
        np.testing.assert_allclose(o.detach().numpy(), z['train_' + n], rtol=1e-4, atol=2e-5)
    _loss_sum(outs).backward()
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        assert abs(p.grad.double().norm().item() - float(z['gn_' + n])) <= 2e-3 * float(z['gn_' + n]) + 1e-7, n
    for k in z.files:
        if k.startswith('rm_'):
            np.testing.assert_allclose(m.state_dict()[k[3:]].numpy(), z[k], rtol=1e-4, atol=1e-6)
    m2 = om.det_fill(om.Unet()).eval()
    with torch.no_grad():
        for n, o in zip(('mask', 'point', 'direction'), m2(x)):
            np.testing.assert_allclose(o.numpy(), z['eval_' + n], rtol=1e-4, atol=2e-5)
        for n, o in zip(('mask', 'point', 'direction'), m2(_x(z['x_ragged_cfg']))):
            np.testing.assert_allclose(o.numpy(), z['evalragged_' + n], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('C', [5, 17])
def test_direction_losses_other_class_counts_match_reference(golden, C):
    """4+1 / 16+1 direction classes (options.py:45; the MandD4 / MandD16 ablation models): weighted NLL + cyclic weighted dice"""
    z = golden('losses_classes')
    B, H, W, tseed, lseed = [int(v) for v in z['cfg']]
    lab, dirn, point, weight = synth.train_targets(B, H, W, tseed)
    d = torch.from_numpy(synth.remap_direction(dirn, C).astype(np.int64))
    assert int(d.max()) == C - 1
    lo_dir = torch.from_numpy((np.random.RandomState(lseed + C).randn(B, C, H, W) * 2).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy(weight).float().div(20).squeeze(1)
    dce = (torch.nn.functional.nll_loss(torch.log_softmax(lo_dir, 1), d, reduction='none') * w).mean()
    oh = torch.nn.functional.one_hot(d, C).permute(0, 3, 1, 2).float()
    wd = ot.weight_multiclass_dice(torch.softmax(lo_dir, 1), oh, w)
    assert abs(float(dce) - float(z['c%d_dce' % C])) < 2e-6 and abs(float(wd) - float(z['c%d_wdice' % C])) < 2e-6
    (dce + wd).backward()
    np.testing.assert_allclose(lo_dir.grad.numpy(), z['c%d_g_dir' % C], rtol=1e-4, atol=1e-9)


def test_losses_match_reference(golden):
    z = golden('losses')
    B, H, W, tseed, lseed = [int(v) for v in z['cfg']]
    lab, dirn, point, weight = synth.train_targets(B, H, W, tseed)
    rs = np.random.RandomState(lseed)
    lo_mask = torch.from_numpy((rs.randn(B, 3, H, W) * 2).astype(np.float32)).requires_grad_(True)
    lo_dir = torch.from_numpy((rs.randn(B, 9, H, W) * 2).astype(np.float32)).requires_grad_(True)
    lo_pt = torch.from_numpy(rs.randn(B, 1, H, W).astype(np.float32)).requires_grad_(True)
    # the golden was made with the plain one-hot of the direction target (loss.py classes called directly)
    L = ot.dam_losses(lo_mask, lo_pt, lo_dir, torch.from_numpy(lab), torch.from_numpy(dirn), torch.from_numpy(point),
                      torch.from_numpy(weight), quirk_sample0=False)
    # quirk off still masks nothing here? plain one-hot keeps background class 0 everywhere:
    oh = torch.nn.functional.one_hot(torch.from_numpy(dirn).long(), 9).permute(0, 3, 1, 2).float()
    w = torch.from_numpy(weight).float().div(20).squeeze(1)
    wd = ot.weight_multiclass_dice(torch.softmax(lo_dir, 1), oh, w)
    for k in ('ce', 'dice', 'dce', 'mse'):
        assert abs(float(L[k]) - float(z[k])) < 2e-6, k
    assert abs(float(wd) - float(z['wdice'])) < 2e-6
    total = L['ce'] + L['dice'] + L['dce'] + wd + L['mse']
    total.backward()
    np.testing.assert_allclose(lo_mask.grad.numpy(), z['g_mask'], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(lo_dir.grad.numpy(), z['g_dir'], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(lo_pt.grad.numpy(), z['g_pt'], rtol=1e-4, atol=1e-9)


def test_train_iteration_matches_reference(golden):
    """two iterations of train_util_dam.train (reference) == oracle train_iteration: losses and parameters after Adam"""
    z = golden('train_iter')
    B, _, H, W, xseed = [int(v) for v in z['x_cfg']]
    lab, dirn, point, weight = synth.train_targets(B, H, W, int(z['tgt_cfg'][3]))
    x = torch.from_numpy(synth.det_input((B, 3, H, W), xseed))
    m = om.det_fill(om.Unet())
    opt = ot.make_adam(m)
    sd = dict(m.named_parameters())
    for it in range(2):
        L = ot.train_iteration(m, opt, x, torch.from_numpy(lab), torch.from_numpy(dirn), torch.from_numpy(point),
                               torch.from_numpy(weight))
        r = z['results'][it]          # [loss, dirCE, dirDice, mse, CE, var, ...] (train_util_dam.py:297-299)
        assert abs(L['total'] - r[0]) < 5e-5 and abs(L['dce'] - r[1]) < 2e-5 and abs(L['wdice'] - r[2]) < 2e-5
        assert abs(L['mse'] - r[3]) < 2e-5 and abs(L['ce'] - r[4]) < 2e-5
        np.testing.assert_allclose(L['metrics'], r[6:11], rtol=1e-9, atol=1e-12)     # pixel accuracy, IoU, recall, precision, F1
        for k in z['pick']:
            got = sd[str(k)].detach().reshape(-1)[:96].numpy()
            np.testing.assert_allclose(got, z['p%d_%s' % (it, k)], rtol=2e-3, atol=2e-5, err_msg=str(k))
    np.testing.assert_allclose(m.state_dict()['backbone.1.running_mean'].numpy(), z['rm_backbone.1.running_mean'],
                               rtol=1e-4, atol=1e-6)


def test_unet_train_iteration_matches_reference(golden):
    """two iterations of train_util.train (reference, plain UNet, default options) == oracle unet_train_iteration"""
    z = golden('unet_train_iter')
    B, _, H, W, xseed = [int(v) for v in z['x_cfg']]
    lab, _, _, weight = synth.train_targets(B, H, W, int(z['tgt_cfg'][3]))
    x = torch.from_numpy(synth.det_input((B, 3, H, W), xseed))
    m = om.det_fill(om.UNet(3))
    opt = ot.make_adam(m)
    sd = dict(m.named_parameters())
    for it in range(2):
        L = ot.unet_train_iteration(m, opt, x, torch.from_numpy(lab), torch.from_numpy(weight))
        r = z['results'][it]          # [loss, loss_CE, ssim(-1), pixel metrics ...] (train_util.py:225)
        assert abs(L['total'] - r[0]) < 5e-5 and abs(L['ce'] - r[1]) < 2e-5
        for k in z['pick']:
            got = sd[str(k)].detach().reshape(-1)[:96].numpy()
            np.testing.assert_allclose(got, z['p%d_%s' % (it, k)], rtol=2e-3, atol=2e-5, err_msg=str(k))


def test_validate_oracle_matches_reference_golden(golden):
    """oracle/train.py:validate_iteration == the reference's train_util_dam.validate on the same sample (whole tile and 64/16
    sliding windows), tests/golden/validate.npz"""
    import torch
    from cdnet_amd import synth
    from oracle import models as om
    from oracle import train as ot
    z = golden('validate')
    B, _, H, W, seed = [int(v) for v in z['x_cfg']]
    lab, dirn, point, weight = synth.train_targets(B, H, W, int(z['tgt_cfg'][3]))
    x = torch.from_numpy(synth.det_input((B, 3, H, W), seed))
    net = om.det_fill(om.Unet())
    t = [torch.from_numpy(a) for a in (lab, dirn, point, weight)]
    got = ot.validate_iteration(net, x, *t)
    np.testing.assert_allclose(got, z['whole'], rtol=2e-5, atol=1e-7)
    size, ov = [int(v) for v in z['win_cfg']]
    got = ot.validate_iteration(net, x[:1], *[a[:1] for a in t], split=(size, ov))
    np.testing.assert_allclose(got, z['split'], rtol=2e-5, atol=1e-7)
