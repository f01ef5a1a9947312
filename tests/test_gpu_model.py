"""GPU parity of the DAM-Unet forward (HIP kernels, bf16 activations / fp32 accumulation) against the fp32
oracle network (oracle/models.py, itself pinned to the reference) and the reference's golden outputs.

Stated tolerance for the bf16 path: logits within 3e-2 * max|logit| (absolute) and mean abs
error < 6e-3 * max|logit|; argmax agreement of mask / direction classes >= 99 %."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _models():
    import torch
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    from oracle import models as om
    ref = om.det_fill(om.Unet())
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    m.load_state_dict(ref.state_dict())
    return m.cuda(), ref


def _check(got, want, name, amin=0.99):
    got, want = got.float().cpu().numpy(), want.numpy() if hasattr(want, 'numpy') else want
    scale = np.abs(want).max()
    err = np.abs(got - want)
    assert err.max() <= 3e-2 * scale, '%s: max err %g vs scale %g' % (name, err.max(), scale)
    assert err.mean() <= 6e-3 * scale, '%s: mean err %g vs scale %g' % (name, err.mean(), scale)
    if got.shape[1] > 1:
        agree = (got.argmax(1) == want.argmax(1)).mean()
        assert agree >= amin, '%s: argmax agreement %g' % (name, agree)


def test_dam_unet_eval_forward_vs_oracle_and_golden(golden):
    import torch
    from cdnet_amd import synth
    m, ref = _models()
    m.eval(); ref.eval()
    z = golden('dam_fwd')
    x = torch.from_numpy(synth.det_input((2, 3, 64, 64), 1))
    with torch.no_grad():
        want = ref(x)
        got = m(x.cuda())
    for n, g, w in zip(('mask', 'point', 'direction'), got, want):
        assert tuple(g.shape) == tuple(w.shape) and g.dtype == torch.float32
        _check(g, w, n)
        _check(g, z['eval_' + n], n + ' (golden)')
    # 256x256 tile against the reference's golden output (stored as f16)
    x = torch.from_numpy(synth.det_input((1, 3, 256, 256), 3, f16_exact=True))
    with torch.no_grad():
        got = m(x.cuda())
    for n, g in zip(('mask', 'point', 'direction'), got):
        _check(g, z['eval256_' + n].astype(np.float32), n + ' 256 (golden)')
    # ragged size: F.pad offsets, partial tiles
    x = torch.from_numpy(synth.det_input((1, 3, 72, 104), 4))
    with torch.no_grad():
        got = m(x.cuda())
    for n, g in zip(('mask', 'point', 'direction'), got):
        _check(g, z['evalragged_' + n], n + ' ragged (golden)')


def _random_models(seed=0):
    """Well-conditioned random initialisation (torch default init + non-trivial BN affine).  The closed-form
    det_fill weights of the golden fixtures make training-mode BatchNorm ill-conditioned (channels with
    1/std up to ~50 amplify any 16-bit rounding), so the batch-statistics path is checked against the fp32
    oracle - which itself is pinned to the reference's golden vectors in fp32 (tests/test_oracle_models.py)."""
    import torch
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    from oracle import models as om
    torch.manual_seed(seed)
    ref = om.Unet()
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.normal_(mod.bias, 0, 0.2)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    m.load_state_dict(ref.state_dict())
    return m.cuda(), ref


def test_dam_unet_train_mode_forward_batch_stats():
    """training-mode forward: batch statistics from the conv epilogue, lazily applied BN+ReLU, running-stat update.
    Tolerance: max |err| <= 8e-2 * max|logit|, mean |err| <= 1e-2 * max|logit|, argmax agreement >= 97 %."""
    import torch
    from cdnet_amd import synth
    m, ref = _random_models()
    m.train(); ref.train()
    for shape, seed in (((2, 3, 64, 64), 1), ((3, 3, 96, 128), 2)):
        x = torch.from_numpy(synth.det_input(shape, seed))
        rm0 = {k: v.clone() for k, v in ref.state_dict().items() if 'running' in k}
        want = ref(x)
        with torch.no_grad():
            got = m(x.cuda())
        for n, g, w in zip(('mask', 'point', 'direction'), got, want):
            g, w = g.float().cpu().numpy(), w.detach().numpy()
            scale = np.abs(w).max()
            err = np.abs(g - w)
            assert err.max() <= 8e-2 * scale and err.mean() <= 1e-2 * scale, (n, err.max(), err.mean(), scale)
            if g.shape[1] > 1:
                assert (g.argmax(1) == w.argmax(1)).mean() >= 0.97, n
        sd, rsd = m.state_dict(), ref.state_dict()
        for k in ('backbone.1.running_mean', 'backbone.1.running_var', 'backbone.41.running_mean',
                  'backbone.41.running_var', 'upsample_blocks.2.bn1.running_var', 'point_feature.bn2.running_mean',
                  'mask_feature.bn1.running_var'):
            a, b = sd[k].cpu().numpy(), rsd[k].numpy()
            step = np.abs(b - rm0[k].numpy()).max() + 1e-6          # size of this update
            assert np.abs(a - b).max() <= 0.05 * step + 2e-3, (k, np.abs(a - b).max(), step)


def test_plain_unet_forward_vs_oracle_and_golden(golden):
    """models/unet.py (config 0): eval forward vs the reference's golden, ceil-mode pools / F.pad on a ragged size,
    training-mode forward (batch statistics) vs the fp32 oracle with the reference's Kaiming init."""
    import torch
    from cdnet_amd import synth
    from cdnet_amd.models.unet import UNet
    from oracle import models as om
    z = golden('unet_fwd')
    ref = om.det_fill(om.UNet(3)).eval()
    m = UNet(3)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(ref.state_dict())
    m = m.cuda().eval()
    x = torch.from_numpy(synth.det_input(tuple(int(v) for v in z['x_cfg'][:4]), int(z['x_cfg'][4])))
    with torch.no_grad():
        _check(m(x.cuda()), z['y_eval'], 'unet eval (golden)')
        xr = torch.from_numpy(synth.det_input(tuple(int(v) for v in z['x_ragged_cfg'][:4]), int(z['x_ragged_cfg'][4])))
        _check(m(xr.cuda()), z['y_eval_ragged'], 'unet eval ragged (golden)')
    torch.manual_seed(0)
    ref = om.UNet(3)
    for mod in ref.modules():
        if isinstance(mod, (torch.nn.Conv2d,)):
            torch.nn.init.kaiming_normal_(mod.weight)
    m = UNet(3)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().train(); ref.train()
    x = torch.from_numpy(synth.det_input((2, 3, 64, 64), 5))
    want = ref(x).detach()
    with torch.no_grad():
        got = m(x.cuda()).float().cpu()
    scale = want.abs().max()
    err = (got - want).abs()
    assert err.max() <= 8e-2 * scale and err.mean() <= 1e-2 * scale, (float(err.max()), float(err.mean()), float(scale))


@pytest.mark.parametrize('name', ['model_unet_MandD', 'model_unet_MandD4', 'model_unet_MandD16', 'model_unet_MandDandP'])
def test_ablation_heads_vs_reference_golden(golden, name):
    """the four ablation models of utils.py:857-874 (heads without attention gates): same state_dict keys as the reference, eval
    forward vs the reference's outputs (tests/golden/ablation.npz); bf16 path tolerance as for the rev1 model"""
    import torch
    from cdnet_amd import synth, utils
    z = golden('ablation')

    class _Opt:
        model = {'modelName': name, 'out_c': 3, 'in_c': 3}
    m = utils.chooseModel(_Opt())
    assert list(m.state_dict().keys()) == [str(k) for k in z['keys_' + name]]
    bn = {n for n, mod in m.named_modules() if isinstance(mod, torch.nn.BatchNorm2d)}
    with torch.no_grad():
        synth.det_fill_state_dict(m.state_dict(), bn)
    m = m.cuda().eval()
    shape = tuple(int(v) for v in z['x_cfg'][:4])
    x = torch.from_numpy(synth.det_input(shape, int(z['x_cfg'][4])))
    with torch.no_grad():
        got = m(x.cuda())
    assert len(got) == int(z['n_' + name])
    for k, g in enumerate(got):
        want = z['%s_%d' % (name, k)].astype(np.float32)
        assert tuple(g.shape) == want.shape
        _check(g, want, '%s output %d' % (name, k), amin=0.97)       # 17 near-tied classes on closed-form weights (bf16 path)


@pytest.mark.gpu
def test_point_logits_from_the_point_feature_launch():
    """eval mode, 16-bit path: the point feature's only reader is point_conv, so its logits leave with the residual unit's launch
    (runtime.RU_EVAL_POINT_DOT, cdnet_conv_args.dot_out) and the feature is never stored.  Same rounded feature values, another order of
    the 64-term fp32 sum: the three outputs agree with the stored-feature form to 1e-5 of the logit scale."""
    import torch
    from cdnet_amd import runtime
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    torch.manual_seed(3)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).cuda().eval()
    x = torch.rand((2, 3, 64, 96), device='cuda')
    with torch.no_grad():
        assert runtime.RU_EVAL_POINT_DOT
        a = m(x)
        assert isinstance(m._last_feats[2], runtime.PointLogit)
        runtime.RU_EVAL_POINT_DOT = False
        try:
            b = m(x)
            assert not isinstance(m._last_feats[2], runtime.PointLogit)
        finally:
            runtime.RU_EVAL_POINT_DOT = True
    for u, v, name in zip(a, b, ('mask', 'point', 'direction')):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-6, name
