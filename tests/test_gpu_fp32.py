"""fp32-precision path (cdnet_amd.set_precision('fp32')): fp32 NHWC activations, every product as three split-bf16 MFMAs,
fp32 accumulation - against plain PyTorch fp32 (CPU) references of the same ops and the fp32 oracle network.

Stated tolerance: a split product carries a relative error <= 2^-16 (the dropped lo*lo term and the bf16 rounding of the
lo halves); after fp32 accumulation of K terms the kernels agree with an fp32 CPU convolution to
|got - want| <= 3e-5 * sum_k |a_k b_k| - asserted here as 4e-5 * (|x| conv |w|) + 1e-6.  End to end (30+ layers):
logits within 2e-4 * max|logit|, argmax agreement >= 99.95 %."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32_mode():
    import cdnet_amd
    old = cdnet_amd.get_precision()
    cdnet_amd.set_precision('fp32')
    yield
    cdnet_amd.set_precision(old)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def _close(got, want, bound, what):
    err = (got - want).abs()
    tol = 4e-5 * bound + 1e-6
    bad = err > tol
    assert not bool(bad.any()), '%s: %d bad, max err %g (tol there %g)' % (what, int(bad.sum()), float(err.max()), float(tol.flatten()[err.flatten().argmax()]))


CASES = [
    # N, Cin, Cout, H, W
    (2, 32, 64, 32, 32),
    (1, 64, 64, 48, 40),          # ragged vs the 16x16 tile
    (2, 16, 64, 33, 17),
    (1, 64, 32, 32, 32),
    (1, 16, 16, 32, 48),          # Cout 16 inside a 32-wide tile
    (2, 128, 256, 8, 8),          # 8x8 tiles
    (1, 128, 64, 12, 20),
]


@pytest.mark.parametrize('case', CASES)
def test_conv3x3_fp32_with_transforms(case):
    """3x3 convolution of relu(x*scale+shift) with bias; stats; eval-mode folded epilogue"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g)
    sc, sh = torch.rand((Cin,), generator=g) + 0.5, torch.randn((Cin,), generator=g) * 0.3
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn((Cout,), generator=g)
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    want = F.conv2d(a.double(), w.double(), None, padding=1)
    bound = F.conv2d(a.abs(), w.abs(), None, padding=1)
    cfg = engine.choose_cfg([Cin], Cout, H, W, N=N, f32=True)
    wp = engine.pack_weights(w.cuda(), cfg, 0, split=True)
    src = engine.Src(_nhwc(x), sc.cuda(), sh.cuda(), relu=True)
    out, stats = engine.conv_forward([src], wp, Cout, cfg, stats=True)
    assert out.dtype == torch.float32
    _close(_nchw(out), want.float(), bound, 'raw')
    s = stats.cpu().double().sum(0)
    np.testing.assert_allclose(s[0].numpy(), want.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3 * float(bound.sum((0, 2, 3)).max()) * 1e-2)
    np.testing.assert_allclose(s[1].numpy(), (want ** 2).sum((0, 2, 3)).numpy(), rtol=2e-4)
    # folded epilogue: (conv + bias) * oscale + oshift, ReLU
    osc, osh = torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.2
    out2, _ = engine.conv_forward([src], wp, Cout, cfg, bias=b.cuda(), oscale=osc.cuda(), oshift=osh.cuda(), orelu=True)
    want2 = torch.relu((want.float() + b.view(1, -1, 1, 1)) * osc.view(1, -1, 1, 1) + osh.view(1, -1, 1, 1))
    _close(_nchw(out2), want2, bound * osc.view(1, -1, 1, 1) + 1e-2, 'folded')


def test_conv_fp32_two_sources_pad_and_transposed():
    """decoder shapes: ConvTranspose2d k4 s2 p1 (sub-pixel), then concat([pad(up), skip]) -> 3x3; 1x1 with the fused residual epilogue"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(5)
    N, Cin, Cout, H, W = 2, 64, 32, 12, 10
    x = torch.randn((N, Cin, H, W), generator=g)
    wt = torch.randn((Cin, Cout, 4, 4), generator=g) * 0.05
    want = F.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1)
    bound = F.conv_transpose2d(x.abs(), wt.abs(), stride=2, padding=1)
    cfg = engine.choose_cfg([Cin], Cout, H, W, taps=4, transposed=True, N=N, f32=True)
    wp = engine.pack_weights(wt.cuda(), cfg, 2, split=True)
    up, _ = engine.conv_forward([engine.Src(_nhwc(x))], wp, Cout, cfg, taps=4, transposed=True)
    assert tuple(up.shape) == (N, 2 * H, 2 * W, Cout)
    _close(_nchw(up), want.float(), bound, 'convT4')
    # concat of a padded small tensor and a skip
    skip = torch.randn((N, 48, 2 * H + 1, 2 * W + 2), generator=g)
    w2 = torch.randn((64, Cout + 48, 3, 3), generator=g) * 0.05
    upc = want.float()
    padded = F.pad(upc, (1, 1, 0, 1))
    cat = torch.cat([padded, skip], 1)
    want2 = F.conv2d(cat.double(), w2.double(), padding=1)
    bound2 = F.conv2d(cat.abs(), w2.abs(), padding=1)
    cfg2 = engine.choose_cfg([Cout, 48], 64, 2 * H + 1, 2 * W + 2, N=N, f32=True)
    wp2 = engine.pack_weights(w2.cuda(), cfg2, 0, split=True)
    s_up = engine.Src(_nhwc(upc), off=(0, 1))
    out2, _ = engine.conv_forward([s_up, engine.Src(_nhwc(skip))], wp2, 64, cfg2, H=2 * H + 1, W=2 * W + 2)
    _close(_nchw(out2), want2.float(), bound2, 'concat+pad')
    # 1x1 with fused residual epilogue: out = relu(e*esc+esh + conv1x1(x)+b)
    e = torch.randn((N, 64, H, W), generator=g)
    esc, esh = torch.rand((64,), generator=g) + 0.5, torch.randn((64,), generator=g) * 0.1
    w1 = torch.randn((64, Cin, 1, 1), generator=g) * 0.1
    b1 = torch.randn((64,), generator=g)
    want3 = torch.relu(e * esc.view(1, -1, 1, 1) + esh.view(1, -1, 1, 1) + F.conv2d(x, w1, b1))
    cfg3 = engine.choose_cfg([Cin], 64, H, W, taps=1, N=N, f32=True)
    wp3 = engine.pack_weights(w1.cuda(), cfg3, 0, split=True)
    out3, _ = engine.conv_forward([engine.Src(_nhwc(x))], wp3, 64, cfg3, taps=1, bias=b1.cuda(),
                                  eres=engine.Src(_nhwc(e), esc.cuda(), esh.cuda(), relu=True))
    _close(_nchw(out3), want3, F.conv2d(x.abs(), w1.abs()) + e.abs() + 1, 'eres')


def test_materialize_fp32_pool():
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine, runtime
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 32, 13, 18), generator=g)
    sc, sh = torch.rand((32,), generator=g) + 0.5, torch.randn((32,), generator=g)
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    for ceil in (False, True):
        want = F.max_pool2d(a, 2, 2, ceil_mode=ceil)
        got = runtime.pooled(engine.Src(_nhwc(x), sc.cuda(), sh.cuda(), relu=True), ceil_mode=ceil)
        assert got.x.dtype == torch.float32
        torch.testing.assert_close(_nchw(got.x), want, rtol=1e-6, atol=1e-6)       # (the kernel's affine is one fma, torch's a mul + add)


def _models(seed=0):
    import torch
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    from oracle import models as om
    torch.manual_seed(seed)
    ref = om.Unet()
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.normal_(mod.bias, 0, 0.2)
            torch.nn.init.uniform_(mod.running_var, 0.5, 1.5)
            torch.nn.init.normal_(mod.running_mean, 0, 0.2)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    m.load_state_dict(ref.state_dict())
    return m.cuda(), ref


def _check_logits(got, want, name, rel=2e-4, amin=0.9995):
    got, want = got.float().cpu().numpy(), want.detach().numpy()
    scale = np.abs(want).max()
    err = np.abs(got - want)
    assert err.max() <= rel * scale, '%s: max err %g vs scale %g' % (name, err.max(), scale)
    if got.shape[1] > 1:
        agree = (got.argmax(1) == want.argmax(1)).mean()
        assert agree >= amin, '%s: argmax agreement %g' % (name, agree)


def test_dam_unet_fp32_eval_forward_vs_oracle_and_golden(golden):
    import torch
    from cdnet_amd import synth
    from oracle import models as om
    m, ref = _models()
    m.eval(); ref.eval()
    for shape, seed in (((2, 3, 64, 64), 1), ((1, 3, 72, 104), 4)):
        x = torch.from_numpy(synth.det_input(shape, seed))
        with torch.no_grad():
            want = ref(x)
            got = m(x.cuda())
        for n, g, w in zip(('mask', 'point', 'direction'), got, want):
            _check_logits(g, w, '%s %s' % (n, shape))
    # the reference's own golden outputs (closed-form weights)
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    refd = om.det_fill(om.Unet())
    md = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    md.load_state_dict(refd.state_dict())
    md = md.cuda().eval()
    z = golden('dam_fwd')
    x = torch.from_numpy(synth.det_input((2, 3, 64, 64), 1))
    with torch.no_grad():
        got = md(x.cuda())
    for n, g in zip(('mask', 'point', 'direction'), got):
        w = torch.from_numpy(z['eval_' + n])
        _check_logits(g, w, n + ' (golden)', rel=5e-4)


def test_dam_unet_fp32_train_mode_forward():
    """batch-statistics forward at fp32 precision: tight agreement with the fp32 oracle incl. the running statistics"""
    import torch
    from cdnet_amd import synth
    m, ref = _models(1)
    m.train(); ref.train()
    x = torch.from_numpy(synth.det_input((3, 3, 96, 128), 2))
    want = ref(x)
    with torch.no_grad():
        got = m(x.cuda())
    for n, g, w in zip(('mask', 'point', 'direction'), got, want):
        _check_logits(g, w, n, rel=1e-3, amin=0.999)
    sd, rsd = m.state_dict(), ref.state_dict()
    for k in ('backbone.1.running_mean', 'backbone.41.running_var', 'upsample_blocks.2.bn1.running_var', 'point_feature.bn2.running_mean'):
        np.testing.assert_allclose(sd[k].cpu().numpy(), rsd[k].numpy(), rtol=2e-4, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------
# training step at fp32 precision: element-wise against the plain fp32 oracle (oracle/train.py, pinned to the reference's
# own train() iterations) - REAL network, ReLUs on, no rounding emulation
# ---------------------------------------------------------------------------------------------------------
def _train_setup(B=2, S=64, W=None, seed=0):
    import torch
    from cdnet_amd import synth
    m, ref = _models(seed)
    W = W or S
    lab, dirn, point, weight = synth.train_targets(B, S, W, 21)
    x = torch.from_numpy(synth.det_input((B, 3, S, W), 9))
    t = [torch.from_numpy(a) for a in (lab, dirn, point, weight)]
    return m, ref, x, t


def _hip_grads(m, x, t):
    import torch
    from cdnet_amd import trainer
    tr = trainer.Trainer(m)
    dev = torch.device('cuda:0')
    o = tr.forward(x.to(dev))
    g = tr.loss_and_grads(o[0], o[1], o[2], t[0].to(dev), t[1].to(dev), t[2].to(dev), t[3][:, 0].contiguous().to(dev))
    tr.backward(*g)
    torch.cuda.synchronize()
    return tr, {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if not n.startswith(m.UNUSED_PREFIXES)}


def _oracle_grads(ref, x, t):
    from oracle import train as ot
    ref.train()
    ref.zero_grad()
    out = ref(x)
    L = ot.dam_losses(out[0], out[1], out[2], t[0], t[1], t[2], t[3])
    L['total'].backward()
    return {k: float(v) for k, v in L.items()}, {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}


def _rel_errors(g, rg):
    rel = {}
    for n, w in rg.items():
        if w.norm() < 1e-6:                  # conv biases in front of BatchNorm: exactly zero gradient
            continue
        rel[n] = float((g[n] - w).norm() / w.norm())
    return rel


@pytest.mark.parametrize('shape', [(64, 64), (72, 88)])
def test_fp32_linearised_network_gradients_elementwise(shape):
    """ReLUs off on both sides (no rounding emulation in the oracle): the whole backward orchestration - BatchNorm backward, dW,
    backward-data, concat / pad / max-pool routing, residual units, head - element-wise.  Losses 2e-5; median parameter-gradient
    error <= 2e-4 of its norm; worst <= 2e-2 (the max-pools still decide: a window whose two largest values differ by less than
    the 1e-5 product error routes its gradient elsewhere - seen on the 8x8 / 4x4 layers of the 64x64 case)."""
    from cdnet_amd import runtime
    from oracle import emulate
    from oracle import train as ot
    runtime.DEBUG_NORELU = emulate.NORELU = True
    emulate.QUANT = False
    try:
        m, ref, x, t = _train_setup(S=shape[0], W=shape[1])
        tr, g = _hip_grads(m, x, t)
        ref.train(); ref.zero_grad()
        out = emulate.dam_unet_forward(ref, x)
        L = ot.dam_losses(out[0], out[1], out[2], t[0], t[1], t[2], t[3])
        L['total'].backward()
        rg = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    finally:
        runtime.DEBUG_NORELU = emulate.NORELU = False
        emulate.QUANT = True
    np.testing.assert_allclose(tr.losses.cpu().numpy()[:5], [float(L[k]) for k in ('total', 'dce', 'wdice', 'mse', 'ce')], rtol=2e-5)
    rel = _rel_errors(g, rg)
    worst = max(rel, key=rel.get)
    print('fp32 linearised gradients %s: worst %s %.2e, median %.2e' % (shape, worst, rel[worst], float(np.median(list(rel.values())))))
    assert rel[worst] <= 2e-2, (worst, rel[worst])
    assert np.median(list(rel.values())) <= 2e-4


def test_fp32_real_network_gradients_within_the_oracles_own_conditioning():
    """ReLUs on.  A ReLU / max-pool network answers a perturbation of relative size e by flipping ~e of its decisions, which moves
    gradients by ~sqrt(e) in L2 - for ANY fp32 implementation.  The envelope is measured, not assumed: the fp32 oracle itself is
    re-run with its weights perturbed by 1e-5 (the size of a split-product error); the HIP path must stay within 3x the oracle's own
    response, per quantile.  Losses to 2e-5; head parameters (no decision below them) to 1e-4."""
    import copy
    import torch
    m, ref, x, t = _train_setup()
    tr, g = _hip_grads(m, x, t)
    L, rg = _oracle_grads(ref, x, t)
    np.testing.assert_allclose(tr.losses.cpu().numpy()[:6], [L[k] for k in ('total', 'dce', 'wdice', 'mse', 'ce', 'dice')], rtol=2e-5)
    ref2 = copy.deepcopy(ref)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in ref2.parameters():
            p.mul_(1 + 1e-5 * torch.randn(p.shape, generator=gen))
    _, rg2 = _oracle_grads(ref2, x, t)
    rel, env = _rel_errors(g, rg), _rel_errors(rg2, rg)
    ours, theirs = np.array(sorted(rel.values())), np.array(sorted(env.values()))
    print('fp32 real-network gradients: ours worst %.2e median %.2e | oracle under a 1e-5 perturbation worst %.2e median %.2e'
          % (ours[-1], np.median(ours), theirs[-1], np.median(theirs)))
    assert ours[-1] <= 3 * theirs[-1] and np.median(ours) <= 3 * np.median(theirs)
    assert ours[-1] <= 0.15
    for n in ('point_conv.weight', 'mask_conv.weight', 'direction_conv.weight', 'directionAtt.Conv1x1.weight', 'maskAtt.Conv1x1.weight'):
        assert rel[n] <= 1e-4, (n, rel[n])


def test_fp32_training_run_matches_oracle_trajectory():
    """8 Adam steps on a fixed batch: first loss to 2e-5, the next two to 2e-3, the whole loss trajectory within 3e-2 of the fp32
    oracle's (train_util_dam.train semantics incl. the sample-0 quirk and Adam with weight decay); the reference's never-used parameters stay untouched"""
    import torch
    from cdnet_amd import trainer
    from oracle import train as ot
    m, ref, x, t = _train_setup()
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    tr = trainer.Trainer(m)
    dev = torch.device('cuda:0')
    batch = (x.to(dev), t[0].to(dev), t[1].to(dev), t[2].to(dev), t[3][:, 0].contiguous().to(dev))
    opt = ot.make_adam(ref)
    ours, theirs = [], []
    for _ in range(8):
        ours.append(float(tr.train_step(*batch)[0]))
        theirs.append(ot.train_iteration(ref, opt, x, *t)['total'])
    print('fp32 trajectory', ours, theirs)
    assert abs(ours[0] - theirs[0]) <= 2e-5 * theirs[0]
    # Adam's first steps are ~lr * sign(g): elements whose gradient is below the noise level take opposite steps, so the
    # trajectories separate at the 1e-3 level within a few steps for any two fp32 implementations and at the 1e-2 level by step 6
    # (measured: 1e-7, 5e-5, 4e-4, 3e-3, 8e-3, 1.1e-2, 3e-3, 6e-3)
    np.testing.assert_allclose(ours[:3], theirs[:3], rtol=2e-3)
    np.testing.assert_allclose(ours, theirs, rtol=3e-2)
    assert ours[-1] < ours[0] * 0.9
    sd = m.state_dict()
    for n, p in ref.named_parameters():
        if n.startswith(m.UNUSED_PREFIXES):
            assert torch.equal(sd[n].cpu(), p0[n])


def test_fp32_plain_unet_train_step(golden):
    """BASELINE config 1 (plain UNet, 3 classes) at fp32 precision vs the oracle iteration pinned to the reference's train_util.train"""
    import torch
    from cdnet_amd import synth, trainer
    from cdnet_amd.models.unet import UNet
    from oracle import models as om
    from oracle import train as ot
    torch.manual_seed(0)
    ref = om.UNet(3)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(mod.weight)
    m = UNet(3)
    m.load_state_dict(ref.state_dict())
    m = m.cuda()
    lab, _, _, weight = synth.train_targets(2, 64, 64, 5)
    x = torch.from_numpy(synth.det_input((2, 3, 64, 64), 5))
    tl, tw = torch.from_numpy(lab), torch.from_numpy(weight)
    tr = trainer.UNetTrainer(m)
    opt = ot.make_adam(ref)
    ours, theirs = [], []
    for _ in range(4):
        ours.append(float(tr.train_step(x.cuda(), tl.cuda(), tw[:, 0].contiguous().cuda())[0]))
        theirs.append(float(ot.unet_train_iteration(ref, opt, x, tl, tw)['total']))
    np.testing.assert_allclose(ours, theirs, rtol=1e-3)
