"""Training step of HRNet18_rev1 on the GPU (forward with batch-statistics BatchNorm, five-term loss, backward through the
residual / fuse sums, the bilinear up-sampling and the stride-2 convolutions, Adam) against the fp32 oracle
(oracle/hrnet.py, pinned to the reference by tests/test_oracle_hrnet.py) and the reference's own two train() iterations
(tests/golden/hrnet_train.npz).

Same three-way comparison as tests/test_gpu_train_step.py: loss values; gradients of the LINEARISED network (tight: proves
the backward orchestration); gradient direction of the real network (bf16 forward flips a few ReLU decisions per layer)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Opt:
    model = {'out_c': 3}


def _setup(B=2, S=64, seed=0, gain=None):
    import torch
    from cdnet_amd import synth
    from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
    from oracle import hrnet as oh
    from oracle import models as om
    torch.manual_seed(seed)
    ref = oh.HighResolutionNet()
    if gain is not None:                                   # the golden's closed-form weights
        om.det_fill(ref)
        with torch.no_grad():
            for mod in ref.modules():
                if isinstance(mod, torch.nn.Conv2d):
                    mod.weight.mul_(gain)
    else:
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
                torch.nn.init.normal_(mod.bias, 0, 0.2)
    m = HighResolutionNet(_Opt())
    m.load_state_dict(ref.state_dict())
    lab, dirn, point, weight = synth.train_targets(B, S, S, 41)
    x = torch.from_numpy(synth.det_input((B, 3, S, S), 15, bf16_exact=True))
    t = [torch.from_numpy(a) for a in (lab, dirn, point, weight)]
    return m.cuda(), ref, x, t


def _dev_batch(x, t):
    import torch
    dev = torch.device('cuda:0')
    return (x.to(dev), t[0].to(dev), t[1].to(dev), t[2].to(dev), t[3][:, 0].contiguous().to(dev))


def _hip_grads(m, x, t):
    import torch
    from cdnet_amd import trainer
    tr = trainer.Trainer(m)
    b = _dev_batch(x, t)
    o = tr.forward(b[0])
    g = tr.loss_and_grads(o[0], o[1], o[2], *b[1:])
    tr.backward(*g)
    torch.cuda.synchronize()
    m.sync_real_parameters()
    named = m.trainer_named_parameters()
    grads = {}
    for n, p in m.named_parameters():
        if n.startswith(m.UNUSED_PREFIXES):
            continue
        pp = named[n]
        if pp is p or p.grad is not None:
            grads[n] = (p.grad if p.grad is not None else pp.grad).detach().float().cpu().clone()
        else:                                              # the two concatenation-reading weights: gather from the padded layout
            gp = pp.grad.detach().float().cpu()
            segs = [s for r, q, s in m._slots if q is pp][0]
            grads[n] = torch.cat([gp[:, p0:p0 + k] for _, k, p0 in segs], 1)
    return tr, grads


def _oracle_grads(ref, x, t):
    from oracle import train as ot
    ref.train()
    ref.zero_grad()
    out = ref(x)
    L = ot.dam_losses(out[0], out[1], out[2], t[0], t[1], t[2], t[3])
    L['total'].backward()
    return {k: float(v) for k, v in L.items()}, {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}


def test_parameters_are_views_of_padded_storage():
    import torch
    m, ref, x, t = _setup()
    m.eval()
    with torch.no_grad():
        m(x.cuda())
    sd, rsd = m.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(rsd.keys())
    for k in sd:
        assert sd[k].shape == rsd[k].shape, k
        assert torch.equal(sd[k].cpu(), rsd[k]), k
    named = m.trainer_named_parameters()
    w, wp = m.stage2[0].branches[0][0].conv1.weight, named['stage2.0.branches.0.0.conv1.weight']
    assert tuple(wp.shape) == (32, 32, 3, 3) and w.untyped_storage().data_ptr() == wp.untyped_storage().data_ptr()
    assert float(wp.detach()[18:].abs().max()) == 0 and float(wp.detach()[:, 18:].abs().max()) == 0


def test_loss_values_match_fp32_oracle():
    m, ref, x, t = _setup()
    tr, _ = _hip_grads(m, x, t)
    L, _ = _oracle_grads(ref, x, t)
    got = tr.losses.cpu().numpy()[:6]
    want = [L[k] for k in ('total', 'dce', 'wdice', 'mse', 'ce', 'dice')]
    np.testing.assert_allclose(got, want, rtol=1.5e-2)          # bf16 storage through ~60 layers (eval parity: tests/test_gpu_hrnet.py)


def test_linearised_network_gradients():
    import torch
    from cdnet_amd import runtime
    F = torch.nn.functional
    relu0 = F.relu
    runtime.DEBUG_NORELU = True
    F.relu = lambda v, inplace=False: v
    try:
        m, ref, x, t = _setup()
        _, g = _hip_grads(m, x, t)
        _, rg = _oracle_grads(ref, x, t)
    finally:
        runtime.DEBUG_NORELU = False
        F.relu = relu0
    rel = {}
    for n, want in rg.items():
        if want.norm() < 1e-6:
            continue
        rel[n] = float((g[n] - want).norm() / want.norm())
    worst = max(rel, key=rel.get)
    print('worst', worst, rel[worst], 'median', np.median(list(rel.values())))
    assert rel[worst] <= 0.15, (worst, rel[worst], sorted(rel.items(), key=lambda kv: -kv[1])[:8])
    assert np.median(list(rel.values())) <= 4e-2


def test_real_network_gradient_direction():
    m, ref, x, t = _setup()
    _, g = _hip_grads(m, x, t)
    _, rg = _oracle_grads(ref, x, t)
    cos = {}
    for n, want in rg.items():
        if want.norm() < 1e-6:
            continue
        cos[n] = float((g[n] * want).sum() / (g[n].norm() * want.norm()))
    print('min', min(cos, key=cos.get), min(cos.values()), 'median', np.median(list(cos.values())))
    for n in ('point_conv.weight', 'mask_conv.weight', 'direction_conv.weight'):
        assert cos[n] >= 0.995, (n, cos[n])
    assert np.median(list(cos.values())) >= 0.85         # 64x64 tiles: the 1/32-resolution branch normalises over 8 values per channel
    assert min(cos.values()) >= 0.6, min(cos, key=cos.get)
    num = sum(float(w.abs()[(g[n] * w) > 0].sum()) for n, w in rg.items())
    den = sum(float(w.abs().sum()) for w in rg.values())
    print('magnitude-weighted sign agreement', num / den)
    assert num / den >= 0.9


def test_two_steps_follow_the_reference_train_loop(golden):
    """the reference's train() on HRNet18_rev1, two iterations (tests/golden/make_golden.py gen_hrnet_train)"""
    import torch
    from cdnet_amd import trainer
    z = golden('hrnet_train')
    B, _, H, W, _ = [int(v) for v in z['x_cfg']]
    m, ref, x, t = _setup(B, H, gain=float(z['gain']))
    tr = trainer.Trainer(m, lr=float(z['lr']))
    batch = _dev_batch(x, t)
    _, rg = _oracle_grads(ref, x, t)                       # fp32 gradient at the initial parameters (oracle pinned on this very case)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for it in range(2):
        got = tr.train_step(*batch).cpu().numpy()
        r = z['results'][it]
        # iteration 1 runs on parameters moved by Adam's first step = lr * sign(gradient): every near-zero gradient whose sign the
        # bf16 forward flips moves its parameter the other way, so the second loss only tracks the reference's to a few %
        np.testing.assert_allclose(got[:5], r[:5], rtol=2e-2 if it == 0 else 1e-1, err_msg='iteration %d' % it)
        if it == 0:
            sd1 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    sd = m.state_dict()
    lr = float(z['lr'])
    # Adam step 1 moves every parameter by lr against the sign of its gradient: agreement with the fp32 gradient, weighted by
    # its magnitude (the signs of near-zero gradients are noise)
    num = den = 0.0
    for k, g in rg.items():
        w = g.abs().double()
        step = (sd1[k] - sd0[k]).double()
        assert float(step.abs().max()) <= 1.05 * lr + 2e-4 * float(sd0[k].abs().max()), k
        num += float(w[torch.sign(step) == -torch.sign(g)].sum())
        den += float(w.sum())
    print('weighted sign agreement of the first Adam step', num / den)
    # closed-form weights: gradients of ~1e5 through 30 un-normalised residual sums, badly conditioned (0.81 measured); the
    # well-conditioned comparison is test_real_network_gradient_direction
    assert num / den >= 0.75, num / den
    moved, agree = 0, 0
    for k in z['pick']:
        k = str(k)
        p0, p2 = z['p_init_' + k], z['p1_' + k]
        got = sd[k].detach().reshape(-1)[:96].cpu().numpy()
        assert np.abs(got - p0).max() <= 2.05 * lr + 1e-4 * np.abs(p0).max(), k       # Adam moves at most ~lr per step
        step_ref, step_got = p2 - p0, got - p0
        big = np.abs(step_ref) > 1.5 * lr                                             # the reference moved the same way twice
        moved += int(big.sum())
        agree += int((np.sign(step_got[big]) == np.sign(step_ref[big])).sum())
    print('two-step direction agreement with the reference snapshots', agree, '/', moved)
    assert moved > 200 and agree >= 0.7 * moved, (moved, agree)
    # padding never leaves zero
    named = m.trainer_named_parameters()
    wp = named['stage2.0.branches.0.0.conv1.weight'].detach()
    assert float(wp[18:].abs().max()) == 0 and float(wp[:, 18:].abs().max()) == 0
    # BatchNorm running statistics use the model's momentum (0.01)
    np.testing.assert_allclose(sd['bn1.running_mean'].cpu().numpy(), z['rm_bn1.running_mean'], rtol=2e-2, atol=2e-4)


def test_eval_after_training_uses_the_trained_weights():
    import torch
    from cdnet_amd import trainer
    m, ref, x, t = _setup()
    m.eval()
    with torch.no_grad():
        before = [o.clone() for o in m(x.cuda())]          # caches packed weights / BatchNorm folds
    tr = trainer.Trainer(m)
    batch = _dev_batch(x, t)
    losses = [float(tr.train_step(*batch)[0]) for _ in range(5)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    ref.load_state_dict(m.state_dict())
    ref.eval()
    m.eval()
    with torch.no_grad():
        got = m(x.cuda())
        want = ref(x)
    assert float((got[0] - before[0]).abs().max()) > 1e-3
    for g_, w_ in zip(got, want):
        scale = float(w_.abs().max())
        assert float((g_.cpu() - w_).abs().max()) <= 7e-2 * scale and float((g_.cpu() - w_).abs().mean()) <= 1.2e-2 * scale


@pytest.fixture
def fp32_mode():
    import cdnet_amd
    before = cdnet_amd.get_precision()
    cdnet_amd.set_precision('fp32')
    yield
    cdnet_amd.set_precision(before)


def test_fp32_mode_loss_values_and_linearised_gradients(fp32_mode):
    """the training step in the fp32 precision mode: losses against the fp32 oracle at 1e-4 (1.5e-2 on the 16-bit path) and the
    gradients of the linearised network per parameter at 2e-3 worst / 2e-4 median (15 % / 4 %)"""
    import torch
    from cdnet_amd import runtime
    m, ref, x, t = _setup()
    tr, _ = _hip_grads(m, x, t)
    L, _ = _oracle_grads(ref, x, t)
    got = tr.losses.cpu().numpy()[:6]
    want = [L[k] for k in ('total', 'dce', 'wdice', 'mse', 'ce', 'dice')]
    np.testing.assert_allclose(got, want, rtol=1e-4)
    F = torch.nn.functional
    relu0 = F.relu
    runtime.DEBUG_NORELU = True
    F.relu = lambda v, inplace=False: v
    try:
        m, ref, x, t = _setup()
        _, g = _hip_grads(m, x, t)
        _, rg = _oracle_grads(ref, x, t)
    finally:
        runtime.DEBUG_NORELU = False
        F.relu = relu0
    rel = {n: float((g[n] - want).norm() / want.norm()) for n, want in rg.items() if want.norm() >= 1e-6}
    worst = max(rel, key=rel.get)
    print('fp32 mode: worst', worst, rel[worst], 'median', np.median(list(rel.values())))
    assert rel[worst] <= 2e-3, (worst, rel[worst], sorted(rel.items(), key=lambda kv: -kv[1])[:8])
    assert np.median(list(rel.values())) <= 2e-4


def test_fp32_mode_two_steps_follow_the_reference_train_loop(golden, fp32_mode):
    """the reference's own two train() iterations on HRNet18_rev1 (tests/golden/hrnet_train.npz), fp32 mode: the first iteration's five
    losses at 2e-4 (2e-2 on the 16-bit path), the second at 5e-2 (1e-1; Adam's first step is lr * sign(g): the signs of near-zero
    gradients decide where those parameters go - measured 0.1-3.4 %)"""
    from cdnet_amd import trainer
    z = golden('hrnet_train')
    B, _, H, W, _ = [int(v) for v in z['x_cfg']]
    m, ref, x, t = _setup(B, H, gain=float(z['gain']))
    tr = trainer.Trainer(m, lr=float(z['lr']))
    batch = _dev_batch(x, t)
    for it in range(2):
        got = tr.train_step(*batch).cpu().numpy()
        np.testing.assert_allclose(got[:5], z['results'][it][:5], rtol=2e-4 if it == 0 else 5e-2, err_msg='iteration %d' % it)
    sd = m.state_dict()
    np.testing.assert_allclose(sd['bn1.running_mean'].cpu().numpy(), z['rm_bn1.running_mean'], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
def test_residual_1x1_backward_beside_the_chain_is_bit_identical_on_hrnet(monkeypatch, precision):
    """HRNet's first residual unit reads the concatenation of the four branches: the input gradient of its 1x1 branch - computed on the
    weight-gradient stream (trainer._RU_1X1_SIDE) - is consumed by FuseNode.backward / Trainer.cat_grad, not by a convolution layer's
    BatchNorm backward.  Every consumer of a gradient list waits for the producing stream's event (Trainer.take): beside == on the chain,
    bit for bit on every parameter gradient (a missed wait reads a half-written buffer and shows up here)."""
    import torch
    import cdnet_amd
    from cdnet_amd import trainer
    before = cdnet_amd.get_precision()
    cdnet_amd.set_precision(precision)
    try:
        grads = []
        for beside in (False, True, True):
            monkeypatch.setattr(trainer, '_RU_1X1_SIDE', beside)
            m, ref, x, t = _setup(B=2, S=128)
            _, g = _hip_grads(m, x, t)
            grads.append(g)
        for g in grads[1:]:
            for n in grads[0]:
                assert torch.equal(grads[0][n], g[n]), n
    finally:
        cdnet_amd.set_precision(before)


def test_cfg5_step_at_its_own_size():
    """BASELINE config 5's per-rank shape: HRNet18_rev1 (seg_hrnet_rev1.py:289-548), 4 tiles of 512x512, training step in the 16-bit mode
    (the one tools/bench_hrnet.py times): deterministic bit for bit from the same state, finite, and five Adam steps learn"""
    import torch
    from cdnet_amd import trainer
    from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet

    class O:
        model = {'out_c': 3}
    dev = torch.device('cuda:0')
    batch = trainer.synthetic_batch(4, dev, seed=5, H=512, W=512)
    runs = []
    for rep in range(2):
        torch.manual_seed(0)
        m = HighResolutionNet(O()).cuda().train()
        tr = trainer.Trainer(m)
        runs.append(torch.stack([tr.train_step(*batch).clone() for _ in range(5)]).cpu().numpy())
        del tr, m
        torch.cuda.empty_cache()
    assert np.isfinite(runs[0]).all()
    assert np.array_equal(runs[0], runs[1]), 'the 4 x 512x512 HRNet step is not deterministic'
    assert runs[0][-1, 0] < runs[0][0, 0], runs[0][:, 0]
    print('cfg 5 4 x 512x512: loss %.5f -> %.5f after 5 steps' % (runs[0][0, 0], runs[0][-1, 0]))
