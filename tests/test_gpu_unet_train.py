"""Plain UNet (models/unet.py, BASELINE config 1) training step on the GPU against the fp32 oracle (oracle/train.py
unet_train_iteration, pinned to the reference's train_util.train by tests/golden/unet_train_iter.npz).
Same three-way comparison as test_gpu_train_step.py: loss values, gradient direction, loss trajectory over Adam steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(B=2, S=64, seed=0):
    import torch
    from cdnet_amd import synth
    from cdnet_amd.models.unet import UNet
    from oracle import models as om
    torch.manual_seed(seed)
    ref = om.UNet(3)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.normal_(mod.bias, 0, 0.2)
        if isinstance(mod, torch.nn.ConvTranspose2d):
            torch.nn.init.normal_(mod.bias, 0, 0.1)
    m = UNet(num_classes=3)
    m.load_state_dict(ref.state_dict())
    lab, _, _, weight = synth.train_targets(B, S, S, 31)
    x = torch.from_numpy(synth.det_input((B, 3, S, S), 12))
    return m.cuda(), ref, x, torch.from_numpy(lab), torch.from_numpy(weight)


@pytest.mark.parametrize('K', [3, 5, 9, 17])
def test_final_conv_backward_kernel(K):
    """K = 3: the plain UNet / mask classifier; 5 / 9 / 17: the direction classifiers of the ablation heads (MandD4 / MandD / MandD16)"""
    import ctypes as C
    import torch
    from cdnet_amd import _lib, runtime
    g = torch.Generator().manual_seed(4)
    N, H, W = 2, 24, 40
    f = torch.randn((N, H, W, 64), generator=g).to(torch.bfloat16)
    w = torch.randn((K, 64), generator=g)
    dl = torch.randn((N, K, H, W), generator=g)
    ff = f.float().requires_grad_(True)
    ww = w.clone().requires_grad_(True)
    b = torch.zeros(K, requires_grad=True)
    out = torch.einsum('nhwc,kc->nkhw', ff, ww) + b.view(1, K, 1, 1)
    out.backward(dl)
    fd = f.cuda()
    src = runtime.Src(fd)
    hf = runtime.head_feat(src)
    lib = _lib.load()
    ws = torch.empty((lib.cdnet_final_conv1x1_backward_workspace_floats(),), dtype=torch.float32, device='cuda')
    df = torch.empty((N, H, W, 64), dtype=torch.bfloat16, device='cuda')
    dw = torch.empty((K, 64), dtype=torch.float32, device='cuda')
    db = torch.empty((K,), dtype=torch.float32, device='cuda')
    wd, dld = w.cuda(), dl.cuda()
    _lib.call('cdnet_final_conv1x1_backward', C.byref(hf), _lib.ptr(wd), _lib.ptr(dld), K, N, H, W, _lib.ptr(df), _lib.ptr(ws),
              ws.numel(), _lib.ptr(dw), _lib.ptr(db), _lib.stream_ptr())
    np.testing.assert_allclose(dw.cpu().numpy(), ww.grad.numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(df.float().cpu().numpy(), ff.grad.numpy(), rtol=1e-2, atol=1e-2)


def test_loss_gradients_and_trajectory():
    import torch
    from cdnet_amd import trainer
    from oracle import train as ot
    m, ref, x, lab, weight = _setup()
    dev = torch.device('cuda:0')
    tr = trainer.UNetTrainer(m)
    xd, labd, wd = x.to(dev), lab.to(dev), weight[:, 0].contiguous().to(dev)
    logits = tr.forward(xd)
    dl = tr.loss_and_grads(logits, labd, wd)
    tr.backward(dl)
    torch.cuda.synchronize()
    got = tr.unet_losses.cpu().numpy()
    ref.train()
    ref.zero_grad()
    L = ot.unet_losses(ref(x), lab, weight)
    L['total'].backward()
    np.testing.assert_allclose(got, [float(L['total']), float(L['ce']), float(L['dice'])], rtol=3e-3)
    # gradient direction (a 16-bit forward flips ReLU decisions: see DESIGN.md section 6)
    cos, worst = [], (1.0, None)
    for n, p in ref.named_parameters():
        if p.grad is None or p.grad.norm() < 1e-6:
            continue
        g = dict(m.named_parameters())[n].grad.detach().float().cpu()
        c = float((g * p.grad).sum() / (g.norm() * p.grad.norm() + 1e-30))
        cos.append(c)
        if c < worst[0]:
            worst = (c, n)
    assert min(cos) > 0.8, worst
    assert np.median(cos) > 0.93
    c_final = [c for c, (n, _) in zip(cos, [(n, p) for n, p in ref.named_parameters() if p.grad is not None and p.grad.norm() >= 1e-6])
               if n.startswith('final_conv')]
    assert min(c_final) > 0.999
    # ConvTranspose2d biases (no BatchNorm behind them): the dedicated bias-gradient kernel
    g = dict(m.named_parameters())['up4.up.bias'].grad.detach().float().cpu()
    want = dict(ref.named_parameters())['up4.up.bias'].grad
    assert float((g * want).sum() / (g.norm() * want.norm())) > 0.98
    # a few Adam steps track the fp32 oracle's loss trajectory
    opt = ot.make_adam(ref)
    ref.zero_grad()
    m2, ref2, x, lab, weight = _setup()
    tr2 = trainer.UNetTrainer(m2)
    opt2 = ot.make_adam(ref2)
    a, b = [], []
    for _ in range(5):
        a.append(float(tr2.train_step(xd, labd, wd)[0]))
        b.append(ot.unet_train_iteration(ref2, opt2, x, lab, weight)['total'])
    np.testing.assert_allclose(a, b, rtol=4e-2)
    assert a[-1] < a[0]


def test_linearised_network_gradients_tight():
    """ReLU off on both sides (the 16-bit forward then flips no ReLU decisions): the backward orchestration of the plain
    UNet - BN backward, dW, backward-data, ceil-mode pool routing, cat([skip, up]) + crop, ConvTranspose2d k2s2 and its
    bias, the classifier - must agree with fp32 autograd up to rounding."""
    import torch
    from cdnet_amd import runtime, trainer
    from oracle import train as ot
    runtime.DEBUG_NORELU = True
    try:
        m, ref, x, lab, weight = _setup()
        for name, mod in list(ref.named_modules()):
            for cn, child in list(mod.named_children()):
                if isinstance(child, torch.nn.ReLU):
                    setattr(mod, cn, torch.nn.Identity())
        dev = torch.device('cuda:0')
        tr = trainer.UNetTrainer(m)
        logits = tr.forward(x.to(dev))
        dl = tr.loss_and_grads(logits, lab.to(dev), weight[:, 0].contiguous().to(dev))
        tr.backward(dl)
        torch.cuda.synchronize()
    finally:
        runtime.DEBUG_NORELU = False
    ref.train()
    ref.zero_grad()
    L = ot.unet_losses(ref(x), lab, weight)
    L['total'].backward()
    cos = {}
    for n, p in ref.named_parameters():
        if p.grad is None or p.grad.norm() < 1e-6:
            continue
        g = dict(m.named_parameters())[n].grad.detach().float().cpu()
        cos[n] = float((g * p.grad).sum() / (g.norm() * p.grad.norm() + 1e-30))
    worst = min(cos, key=cos.get)
    assert cos[worst] > 0.97, (worst, cos[worst])
    assert np.median(list(cos.values())) > 0.995


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_cfg1_at_its_own_size(precision):
    """BASELINE config 1 as named: plain UNet, 4 x 256x256x3 tiles (models/unet.py:53-106, train_util.py:58-200) - the 512 -> 1024 -> 1024 middle
    runs on 16x16 pixels here (4x4 at the fixture size of the other tests).  The first step's loss against the fp32 CPU oracle on the same
    weights and batch; the step is deterministic bit for bit (two trainers from the same state); five Adam steps learn."""
    import torch
    import cdnet_amd
    from cdnet_amd import trainer
    from cdnet_amd.models.unet import UNet
    from oracle import models as om
    from oracle import train as ot
    before = cdnet_amd.get_precision()
    cdnet_amd.set_precision(precision)
    try:
        dev = torch.device('cuda:0')
        torch.manual_seed(3)
        ref = om.UNet(3)
        x, lab, _, _, weight = trainer.synthetic_batch(4, torch.device('cpu'), seed=11)
        runs = []
        for rep in range(2):
            m = UNet(num_classes=3)
            m.load_state_dict(ref.state_dict())
            tr = trainer.UNetTrainer(m.cuda())
            xd, ld, wd = x.to(dev), lab.to(dev), weight.to(dev)
            runs.append([tr.train_step(xd, ld, wd).clone() for _ in range(5)])
        torch.cuda.synchronize()
        a = torch.stack(runs[0]).cpu().numpy()
        b = torch.stack(runs[1]).cpu().numpy()
        assert np.array_equal(a, b), 'the 4 x 256x256 UNet step is not deterministic'
        ref.train()
        with torch.no_grad():
            L = ot.unet_losses(ref(x), lab, weight)
        want = [float(L['total']), float(L['ce']), float(L['dice'])]
        np.testing.assert_allclose(a[0], want, rtol=2e-4 if precision == 'fp32' else 3e-3)
        assert a[-1, 0] < a[0, 0], a[:, 0]
        print('cfg 1 [%s] 4 x 256x256: loss %.5f (oracle %.5f) -> %.5f after 5 steps' % (precision, a[0, 0], want[0], a[-1, 0]))
    finally:
        cdnet_amd.set_precision(before)
