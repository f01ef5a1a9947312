"""Training step of the ablation heads (models/dam/model_unet_MandD.py / MandD4 / MandD16 / MandDandP through train_util_dam.train):
cdnet_amd.trainer.AblationTrainer against the oracle (oracle.models.Unet(variant=...), oracle.train.ablation_losses).
  * loss values vs the fp32 oracle, 2e-3 relative (bf16 path);
  * gradients of the linearised network in fp32 precision vs the oracle's autograd: median relative error <= 2e-3, worst <= 5e-2
    (the conditioning argument of tests/test_gpu_fp32.py);
  * the parameters the reference's forward never touches get no gradient and are never stepped; six Adam steps track the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(variant, B=2, S=64, seed=0):
    import importlib
    import torch
    from cdnet_amd import synth
    from oracle import models as om
    torch.manual_seed(seed)
    classes = {'MandD4': 5, 'MandD16': 17}.get(variant, 9)          # options.py:45 direction_classes of the 4- / 16-direction ablations
    ref = om.Unet(variant='MandD' if classes != 9 else variant, direction_classes=classes)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.normal_(mod.bias, 0, 0.2)
    Unet = importlib.import_module('cdnet_amd.models.dam.model_unet_' + variant).Unet
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(ref.state_dict())
    lab, dirn, point, weight = synth.train_targets(B, S, S, 21)
    dirn = synth.remap_direction(dirn, classes)
    x = torch.from_numpy(synth.det_input((B, 3, S, S), 9))
    t = [torch.from_numpy(a) for a in (lab, dirn, point, weight)]
    return m.cuda(), ref, x, t


def _hip(m, x, t):
    import torch
    from cdnet_amd import trainer
    tr = trainer.AblationTrainer(m)
    dev = torch.device('cuda:0')
    out = tr.forward(x.to(dev))
    g = tr.loss_and_grads(out, t[0].to(dev), t[1].to(dev), t[2].to(dev), t[3][:, 0].contiguous().to(dev))
    tr.backward(*g)
    torch.cuda.synchronize()
    return tr, {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if not n.startswith(m.UNUSED_PREFIXES)}


def _oracle(ref, x, t, linear=False):
    from oracle import emulate
    from oracle import train as ot
    ref.train()
    ref.zero_grad()
    if linear:
        emulate.QUANT, emulate.NORELU = False, True
        try:
            out = emulate.dam_unet_forward(ref, x)
        finally:
            emulate.QUANT, emulate.NORELU = True, False
    else:
        out = ref(x)
    L = ot.ablation_losses(out, t[0], t[1], t[2], t[3])
    L['total'].backward()
    return {k: float(v) for k, v in L.items()}, {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('variant', ['MandD', 'MandDandP', 'MandD4', 'MandD16'])
def test_ablation_loss_values_and_unused_parameters(variant):
    m, ref, x, t = _setup(variant)
    tr, g = _hip(m, x, t)
    L, rg = _oracle(ref, x, t)
    got = tr.losses.cpu().numpy()[:6]
    want = [L[k] for k in ('total', 'dce', 'wdice', 'mse', 'ce', 'dice')]
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=1e-6)
    if variant != 'MandDandP':
        assert want[3] == 0.0 and got[3] == 0.0                    # no point branch, no MSE term
    # exactly the parameters autograd reaches in the reference get a gradient here
    assert set(g.keys()) == set(rg.keys()), set(g.keys()) ^ set(rg.keys())


@pytest.mark.parametrize('variant', ['MandD', 'MandDandP', 'MandD4', 'MandD16'])
def test_ablation_linearised_gradients_fp32(variant):
    import cdnet_amd
    from cdnet_amd import runtime
    cdnet_amd.set_precision('fp32')
    runtime.DEBUG_NORELU = True
    try:
        m, ref, x, t = _setup(variant)
        tr, g = _hip(m, x, t)
        L, rg = _oracle(ref, x, t, linear=True)
    finally:
        runtime.DEBUG_NORELU = False
        cdnet_amd.set_precision('bf16')
    np.testing.assert_allclose(tr.losses.cpu().numpy()[:5], [L[k] for k in ('total', 'dce', 'wdice', 'mse', 'ce')], rtol=2e-4, atol=1e-7)
    rel = {n: float((g[n] - w).norm() / w.norm()) for n, w in rg.items() if w.norm() >= 1e-6}
    worst = max(rel, key=rel.get)
    assert np.median(list(rel.values())) <= 2e-3, np.median(list(rel.values()))
    assert rel[worst] <= 5e-2, (worst, rel[worst])
    for n in ('mask_conv.weight', 'direction_conv.weight', 'residual.conv2.weight', 'mask_feature.conv1.weight'):
        assert rel[n] <= 2e-3, (n, rel[n])


@pytest.mark.parametrize('variant', ['MandD', 'MandD16'])
def test_ablation_short_training_run_tracks_the_oracle(variant):
    import torch
    from cdnet_amd import trainer
    from oracle import train as ot
    m, ref, x, t = _setup(variant)
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    tr = trainer.AblationTrainer(m)
    dev = torch.device('cuda:0')
    batch = (x.to(dev), t[0].to(dev), t[1].to(dev), t[2].to(dev), t[3][:, 0].contiguous().to(dev))
    opt = ot.make_adam(ref)
    ours, theirs = [], []
    for _ in range(6):
        ours.append(float(tr.train_step(*batch)[0]))
        ref.train()
        L = ot.ablation_losses(ref(x), *t)
        opt.zero_grad()
        L['total'].backward()
        opt.step()
        theirs.append(float(L['total']))
    print(variant, 'trajectory', ours, theirs)
    np.testing.assert_allclose(ours[0], theirs[0], rtol=2e-3)
    np.testing.assert_allclose(ours, theirs, rtol=5e-2)
    assert ours[-1] < ours[0] * 0.95
    sd = m.state_dict()
    for n, p in ref.named_parameters():
        if n.startswith(m.UNUSED_PREFIXES):
            assert torch.equal(sd[n].cpu(), p0[n]), n


@pytest.mark.parametrize('variant', ['MandD', 'MandD4', 'MandD16'])
def test_ablation_validate_matches_oracle(variant):
    """train_util_dam.validate on the two-output models (cdnet_dam_val_sums_classes with 9 / 5 / 17 direction classes) vs
    oracle.train.validate_losses on the oracle network's eval outputs, fp32 precision: losses 2e-4, pixel metrics 5e-3"""
    import torch
    import cdnet_amd
    from cdnet_amd import train_util_dam
    from cdnet_amd.options import Options
    from oracle import train as ot
    cdnet_amd.set_precision('fp32')
    try:
        m, ref, x, t = _setup(variant)
        lab, dirn, point, weight = t
        classes = m.DIRECTION_OUT
        assert len(torch.unique(dirn)) == classes
        opt = Options(isTrain=True).parse([])
        opt.direction_classes = classes
        target0 = (lab.long() * 127 + (lab == 2).long()).unsqueeze(1)
        got = train_util_dam.validate([(x, weight, target0, point, dirn)], m, None, opt, None, all_img_test=1)
    finally:
        cdnet_amd.set_precision('bf16')
    ref.eval()
    with torch.no_grad():
        mask, direction = ref(x)
        L = ot.validate_losses(mask, torch.zeros_like(mask[:, :1]), direction, lab, dirn, torch.zeros_like(point), weight)
    want = [float(L[k]) for k in ('total', 'dce', 'ddice', 'mse')]
    np.testing.assert_allclose(got[:4], want, rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(got[4:9], ot.pixel_metrics(mask.argmax(1).numpy(), lab.numpy()), atol=5e-3)


def test_train_entry_with_two_output_model_and_reference_options():
    """train_util_dam.train with model_unet_MandD4 chosen through utils.chooseModel / get_optimizer and the option combination the
    reference needs for a two-output model (direction = 1, mseloss = 0, direction_classes = 5; train_util_dam.py:157-163) == one
    oracle iteration: losses 3e-3 (bf16 path), no point term"""
    import torch
    from cdnet_amd import synth, train_util_dam, utils
    from cdnet_amd.options import Options
    from oracle import train as ot
    _, ref, x, t = _setup('MandD4')
    opt = Options(isTrain=True)
    opt.model['modelName'], opt.model['mseloss'], opt.direction_classes = 'model_unet_MandD4', 0, 5
    m = utils.chooseModel(opt)
    m.load_state_dict(ref.state_dict())
    m = m.cuda()
    trainer, _ = utils.get_optimizer(opt, m)
    lab, dirn, point, weight = t
    target0 = (lab.long() * 127 + (lab == 2).long()).unsqueeze(1)
    got = train_util_dam.train([(x, weight, target0, point, dirn)], m, trainer, None, 0, opt, None)
    ref.train()
    L = ot.ablation_losses(ref(x), lab, dirn, point, weight)
    assert got.shape == (11,) and got[3] == 0.0 and got[5] == -1.0
    np.testing.assert_allclose(got[:5], [float(L[k]) for k in ('total', 'dce', 'wdice', 'mse', 'ce')], rtol=3e-3, atol=1e-7)
