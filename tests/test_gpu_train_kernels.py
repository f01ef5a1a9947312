"""GPU numerics of the training kernels against plain PyTorch fp32 (CPU autograd) references of the same ops."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf(x):
    import torch
    return x.to(torch.bfloat16).to(torch.float32)


def _nhwc(x, dtype=None):
    import torch
    return x.permute(0, 2, 3, 1).contiguous().to(dtype or torch.bfloat16).cuda()


def _nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


def _wgrad(srcs, g, weight_shape, kind, H, W, cin_real=None, deferred=None):
    """run cdnet_conv_backward_weight for every source; returns dW (cpu) - or, with a list in `deferred`, leaves the slabs, appends
    the reduce descriptors' arguments and returns dW on the device (to be filled by _reduce_batch)"""
    import torch
    from cdnet_amd import _lib, engine, trainer
    lib = _lib.load()
    dw = torch.zeros(weight_shape, dtype=torch.float32, device='cuda')
    mode = {'conv3': 0, 'conv1': 0, 'convT4': 2, 'convT2': 3}[kind]
    taps = {'conv3': 9, 'conv1': 1, 'convT4': 4, 'convT2': 1}[kind]
    tr = kind.startswith('convT')
    npar, ostride = (4, 2) if tr else (1, 1)
    Cout = weight_shape[1] if tr else weight_shape[0]
    cin_real = cin_real or (weight_shape[0] if tr else weight_shape[1])
    N = g.shape[0]
    coff = 0
    for s in srcs:
        ci_t = trainer._choose_ci_tiles(s.C, Cout)
        ksplit = 7
        slab = torch.full((lib.cdnet_conv_wgrad_slab_floats(s.C, Cout, taps, npar, ci_t, ksplit),), float('nan'), dtype=torch.float32, device='cuda')
        cs = engine.ConvSrc()
        s.fill(cs)
        _lib.call('cdnet_conv_backward_weight', C.byref(cs), coff, min(s.C, cin_real - coff), cin_real, _lib.ptr(g), Cout, N, H, W,
                  taps, npar, ostride, ci_t, ksplit, _lib.ptr(slab), _lib.ptr(dw), mode | (0x100 if deferred is not None else 0), _lib.stream_ptr())
        if deferred is not None:
            deferred.append(((s.C, coff, min(s.C, cin_real - coff), cin_real, Cout, taps, npar, ci_t, ksplit, _lib.ptr(slab), _lib.ptr(dw), mode), slab))
        coff += s.C
    return dw.cpu() if deferred is None else dw


def _reduce_batch(deferred):
    import torch
    from cdnet_amd import _lib
    descs = (_lib.WgradReduceDesc * len(deferred))()
    b0 = 0
    for i, (a, _) in enumerate(deferred):
        _lib.call('cdnet_wgrad_reduce_desc_fill', *a, b0, C.byref(descs[i]))
        assert descs[i].block0 == b0 and descs[i].blocks > 0
        b0 += descs[i].blocks
    tab = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).cuda()
    _lib.call('cdnet_wgrad_reduce_batch', _lib.ptr(tab), len(deferred), b0, _lib.stream_ptr())
    torch.cuda.synchronize()


@pytest.mark.parametrize('case', [(2, 64, 64, 24, 40), (1, 32, 128, 16, 16), (2, 128, 32, 9, 21), (1, 16, 64, 32, 32),
                                  (2, 64, 16, 16, 48), (1, 256, 64, 8, 8), (2, 32, 32, 24, 40), (3, 48, 24, 17, 33), (1, 16, 16, 64, 64)])
def test_wgrad_conv3x3_plain(case):
    import torch
    from cdnet_amd import engine
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn((N, Cin, H, W), generator=g))
    dy = _bf(torch.randn((N, Cout, H, W), generator=g))
    want = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1)
    got = _wgrad([engine.Src(_nhwc(x))], _nhwc(dy), (Cout, Cin, 3, 3), 'conv3', H, W)
    assert _rel(got, want) < 2e-3, _rel(got, want)


def test_wgrad_fused_sources_and_stem():
    """two concat sources: pooled+affine+relu (fp16 raw) and padded residual source; plus the zero-padded RGB stem"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(5)
    N, Ca, Cb, Cout, H, W = 2, 32, 64, 64, 20, 28
    a = torch.randn((N, Ca, 2 * H, 2 * W), generator=g).half().float()
    b = torch.randn((N, Cb, H - 1, W - 3), generator=g).half().float()
    res = torch.randn((N, Cb, H - 1, W - 3), generator=g).half().float()
    sa, ha = torch.rand((Ca,), generator=g) + 0.5, torch.randn((Ca,), generator=g) * 0.3
    sa[::3] *= -1
    sb, hb = torch.rand((Cb,), generator=g) + 0.5, torch.randn((Cb,), generator=g) * 0.3
    dy = _bf(torch.randn((N, Cout, H, W), generator=g))
    ta = _bf(F.max_pool2d(_bf(F.relu(a * sa.view(1, -1, 1, 1) + ha.view(1, -1, 1, 1))), 2))
    tb = F.pad(_bf(F.relu(b * sb.view(1, -1, 1, 1) + hb.view(1, -1, 1, 1) + res)), (1, 2, 0, 1))
    want = torch.nn.grad.conv2d_weight(torch.cat([ta, tb], 1), (Cout, Ca + Cb, 3, 3), dy, padding=1)
    srcs = [engine.Src(_nhwc(a, torch.float16), sa.cuda(), ha.cuda(), relu=True, pool=True),
            engine.Src(_nhwc(b, torch.float16), sb.cuda(), hb.cuda(), relu=True, res=_nhwc(res, torch.float16), off=(0, 1))]
    got = _wgrad(srcs, _nhwc(dy), (Cout, Ca + Cb, 3, 3), 'conv3', H, W)
    assert _rel(got, want) < 3e-3, _rel(got, want)
    # stem: 3 real channels stored as 16
    x = _bf(torch.rand((2, 3, 32, 32), generator=g))
    dy = _bf(torch.randn((2, 64, 32, 32), generator=g))
    want = torch.nn.grad.conv2d_weight(x, (64, 3, 3, 3), dy, padding=1)
    x16 = torch.zeros((2, 16, 32, 32)); x16[:, :3] = x
    got = _wgrad([engine.Src(_nhwc(x16))], _nhwc(dy), (64, 3, 3, 3), 'conv3', 32, 32, cin_real=3)
    assert _rel(got, want) < 2e-3


def test_wgrad_conv1x1_and_transposed():
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(7)
    x = _bf(torch.randn((2, 64, 24, 16), generator=g))
    dy = _bf(torch.randn((2, 64, 24, 16), generator=g))
    want = torch.nn.grad.conv2d_weight(x, (64, 64, 1, 1), dy)
    got = _wgrad([engine.Src(_nhwc(x))], _nhwc(dy), (64, 64, 1, 1), 'conv1', 24, 16)
    assert _rel(got, want) < 2e-3
    for (N, Cin, Cout, H, W) in [(2, 64, 32, 8, 8), (1, 32, 16, 24, 40)]:
        x = _bf(torch.randn((N, Cin, H, W), generator=g)).requires_grad_(False)
        w = torch.zeros((Cin, Cout, 4, 4), requires_grad=True)
        dy = _bf(torch.randn((N, Cout, 2 * H, 2 * W), generator=g))
        F.conv_transpose2d(x, w, None, stride=2, padding=1).backward(dy)
        got = _wgrad([engine.Src(_nhwc(x))], _nhwc(dy), (Cin, Cout, 4, 4), 'convT4', H, W)
        assert _rel(got, w.grad) < 2e-3, ('convT4', _rel(got, w.grad))
    x = _bf(torch.randn((2, 64, 12, 20), generator=g))
    w = torch.zeros((64, 32, 2, 2), requires_grad=True)
    dy = _bf(torch.randn((2, 32, 24, 40), generator=g))
    F.conv_transpose2d(x, w, None, stride=2).backward(dy)
    got = _wgrad([engine.Src(_nhwc(x))], _nhwc(dy), (64, 32, 2, 2), 'convT2', 12, 20)
    assert _rel(got, w.grad) < 2e-3


@pytest.mark.parametrize('f32', [False, True])
def test_wgrad_deferred_batch_reduce_is_bit_identical(f32):
    """CDNET_WGRAD_DEFER_REDUCE + one cdnet_wgrad_reduce_batch launch over the slabs of six calls (3x3 with two concat sources and a
    zero-padded stem, 1x1, both transposed forms; 16-bit and fp32 tensors) == the reduce inside every call, bit for bit; the
    untouched dW elements of a call stay untouched"""
    import torch
    from cdnet_amd import engine
    gen = torch.Generator().manual_seed(11)
    cast = (lambda t: t.float().cuda().contiguous()) if f32 else (lambda t: t)
    nh = lambda t: cast(_nhwc(_bf(t)))
    jobs = []
    a, b = torch.randn((2, 32, 20, 28), generator=gen), torch.randn((2, 64, 20, 28), generator=gen)
    jobs.append(([engine.Src(nh(a)), engine.Src(nh(b))], nh(torch.randn((2, 64, 20, 28), generator=gen)), (64, 96, 3, 3), 'conv3', 20, 28, None))
    x16 = torch.zeros((2, 16, 32, 32)); x16[:, :3] = torch.rand((2, 3, 32, 32), generator=gen)
    jobs.append(([engine.Src(nh(x16))], nh(torch.randn((2, 64, 32, 32), generator=gen)), (64, 3, 3, 3), 'conv3', 32, 32, 3))
    jobs.append(([engine.Src(nh(torch.randn((2, 64, 24, 16), generator=gen)))], nh(torch.randn((2, 64, 24, 16), generator=gen)), (64, 64, 1, 1), 'conv1', 24, 16, None))
    jobs.append(([engine.Src(nh(torch.randn((2, 64, 8, 8), generator=gen)))], nh(torch.randn((2, 32, 16, 16), generator=gen)), (64, 32, 4, 4), 'convT4', 8, 8, None))
    jobs.append(([engine.Src(nh(torch.randn((2, 64, 12, 20), generator=gen)))], nh(torch.randn((2, 32, 24, 40), generator=gen)), (64, 32, 2, 2), 'convT2', 12, 20, None))
    want = [_wgrad(srcs, g, shape, kind, H, W, cin_real=cr) for srcs, g, shape, kind, H, W, cr in jobs]
    deferred = []
    dws = [_wgrad(srcs, g, shape, kind, H, W, cin_real=cr, deferred=deferred) for srcs, g, shape, kind, H, W, cr in jobs]
    assert len(deferred) == 6 and all(float(d.abs().sum()) == 0.0 for d in dws)        # nothing summed yet
    _reduce_batch(deferred)
    for w, d in zip(want, dws):
        assert float(w.abs().sum()) > 0 and torch.equal(w, d.cpu())


def test_backward_data_of_transposed_conv_via_space_to_depth():
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(9)
    for (N, Cin, Cout, H, W, k) in [(2, 64, 32, 8, 8, 4), (1, 32, 16, 24, 40, 4), (2, 64, 32, 12, 20, 2)]:
        x = torch.randn((N, Cin, H, W), generator=g).requires_grad_(True)
        w = _bf(torch.randn((Cin, Cout, k, k), generator=g) * 0.1)
        dy = _bf(torch.randn((N, Cout, 2 * H, 2 * W), generator=g))
        F.conv_transpose2d(x, w, None, stride=2, padding=1 if k == 4 else 0).backward(dy)
        cfg = engine.choose_cfg([2 * Cout, 2 * Cout], Cin, H, W)
        wp = engine.pack_weights(w.cuda(), cfg, 4 if k == 4 else 5)
        gy = _nhwc(dy)
        views = [engine.Src(gy, view=(a * 2 * W * Cout, H, W, 2 * Cout, 4 * W * Cout)) for a in (0, 1)]
        out, _ = engine.conv_forward(views, wp, Cin, cfg, taps=9 if k == 4 else 1, H=H, W=W)
        assert _rel(_nchw(out), x.grad) < 6e-3, (k, _rel(_nchw(out), x.grad))


def _bn_case(pooled, with_res, two_grads, seed):
    import torch
    import torch.nn.functional as F
    from cdnet_amd import _lib, trainer
    g = torch.Generator().manual_seed(seed)
    N, Cc, H, W = 2, 32, 12, 20
    raw = torch.randn((N, Cc, H, W), generator=g).half().float().requires_grad_(True)
    res = torch.randn((N, Cc, H, W), generator=g).half().float().requires_grad_(True) if with_res else None
    gamma = (torch.rand((Cc,), generator=g) + 0.5).requires_grad_(True)
    gamma.data[::4] *= -1
    beta = (torch.randn((Cc,), generator=g) * 0.2).requires_grad_(True)
    mean = raw.detach().mean((0, 2, 3))
    var = raw.detach().var((0, 2, 3), unbiased=False)
    y = F.batch_norm(raw, None, None, gamma, beta, training=True, eps=1e-5)
    if with_res:
        y = y + res
    a = F.relu(y)
    a = a + (_bf(a.detach()) - a.detach())      # consumers see the activation rounded to bf16 (as the conv staging does)
    total = 0
    gins = []
    if pooled:
        p = F.max_pool2d(a, 2)
        gp = _bf(torch.randn(p.shape, generator=g))
        total = total + (p * gp).sum()
        gins.append(trainer._G(_nhwc(gp), p.shape[2], p.shape[3], pooled=1))
    if two_grads or not pooled:
        # consumer that read the tensor through F.pad offsets (1, 2) and as a channel slice of a wider gradient
        ap = F.pad(a, (2, 1, 1, 0))
        gfull = _bf(torch.randn((N, Cc + 16, H + 1, W + 3), generator=g))
        total = total + (ap * gfull[:, 8:8 + Cc]).sum()
        gins.append(trainer._G(_nhwc(gfull), H + 1, W + 3, oy=1, ox=2, coff=8, cstride=Cc + 16))
    total.backward()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = (gamma.detach() * invstd)
    shift = beta.detach() - mean * scale
    A = trainer.BnBwdArgs()
    raw_d = _nhwc(raw.detach(), torch.float16)
    res_d = _nhwc(res.detach(), torch.float16) if with_res else None
    keep = [raw_d, res_d]
    A.raw, A.res = raw_d.data_ptr(), (res_d.data_ptr() if with_res else None)
    dev = lambda t: t.detach().float().cuda().contiguous()
    sc, sh, mu, iv, gm = dev(scale), dev(shift), dev(mean), dev(invstd), dev(gamma)
    A.scale, A.shift, A.mean, A.invstd = sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), iv.data_ptr()
    A.ngin = len(gins)
    for k, gi in enumerate(gins):
        A.gin[k].g = gi.t.data_ptr()
        A.gin[k].Hg, A.gin[k].Wg, A.gin[k].oy, A.gin[k].ox = gi.Hg, gi.Wg, gi.oy, gi.ox
        A.gin[k].pooled, A.gin[k].coff, A.gin[k].cstride = gi.pooled, gi.coff, gi.cstride or Cc
    A.f16, A.relu, A.N, A.H, A.W, A.C = 1, 1, N, H, W, Cc
    ws = torch.empty((_lib.load().cdnet_bn_backward_workspace_floats(Cc),), dtype=torch.float32, device='cuda')
    dgamma, dbeta = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    draw = torch.empty((N, H, W, Cc), dtype=torch.bfloat16, device='cuda')
    dz = torch.empty((N, H, W, Cc), dtype=torch.bfloat16, device='cuda')
    _lib.call('cdnet_bn_backward', C.byref(A), _lib.ptr(gm), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), ws.numel(),
              _lib.ptr(draw), _lib.ptr(dz) if with_res else None, _lib.stream_ptr())
    assert _rel(_nchw(draw), raw.grad) < 1e-2, ('draw', _rel(_nchw(draw), raw.grad))
    assert _rel(dgamma.cpu(), gamma.grad) < 3e-3 and _rel(dbeta.cpu(), beta.grad) < 3e-3
    if with_res:
        assert _rel(_nchw(dz), res.grad) < 6e-3


@pytest.mark.parametrize('cfg', [(False, False, False), (True, False, False), (True, False, True), (False, True, False),
                                 (True, True, True)])
def test_bn_relu_pool_pad_backward(cfg):
    _bn_case(*cfg, seed=11)


def test_loss_values_and_gradients(golden):
    import torch
    from cdnet_amd import synth, _lib
    from oracle import train as ot
    for quirk, (B, H, W) in ((1, (3, 40, 48)), (0, (2, 64, 64)), (1, (16, 32, 32))):
        lab, dirn, point, weight = synth.train_targets(B, H, W, 11)
        if quirk:
            dirn[B - 1] = 0                       # a sample whose direction map is constant (train_util_dam.py:141)
        rs = np.random.RandomState(5)
        lm = torch.from_numpy((rs.randn(B, 3, H, W) * 2).astype(np.float32)).requires_grad_(True)
        ld = torch.from_numpy((rs.randn(B, 9, H, W) * 2).astype(np.float32)).requires_grad_(True)
        lp = torch.from_numpy(rs.randn(B, 1, H, W).astype(np.float32)).requires_grad_(True)
        L = ot.dam_losses(lm, lp, ld, torch.from_numpy(lab), torch.from_numpy(dirn), torch.from_numpy(point),
                          torch.from_numpy(weight), quirk_sample0=bool(quirk))
        L['total'].backward()
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        ws = torch.empty((_lib.load().cdnet_dam_loss_workspace_floats(B, H * W),), dtype=torch.float32, device='cuda')
        losses = torch.zeros(11, device='cuda')
        dm, dp, dd = torch.empty_like(lm, device='cuda'), torch.empty_like(lp, device='cuda'), torch.empty_like(ld, device='cuda')
        keep = [lm.detach().cuda(), lp.detach().cuda(), ld.detach().cuda(), dev(lab), dev(dirn), dev(point), dev(weight[:, 0])]
        _lib.call('cdnet_dam_loss', *[_lib.ptr(t) for t in keep], B, H, W, quirk,
                  _lib.ptr(ws), ws.numel(), _lib.ptr(losses), _lib.ptr(dm), _lib.ptr(dp), _lib.ptr(dd), _lib.stream_ptr())
        got = losses.cpu().numpy()[:6]
        want = [float(L[k]) for k in ('total', 'dce', 'wdice', 'mse', 'ce', 'dice')]
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(losses.cpu().numpy()[6:], ot.pixel_metrics(ld.detach().argmax(1).numpy(), dirn), rtol=1e-6, atol=1e-7)
        assert _rel(dm.cpu(), lm.grad) < 1e-4 and _rel(dd.cpu(), ld.grad) < 1e-4 and _rel(dp.cpu(), lp.grad) < 1e-4


@pytest.mark.parametrize('C', [5, 17])
def test_loss_other_direction_class_counts(golden, C):
    """cdnet_dam_loss_classes on 4+1 / 16+1 direction classes (model_unet_MandD4 / MandD16): the direction terms and their gradient
    against the reference's golden (plain one-hot target = quirk off, every pixel labelled foreground so that no one-hot row is
    masked), then the masked / quirk variants against the oracle"""
    import torch
    from cdnet_amd import synth, _lib
    from oracle import train as ot
    z = golden('losses_classes')
    B, H, W, tseed, lseed = [int(v) for v in z['cfg']]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def run(lm, lp, ld, lab, dirn, point, weight, quirk):
        ws = torch.empty((_lib.load().cdnet_dam_loss_classes_workspace_floats(B, H * W, C),), dtype=torch.float32, device='cuda')
        losses = torch.zeros(11, device='cuda')
        dm, dp, dd = torch.empty_like(lm, device='cuda'), torch.empty_like(lp, device='cuda'), torch.empty_like(ld, device='cuda')
        keep = [lm.detach().cuda(), lp.detach().cuda(), ld.detach().cuda(), dev(lab), dev(dirn), dev(point), dev(weight[:, 0])]
        _lib.call('cdnet_dam_loss_classes', *[_lib.ptr(t) for t in keep], B, H, W, C, quirk,
                  _lib.ptr(ws), ws.numel(), _lib.ptr(losses), _lib.ptr(dm), _lib.ptr(dp), _lib.ptr(dd), _lib.stream_ptr())
        return losses.cpu().numpy(), dm.cpu(), dp.cpu(), dd.cpu()

    lab, dirn, point, weight = synth.train_targets(B, H, W, tseed)
    dirn = synth.remap_direction(dirn, C)
    ld = torch.from_numpy((np.random.RandomState(lseed + C).randn(B, C, H, W) * 2).astype(np.float32)).requires_grad_(True)
    rs = np.random.RandomState(3)
    lm = torch.from_numpy((rs.randn(B, 3, H, W) * 2).astype(np.float32)).requires_grad_(True)
    lp = torch.from_numpy(rs.randn(B, 1, H, W).astype(np.float32)).requires_grad_(True)
    # golden: one-hot of the direction map without the foreground mask == label 1 everywhere
    losses, _, _, dd = run(lm, lp, ld, np.ones_like(lab), dirn, point, weight, 0)
    np.testing.assert_allclose(losses[1:3], [float(z['c%d_dce' % C]), float(z['c%d_wdice' % C])], rtol=2e-5, atol=2e-6)
    assert _rel(dd, torch.from_numpy(z['c%d_g_dir' % C])) < 1e-4
    # masked one-hot, sample-0 quirk, a constant sample: vs the oracle
    for quirk in (1, 0):
        d2 = dirn.copy()
        if quirk:
            d2[B - 1] = 0
        for t in (lm, lp, ld):
            t.grad = None
        L = ot.dam_losses(lm, lp, ld, torch.from_numpy(lab), torch.from_numpy(d2), torch.from_numpy(point), torch.from_numpy(weight),
                          quirk_sample0=bool(quirk))
        L['total'].backward()
        losses, dm, dp, dd = run(lm, lp, ld, lab, d2, point, weight, quirk)
        np.testing.assert_allclose(losses[:6], [float(L[k]) for k in ('total', 'dce', 'wdice', 'mse', 'ce', 'dice')], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(losses[6:], ot.pixel_metrics(ld.detach().argmax(1).numpy(), d2), rtol=1e-6, atol=1e-7)
        assert _rel(dm, lm.grad) < 1e-4 and _rel(dd, ld.grad) < 1e-4 and _rel(dp, lp.grad) < 1e-4
    # a direction class beyond the map's range poisons the losses (no silent clamp)
    bad = dirn.copy()
    bad[0, 0, 0] = C
    assert np.isnan(run(lm, lp, ld, lab, bad, point, weight, 1)[0]).all()


def test_adam_matches_torch():
    import torch
    from cdnet_amd import _lib
    g = torch.Generator().manual_seed(1)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.99), weight_decay=1e-4)
    p, m, v = p0.clone().cuda(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * (0.1 if step != 2 else 1e-6)
        ref.grad = gr.clone()
        opt.step()
        grd = gr.cuda()
        _lib.call('cdnet_adam_step', _lib.ptr(p), _lib.ptr(grd), _lib.ptr(m), _lib.ptr(v), n, 1e-3, 0.9, 0.99, 1e-8, 1e-4,
                  step, 1.0, _lib.stream_ptr())
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-7)


def test_dam_head_backward():
    import torch
    from cdnet_amd import _lib, runtime, engine
    from oracle import models as om
    torch.manual_seed(3)
    net = om.Unet()
    N, H, W = 2, 24, 20
    F = [_bf(torch.randn((N, 64, H, W))).requires_grad_(True) for _ in range(3)]
    point = net.point_conv(F[2])
    direction = net.direction_conv(net.directionAtt(F[1], point))
    mask = net.mask_conv(net.maskAtt(F[0], direction))
    gm, gp, gd = torch.randn_like(mask), torch.randn_like(point), torch.randn_like(direction)
    ((mask * gm).sum() + (point * gp).sum() + (direction * gd).sum()).backward()
    ps = [net.point_conv.weight, net.direction_conv.weight, net.mask_conv.weight, net.point_conv.bias, net.direction_conv.bias,
          net.mask_conv.bias, net.directionAtt.Conv1x1.weight, net.maskAtt.Conv1x1.weight]
    hw = torch.cat([p.detach().reshape(-1) for p in ps]).cuda()
    want_dw = torch.cat([p.grad.reshape(-1) for p in ps])
    feats = [engine.Src(_nhwc(f.detach())) for f in F]
    hf = [runtime.head_feat(f) for f in feats]
    df = [torch.empty((N, H, W, 64), dtype=torch.bfloat16, device='cuda') for _ in range(3)]
    ws = torch.empty((_lib.load().cdnet_dam_head_backward_workspace_floats(N, H, W),), dtype=torch.float32, device='cuda')
    dhw = torch.zeros(855, device='cuda')
    gmd, gpd, gdd = gm.cuda(), gp.cuda(), gd.cuda()
    _lib.call('cdnet_dam_head_backward', C.byref(hf[0]), C.byref(hf[1]), C.byref(hf[2]), _lib.ptr(hw), _lib.ptr(gmd),
              _lib.ptr(gpd), _lib.ptr(gdd), N, H, W, _lib.ptr(df[0]), _lib.ptr(df[1]), _lib.ptr(df[2]), _lib.ptr(ws),
              ws.numel(), _lib.ptr(dhw), _lib.stream_ptr())
    for k in range(3):
        assert _rel(_nchw(df[k]), F[k].grad) < 6e-3, (k, _rel(_nchw(df[k]), F[k].grad))
    got = dhw.cpu()
    off = 0
    for p in ps:
        n = p.numel()
        assert _rel(got[off:off + n], p.grad.reshape(-1)) < 2e-3, (off, _rel(got[off:off + n], p.grad.reshape(-1)))
        off += n
