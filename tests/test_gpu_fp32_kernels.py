"""fp32-precision training kernels (fp32 tensors, split-bf16 x3 MFMA products) against plain PyTorch fp32 / fp64 (CPU autograd)
references of the same ops.  Tolerances: MFMA kernels 5e-5 relative (2^-16 per product), streaming kernels 1e-5."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().float().cuda()


def _nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _wgrad(srcs, g, weight_shape, kind, H, W, cin_real=None, ksplit=7):
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
        slab = torch.full((lib.cdnet_conv_wgrad_slab_floats(s.C, Cout, taps, npar, ci_t, ksplit),), float('nan'), dtype=torch.float32, device='cuda')   # (a slice without tiles must still write its slab)
        cs = engine.ConvSrc()
        s.fill(cs)
        assert cs.f16 == 2
        _lib.call('cdnet_conv_backward_weight', C.byref(cs), coff, min(s.C, cin_real - coff), cin_real, _lib.ptr(g), Cout, N, H, W,
                  taps, npar, ostride, ci_t, ksplit, _lib.ptr(slab), _lib.ptr(dw), mode, _lib.stream_ptr())
        coff += s.C
    return dw.cpu()


@pytest.mark.parametrize('case', [(2, 64, 64, 24, 40), (1, 32, 128, 16, 16), (2, 128, 32, 9, 21), (1, 16, 64, 32, 32),
                                  (2, 64, 16, 16, 48), (1, 256, 64, 8, 8), (2, 16, 16, 24, 40), (1, 32, 32, 17, 33), (3, 32, 16, 8, 64)])
def test_wgrad_fp32_conv3x3(case):
    import torch
    from cdnet_amd import engine
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, Cin, H, W), generator=g)
    dy = torch.randn((N, Cout, H, W), generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)
    got = _wgrad([engine.Src(_nhwc(x))], _nhwc(dy), (Cout, Cin, 3, 3), 'conv3', H, W)
    assert _rel(got, want) < 5e-5, _rel(got, want)


@pytest.mark.parametrize('case', [(2, 64, 64, 24, 16, False), (3, 64, 64, 10, 20, True), (1, 128, 64, 7, 33, True), (2, 64, 128, 16, 48, False),
                                  (2, 16, 16, 24, 16, False), (3, 32, 64, 10, 20, True), (2, 64, 32, 9, 21, True)])
def test_wgrad_fp32_conv1x1_ws32(case):
    """1x1 weight gradient on wgrad_ws32_kernel<XF, 1> (64 x 64 channel blocks): plain and BatchNorm + ReLU sources, ragged sizes (tiles
    of 4 x 16 pixels cut by the image), several channel blocks, more slices than tiles"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, Cin, Cout, H, W, fused = case
    g = torch.Generator().manual_seed(17 + H)
    x = torch.randn((N, Cin, H, W), generator=g)
    dy = torch.randn((N, Cout, H, W), generator=g)
    if fused:
        sc, sh = torch.rand((Cin,), generator=g) + 0.5, torch.randn((Cin,), generator=g) * 0.3
        t = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        src = engine.Src(_nhwc(x), sc.cuda(), sh.cuda(), relu=True)
    else:
        t, src = x, engine.Src(_nhwc(x))
    want = torch.nn.grad.conv2d_weight(t.double(), (Cout, Cin, 1, 1), dy.double())
    got = _wgrad([src], _nhwc(dy), (Cout, Cin, 1, 1), 'conv1', H, W)
    assert _rel(got, want) < 5e-5, _rel(got, want)


def test_wgrad_fp32_transformed_sources_1x1_and_transposed():
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(5)
    # two concat sources: affine+relu, and padded affine+residual+relu
    N, Ca, Cb, Cout, H, W = 2, 32, 64, 64, 20, 28
    a = torch.randn((N, Ca, H, W), generator=g)
    b = torch.randn((N, Cb, H - 1, W - 3), generator=g)
    res = torch.randn((N, Cb, H - 1, W - 3), generator=g)
    sa, ha = torch.rand((Ca,), generator=g) + 0.5, torch.randn((Ca,), generator=g) * 0.3
    sb, hb = torch.rand((Cb,), generator=g) + 0.5, torch.randn((Cb,), generator=g) * 0.3
    dy = torch.randn((N, Cout, H, W), generator=g)
    ta = F.relu(a * sa.view(1, -1, 1, 1) + ha.view(1, -1, 1, 1))
    tb = F.pad(F.relu(b * sb.view(1, -1, 1, 1) + hb.view(1, -1, 1, 1) + res), (1, 2, 0, 1))
    want = torch.nn.grad.conv2d_weight(torch.cat([ta, tb], 1).double(), (Cout, Ca + Cb, 3, 3), dy.double(), padding=1)
    srcs = [engine.Src(_nhwc(a), sa.cuda(), ha.cuda(), relu=True),
            engine.Src(_nhwc(b), sb.cuda(), hb.cuda(), relu=True, res=_nhwc(res), off=(0, 1))]
    got = _wgrad(srcs, _nhwc(dy), (Cout, Ca + Cb, 3, 3), 'conv3', H, W)
    assert _rel(got, want) < 5e-5, _rel(got, want)
    # stem: 3 real channels stored as 16
    x = torch.rand((2, 3, 32, 32), generator=g)
    dy = torch.randn((2, 64, 32, 32), generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 3, 3), dy.double(), padding=1)
    x16 = torch.zeros((2, 16, 32, 32)); x16[:, :3] = x
    got = _wgrad([engine.Src(_nhwc(x16))], _nhwc(dy), (64, 3, 3, 3), 'conv3', 32, 32, cin_real=3)
    assert _rel(got, want) < 5e-5
    # 1x1
    x = torch.randn((2, 64, 24, 16), generator=g)
    dy = torch.randn((2, 64, 24, 16), generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (64, 64, 1, 1), dy.double())
    got = _wgrad([engine.Src(_nhwc(x))], _nhwc(dy), (64, 64, 1, 1), 'conv1', 24, 16)
    assert _rel(got, want) < 5e-5
    # transposed k4 s2 p1 and k2 s2
    for (N, Cin, Cout, H, W, k) in [(2, 64, 32, 8, 8, 4), (1, 32, 16, 24, 40, 4), (2, 64, 32, 12, 20, 2)]:
        x = torch.randn((N, Cin, H, W), generator=g).double()
        w = torch.zeros((Cin, Cout, k, k), dtype=torch.float64, requires_grad=True)
        dy = torch.randn((N, Cout, 2 * H, 2 * W), generator=g)
        F.conv_transpose2d(x, w, None, stride=2, padding=1 if k == 4 else 0).backward(dy.double())
        got = _wgrad([engine.Src(_nhwc(x))], _nhwc(dy), (Cin, Cout, k, k), 'convT4' if k == 4 else 'convT2', H, W)
        assert _rel(got, w.grad) < 5e-5, (k, _rel(got, w.grad))


def test_backward_data_fp32():
    """input gradients: 3x3 / 1x1 with the flipped pack (mode 1), transposed convolutions through the space-to-depth view"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(9)
    for (N, Cin, Cout, H, W, k) in [(2, 64, 64, 24, 20, 3), (1, 128, 32, 16, 16, 3), (2, 64, 64, 12, 12, 1)]:
        x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64).requires_grad_(True)
        w = torch.randn((Cout, Cin, k, k), generator=g) * 0.1
        dy = torch.randn((N, Cout, H, W), generator=g)
        F.conv2d(x, w.double(), padding=k // 2).backward(dy.double())
        cfg = engine.choose_cfg([Cout], Cin, H, W, taps=k * k, N=N, f32=True)
        wp = engine.pack_weights(w.cuda(), cfg, 1, split=True)
        out, _ = engine.conv_forward([engine.Src(_nhwc(dy))], wp, Cin, cfg, taps=k * k, H=H, W=W)
        assert _rel(_nchw(out), x.grad) < 5e-5, (k, _rel(_nchw(out), x.grad))
    for (N, Cin, Cout, H, W, k) in [(2, 64, 32, 8, 8, 4), (1, 32, 16, 24, 40, 4), (2, 64, 32, 12, 20, 2)]:
        x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64).requires_grad_(True)
        w = torch.randn((Cin, Cout, k, k), generator=g) * 0.1
        dy = torch.randn((N, Cout, 2 * H, 2 * W), generator=g)
        F.conv_transpose2d(x, w.double(), None, stride=2, padding=1 if k == 4 else 0).backward(dy.double())
        cfg = engine.choose_cfg([2 * Cout, 2 * Cout], Cin, H, W, taps=9 if k == 4 else 1, N=N, f32=True)
        wp = engine.pack_weights(w.cuda(), cfg, 4 if k == 4 else 5, split=True)
        gy = _nhwc(dy)
        views = [engine.Src(gy, view=(a * 2 * W * Cout, H, W, 2 * Cout, 4 * W * Cout)) for a in (0, 1)]
        out, _ = engine.conv_forward(views, wp, Cin, cfg, taps=9 if k == 4 else 1, H=H, W=W)
        assert _rel(_nchw(out), x.grad) < 5e-5, (k, _rel(_nchw(out), x.grad))


def _bn_case(pooled, with_res, two_grads, seed, outmask=False, skip=0, size=(12, 20), pool_first=True):
    import torch
    import torch.nn.functional as F
    from cdnet_amd import _lib, trainer
    g = torch.Generator().manual_seed(seed)
    N, Cc, (H, W) = 2, 32, size
    raw = torch.randn((N, Cc, H, W), generator=g).requires_grad_(True)
    res = torch.randn((N, Cc, H, W), generator=g).requires_grad_(True) if with_res else None
    gamma = (torch.rand((Cc,), generator=g) + 0.5).requires_grad_(True)
    gamma.data[::4] *= -1
    beta = (torch.randn((Cc,), generator=g) * 0.2).requires_grad_(True)
    mean = raw.detach().mean((0, 2, 3))
    var = raw.detach().var((0, 2, 3), unbiased=False)
    y = F.batch_norm(raw, None, None, gamma, beta, training=True, eps=1e-5)
    if with_res:
        y = y + res
    a = F.relu(y)
    total = 0
    gins = []
    if pooled:
        p = F.max_pool2d(a, 2)
        gp = torch.randn(p.shape, generator=g)
        total = total + (p * gp).sum()
        gins.append(trainer._G(_nhwc(gp), p.shape[2], p.shape[3], pooled=1))
    for _ in range(skip):
        # a same-size consumer that reads the activation as a channel slice of a wider tensor (the decoder's torch.cat with the skip)
        gwide = torch.randn((N, Cc + 16, H, W), generator=g)
        total = total + (a * gwide[:, 16:16 + Cc]).sum()
        gins.append(trainer._G(_nhwc(gwide), H, W, coff=16, cstride=Cc + 16))
    if not pool_first:
        gins = gins[1:] + gins[:1]
    if (two_grads or not pooled) and not skip:
        if outmask:
            gfull = torch.randn((N, Cc, H, W), generator=g)
            total = total + (a * gfull).sum()
            gins.append(trainer._G(_nhwc(gfull), H, W))
        else:
            ap = F.pad(a, (2, 1, 1, 0))
            gfull = torch.randn((N, Cc + 16, H + 1, W + 3), generator=g)
            total = total + (ap * gfull[:, 8:8 + Cc]).sum()
            gins.append(trainer._G(_nhwc(gfull), H + 1, W + 3, oy=1, ox=2, coff=8, cstride=Cc + 16))
    total.backward()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = (gamma.detach() * invstd)
    shift = beta.detach() - mean * scale
    A = trainer.BnBwdArgs()
    raw_d = _nhwc(raw.detach())
    # relu = 2: `res` is the stored post-ReLU output of the unit (the fused residual epilogue), the mask is read from it
    res_d = (_nhwc(a.detach()) if outmask else _nhwc(res.detach())) if with_res else None
    A.raw, A.res = raw_d.data_ptr(), (res_d.data_ptr() if with_res else None)
    dev = lambda t: t.detach().float().cuda().contiguous()
    sc, sh, mu, iv, gm = dev(scale), dev(shift), dev(mean), dev(invstd), dev(gamma)
    A.scale, A.shift, A.mean, A.invstd = sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), iv.data_ptr()
    A.ngin = len(gins)
    for k, gi in enumerate(gins):
        A.gin[k].g = gi.t.data_ptr()
        A.gin[k].Hg, A.gin[k].Wg, A.gin[k].oy, A.gin[k].ox = gi.Hg, gi.Wg, gi.oy, gi.ox
        A.gin[k].pooled, A.gin[k].coff, A.gin[k].cstride = gi.pooled, gi.coff, gi.cstride or Cc
    A.f16, A.relu, A.N, A.H, A.W, A.C = 2, (2 if outmask else 1), N, H, W, Cc
    ws = torch.empty((_lib.load().cdnet_bn_backward_workspace_floats(Cc),), dtype=torch.float32, device='cuda')
    dgamma, dbeta = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    draw = torch.empty((N, H, W, Cc), dtype=torch.float32, device='cuda')
    dz = torch.empty((N, H, W, Cc), dtype=torch.float32, device='cuda')
    _lib.call('cdnet_bn_backward', C.byref(A), _lib.ptr(gm), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), ws.numel(),
              _lib.ptr(draw), _lib.ptr(dz) if with_res else None, _lib.stream_ptr())
    assert _rel(_nchw(draw), raw.grad) < 1e-5, ('draw', _rel(_nchw(draw), raw.grad))
    assert _rel(dgamma.cpu(), gamma.grad) < 1e-5 and _rel(dbeta.cpu(), beta.grad) < 1e-5
    if with_res:
        assert _rel(_nchw(dz), res.grad) < 1e-5


@pytest.mark.parametrize('cfg', [(False, False, False), (True, False, False), (True, False, True), (False, True, False),
                                 (True, True, True)])
def test_bn_backward_fp32(cfg):
    _bn_case(*cfg, seed=11)


@pytest.mark.parametrize('cfg', [(1, (12, 20), True), (1, (13, 21), False), (2, (10, 18), False), (0, (9, 7), True)])
def test_bn_backward_fp32_pool_window(cfg):
    """the encoder's layers in front of a max-pool (one pooled consumer + the decoder's skip slice): bn_bwd_window32_kernel, odd sizes
    (a last window row / column without a pooled gradient), the pooled source first or last in the argument order"""
    skip, size, pool_first = cfg
    _bn_case(True, False, False, seed=13, skip=skip, size=size, pool_first=pool_first)


def test_bn_backward_fp32_mask_from_stored_output():
    _bn_case(False, True, False, seed=12, outmask=True)


def test_head_forward_backward_fp32():
    """DAM head on fp32 features vs PyTorch autograd (float64)"""
    import torch
    from cdnet_amd import _lib, runtime, engine
    from oracle import models as om
    torch.manual_seed(3)
    ref = om.Unet().double()
    N, H, W = 2, 24, 20
    f = [torch.randn((N, 64, H, W), dtype=torch.float64).requires_grad_(True) for _ in range(3)]
    # model_unet_rev1.py:258-263
    x_point = ref.point_conv(f[2])
    x_dir = ref.direction_conv(ref.directionAtt(f[1], x_point))
    x_mask = ref.mask_conv(ref.maskAtt(f[0], x_dir))
    gm, gp, gd = torch.randn(x_mask.shape, dtype=torch.float64), torch.randn(x_point.shape, dtype=torch.float64), torch.randn(x_dir.shape, dtype=torch.float64)
    ((x_mask * gm).sum() + (x_point * gp).sum() + (x_dir * gd).sum()).backward()
    ps = [ref.point_conv.weight, ref.direction_conv.weight, ref.mask_conv.weight, ref.point_conv.bias, ref.direction_conv.bias,
          ref.mask_conv.bias, ref.directionAtt.Conv1x1.weight, ref.maskAtt.Conv1x1.weight]
    hw = torch.cat([p.detach().reshape(-1).float() for p in ps]).cuda().contiguous()
    feats = [engine.Src(_nhwc(t.detach())) for t in f]
    hf = [runtime.head_feat(s) for s in feats]
    mask = torch.empty((N, 3, H, W), device='cuda'); point = torch.empty((N, 1, H, W), device='cuda'); dirn = torch.empty((N, 9, H, W), device='cuda')
    _lib.call('cdnet_dam_head_forward', C.byref(hf[0]), C.byref(hf[1]), C.byref(hf[2]), _lib.ptr(hw), N, H, W, _lib.ptr(mask), _lib.ptr(point),
              _lib.ptr(dirn), _lib.stream_ptr())
    assert _rel(mask.cpu(), x_mask.detach()) < 1e-5 and _rel(point.cpu(), x_point.detach()) < 1e-5 and _rel(dirn.cpu(), x_dir.detach()) < 1e-5
    df = [torch.empty((N, H, W, 64), dtype=torch.float32, device='cuda') for _ in range(3)]
    need = _lib.load().cdnet_dam_head_backward_workspace_floats(N, H, W)
    ws = torch.empty((need,), device='cuda')
    dhw = torch.zeros((855,), device='cuda')
    dm, dp, dd = [t.float().cuda().contiguous() for t in (gm, gp, gd)]
    _lib.call('cdnet_dam_head_backward', C.byref(hf[0]), C.byref(hf[1]), C.byref(hf[2]), _lib.ptr(hw), _lib.ptr(dm), _lib.ptr(dp), _lib.ptr(dd),
              N, H, W, _lib.ptr(df[0]), _lib.ptr(df[1]), _lib.ptr(df[2]), _lib.ptr(ws), ws.numel(), _lib.ptr(dhw), _lib.stream_ptr())
    for k in range(3):
        assert _rel(_nchw(df[k]), f[k].grad) < 1e-5, (k, _rel(_nchw(df[k]), f[k].grad))
    want = torch.cat([p.grad.reshape(-1) for p in ps])
    assert _rel(dhw.cpu(), want) < 1e-5


@pytest.mark.parametrize('shape', [(2, 24, 20), (1, 16, 16), (3, 7, 9)])
def test_head_forward_plain_bf16_features_on_the_matrix_cores(shape):
    """eval-mode head over plain bf16 features (dam_head_mfma_kernel: weights as hi + lo bf16 pairs on `v_mfma_f32_16x16x32_bf16`) vs the
    oracle's head in float64 over the same bf16-representable features: 2e-5 of the logit scale; the same with the point logits given
    instead of the third feature (f3.raw = NULL, cdnet_conv_args.dot_out's consumer).  Ragged pixel counts cover the clamped last group."""
    import torch
    from cdnet_amd import _lib, runtime, engine
    from oracle import models as om
    torch.manual_seed(5)
    ref = om.Unet().double()
    N, H, W = shape
    f = [torch.randn((N, 64, H, W)).to(torch.bfloat16).double() for _ in range(3)]
    x_point = ref.point_conv(f[2])
    x_dir = ref.direction_conv(ref.directionAtt(f[1], x_point))
    x_mask = ref.mask_conv(ref.maskAtt(f[0], x_dir))
    ps = [ref.point_conv.weight, ref.direction_conv.weight, ref.mask_conv.weight, ref.point_conv.bias, ref.direction_conv.bias,
          ref.mask_conv.bias, ref.directionAtt.Conv1x1.weight, ref.maskAtt.Conv1x1.weight]
    hw = torch.cat([p.detach().reshape(-1).float() for p in ps]).cuda().contiguous()
    feats = [engine.Src(t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()) for t in f]
    hf = [runtime.head_feat(s) for s in feats]
    for given in (False, True):
        mask = torch.full((N, 3, H, W), 7.0, device='cuda'); dirn = torch.full((N, 9, H, W), 7.0, device='cuda')
        point = x_point.detach().float().cuda().contiguous() if given else torch.full((N, 1, H, W), 7.0, device='cuda')
        h3 = runtime.HeadFeat() if given else hf[2]
        _lib.call('cdnet_dam_head_forward', C.byref(hf[0]), C.byref(hf[1]), C.byref(h3), _lib.ptr(hw), N, H, W, _lib.ptr(mask), _lib.ptr(point),
                  _lib.ptr(dirn), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert _rel(mask.cpu(), x_mask.detach()) < 2e-5 and _rel(point.cpu(), x_point.detach()) < 2e-5 and _rel(dirn.cpu(), x_dir.detach()) < 2e-5, \
            (given, _rel(mask.cpu(), x_mask.detach()), _rel(point.cpu(), x_point.detach()), _rel(dirn.cpu(), x_dir.detach()))


@pytest.mark.parametrize('eres', [False, True])
def test_conv_fp32_full_tile_epilogue_is_bit_identical_to_the_general_one(eres):
    """conv_f32_kernel's full-tile epilogue (one base pointer, 32-bit offsets, statistics without bounds tests) against the general
    per-element one (cdnet_conv_args.debug bit 16): the same arithmetic, so outputs and per-tile statistics agree bit for bit"""
    import torch
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(5)
    N, Cin, Cout, H, W, cfg = 2, 64, 64, 48, 64, (16, 16, 64)
    x = torch.randn((N, Cin, H, W), generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * 0.1
    wp = engine.pack_weights(w.cuda(), cfg, 0, split=True)
    src = engine.Src(_nhwc(x), sc.cuda(), sh.cuda(), relu=True)
    kw = {}
    if eres:
        kw = dict(oscale=(torch.rand(Cout, generator=g) + 0.5).cuda(), oshift=(torch.randn(Cout, generator=g) * 0.1).cuda(),
                  eres=engine.Src(_nhwc(torch.randn((N, Cout, H, W), generator=g)), relu=True))
    outs = []
    for dbg in (16, 0):
        engine.CONV_DEBUG = dbg
        try:
            out = torch.empty((N, H, W, Cout), dtype=torch.float32, device='cuda')
            o, st = engine.conv_forward([src], wp, Cout, cfg, taps=9, out=out, stats=None if eres else True, **kw)
            torch.cuda.synchronize()
        finally:
            engine.CONV_DEBUG = 0
        outs.append((o.clone(), None if st is None else st.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    if not eres:
        assert torch.equal(outs[0][1], outs[1][1])


# ----------------------------------------------------------------------------------------------------------------------------------
# conv_ws32_kernel (csrc/conv32ws.hip): the wave-specialised persistent form of the fp32-precision 3x3 convolution.  Same MFMA
# sequence per accumulator as conv_f32_kernel -> bit-identical outputs; the per-tile statistics are summed in another order (1e-6).
# Every case runs with few persistent workgroups (debug >> 8), so that a workgroup walks several tiles (odd and even numbers of
# chunk intervals, the deferred two-half epilogue, the serial epilogue of the last tile), and with the default grid.
# ----------------------------------------------------------------------------------------------------------------------------------
def _ws32_case(N, cins, Cout, H, W, xf, stats=False, fold=False, offs=None, res=False, coff=0, cstride=None, seed=0, bn=None):
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(100 + seed)
    srcs, parts = [], []
    for k, c in enumerate(cins):
        hs, ws = (H, W) if not offs or k == 0 else (H - offs[0] - 1, W - offs[1] - 2)
        x = torch.randn((N, c, hs, ws), generator=g)
        sc = sh = r = None
        t = x
        if xf >= 1:
            sc, sh = torch.rand((c,), generator=g) + 0.5, torch.randn((c,), generator=g) * 0.3
            t = t * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        if res and k == len(cins) - 1:
            r = torch.randn((N, c, hs, ws), generator=g)
            t = t + r
        if xf >= 1:
            t = F.relu(t)
        off = (0, 0)
        if offs and k > 0:
            off = offs
            t = F.pad(t, (offs[1], W - ws - offs[1], offs[0], H - hs - offs[0]))
        parts.append(t)
        srcs.append(engine.Src(_nhwc(x), None if sc is None else sc.cuda(), None if sh is None else sh.cuda(), relu=xf >= 1,
                               res=None if r is None else _nhwc(r), off=off))
    cin = sum(cins)
    w = torch.randn((Cout, cin, 3, 3), generator=g) * (1.5 / (9 * cin) ** 0.5)
    bias = torch.randn((Cout,), generator=g) * 0.1 if fold else None
    osc = (torch.rand((Cout,), generator=g) + 0.5) if fold else None
    osh = torch.randn((Cout,), generator=g) * 0.2 if fold else None
    want = F.conv2d(torch.cat(parts, 1).double(), w.double(), None if bias is None else bias.double(), padding=1)
    raw = want
    if fold:
        want = F.relu(want * osc.double().view(1, -1, 1, 1) + osh.double().view(1, -1, 1, 1))
    cfg = (16, 16, bn or (64 if Cout > 32 else 32))         # (choose_cfg would take 8x8 tiles for the few-pixel cases)
    wp = engine.pack_weights(w.cuda(), cfg, 0, split=True)
    outs = {}
    for name, dbg in (('old', 32), ('ws_few', 64 | (3 << 8)), ('ws_five', 64 | (5 << 8)), ('ws', 64)):
        engine.CONV_DEBUG = dbg
        try:
            cs = cstride or Cout
            out = torch.full((N, H, W, cs), 7.0, dtype=torch.float32, device='cuda')
            if cs != Cout:
                # channel slice of a wider tensor: call the ABI with out_cstride / out_coff through a view-less path
                import ctypes as C
                from cdnet_amd import _lib
                a = engine.ConvArgs()
                for i, s in enumerate(srcs):
                    s.fill(a.src[i])
                a.nsrc, a.w = len(srcs), wp.data_ptr()
                a.out, a.Cout, a.out_cstride, a.out_coff = out.data_ptr(), Cout, cs, coff
                a.N, a.H, a.W, a.taps, a.npar, a.ostride, a.nchunk = N, H, W, 9, 1, 1, cin // 16
                a.tile, a.CK, a.BN, a.f32, a.debug = cfg[0], cfg[1], cfg[2], 1, dbg
                _lib.call('cdnet_conv_forward', C.byref(a), _lib.stream_ptr())
                st = None
                got = out[..., coff:coff + Cout]
                assert float((out[..., :coff] - 7.0).abs().max() if coff else 0.0) == 0.0 and float((out[..., coff + Cout:] - 7.0).abs().max()) == 0.0
            else:
                got, st = engine.conv_forward(srcs, wp, Cout, cfg, bias=None if bias is None else bias.cuda(), oscale=None if osc is None else osc.cuda(),
                                              oshift=None if osh is None else osh.cuda(), orelu=fold, out=out, stats=True if stats else None, H=H, W=W)
            torch.cuda.synchronize()
            outs[name] = (got.clone(), None if st is None else st.clone())
        finally:
            engine.CONV_DEBUG = 0
    ref = want.permute(0, 2, 3, 1).contiguous()
    for name, (got, st) in outs.items():
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 5e-5, (name, err)
        if name != 'old':
            assert torch.equal(got, outs['old'][0]), '%s differs from conv_f32_kernel' % name       # the same MFMA sequence per accumulator
        if stats:
            T = st.shape[0]
            rr = raw.permute(0, 2, 3, 1).reshape(N, H // 16, 16, W // 16, 16, Cout).permute(0, 1, 3, 2, 4, 5).reshape(T, 256, Cout)
            assert float((st[:, 0].double().cpu() - rr.sum(1)).abs().max()) < 2e-4 * float(rr.abs().sum(1).max())
            assert float((st[:, 1].double().cpu() - (rr * rr).sum(1)).abs().max()) < 2e-4 * float((rr * rr).sum(1).max())


@pytest.mark.parametrize('case', [
    dict(N=2, cins=(64,), Cout=64, H=32, W=48, xf=0),                                   # plain source (stored activations, gradients)
    dict(N=2, cins=(64,), Cout=64, H=32, W=48, xf=1, stats=True),                       # training: BatchNorm source, raw output + statistics
    dict(N=1, cins=(64,), Cout=64, H=16, W=16, xf=0),                                   # one tile
    dict(N=3, cins=(128,), Cout=128, H=32, W=32, xf=1, fold=True),                      # two output-channel tiles, folded BatchNorm + ReLU
    dict(N=2, cins=(64, 16), Cout=16, H=32, W=32, xf=1, stats=True, offs=(1, 2)),       # decoder: two sources, odd chunk count, pad offsets, 16 couts
    dict(N=2, cins=(64, 32), Cout=32, H=48, W=32, xf=1, stats=True),                    # six chunks, 32 couts
    dict(N=1, cins=(256,), Cout=256, H=32, W=32, xf=0),                                 # long channel loop
    dict(N=2, cins=(64,), Cout=64, H=32, W=32, xf=2, res=True),                         # residual operand (run-time transform flags)
    dict(N=2, cins=(64,), Cout=32, H=32, W=32, xf=0, coff=16, cstride=64),              # channel slice of a wider output tensor
    dict(N=2, cins=(16,), Cout=64, H=32, W=48, xf=0, stats=True),                       # ONE chunk per tile (the stem): the whole epilogue in one interval
    dict(N=3, cins=(16,), Cout=64, H=32, W=32, xf=1, stats=True),                       # one chunk, BatchNorm source
    dict(N=2, cins=(16,), Cout=80, H=32, W=32, xf=0),                                   # one chunk, two output-channel tiles, ragged couts (backward of 80 -> 16)
    dict(N=2, cins=(32,), Cout=64, H=32, W=32, xf=1, stats=True),                       # two chunks
    dict(N=2, cins=(32, 16), Cout=64, H=32, W=48, xf=1, fold=True, offs=(1, 2)),        # three chunks, two sources
    dict(N=1, cins=(48,), Cout=128, H=16, W=32, xf=0),                                  # three chunks, one source
    # the route of the timed workload's bottleneck (engine.choose_cfg, CDNET_F32_WS16: 16 x 16-pixel layers with >= 256 channels at
    # 8 <= N <= 32 take 32-cout tiles on conv_ws32_kernel): the 768 -> 256 decoder convolution (model_unet_rev1.py:119-143; two sources,
    # a 48-chunk loop) and a 512 -> 512 encoder layer (32 chunks), one image tile per image
    dict(N=16, cins=(256, 512), Cout=256, H=16, W=16, xf=1, stats=True, bn=32),
    dict(N=16, cins=(512,), Cout=512, H=16, W=16, xf=1, stats=True, bn=32),
    dict(N=8, cins=(512,), Cout=512, H=16, W=16, xf=0, bn=32),                          # its backward-data form (plain gradient source)
])
def test_conv_ws32_matches_conv_f32_and_fp64(case):
    _ws32_case(**case)


def test_conv_ws32_backward_data_of_transposed_conv_views():
    """the space-to-depth backward of ConvTranspose2d(k4, s2, p1): two strided view sources (row parities) on conv_ws32_kernel"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(77)
    N, Cin, Cout, H, W = 2, 64, 32, 16, 32
    x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.randn((Cin, Cout, 4, 4), generator=g) * 0.1
    dy = torch.randn((N, Cout, 2 * H, 2 * W), generator=g)
    F.conv_transpose2d(x, w.double(), None, stride=2, padding=1).backward(dy.double())
    cfg = engine.choose_cfg([2 * Cout, 2 * Cout], Cin, H, W, taps=9, N=N, f32=True)
    wp = engine.pack_weights(w.cuda(), cfg, 4, split=True)
    gy = _nhwc(dy)
    views = [engine.Src(gy, view=(a * 2 * W * Cout, H, W, 2 * Cout, 4 * W * Cout)) for a in (0, 1)]
    outs = []
    for dbg in (32, 64 | (2 << 8)):
        engine.CONV_DEBUG = dbg
        try:
            out, _ = engine.conv_forward(views, wp, Cin, cfg, taps=9, H=H, W=W)
        finally:
            engine.CONV_DEBUG = 0
        assert _rel(_nchw(out), x.grad) < 5e-5
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('case', [dict(N=2, Cin=64, Cout=64, H=32, W=48, G=3), dict(N=1, Cin=128, Cout=128, H=32, W=32, G=0),
                                  dict(N=3, Cin=64, Cout=64, H=16, W=16, G=2)])
def test_conv_ws32_bn_backward_statistics_epilogue(case):
    """cdnet_conv_args.ws = 2 on conv_ws32_kernel: a backward-data launch whose output dY is the gradient w.r.t. the activated output of
    a BatchNorm + ReLU layer also leaves that layer's first BatchNorm-backward pass - partial rows of sum(dz) and sum(dz * xhat) with
    dz = dY * [raw * scale + shift > 0] - for cdnet_bn_backward_finalize.  Output bit-identical to the plain launch; the sums against fp64."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, Cin, Cout, H, W, G = [case[k] for k in ('N', 'Cin', 'Cout', 'H', 'W', 'G')]
    g = torch.Generator().manual_seed(31 + H + Cout)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (1.5 / (9 * Cin) ** 0.5)
    raw = torch.randn((N, Cout, H, W), generator=g) * 1.5 + 0.3
    sc, sh = torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.4
    mu, inv = torch.randn((Cout,), generator=g) * 0.3, torch.rand((Cout,), generator=g) + 0.5
    cfg = (16, 16, 64)
    wp = engine.pack_weights(w.cuda(), cfg, 0, split=True)
    src = engine.Src(_nhwc(x))
    engine.CONV_DEBUG = 64
    try:
        plain, _ = engine.conv_forward([src], wp, Cout, cfg, H=H, W=W)
        engine.CONV_DEBUG = 64 | (G << 8)
        part = torch.zeros((1024, 2, Cout), dtype=torch.float32, device='cuda')
        raw_d = _nhwc(raw)
        bns = (raw_d, sc.cuda(), sh.cuda(), mu.cuda(), inv.cuda(), part)
        assert engine.conv_forward([src], wp, Cout, cfg, H=H, W=W, query_ws=True, bns=bns)
        out, _ = engine.conv_forward([src], wp, Cout, cfg, H=H, W=W, bns=bns)
        out2, _ = engine.conv_forward([src], wp, Cout, cfg, H=H, W=W, bns=bns)          # (every row is rewritten by every launch)
    finally:
        engine.CONV_DEBUG = 0
    torch.cuda.synchronize()
    assert torch.equal(out, plain) and torch.equal(out2, plain)
    dy = _nchw(out).double()
    act = raw.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    dz = dy * (act > 0)
    xhat = (raw.double() - mu.double().view(1, -1, 1, 1)) * inv.double().view(1, -1, 1, 1)
    want1, want2 = dz.sum((0, 2, 3)), (dz * xhat).sum((0, 2, 3))
    got = part.double().cpu().sum(0)
    scale1, scale2 = dz.abs().sum((0, 2, 3)).max(), (dz * xhat).abs().sum((0, 2, 3)).max()
    assert float((got[0] - want1).abs().max() / scale1) < 1e-5, float((got[0] - want1).abs().max() / scale1)
    assert float((got[1] - want2).abs().max() / scale2) < 1e-5, float((got[1] - want2).abs().max() / scale2)


@pytest.mark.parametrize('case', [dict(N=2, C1=64, H=32, W=48, G=3), dict(N=3, C1=16, H=32, W=32, G=0), dict(N=1, C1=64, H=16, W=16, G=1)])
def test_conv_ws32_one_tap_second_source(case):
    """cdnet_conv_args.taps1 = 1 on conv_ws32_kernel (fp32 mode): relu(conv3x3(h; w2 * scale) + conv1x1(x; w1) + shift) in one launch - a
    residual unit's second convolution with its BatchNorm scale folded into the hi | lo split pack and its 1x1 branch as one-tap chunks of a
    second source (model_unet_rev1.py:161-170, eval mode) - against the fp64 composition of the two convolutions, 5e-5 of the output scale."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, C1, H, W, G = [case[k] for k in ('N', 'C1', 'H', 'W', 'G')]
    g = torch.Generator().manual_seed(41 + C1)
    h = F.relu(torch.randn((N, 64, H, W), generator=g))
    x = F.relu(torch.randn((N, C1, H, W), generator=g))
    w2 = torch.randn((64, 64, 3, 3), generator=g) * (1.5 / (9 * 64) ** 0.5)
    w1 = torch.randn((64, C1, 1, 1), generator=g) * (1.5 / C1 ** 0.5)
    sc = torch.rand((64,), generator=g) + 0.5
    shift = torch.randn((64,), generator=g) * 0.3
    want = F.relu(F.conv2d(h.double(), w2.double() * sc.double().view(-1, 1, 1, 1), None, padding=1) + F.conv2d(x.double(), w1.double(), None) +
                  shift.double().view(1, -1, 1, 1))
    cfg = (16, 16, 64)
    wp = torch.cat([engine.pack_weights(w2.cuda(), cfg, 0, split=True, cout_scale=sc.cuda()), engine.pack_weights(w1.cuda(), cfg, 0, split=True)])
    srcs = [engine.Src(_nhwc(h)), engine.Src(_nhwc(x))]
    engine.CONV_DEBUG = 64 | (G << 8)
    try:
        assert engine.conv_forward(srcs, wp, 64, cfg, oshift=shift.cuda(), orelu=True, H=H, W=W, taps1=1, query_ws=True) == 1
        out, _ = engine.conv_forward(srcs, wp, 64, cfg, oshift=shift.cuda(), orelu=True, H=H, W=W, taps1=1)
        torch.cuda.synchronize()
        engine.CONV_DEBUG = 32                                # the one-tile kernels have no such form: the ABI says so instead of computing a 3x3
        with pytest.raises(RuntimeError):
            engine.conv_forward(srcs, wp, 64, cfg, oshift=shift.cuda(), orelu=True, H=H, W=W, taps1=1)
    finally:
        engine.CONV_DEBUG = 0
    assert _rel(_nchw(out), want) < 5e-5


@pytest.mark.parametrize('case', [dict(N=2, Cin=64, Cout=64, H=32, W=48, G=3, neg=False), dict(N=1, Cin=128, Cout=128, H=32, W=32, G=0, neg=True),
                                  dict(N=3, Cin=64, Cout=64, H=16, W=16, G=1, neg=True)])
def test_conv_ws32_fused_max_pool_output(case):
    """cdnet_conv_args.pool_out in fp32 mode: nn.MaxPool2d(2, 2) of relu(conv * scale + shift) from conv_ws32_kernel's epilogue (a lane
    holds whole 2x2 windows) - the 'M' layers of the VGG16-BN encoder (model_unet_rev1.py:40-41) without a cdnet_src_materialize pass.  The
    full-resolution output is bit-identical to the launch without it, the pooled tensor bit-identical to torch's max_pool2d of that output
    (a maximum of stored values: exact) - also with negative epilogue scales (the affine comes before the maximum); the one-tile kernels
    and launches with statistics say they do not serve it."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, Cin, Cout, H, W, G, neg = [case[k] for k in ('N', 'Cin', 'Cout', 'H', 'W', 'G', 'neg')]
    g = torch.Generator().manual_seed(23 + Cin + H)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (1.5 / (9 * Cin) ** 0.5)
    sc = torch.rand((Cout,), generator=g) + 0.5
    if neg:
        sc = sc * torch.where(torch.rand((Cout,), generator=g) < 0.3, -1.0, 1.0)
    sh = torch.randn((Cout,), generator=g) * 0.3
    cfg = (16, 16, 64)
    wp = engine.pack_weights(w.cuda(), cfg, 0, split=True)
    src = [engine.Src(_nhwc(x))]
    kw = dict(oscale=sc.cuda(), oshift=sh.cuda(), orelu=True, H=H, W=W)
    engine.CONV_DEBUG = 64 | (G << 8)
    try:
        plain, _ = engine.conv_forward(src, wp, Cout, cfg, **kw)
        out = torch.empty_like(plain)
        pout = torch.full((N, H // 2, W // 2, Cout), 7.0, dtype=torch.float32, device='cuda')
        assert engine.conv_forward(src, wp, Cout, cfg, out=out, pool_out=pout, query_ws=True, **kw) == 1
        engine.conv_forward(src, wp, Cout, cfg, out=out, pool_out=pout, **kw)
        torch.cuda.synchronize()
        st = torch.empty((N * (H // 16) * (W // 16), 2, Cout), dtype=torch.float32, device='cuda')
        assert engine.conv_forward(src, wp, Cout, cfg, out=out.clone(), pool_out=pout.clone(), query_ws=True, H=H, W=W, orelu=True, stats=st) == 0
        engine.CONV_DEBUG = 32
        with pytest.raises(RuntimeError):
            engine.conv_forward(src, wp, Cout, cfg, out=out.clone(), pool_out=pout.clone(), **kw)
    finally:
        engine.CONV_DEBUG = 0
    assert torch.equal(out, plain)
    assert torch.equal(_nchw(pout), F.max_pool2d(_nchw(plain), 2))
    want = F.relu(F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    assert _rel(_nchw(plain), want) < 5e-5
