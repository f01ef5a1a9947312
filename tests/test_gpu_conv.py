"""GPU numerics of the MFMA convolution kernels against a plain PyTorch fp32 (CPU) reference of the same op.
Inputs and weights are bf16-representable, accumulation is fp32, the output is rounded to bf16:
tolerance |got - want| <= 2^-7 * |want| + 2e-3 (one bf16 ulp + accumulation-order slack)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf(x):
    import torch
    return x.to(torch.bfloat16).to(torch.float32)


def _close(got, want, what=''):
    import torch
    err = (got - want).abs()
    tol = want.abs() * 2 ** -7 + 2e-3
    bad = err > tol
    assert not bool(bad.any()), '%s: %d bad, max err %g at want %g' % (
        what, int(bad.sum()), float(err.max()), float(want.flatten()[err.flatten().argmax()]))


def _nhwc(x, dtype=None):     # NCHW fp32 cpu -> NHWC bf16 (or given dtype) cuda
    import torch
    return x.permute(0, 2, 3, 1).contiguous().to(dtype or torch.bfloat16).cuda()


def _nchw(y):     # NHWC bf16 cuda -> NCHW fp32 cpu
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


CASES_3x3 = [
    # N, Cin, Cout, H, W, cfg
    (2, 32, 64, 32, 32, (16, 32, 64)),
    (1, 64, 64, 48, 40, (16, 32, 64)),        # ragged vs the 16x16 tile
    (2, 16, 64, 33, 17, (16, 16, 64)),
    (1, 64, 32, 32, 32, (16, 32, 32)),
    (1, 16, 16, 32, 48, (16, 16, 32)),        # Cout 16 padded to a 32-wide tile
    (1, 128, 128, 16, 16, (16, 32, 128)),
    (2, 64, 64, 16, 32, (16, 64, 64)),
    (2, 128, 256, 8, 8, (8, 32, 128)),
    (1, 64, 64, 8, 16, (8, 64, 64)),
    (1, 128, 64, 12, 20, (8, 32, 64)),
    (1, 64, 128, 8, 8, (8, 32, 128)),
]


@pytest.mark.parametrize('case', CASES_3x3)
def test_conv3x3_plain(case):
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, Cin, Cout, H, W, cfg = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = _bf(torch.randn((N, Cin, H, W), generator=g))
    w = _bf(torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5)
    b = torch.randn((Cout,), generator=g)
    want = F.conv2d(x, w, b, padding=1)
    wp = engine.pack_weights(w.cuda(), cfg, 0)
    out, stats = engine.conv_forward([engine.Src(_nhwc(x))], wp, Cout, cfg, bias=b.cuda(), stats=True)
    _close(_nchw(out), want, 'conv3x3')
    # per-tile statistics of the bias-free fp32 accumulators
    raw = F.conv2d(x, w, None, padding=1)
    s = stats.sum(0).cpu()
    np.testing.assert_allclose(s[0].numpy(), raw.sum((0, 2, 3)).numpy(), rtol=2e-4, atol=2e-2)
    np.testing.assert_allclose(s[1].numpy(), (raw * raw).sum((0, 2, 3)).numpy(), rtol=2e-4, atol=2e-2)


def test_conv3x3_fused_bn_relu_pool_concat_pad_epilogue():
    """every staging transform at once: source A = maxpool2x2(relu(bn(a))) ; source B = relu(bn(b) + res) with an
    F.pad offset; epilogue = folded eval BN + ReLU."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(5)
    N, Ca, Cb, Cout, H, W = 2, 32, 64, 64, 20, 28
    a = _bf(torch.randn((N, Ca, 2 * H, 2 * W), generator=g))
    b = _bf(torch.randn((N, Cb, H - 1, W - 3), generator=g))          # smaller: padded by (0,1) rows, (1,2) cols
    res = _bf(torch.randn((N, Cb, H - 1, W - 3), generator=g))
    sa, ha = torch.rand((Ca,), generator=g) + 0.5, torch.randn((Ca,), generator=g) * 0.3
    sa[::3] *= -1                                                      # negative gamma: pool must follow the affine
    sb, hb = torch.rand((Cb,), generator=g) + 0.5, torch.randn((Cb,), generator=g) * 0.3
    w = _bf(torch.randn((Cout, Ca + Cb, 3, 3), generator=g) * (2.0 / (9 * (Ca + Cb))) ** 0.5)
    osc, osh = torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.2
    bias = torch.randn((Cout,), generator=g) * 0.1
    ta = _bf(F.max_pool2d(_bf(F.relu(a * sa.view(1, -1, 1, 1) + ha.view(1, -1, 1, 1))), 2))
    tb = _bf(F.relu(b * sb.view(1, -1, 1, 1) + hb.view(1, -1, 1, 1) + res))
    tb = F.pad(tb, (1, 2, 0, 1))
    want = F.relu((F.conv2d(torch.cat([ta, tb], 1), w, bias, padding=1)) * osc.view(1, -1, 1, 1) + osh.view(1, -1, 1, 1))
    cfg = (16, 32, 64)
    wp = engine.pack_weights(w.cuda(), cfg, 0)
    srcs = [engine.Src(_nhwc(a), sa.cuda(), ha.cuda(), relu=True, pool=True),
            engine.Src(_nhwc(b), sb.cuda(), hb.cuda(), relu=True, res=_nhwc(res), off=(0, 1))]
    out, _ = engine.conv_forward(srcs, wp, Cout, cfg, bias=bias.cuda(), oscale=osc.cuda(), oshift=osh.cuda(),
                                 orelu=True, H=H, W=W)
    _close(_nchw(out), want, 'fused')


@pytest.mark.parametrize('case', [(2, 32, 64, 24, 24, (16, 32, 64)), (1, 16, 64, 40, 24, (16, 16, 64)),
                                  (2, 64, 128, 8, 8, (8, 32, 128))])
def test_conv1x1(case):
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, Cin, Cout, H, W, cfg = case
    g = torch.Generator().manual_seed(11)
    x = _bf(torch.randn((N, Cin, H, W), generator=g))
    w = _bf(torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5)
    b = torch.randn((Cout,), generator=g)
    want = F.conv2d(x, w, b)
    wp = engine.pack_weights(w.cuda(), cfg, 0)
    out, _ = engine.conv_forward([engine.Src(_nhwc(x))], wp, Cout, cfg, taps=1, bias=b.cuda())
    _close(_nchw(out), want, 'conv1x1')


@pytest.mark.parametrize('case', [(2, 64, 32, 8, 8, (8, 64, 64)), (1, 32, 16, 24, 40, (16, 32, 32)),
                                  (2, 128, 64, 16, 16, (16, 32, 64)), (1, 512, 256, 8, 8, (8, 32, 128))])
def test_conv_transpose_k4s2p1(case):
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, Cin, Cout, H, W, cfg = case
    g = torch.Generator().manual_seed(13)
    x = _bf(torch.randn((N, Cin, H, W), generator=g))
    w = _bf(torch.randn((Cin, Cout, 4, 4), generator=g) * (2.0 / (4 * Cin)) ** 0.5)
    want = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    wp = engine.pack_weights(w.cuda(), cfg, 2)
    out, stats = engine.conv_forward([engine.Src(_nhwc(x))], wp, Cout, cfg, taps=4, transposed=True, stats=True)
    assert tuple(out.shape) == (N, 2 * H, 2 * W, Cout)
    _close(_nchw(out), want, 'convT4')
    np.testing.assert_allclose(stats.sum(0)[0].cpu().numpy(), want.sum((0, 2, 3)).numpy(), rtol=2e-4, atol=3e-2)


def test_conv_transpose_k2s2():
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(17)
    N, Cin, Cout, H, W, cfg = 2, 64, 32, 12, 20, (16, 32, 32)
    x = _bf(torch.randn((N, Cin, H, W), generator=g))
    w = _bf(torch.randn((Cin, Cout, 2, 2), generator=g) * (2.0 / Cin) ** 0.5)
    b = torch.randn((Cout,), generator=g)
    want = F.conv_transpose2d(x, w, b, stride=2)
    wp = engine.pack_weights(w.cuda(), cfg, 3)
    out, _ = engine.conv_forward([engine.Src(_nhwc(x))], wp, Cout, cfg, taps=1, transposed=True, bias=b.cuda())
    _close(_nchw(out), want, 'convT2')


def test_conv_backward_data_pack():
    """backward-data of a 3x3 conv = forward conv with the flipped/transposed pack (mode 1)"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(19)
    N, Cin, Cout, H, W = 2, 32, 64, 24, 16
    w = _bf(torch.randn((Cout, Cin, 3, 3), generator=g) * 0.1)
    dy = _bf(torch.randn((N, Cout, H, W), generator=g))
    want = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy, padding=1)
    cfg = (16, 32, 32)
    wp = engine.pack_weights(w.cuda(), cfg, 1)
    out, _ = engine.conv_forward([engine.Src(_nhwc(dy))], wp, Cin, cfg)
    _close(_nchw(out), want, 'bwd-data')


@pytest.mark.parametrize('ceil_mode,hw', [(False, (64, 96)), (True, (37, 51))])
def test_materialized_pool_equals_on_the_fly_pool(ceil_mode, hw):
    """cdnet_src_materialize writes exactly what the convolution's staging code computes for a BatchNorm + ReLU + 2x2 max-pool
    source: the convolution over the stored copy is bit-identical to the one over the lazy view (and both match PyTorch)."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine, runtime
    g = torch.Generator().manual_seed(3)
    N, Cc, Cout = 2, 64, 32
    H, W = hw
    raw = (torch.randn((N, H, W, Cc), generator=g) * 2).half().cuda()
    sc = (torch.rand((Cc,), generator=g) + 0.5).cuda()
    sh = (torch.randn((Cc,), generator=g) * 0.3).cuda()
    w = (torch.randn((Cout, Cc, 3, 3), generator=g) * 0.05).cuda()
    src = engine.Src(raw, sc, sh, relu=True)
    lazy = runtime.pooled(src, ceil_mode, materialized=False)
    stored = runtime.pooled(src, ceil_mode, materialized=True)
    assert stored.grad_to[0] is raw and stored.x.dtype == torch.bfloat16 and stored.scale is None
    Hp, Wp = lazy.logical_hw()
    assert tuple(stored.x.shape) == (N, Hp, Wp, Cc)
    cfg = engine.choose_cfg([Cc], Cout, Hp, Wp)
    wp = engine.pack_weights(w, cfg, 0)
    a, _ = engine.conv_forward([lazy], wp, Cout, cfg)
    b, _ = engine.conv_forward([stored], wp, Cout, cfg)
    assert torch.equal(a, b)
    act = F.relu(raw.float() * sc + sh).permute(0, 3, 1, 2)
    want = F.max_pool2d(act, 2, 2, ceil_mode=ceil_mode)
    # (fused multiply-add here vs multiply + add in PyTorch: at most one bf16 ulp)
    np.testing.assert_allclose(stored.x.float().permute(0, 3, 1, 2).cpu().numpy(), want.cpu().numpy(), rtol=8e-3, atol=1e-6)


@pytest.mark.parametrize('affine', [True, False])
def test_fused_residual_epilogue_equals_conv_plus_materialize(affine):
    """cdnet_conv_args.eres (ResidualUnit: relu2(bn2(raw2) + conv_1x1(x)) in the epilogue of conv_1x1) is bit-identical to the
    separate path: conv_1x1 -> fp16 residual tensor, then cdnet_src_materialize over the (raw2, residual) pair"""
    import torch
    from cdnet_amd import engine, runtime
    g = torch.Generator().manual_seed(11)
    N, H, W, Cin, Cout = 2, 37, 53, 16, 64
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).cuda()
    raw2 = (torch.randn((N, H, W, Cout), generator=g) * 2).half().cuda()
    sc = (torch.rand((Cout,), generator=g) + 0.5).cuda() if affine else None
    sh = (torch.randn((Cout,), generator=g) * 0.3).cuda() if affine else None
    w = (torch.randn((Cout, Cin, 1, 1), generator=g) * 0.2).cuda()
    b = torch.randn((Cout,), generator=g).cuda()
    cfg = engine.choose_cfg([Cin], Cout, H, W, taps=1)
    wp = engine.pack_weights(w, cfg, 0)
    r, _ = engine.conv_forward([engine.Src(x)], wp, Cout, cfg, taps=1, bias=b, out_dtype=torch.float16)
    want = runtime.materialize(engine.Src(raw2, sc, sh, relu=True, res=r))
    got, _ = engine.conv_forward([engine.Src(x)], wp, Cout, cfg, taps=1, bias=b, eres=engine.Src(raw2, sc, sh, relu=True))
    assert got.dtype == torch.bfloat16 and torch.equal(got, want.x)
    ref = torch.relu((raw2.float() * sc + sh if affine else raw2.float()) + r.float())
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.cpu().numpy(), rtol=8e-3, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------
# conv_ws_kernel (wave-specialised, weight-stationary, persistent) == conv_fwd_kernel, bit for bit
# ---------------------------------------------------------------------------------------------------------
def _run_both(fn):
    """fn() launches convolutions through engine.conv_forward; returns its outputs once per kernel choice"""
    import torch
    from cdnet_amd import engine
    outs = []
    for dbg in (32, 64):                      # 32: conv_fwd_kernel only; 64: conv_ws_kernel even for small launches
        engine.CONV_DEBUG = dbg
        try:
            outs.append(fn())
            torch.cuda.synchronize()
        finally:
            engine.CONV_DEBUG = 0
    return outs


@pytest.mark.parametrize('case', [(2, 64, 64, 64, 64), (1, 64, 64, 40, 56), (3, 16, 64, 33, 17), (2, 32, 32, 48, 32), (2, 64, 16, 32, 48),
                                  (16, 64, 64, 256, 256),
                                  # streamed weights (more than four 16-channel chunks per tile): 128, 256, 192 input channels
                                  (2, 128, 64, 64, 64), (1, 256, 128, 32, 48), (2, 192, 32, 32, 32), (4, 512, 64, 16, 32)])
def test_ws_kernel_bit_identical_plain_stats_and_folded(case):
    import torch
    from cdnet_amd import engine
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(N + Cin + H)
    x = _nhwc(torch.randn((N, Cin, H, W), generator=g))
    raw = _nhwc(torch.randn((N, Cin, H, W), generator=g), torch.float16)
    sc, sh = (torch.rand((Cin,), generator=g) + 0.5).cuda(), (torch.randn((Cin,), generator=g) * 0.3).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
    b = torch.randn((Cout,), generator=g).cuda()
    osc, osh = (torch.rand((Cout,), generator=g) + 0.5).cuda(), (torch.randn((Cout,), generator=g) * 0.2).cuda()
    cfg = (16, 16, 64 if Cout > 32 else 32)
    wp = engine.pack_weights(w, cfg, 0)

    def run():
        o1, st = engine.conv_forward([engine.Src(x)], wp, Cout, cfg, bias=b, stats=True, out_dtype=torch.float16)          # training-mode raw + stats
        o2, _ = engine.conv_forward([engine.Src(raw, sc, sh, relu=True)], wp, Cout, cfg, oscale=osc, oshift=osh, orelu=True)   # lazily transformed source, folded epilogue
        return o1.clone(), st.clone(), o2.clone()
    a, bb = _run_both(run)
    for u, v in zip(a, bb):
        assert torch.equal(u, v)
    if case[0] == 16:
        return
    # and it is right: against PyTorch fp32
    import torch.nn.functional as F
    want = F.conv2d(_nchw(x), _bf(w.cpu()), b.cpu(), padding=1)
    _close(_nchw(bb[0]), want, 'ws plain')


@pytest.mark.parametrize('chans', [(32, 32), (64, 64), (64, 128)])
def test_ws_kernel_bit_identical_two_sources_residual_and_fused_epilogue(chans):
    """two lazily transformed sources (a padded small one with an offset, one with a residual branch): 32 + 32 channels = the resident
    form, 64 + 64 and 64 + 128 = the streamed-weights form; full 16x16 tiles so that the producer / consumer kernel takes the launch
    (its movers request by chunk pairs since round 5: every source an even number of 16-channel chunks)"""
    import torch
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 48, 64
    ca, cb = chans
    a = _nhwc(torch.randn((N, ca, H - 1, W - 2), generator=g), torch.float16)          # padded small source (decoder up branch)
    sa, ha = (torch.rand((ca,), generator=g) + 0.5).cuda(), (torch.randn((ca,), generator=g) * 0.3).cuda()
    bsrc = _nhwc(torch.randn((N, cb, H, W), generator=g), torch.float16)
    res = _nhwc(torch.randn((N, cb, H, W), generator=g), torch.float16)
    sb, hb = (torch.rand((cb,), generator=g) + 0.5).cuda(), (torch.randn((cb,), generator=g) * 0.3).cuda()
    w = (torch.randn((64, ca + cb, 3, 3), generator=g) * 0.05).cuda()
    cfg = (16, 16, 64)
    wp = engine.pack_weights(w, cfg, 0)
    e = _nhwc(torch.randn((N, 64, H, W), generator=g), torch.float16)
    esc, esh = (torch.rand((64,), generator=g) + 0.5).cuda(), (torch.randn((64,), generator=g) * 0.1).cuda()
    bias = torch.randn((64,), generator=g).cuda()

    def run():
        srcs = [engine.Src(a, sa, ha, relu=True, off=(1, 1)), engine.Src(bsrc, sb, hb, relu=True, res=res)]
        o1, st = engine.conv_forward(srcs, wp, 64, cfg, stats=True, H=H, W=W, out_dtype=torch.float16)
        o2, _ = engine.conv_forward(srcs, wp, 64, cfg, bias=bias, H=H, W=W, eres=engine.Src(e, esc, esh, relu=True))       # fused residual epilogue (conv_fwd_kernel in both runs)
        # the launch really is the producer / consumer kernel's
        assert engine.conv_forward(srcs, wp, 64, cfg, stats=True, H=H, W=W, out_dtype=torch.float16, query_ws=True) == (engine.CONV_DEBUG == 64)
        return o1.clone(), st.clone(), o2.clone()
    x, y = _run_both(run)
    for u, v in zip(x, y):
        assert torch.equal(u, v)


def test_ws_kernel_backward_data_view_sources():
    """backward-data of ConvTranspose2d(32 -> 16, k4 s2 p1) = 3x3 convolution over the two row-parity views of the gradient
    (row_stride sources): both kernels, bit for bit"""
    import torch
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(4)
    N, Cin, Cout, H, W = 2, 32, 16, 32, 48          # full 16x16 tiles: the launch is the producer / consumer kernel's
    w = (torch.randn((Cin, Cout, 4, 4), generator=g) * 0.1).cuda()
    dy = _nhwc(torch.randn((N, Cout, 2 * H, 2 * W), generator=g))
    cfg = (16, 16, 32)
    wp = engine.pack_weights(w, cfg, 4)

    def run():
        views = [engine.Src(dy, view=(a * 2 * W * Cout, H, W, 2 * Cout, 4 * W * Cout)) for a in (0, 1)]
        out, _ = engine.conv_forward(views, wp, Cin, cfg, taps=9, H=H, W=W)
        return (out.clone(),)
    x, y = _run_both(run)
    assert torch.equal(x[0], y[0])
