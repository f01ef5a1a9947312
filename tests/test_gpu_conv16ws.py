"""conv_ws16_kernel (csrc/conv16ws.hip): the persistent 3x3 convolution of the 16-bit path for launches without BatchNorm statistics -
swapped operand roles (weights = MFMA A operand), stores straight from the accumulators, any chunk count, resident or streamed weights,
an optional one-tap second source (a residual unit's 1x1 branch inside its second 3x3 convolution).

Every case runs on the one-tile kernel (debug 32: conv_fwd_kernel, the fp32-CPU-checked baseline of tests/test_gpu_conv.py), on
conv_ws16_kernel with a few persistent workgroups (debug >> 8: a workgroup walks several tiles - odd and even numbers of run chunks,
the deferred epilogue, the serial epilogue of the last tile) and with the default grid.  Without bias the accumulation order per
output is conv_fwd_kernel's: bit-identical.  With bias the sum starts from the bias (one rounding moved): one bf16 ulp.
Reference of every case: PyTorch fp32 on the CPU (F.conv2d over bf16-representable inputs), tolerance 2^-7 |want| + 2e-3."""
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


def _close(got, want, what=''):
    err = (got - want).abs()
    tol = want.abs() * 2 ** -7 + 2e-3
    bad = err > tol
    assert not bool(bad.any()), '%s: %d bad, max err %g at want %g' % (
        what, int(bad.sum()), float(err.max()), float(want.flatten()[err.flatten().argmax()]))


def _case(N, cins, Cout, H, W, xf=0, bias=False, relu=False, offs=None, res=False, coff=0, cstride=None, seed=0, grids=(3, 5, 0)):
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(300 + seed)
    srcs, parts = [], []
    for k, c in enumerate(cins):
        hs, ws = (H, W) if not offs or k == 0 else (H - offs[0] - 1, W - offs[1] - 2)
        x = _bf(torch.randn((N, c, hs, ws), generator=g))
        sc = sh = r = None
        t = x
        dt = torch.bfloat16
        if xf:                                   # a training-mode source: raw fp16 tensor, BatchNorm scale / shift, ReLU
            dt = torch.float16
            x = x.to(torch.float16).float()
            sc, sh = torch.rand((c,), generator=g) + 0.5, torch.randn((c,), generator=g) * 0.3
            t = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
            if res and k == len(cins) - 1:
                r = torch.randn((N, c, hs, ws), generator=g).to(torch.float16).float()
                t = t + r
            t = _bf(F.relu(t))
        off = (0, 0)
        if offs and k > 0:
            off = offs
            t = F.pad(t, (offs[1], W - ws - offs[1], offs[0], H - hs - offs[0]))
        parts.append(t)
        srcs.append(engine.Src(_nhwc(x, dt), None if sc is None else sc.cuda(), None if sh is None else sh.cuda(), relu=bool(xf),
                               res=None if r is None else _nhwc(r, dt), off=off))
    cin = sum(cins)
    w = _bf(torch.randn((Cout, cin, 3, 3), generator=g) * (1.5 / (9 * cin) ** 0.5))
    b = torch.randn((Cout,), generator=g) * 0.3 if bias else None
    want = F.conv2d(torch.cat(parts, 1), w, b, padding=1)
    if relu:
        want = F.relu(want)
    cfg = (16, 16, 64 if Cout > 32 else 32)
    wp = engine.pack_weights(w.cuda(), cfg, 0)
    outs = {}
    # (debug bit 16: the quad-request form of the movers, which production takes only for tensors beyond the Infinity Cache)
    for name, dbg in [('one-tile', 32)] + [('ws16_%d' % gr, 64 | (gr << 8)) for gr in grids] + [('ws16quad_%d' % grids[0], 64 | 16 | (grids[0] << 8))]:
        engine.CONV_DEBUG = dbg
        try:
            cs = cstride or Cout
            out = torch.full((N, H, W, cs), 7.0, dtype=torch.bfloat16, device='cuda')
            if name != 'one-tile':
                assert _eligible(engine, srcs, wp, Cout, cfg, H, W), 'not taken by conv_ws16_kernel'
            if cs != Cout:
                import ctypes as C
                from cdnet_amd import _lib
                a = engine.ConvArgs()
                for i, s in enumerate(srcs):
                    s.fill(a.src[i])
                a.nsrc, a.w, a.bias = len(srcs), wp.data_ptr(), (None if b is None else b.cuda().data_ptr())
                a.orelu = int(relu)
                a.out, a.Cout, a.out_cstride, a.out_coff = out.data_ptr(), Cout, cs, coff
                a.N, a.H, a.W, a.taps, a.npar, a.ostride, a.nchunk = N, H, W, 9, 1, 1, cin // 16
                a.tile, a.CK, a.BN, a.debug = cfg[0], cfg[1], cfg[2], dbg
                bkeep = None if b is None else b.cuda()
                if bkeep is not None:
                    a.bias = bkeep.data_ptr()
                _lib.call('cdnet_conv_forward', C.byref(a), _lib.stream_ptr())
                got = out[..., coff:coff + Cout]
                torch.cuda.synchronize()
                assert float((out[..., :coff].float() - 7.0).abs().max() if coff else 0.0) == 0.0
                assert float((out[..., coff + Cout:].float() - 7.0).abs().max()) == 0.0
            else:
                got, _ = engine.conv_forward(srcs, wp, Cout, cfg, bias=None if b is None else b.cuda(), orelu=relu, out=out, H=H, W=W)
            torch.cuda.synchronize()
            outs[name] = got.clone()
        finally:
            engine.CONV_DEBUG = 0
    for name, got in outs.items():
        _close(_nchw(got), want, name)
        if name != 'one-tile':
            if b is None and torch.equal(got, outs['one-tile']):
                pass                                                  # the 32x32x16 forms: the same sum order per output, bit-identical
            else:
                # with a bias the sum starts from it; the 16x16x32 consumers of the out-image form (even chunk counts >= 4) add 32 products per
                # MFMA instead of 16: another rounding sequence of the fp32 sums - one bf16 ulp at most
                d = (got.float() - outs['one-tile'].float()).abs()
                assert float((d / (outs['one-tile'].float().abs() + 1e-2)).max()) <= 2 ** -7, name
            assert torch.equal(got, outs['ws16_%d' % grids[0]]), '%s differs between grids' % name


def _eligible(engine, srcs, wp, Cout, cfg, H, W, **kw):
    return engine.conv_forward(srcs, wp, Cout, cfg, H=H, W=W, query_ws=True, **kw) == 2


@pytest.mark.parametrize('case', [
    dict(N=2, cins=(64,), Cout=64, H=32, W=48),                                          # the dominant layer's shape, plain source
    dict(N=2, cins=(64,), Cout=64, H=32, W=48, bias=True, relu=True),                    # eval-mode epilogue: shift in the accumulators' start, ReLU
    dict(N=1, cins=(64,), Cout=64, H=16, W=16),                                          # one tile
    dict(N=3, cins=(16,), Cout=64, H=32, W=32, bias=True, relu=True),                    # ONE chunk per tile (the stem): the whole epilogue in one step
    dict(N=2, cins=(32,), Cout=64, H=32, W=48),                                          # two chunks
    dict(N=1, cins=(48,), Cout=128, H=16, W=32),                                         # three chunks (odd: barriers fall inside tiles), two cout tiles
    dict(N=2, cins=(64, 16), Cout=16, H=32, W=32, offs=(1, 2), bias=True, relu=True),    # decoder 80 -> 16: five chunks, pad offsets, 16 couts
    dict(N=2, cins=(64, 32), Cout=32, H=48, W=32),                                       # 96 -> 32: six chunks, 32-cout tiles
    dict(N=2, cins=(16,), Cout=80, H=32, W=32),                                          # backward-data of 80 -> 16: ragged second cout tile
    dict(N=1, cins=(256,), Cout=256, H=32, W=32),                                        # streamed weights (16 chunks), four cout tiles
    dict(N=1, cins=(128, 64), Cout=64, H=32, W=32, bias=True, relu=True),                # streamed, two sources (decoder 192 -> 64)
    dict(N=2, cins=(64,), Cout=64, H=32, W=32, xf=1),                                    # run-time source transform: fp16 raw x scale + shift, ReLU
    dict(N=2, cins=(64,), Cout=64, H=32, W=32, xf=1, res=True),                          # ... with a residual operand
    dict(N=2, cins=(64,), Cout=32, H=32, W=32, coff=16, cstride=64),                     # channel slice of a wider output tensor
    dict(N=2, cins=(160,), Cout=32, H=32, W=32),                                         # ten chunks of 32-cout weights: resident
])
def test_conv_ws16_matches_conv_fwd_and_fp32(case):
    _case(**case)


def test_conv_ws16_backward_data_of_transposed_conv_views():
    """the space-to-depth backward of ConvTranspose2d(k4, s2, p1): two strided view sources (row parities) on conv_ws16_kernel"""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    g = torch.Generator().manual_seed(77)
    N, Cin, Cout, H, W = 2, 64, 32, 16, 32
    x = _bf(torch.randn((N, Cin, H, W), generator=g)).requires_grad_(True)
    w = _bf(torch.randn((Cin, Cout, 4, 4), generator=g) * 0.1)
    dy = _bf(torch.randn((N, Cout, 2 * H, 2 * W), generator=g))
    F.conv_transpose2d(x, w, None, stride=2, padding=1).backward(dy)
    cfg = (16, 16, 64)
    wp = engine.pack_weights(w.cuda(), cfg, 4)
    gy = _nhwc(dy)
    views = [engine.Src(gy, view=(a * 2 * W * Cout, H, W, 2 * Cout, 4 * W * Cout)) for a in (0, 1)]
    outs = []
    for dbg in (32, 64 | (2 << 8), 64):
        engine.CONV_DEBUG = dbg
        try:
            if dbg != 32:
                assert engine.conv_forward(views, wp, Cin, cfg, taps=9, H=H, W=W, query_ws=True) == 2
            out, _ = engine.conv_forward(views, wp, Cin, cfg, taps=9, H=H, W=W)
        finally:
            engine.CONV_DEBUG = 0
        _close(_nchw(out), x.grad, 'convT backward')
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize('case', [dict(N=2, C1=64, H=32, W=48, G=3), dict(N=3, C1=16, H=32, W=32, G=0), dict(N=1, C1=64, H=16, W=16, G=1)])
def test_conv_ws16_one_tap_second_source(case):
    """cdnet_conv_args.taps1 = 1: relu(conv3x3(h; w2 * scale) + conv1x1(x; w1) + shift) in one launch - a residual unit's second
    convolution with its BatchNorm folded and its 1x1 branch as extra K steps (model_unet_rev1.py:161-170, eval mode) - against the
    PyTorch fp32 composition of the two convolutions."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, C1, H, W, G = [case[k] for k in ('N', 'C1', 'H', 'W', 'G')]
    g = torch.Generator().manual_seed(41 + C1)
    h = _bf(F.relu(torch.randn((N, 64, H, W), generator=g)))
    x = _bf(F.relu(torch.randn((N, C1, H, W), generator=g)))
    w2 = _bf(torch.randn((64, 64, 3, 3), generator=g) * (1.5 / (9 * 64) ** 0.5))
    w1 = _bf(torch.randn((64, C1, 1, 1), generator=g) * (1.5 / C1 ** 0.5))
    shift = torch.randn((64,), generator=g) * 0.3
    want = F.relu(F.conv2d(h, w2, None, padding=1) + F.conv2d(x, w1, None) + shift.view(1, -1, 1, 1))
    cfg = (16, 16, 64)
    wp = torch.cat([engine.pack_weights(w2.cuda(), cfg, 0), engine.pack_weights(w1.cuda(), cfg, 0)])
    srcs = [engine.Src(_nhwc(h)), engine.Src(_nhwc(x))]
    engine.CONV_DEBUG = 64 | (G << 8)
    try:
        assert engine.conv_forward(srcs, wp, 64, cfg, oshift=shift.cuda(), orelu=True, H=H, W=W, taps1=1, query_ws=True) == 2
        out, _ = engine.conv_forward(srcs, wp, 64, cfg, oshift=shift.cuda(), orelu=True, H=H, W=W, taps1=1)
    finally:
        engine.CONV_DEBUG = 0
    _close(_nchw(out), want, 'fused residual unit')


def test_launches_outside_its_scope_fall_back():
    """statistics, an epilogue scale, fp16 outputs, ragged sizes: not conv_ws16_kernel's (cdnet_conv_forward takes the older kernels)"""
    import torch
    from cdnet_amd import engine
    x = engine.Src(torch.zeros((2, 32, 32, 64), dtype=torch.bfloat16, device='cuda'))
    cfg = (16, 16, 64)
    wp = engine.pack_weights(torch.zeros((64, 64, 3, 3), device='cuda'), cfg, 0)
    import ctypes as C
    from cdnet_amd import _lib

    def which(**kw):
        """cdnet_conv_ws_eligible: 2 = conv_ws16_kernel, 1 = an older persistent kernel, 0 = the one-tile kernel"""
        a = engine.ConvArgs()
        x.fill(a.src[0])
        out = torch.empty((2, 32, 32, 64), dtype=torch.bfloat16, device='cuda')
        a.nsrc, a.w, a.out, a.Cout, a.out_cstride = 1, wp.data_ptr(), out.data_ptr(), 64, 64
        a.N, a.H, a.W, a.taps, a.npar, a.ostride, a.nchunk = 2, 32, 32, 9, 1, 1, 4
        a.tile, a.CK, a.BN, a.debug = 16, 16, 64, 64
        keep = []
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                keep.append(v)
                v = v.data_ptr()
            setattr(a, k, v)
        return int(_lib.load().cdnet_conv_ws_eligible(C.byref(a)))
    z = torch.zeros((64,), device='cuda')
    st = torch.zeros((8, 2, 64), device='cuda')
    assert which() == 2                                       # the plain launch is its own
    assert which(bias=z, oshift=z, orelu=1) == 2
    assert which(oscale=z) == 1                               # an epilogue scale (the host folds it into the weights instead)
    assert which(out_f16=1) == 1
    assert which(stats=st) == 1                               # training forward: BatchNorm statistics stay on conv_ws_kernel
    assert which(H=24) == 0                                   # ragged tiles: the one-tile kernel
    assert which(debug=32) == 0


@pytest.mark.parametrize('case', [dict(N=2, Cin=64, Cout=64, H=32, W=48, G=3), dict(N=1, Cin=128, Cout=128, H=32, W=32, G=0),
                                  dict(N=3, Cin=64, Cout=32, H=16, W=32, G=2), dict(N=1, Cin=64, Cout=64, H=16, W=16, G=1)])
def test_conv_ws16_fused_max_pool_output(case):
    """cdnet_conv_args.pool_out: nn.MaxPool2d(2, 2) of the ReLU-activated output from the movers' store path of conv_ws16_kernel's
    out-image form (the 'M' layers of the VGG16-BN encoder, model_unet_rev1.py:40-41) - the full-resolution output bit-identical to the
    launch without it, the pooled tensor bit-identical to torch's max_pool2d of that output (a maximum of stored values: exact)."""
    import os
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    if os.environ.get('CDNET_WS16_OUT', '1') == '0':
        pytest.skip('the fused max-pool rides in the out-image form, which this environment switches off')
    N, Cin, Cout, H, W, G = [case[k] for k in ('N', 'Cin', 'Cout', 'H', 'W', 'G')]
    g = torch.Generator().manual_seed(7 + Cin + H)
    x = _bf(torch.randn((N, Cin, H, W), generator=g))
    w = _bf(torch.randn((Cout, Cin, 3, 3), generator=g) * (1.5 / (9 * Cin) ** 0.5))
    b = torch.randn((Cout,), generator=g) * 0.3
    cfg = (16, 16, 64 if Cout > 32 else 32)
    wp = engine.pack_weights(w.cuda(), cfg, 0)
    src = [engine.Src(_nhwc(x))]
    engine.CONV_DEBUG = 64 | (G << 8)
    try:
        plain, _ = engine.conv_forward(src, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W)
        pout = torch.full((N, H // 2, W // 2, Cout), 7.0, dtype=torch.bfloat16, device='cuda')
        out = torch.empty_like(plain)
        assert engine.conv_forward(src, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, out=out, pool_out=pout, query_ws=True) == 2
        engine.conv_forward(src, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, out=out, pool_out=pout)
        torch.cuda.synchronize()
        # without the ReLU (signed values) or on the one-tile kernels the pooled output does not exist: the ABI says so
        assert engine.conv_forward(src, wp, Cout, cfg, oshift=b.cuda(), orelu=False, H=H, W=W, out=out, pool_out=pout, query_ws=True) == 0
        engine.CONV_DEBUG = 32
        with pytest.raises(RuntimeError):
            engine.conv_forward(src, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, out=out, pool_out=pout)
    finally:
        engine.CONV_DEBUG = 0
    assert torch.equal(out, plain)
    want = F.max_pool2d(_nchw(plain), 2)
    assert torch.equal(_nchw(pout), want)
    _close(_nchw(plain), F.relu(F.conv2d(x, w, b, padding=1)), 'conv + relu')


@pytest.mark.parametrize('case', [dict(N=2, Cin=64, Cout=64, H=32, W=48, G=3, mix=False), dict(N=1, Cin=64, Cout=64, H=32, W=32, G=0, mix=True),
                                  dict(N=3, Cin=64, Cout=64, H=16, W=32, G=2, mix=True), dict(N=1, Cin=64, Cout=32, H=16, W=16, G=1, mix=False)])
def test_conv_ws16_fused_classifier_output(case):
    """cdnet_conv_args.dot_w / dot_b / dot_out: the 1x1 classifier over the activated, bf16-rounded output from the movers' store path of
    conv_ws16_kernel's out-image form - the DAM head's point logit (model_unet_rev1.py:252-253) without a stored point feature.  The
    logits equal the fp32 dot product over the output of the plain launch (same rounded values, another summation order: 1e-5 of the
    magnitude), with or without the feature stored beside them; a launch the form does not serve says so."""
    import os
    import torch
    from cdnet_amd import engine
    if os.environ.get('CDNET_WS16_OUT', '1') == '0':
        pytest.skip('the fused classifier rides in the out-image form, which this environment switches off')
    N, Cin, Cout, H, W, G, mix = [case[k] for k in ('N', 'Cin', 'Cout', 'H', 'W', 'G', 'mix')]
    g = torch.Generator().manual_seed(11 + Cin + H + Cout)
    x = _bf(torch.randn((N, Cin, H, W), generator=g))
    w = _bf(torch.randn((Cout, Cin, 3, 3), generator=g) * (1.5 / (9 * Cin) ** 0.5))
    b = torch.randn((Cout,), generator=g) * 0.3
    dw = (torch.randn((Cout,), generator=g) * 0.2).cuda()
    db = torch.tensor([0.37], device='cuda')
    cfg = (16, 16, 64 if Cout > 32 else 32)
    srcs, kw = [engine.Src(_nhwc(x))], {}
    wp = engine.pack_weights(w.cuda(), cfg, 0)
    if mix:                                   # a residual unit's second launch: one-tap chunks of a second source behind the nine-tap ones
        x2 = _bf(torch.randn((N, Cin, H, W), generator=g))
        w1 = _bf(torch.randn((Cout, Cin, 1, 1), generator=g) * (1.0 / Cin ** 0.5))
        wp1 = engine.pack_weights(w1.cuda(), cfg, 0)
        nt = -(-Cout // cfg[2])
        wp = torch.cat([wp.view(nt, -1), wp1.view(nt, -1)], 1).contiguous().view(-1)
        srcs.append(engine.Src(_nhwc(x2)))
        kw = dict(taps1=1)
    engine.CONV_DEBUG = 64 | (G << 8)
    try:
        plain, _ = engine.conv_forward(srcs, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, **kw)
        torch.cuda.synchronize()
        engine.CONV_DEBUG = 64 | 16 | (G << 8)                   # the quad-request form of the movers: the same bits
        quad, _ = engine.conv_forward(srcs, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, **kw)
        torch.cuda.synchronize()
        assert torch.equal(quad, plain)
        engine.CONV_DEBUG = 64 | (G << 8)
        want = (plain.float() * dw.view(1, 1, 1, -1)).sum(3) + db
        pt = torch.full((N, 1, H, W), 7.0, device='cuda')
        assert engine.conv_forward(srcs, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, dot=(dw, db, pt), query_ws=True, **kw) == 2
        engine.conv_forward(srcs, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, dot=(dw, db, pt), **kw)          # the output is not stored
        torch.cuda.synchronize()
        err = (pt[:, 0] - want).abs().max()
        assert float(err) <= 1e-5 * float(want.abs().max()) + 1e-6, float(err)
        # both outputs: the feature beside its logits
        import ctypes as C
        from cdnet_amd import _lib
        a = engine.ConvArgs()
        for i, s in enumerate(srcs):
            s.fill(a.src[i])
        out = torch.empty_like(plain)
        pt2 = torch.zeros_like(pt)
        bb = b.cuda()
        a.nsrc, a.w, a.oshift, a.orelu = len(srcs), wp.data_ptr(), bb.data_ptr(), 1
        a.out, a.Cout, a.out_cstride, a.out_coff = out.data_ptr(), Cout, Cout, 0
        a.N, a.H, a.W, a.taps, a.npar, a.ostride, a.nchunk = N, H, W, 9, 1, 1, sum(s.C for s in srcs) // 16
        a.tile, a.CK, a.BN, a.debug, a.taps1 = cfg[0], cfg[1], cfg[2], engine.CONV_DEBUG, (1 if mix else 0)
        a.dot_w, a.dot_b, a.dot_out = dw.data_ptr(), db.data_ptr(), pt2.data_ptr()
        _lib.call('cdnet_conv_forward', C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(out, plain) and torch.equal(pt2, pt)
        # not served: the one-tile kernels, a pooled output beside it
        engine.CONV_DEBUG = 32
        assert engine.conv_forward(srcs, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, dot=(dw, db, pt), query_ws=True, **kw) == 0
        with pytest.raises(RuntimeError):
            engine.conv_forward(srcs, wp, Cout, cfg, oshift=b.cuda(), orelu=True, H=H, W=W, dot=(dw, db, pt), **kw)
    finally:
        engine.CONV_DEBUG = 0


@pytest.mark.parametrize('case', [dict(N=2, H=32, W=48, G=3), dict(N=1, H=16, W=32, G=0)])
def test_residual_unit_with_16_channel_input_takes_the_out_image_form(case):
    """the first residual unit of the DAM head (model_unet_rev1.py:161-170, x has 16 channels): ONE one-tap chunk behind the four nine-tap ones is
    an odd chunk count - with one more one-tap chunk of zero weights (engine.conv_forward(pad_chunks=1)) the launch takes conv_ws16_kernel's out-image
    form.  What the movers read for the padding chunk is the neighbouring pixel's channels (zeros past the tensor's end); its weights are zeros, the
    result equals the unpadded launch bit for bit and the float reference to bf16 rounding."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, H, W, G = [case[k] for k in ('N', 'H', 'W', 'G')]
    g = torch.Generator().manual_seed(5 + H + W)
    h = _bf(torch.randn((N, 64, H, W), generator=g))
    x = _bf(torch.randn((N, 16, H, W), generator=g))
    w = _bf(torch.randn((64, 64, 3, 3), generator=g) * (1.5 / (9 * 64) ** 0.5))
    w1 = _bf(torch.randn((64, 16, 1, 1), generator=g) * 0.25)
    b = torch.randn((64,), generator=g) * 0.3
    cfg = (16, 16, 64)
    a, b1 = engine.pack_weights(w.cuda(), cfg, 0).view(1, -1), engine.pack_weights(w1.cuda(), cfg, 0).view(1, -1)
    plain_pack = torch.cat([a, b1], 1).contiguous().view(-1)
    padded_pack = torch.cat([a, b1, torch.zeros((1, 16 * 64), dtype=a.dtype, device=a.device)], 1).contiguous().view(-1)
    srcs = [engine.Src(_nhwc(h)), engine.Src(_nhwc(x))]
    engine.CONV_DEBUG = 64 | (G << 8)
    try:
        kw = dict(oshift=b.cuda(), orelu=True, H=H, W=W, taps1=1)
        assert engine.conv_forward(srcs, padded_pack, 64, cfg, 9, query_ws=True, pad_chunks=1, **kw) == 2
        plain, _ = engine.conv_forward(srcs, plain_pack, 64, cfg, 9, **kw)
        padded, _ = engine.conv_forward(srcs, padded_pack, 64, cfg, 9, pad_chunks=1, **kw)
        torch.cuda.synchronize()
    finally:
        engine.CONV_DEBUG = 0
    assert torch.equal(padded, plain)
    want = F.relu(F.conv2d(h, w, None, padding=1) + F.conv2d(x, w1) + b.view(1, -1, 1, 1))
    _close(_nchw(padded), want, 'padded one-tap chunk')


@pytest.mark.parametrize('case', [dict(N=2, H=32, W=48, G=3, Cout=16), dict(N=1, H=16, W=32, G=0, Cout=64)])
def test_decoder_block_with_swapped_sources_and_a_padding_chunk(case):
    """cat([x (16 channels), skip (64)]) -> 3x3 convolution (model_unet_rev1.py:133-141) as [skip, x] with the weight's input channels permuted and one
    padding chunk of zero weights behind x (runtime.ConvLayer.forward_eval_swapped): five chunks become six - conv_ws16_kernel's out-image form
    with pair requests.  Another summation order than the [x, skip] launch (bf16 rounding apart), the same convolution."""
    import torch
    import torch.nn.functional as F
    from cdnet_amd import engine
    N, H, W, G, Cout = [case[k] for k in ('N', 'H', 'W', 'G', 'Cout')]
    g = torch.Generator().manual_seed(17 + H + Cout)
    x = _bf(torch.randn((N, 16, H, W), generator=g))
    skip = _bf(torch.randn((N, 64, H, W), generator=g))
    w = _bf(torch.randn((Cout, 80, 3, 3), generator=g) * (1.5 / (9 * 80) ** 0.5))
    b = torch.randn((Cout,), generator=g) * 0.3
    cfg = (16, 16, 64 if Cout > 32 else 32)
    wperm = torch.cat([w[:, 16:], w[:, :16], torch.zeros((Cout, 16, 3, 3))], 1).contiguous()
    wp = engine.pack_weights(wperm.cuda(), cfg, 0)
    srcs = [engine.Src(_nhwc(skip)), engine.Src(_nhwc(x))]
    engine.CONV_DEBUG = 64 | (G << 8)
    try:
        kw = dict(oshift=b.cuda(), orelu=True, H=H, W=W, pad_chunks=1)
        assert engine.conv_forward(srcs, wp, Cout, cfg, 9, query_ws=True, **kw) == 2
        got, _ = engine.conv_forward(srcs, wp, Cout, cfg, 9, **kw)
        torch.cuda.synchronize()
    finally:
        engine.CONV_DEBUG = 0
    want = F.relu(F.conv2d(torch.cat([x, skip], 1), w, b, padding=1))
    _close(_nchw(got), want, 'swapped sources + padding chunk')
