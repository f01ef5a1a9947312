"""Kernel-level parity of the HRNet-specific pieces against PyTorch fp32 on identical (bf16-exact) inputs:
stride-2 3x3 convolution through the space-to-depth view (forward pack mode 6, backward-data pack mode 7 + permute,
weight gradient reduce mode 6), cdnet_fuse_sum with affine / fp16 terms, the bilinear up-sampling transpose."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf(x):
    import torch
    return x.to(torch.bfloat16).float()


def _nhwc(x, dtype=None):
    import torch
    return x.permute(0, 2, 3, 1).contiguous().to(dtype or torch.bfloat16).cuda()


def _nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def test_stride2_conv_forward_backward():
    import torch
    import torch.nn.functional as F
    from cdnet_amd import _lib, engine, trainer
    g = torch.Generator().manual_seed(2)
    N, Cp, Cout, H, W = 2, 32, 48, 24, 40
    x = _bf(torch.randn((N, Cp, H, W), generator=g)).requires_grad_(True)
    w = _bf(torch.randn((Cout, Cp, 3, 3), generator=g) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=2, padding=1)
    dy = _bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    xd = _nhwc(x.detach())
    views = [engine.Src(xd, view=(a * W * Cp, H // 2, W // 2, 2 * Cp, 2 * W * Cp)) for a in (0, 1)]
    cfg = engine.choose_cfg([2 * Cp, 2 * Cp], Cout, H // 2, W // 2)
    wp = engine.pack_weights(w.detach().cuda().contiguous(), cfg, 6)
    out, _ = engine.conv_forward(views, wp, Cout, cfg, taps=9, H=H // 2, W=W // 2)
    np.testing.assert_allclose(_nchw(out).numpy(), y.detach().numpy(), rtol=2e-2, atol=2e-2)
    # backward-data: conv over dY with pack mode 7 -> space-to-depth gradient -> permute
    dyd = _nhwc(dy)
    cfgb = engine.choose_cfg([Cout], 4 * Cp, H // 2, W // 2)
    wpb = engine.pack_weights(w.detach().cuda().contiguous(), cfgb, 7)
    gs2d, _ = engine.conv_forward([engine.Src(dyd)], wpb, 4 * Cp, cfgb, taps=9, H=H // 2, W=W // 2)
    gin = torch.empty((N, H, W, Cp), dtype=torch.bfloat16, device='cuda')
    _lib.call('cdnet_s2d_to_nhwc', _lib.ptr(gs2d), N, H // 2, W // 2, Cp, _lib.ptr(gin), _lib.stream_ptr())
    np.testing.assert_allclose(_nchw(gin).numpy(), x.grad.numpy(), rtol=2e-2, atol=3e-2)
    # weight gradient: one launch per row-parity source, reduce mode 6
    lib = _lib.load()
    dw = torch.zeros((Cout, Cp, 3, 3), dtype=torch.float32, device='cuda')
    for a, s in enumerate(views):
        ci_t = trainer._choose_ci_tiles(s.C, Cout)
        ks = 5
        slab = torch.full((lib.cdnet_conv_wgrad_slab_floats(s.C, Cout, 9, 1, ci_t, ks),), float('nan'), dtype=torch.float32, device='cuda')
        cs = engine.ConvSrc()
        s.fill(cs)
        _lib.call('cdnet_conv_backward_weight', C.byref(cs), a * 2 * Cp, s.C, 4 * Cp, _lib.ptr(dyd), Cout, N, H // 2, W // 2, 9, 1, 1, ci_t, ks,
                  _lib.ptr(slab), _lib.ptr(dw), 6, _lib.stream_ptr())
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), rtol=3e-3, atol=3e-2)


def test_fuse_sum_affine_terms_and_upsample_backward():
    import torch
    import torch.nn.functional as F
    from cdnet_amd import _lib
    from cdnet_amd.models.dam.seg_hrnet_rev1 import FuseTerm
    g = torch.Generator().manual_seed(4)
    N, Cc, H, W = 2, 32, 16, 24
    a = _bf(torch.randn((N, Cc, H, W), generator=g))
    raw = torch.randn((N, Cc, H // 4, W // 4), generator=g).half().float().requires_grad_(True)
    sc, sh = torch.rand((Cc,), generator=g) + 0.5, torch.randn((Cc,), generator=g) * 0.3
    low = raw * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    up = F.interpolate(low, size=(H, W), mode='bilinear', align_corners=False)
    want = F.relu(a + up)
    terms = (FuseTerm * 2)()
    ad, rd = _nhwc(a), _nhwc(raw.detach(), torch.float16)
    scd, shd = sc.cuda(), sh.cuda()
    terms[0].x, terms[0].Hs, terms[0].Ws = ad.data_ptr(), H, W
    terms[1].x, terms[1].Hs, terms[1].Ws, terms[1].scale, terms[1].shift, terms[1].f16 = rd.data_ptr(), H // 4, W // 4, scd.data_ptr(), shd.data_ptr(), 1
    out = torch.empty((N, H, W, Cc), dtype=torch.bfloat16, device='cuda')
    _lib.call('cdnet_fuse_sum', C.byref(terms), 2, N, H, W, Cc, 1, _lib.ptr(out), Cc, 0, _lib.stream_ptr())
    np.testing.assert_allclose(_nchw(out).numpy(), want.detach().numpy(), rtol=1e-2, atol=1e-2)
    # transpose of the up-sampling
    d = _bf(torch.randn((N, Cc, H, W), generator=g))
    up.backward(d)
    want_low = raw.grad / sc.view(1, -1, 1, 1)             # gradient w.r.t. the affine output
    dd = _nhwc(d)
    din = torch.empty((N, H // 4, W // 4, Cc), dtype=torch.bfloat16, device='cuda')
    _lib.call('cdnet_upsample_bilinear_backward', _lib.ptr(dd), N, H, W, Cc, Cc, 0, H // 4, W // 4, _lib.ptr(din), _lib.stream_ptr())
    np.testing.assert_allclose(_nchw(din).numpy(), want_low.numpy(), rtol=1e-2, atol=2e-2)
