"""Host-side plumbing for the convolution stack: ctypes mirrors of the ABI structs, weight packing, per-layer
kernel configuration and thin call wrappers.  No compute happens here - every function ends in a C-ABI call
into libcdnet_hip.so (torch tensors only provide device memory and the current stream).
"""
import ctypes as C
import os
import torch

from . import _lib


class ConvSrc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('res', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('C', C.c_int), ('Hs', C.c_int), ('Ws', C.c_int), ('pool', C.c_int), ('relu', C.c_int),
                ('off_y', C.c_int), ('off_x', C.c_int), ('f16', C.c_int), ('row_stride', C.c_int)]


class ConvArgs(C.Structure):
    _fields_ = [('src', ConvSrc * 2), ('nsrc', C.c_int), ('w', C.c_void_p), ('bias', C.c_void_p),
                ('oscale', C.c_void_p), ('oshift', C.c_void_p), ('orelu', C.c_int), ('out', C.c_void_p),
                ('Cout', C.c_int), ('out_cstride', C.c_int), ('out_coff', C.c_int), ('stats', C.c_void_p),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('taps', C.c_int), ('npar', C.c_int),
                ('ostride', C.c_int), ('nchunk', C.c_int), ('tile', C.c_int), ('CK', C.c_int), ('BN', C.c_int),
                ('out_f16', C.c_int), ('debug', C.c_int), ('ws', C.c_int), ('f32', C.c_int),
                ('eres', C.c_void_p), ('eres_scale', C.c_void_p), ('eres_shift', C.c_void_p), ('eres_f16', C.c_int), ('eres_relu', C.c_int),
                ('taps1', C.c_int), ('pad_', C.c_int), ('pool_out', C.c_void_p),
                ('dot_w', C.c_void_p), ('dot_b', C.c_void_p), ('dot_out', C.c_void_p)]


def _dp(t):
    return None if t is None else t.data_ptr()


class Src:
    """One convolution source: a 16-bit NHWC tensor [N,Hs,Ws,C] (bf16, or fp16 for raw pre-BatchNorm outputs) plus the
    producer's lazily-applied transform.  `view=(ptr_offset_elems, Hs, Ws, C, row_stride)` describes a strided window of
    `x` instead of the dense tensor (space-to-depth view of an upsampled gradient)."""

    def __init__(self, x, scale=None, shift=None, relu=False, pool=False, res=None, off=(0, 0), view=None):
        assert x.dtype in (torch.bfloat16, torch.float16, torch.float32) and x.is_contiguous()
        # a source keeps its residual branch in the storage type of x
        assert res is None or res.dtype == x.dtype
        self.x, self.scale, self.shift, self.relu, self.pool, self.res, self.off = x, scale, shift, relu, pool, res, off
        if view is None:
            assert x.dim() == 4
            self.N, self.Hs, self.Ws, self.Cc = x.shape
            self.ptr_off, self.row_stride = 0, 0
        else:
            self.ptr_off, self.Hs, self.Ws, self.Cc, self.row_stride = view
            self.N = x.shape[0]

    @property
    def f16(self):
        """storage code of the ABI: 0 bf16, 1 fp16, 2 fp32"""
        return {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[self.x.dtype]

    @property
    def f32(self):
        return self.x.dtype == torch.float32

    @property
    def C(self):
        return self.Cc

    def ptr(self):
        return self.x.data_ptr() + self.x.element_size() * self.ptr_off

    def logical_hw(self):
        h, w = self.Hs, self.Ws
        if self.pool:                      # 1/True: floor mode, 2: ceil mode
            c = 1 if int(self.pool) == 2 else 0
            return ((h + c) // 2, (w + c) // 2)
        return (h, w)

    def fill(self, cs):
        """fill a ConvSrc ctypes struct"""
        cs.x = self.ptr()
        cs.res = _dp(self.res)
        cs.scale, cs.shift = _dp(self.scale), _dp(self.shift)
        cs.C, cs.Hs, cs.Ws = self.C, self.Hs, self.Ws
        cs.pool, cs.relu = int(self.pool), int(self.relu)
        cs.off_y, cs.off_x = self.off
        cs.f16 = int(self.f16)
        cs.row_stride = int(self.row_stride)


_SUPPORTED = {(16, 16, 32), (16, 16, 64), (16, 32, 32), (16, 32, 64), (16, 32, 128), (16, 64, 64),
              (8, 32, 64), (8, 32, 128), (8, 64, 64)}


def choose_cfg(src_channels, Cout, H, W, override=None, taps=9, transposed=False, N=16, f32=False):
    """(tile, CK, BN) for a layer.  CK must divide every source's channel count.  N = images per launch."""
    if f32:
        # fp32-precision kernels (csrc/conv32.hip): 16-channel chunks, 16x16 tiles (8x8 for the few-pixel layers)
        assert all(c % 16 == 0 for c in src_channels), src_channels
        assert override is None or (tuple(override)[1] == 16 and tuple(override) in ((16, 16, 64), (16, 16, 32), (8, 16, 64))), \
            'fp32 precision: configuration override %s is not one of the fp32 kernels' % (override,)
        if override is not None:
            return tuple(override)
        small = min(H, W) <= 8 or (min(H, W) <= 16 and N <= 32)
        if small and Cout > 32:
            # 16 x 16-pixel layers with long channel loops (the 512-channel bottleneck): 32-cout tiles give the persistent producer /
            # consumer kernel one workgroup per CU (16 tiles x 16 cout tiles at 16 images) instead of 8 x 8 tiles on the one-tile kernel
            if os.environ.get('CDNET_F32_WS16', '1') == '1' and taps == 9 and not transposed and H == 16 and W == 16 and \
                    sum(src_channels) >= 256 and Cout >= 256 and 8 <= N <= 32:
                return (16, 16, 32)
            return (8, 16, 64)
        return (16, 16, 64 if Cout > 32 else 32)
    if override is not None:
        assert tuple(override) in _SUPPORTED, override
        return tuple(override)
    ctot = sum(src_channels)
    # few pixels: 8x8 tiles give the grid more workgroups (measured: 16x16 layers at 16 images, 8x8 layers always)
    small = min(H, W) <= 8 or (min(H, W) <= 16 and N <= 32)
    ck = 64
    while any(c % ck for c in src_channels):
        ck //= 2
    assert ck >= 16, 'source channels must be multiples of 16: %s' % (src_channels,)
    if small:
        # 16x16-pixel layers with long channel loops (the bottleneck convolutions): the persistent producer / consumer kernel keeps four
        # chunks of loads in flight per workgroup - 128 workgroups of it beat 512 workgroups that each expose every chunk's latency
        # (512 -> 512 @16x16 x16: 54 -> measured in tools/bench_conv_stream.py)
        ws16 = taps == 9 and not transposed and H % 16 == 0 and W % 16 == 0 and min(H, W) == 16 and ctot % 64 == 0 and 256 <= ctot <= 768
        if ck >= 32 and not ws16:
            return (8, 32, 64)
    if ck == 64:
        ck = 32                       # 2 workgroups per CU (LDS) beat one fat one
    bn = 64 if Cout > 32 else 32
    if ck >= 16:
        ck = 16                       # 34 KB of LDS and <= 168 VGPRs: three workgroups per CU hide the staging latency (measured)
    if ck == 16 and bn > 64:
        bn = 64
    return (16, ck, bn)


def dominant_kernel_name(precision, tiles=16):
    """the kernel cdnet_conv_forward runs the dominant layer (3x3 64 -> 64 on full 16x16 tiles) on - what bench.py's `roofline` object and
    the profile filters name.  The 16-bit kernel's instantiation depends on the launch's size: chunk-PAIR requests while the tensors fit the
    256 MB Infinity Cache, QUAD requests beyond it (csrc/conv16ws.hip: launch_ws16's `quad` test) - the 16- and 64-tile figures of
    `roofline_bf16*` come from two different instantiations"""
    if precision == 'fp32':
        return 'conv_ws32_kernel<64,0,false>'
    quad = tiles * 256 * 256 * (64 + 64) * 2 >= (512 << 20)         # io_bytes of launch_ws16
    return 'conv_ws16_kernel<64,0,false,false,0,4,4,true,false,true,%s>' % ('true' if quad else 'false')


def packed_elems(Cout, nchunk, taps, CK, BN, npar):
    return _lib.load().cdnet_conv_packed_weight_elems(Cout, nchunk, taps, CK, BN, npar)


def _pack_dims(w, cfg, mode):
    """(Cout, Cin, KH, KW, CK, BN, taps, npar) as cdnet_pack_conv_weights wants them (GEMM roles, see include/cdnet_hip.h)"""
    _, CK, BN = cfg[:3]
    if mode == 0:
        Cout, Cin, KH, KW = w.shape
    elif mode == 1:
        Cin, Cout, KH, KW = w.shape          # roles swap: GEMM-Cout = original in_channels
    elif mode in (4, 5, 6):
        Cout, ct, KH, KW = w.shape           # backward-data of a transposed conv: GEMM-Cout = its in_channels,
        Cin = 4 * ct                         # GEMM-Cin = space-to-depth of its out_channels (6: stride-2 forward conv)
    elif mode == 7:
        Cin, ct, KH, KW = w.shape            # backward-data of the stride-2 conv (mode 6): GEMM-Cout = 4 * its in_channels
        Cout = 4 * ct
    else:
        Cin, Cout, KH, KW = w.shape
    taps = {2: 4, 3: 1, 4: 9, 5: 1, 6: 9, 7: 9}.get(mode, KH * KW)
    npar = 4 if mode in (2, 3) else 1
    return Cout, Cin, KH, KW, CK, BN, taps, npar


class PackJob(C.Structure):
    _fields_ = [('w', C.c_void_p), ('packed', C.c_void_p), ('Cout', C.c_int), ('Cin', C.c_int), ('KH', C.c_int), ('KW', C.c_int),
                ('CK', C.c_int), ('BN', C.c_int), ('mode', C.c_int), ('pad_', C.c_int)]


def pack_job(w, cfg, mode, out, split=False):
    """descriptor of one re-pack of `w` into the existing packed buffer `out` (cdnet_pack_conv_weights_batch)"""
    Cout, Cin, KH, KW, CK, BN, taps, npar = _pack_dims(w, cfg, mode)
    assert out.numel() == packed_elems(Cout, Cin // CK, taps, CK, BN, npar) * (2 if split else 1)
    j = PackJob()
    j.w, j.packed, j.Cout, j.Cin, j.KH, j.KW, j.CK, j.BN = w.data_ptr(), out.data_ptr(), Cout, Cin, KH, KW, CK, BN
    j.mode = mode | (PACK_SPLIT if split else 0)
    return j


PACK_SPLIT = 16       # CDNET_PACK_SPLIT


def pack_weights(w, cfg, mode, Cin_pad=None, out=None, split=False, cout_scale=None):
    """w: fp32 cuda tensor. mode 0 Conv2d fwd [Cout,Cin,KH,KW]; 1 Conv2d bwd-data; 2 ConvT k4s2p1; 3 ConvT k2s2.
    Cin_pad: Cin rounded up to the chunk grid (e.g. 3 -> 16 for the RGB input).  Returns a bf16-bits int16 tensor."""
    assert w.dtype == torch.float32 and w.is_cuda and w.is_contiguous()
    Cout, Cin, KH, KW, CK, BN, taps, npar = _pack_dims(w, cfg, mode)
    Cin_p = Cin if Cin_pad is None else Cin_pad
    if Cin_p != Cin:
        # zero-extend the reduction channels (only the RGB stem needs this)
        if mode == 0:
            wp = torch.zeros((Cout, Cin_p, KH, KW), dtype=torch.float32, device=w.device)
            wp[:, :Cin] = w
        else:
            raise NotImplementedError
        w, Cin = wp.contiguous(), Cin_p
    n = packed_elems(Cout, Cin // CK, taps, CK, BN, npar) * (2 if split else 1)
    if out is None or out.numel() != n:
        out = torch.empty((n,), dtype=torch.int16, device=w.device)
    if cout_scale is not None:           # eval-mode BatchNorm scale folded into the weights (mode 0): w[cout] * cout_scale[cout], then rounded
        assert mode == 0 and cout_scale.dtype == torch.float32 and cout_scale.numel() == Cout
        _lib.call('cdnet_pack_conv_weights_scaled', _lib.ptr(w), _lib.ptr(cout_scale), _lib.ptr(out), Cout, Cin, KH, KW, CK, BN,
                  mode | (PACK_SPLIT if split else 0), _lib.stream_ptr())
        return out
    _lib.call('cdnet_pack_conv_weights', _lib.ptr(w), _lib.ptr(out), Cout, Cin, KH, KW, CK, BN, mode | (PACK_SPLIT if split else 0),
              _lib.stream_ptr())
    return out


CONV_DEBUG = 0        # cdnet_conv_args.debug of every launch (tests: 32 = conv_fwd_kernel only, 64 = force conv_ws_kernel)


def conv_forward(srcs, wpacked, Cout, cfg, taps=9, transposed=False, bias=None, oscale=None, oshift=None,
                 orelu=False, out=None, stats=None, H=None, W=None, out_dtype=torch.bfloat16, eres=None, query_ws=False, bns=None, taps1=0, pool_out=None, dot=None,
                 pad_chunks=0, debug_or=0):
    """Launch one convolution.  srcs: list of Src (1 or 2).  Returns (out, stats).  fp32 sources select the fp32-precision
    kernels (`wpacked` must then be the split pack and the output is fp32)."""
    tile, CK, BN = cfg[:3]
    s0 = srcs[0]
    f32 = s0.f32
    assert all(s.f32 == f32 for s in srcs)
    if f32:
        out_dtype = torch.float32
    N = s0.N
    if H is None:
        H, W = s0.logical_hw()
        H, W = H + 0, W + 0
    ostride = 2 if transposed else 1
    npar = 4 if transposed else 1
    a = ConvArgs()
    nchunk = 0
    for i, s in enumerate(srcs):
        assert s.C % CK == 0, (s.C, CK)
        s.fill(a.src[i])
        nchunk += s.C // CK
    a.nsrc = len(srcs)
    if dot is not None:
        # (weights [Cout] f32, bias [1] f32, logits f32 [N,1,H,W]): the 1x1 classifier over the activated output leaves with the stores and
        # the output itself is not stored (cdnet_conv_args.dot_out; conv_ws16_kernel's out-image form - ask query_ws first)
        assert out is None and stats is None and eres is None and bns is None and pool_out is None and not f32 and not transposed
        a.dot_w, a.dot_b, a.dot_out = dot[0].data_ptr(), dot[1].data_ptr(), dot[2].data_ptr()
        assert dot[0].numel() == Cout and dot[0].dtype == torch.float32 and dot[2].dtype == torch.float32 and dot[2].numel() == N * H * W
    elif out is None:
        out = torch.empty((N, H * ostride, W * ostride, Cout), dtype=out_dtype, device=s0.x.device)
    ntiles = ((H + tile - 1) // tile) * ((W + tile - 1) // tile)
    if stats is True:
        stats = torch.empty((N * npar * ntiles, 2, Cout), dtype=torch.float32, device=s0.x.device)
    a.w, a.bias, a.oscale, a.oshift = _dp(wpacked), _dp(bias), _dp(oscale), _dp(oshift)
    a.orelu = int(orelu)
    a.out, a.Cout, a.out_cstride, a.out_coff = (None if out is None else out.data_ptr()), Cout, (Cout if out is None else out.shape[3]), 0
    a.stats = _dp(stats)
    a.N, a.H, a.W = N, H, W
    # pad_chunks (two plain 16-bit sources only): that many more chunks than the second source has channels for - their packed
    # weights are zeros (the caller's pack holds them), what the movers read for them is the neighbouring pixel's channels (zeros past the tensor's
    # end); it makes the chunk count even, which conv_ws16_kernel's out-image form with pair requests needs
    assert pad_chunks == 0 or (len(srcs) == 2 and taps == 9 and not f32 and all(s.scale is None and not s.relu and s.res is None and not s.f16 for s in srcs))
    a.taps, a.npar, a.ostride, a.nchunk = taps, npar, ostride, nchunk + pad_chunks
    a.tile, a.CK, a.BN = tile, CK, BN
    a.out_f16 = int(out is not None and out.dtype == torch.float16)
    a.ws = 0
    a.taps1 = taps1
    a.pool_out = _dp(pool_out)           # nn.MaxPool2d(2, 2) of the activated output beside it (conv_ws16_kernel's out-image form; ask query_ws first)
    a.debug = CONV_DEBUG | debug_or      # (debug_or: a caller's kernel-selection bits for THIS launch - no global is touched)
    a.f32 = int(f32)
    assert out is None or (out.dtype == torch.float32) == f32
    if eres is not None:                 # fused residual epilogue: eres = Src(other branch[, scale, shift], relu=...)
        assert not transposed and stats is None and not orelu and out.dtype == (torch.float32 if f32 else torch.bfloat16) and tuple(eres.x.shape) == tuple(out.shape)
        assert eres.f32 == f32
        a.eres, a.eres_scale, a.eres_shift = eres.x.data_ptr(), _dp(eres.scale), _dp(eres.shift)
        a.eres_f16, a.eres_relu = int(eres.f16), int(eres.relu)
    if bns is not None:                  # BatchNorm-backward statistics of the output beside the stores (cdnet_conv_args.ws = 2)
        raw, sc, sh, mu, inv, partial = bns
        assert eres is None and stats is None and bias is None and oscale is None and oshift is None and tuple(raw.shape) == tuple(out.shape)
        a.ws = 2
        a.eres, a.oscale, a.oshift, a.eres_scale, a.eres_shift, a.stats = raw.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), inv.data_ptr(), partial.data_ptr()
    if query_ws:                         # would this launch run on a producer / consumer kernel? 2 = conv_ws16_kernel, 1 = conv_ws_kernel / conv_ws32_kernel (nothing is launched)
        return int(_lib.load().cdnet_conv_ws_eligible(C.byref(a)))
    _lib.call('cdnet_conv_forward', C.byref(a), _lib.stream_ptr())
    return out, stats
