"""Execution runtime of the convolutional networks on the HIP kernels.

A network (cdnet_amd/models/...) owns its parameters as fp32 torch tensors with the reference's state_dict names;
this module turns them into packed bf16 MFMA weights and drives the C-ABI calls.  Nothing here computes on torch
tensors: torch is the allocator / stream provider.

Activation convention: NHWC bf16.  A tensor handed from layer to layer is an `engine.Src`: the stored tensor plus
the producer's not-yet-applied per-channel affine / residual / ReLU / 2x2 max-pool, which the consumer's staging
code applies on the fly.  In eval mode convolutions fold BatchNorm into their epilogue and store activated values;
in train mode they store the raw convolution output and emit BN statistics, and consumers apply
relu(raw*scale+shift).
"""
import ctypes as C
import weakref
import torch

from . import _lib, engine
from .engine import Src

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

import os as _os

# 'bf16': NHWC bf16 activations (fp16 for raw pre-BatchNorm outputs), bf16 MFMA operands, fp32 accumulation
# 'fp32': NHWC fp32 activations / gradients, split-bf16 (hi, lo) operands, three MFMAs per product, fp32 accumulation
PRECISION = _os.environ.get('CDNET_PRECISION', 'bf16')
assert PRECISION in ('bf16', 'fp32'), PRECISION


def set_precision(p):
    global PRECISION
    assert p in ('bf16', 'fp32'), p
    if p != PRECISION:
        PRECISION = p
        WEIGHTS_EPOCH[0] += 1            # every packed weight copy is stale (hi-only vs hi|lo packs)


def act_dtype():
    """stored activations / gradients that are MFMA operands"""
    return torch.float32 if PRECISION == 'fp32' else torch.bfloat16


def raw_dtype():
    """raw pre-BatchNorm convolution outputs and residual branches"""
    return torch.float32 if PRECISION == 'fp32' else torch.float16


DEBUG_NORELU = False   # debug aid (tools/debug_train.py): linearised network
TAPE = None              # set by cdnet_amd.trainer around a training forward: layers append themselves in execution order
WEIGHTS_EPOCH = [0]      # bumped by the fused Adam step (it updates parameters behind torch's version counters)
LAYERS = weakref.WeakSet()   # every live ConvLayer (the trainer batches their weight re-packs)


class HeadFeat(C.Structure):
    _fields_ = [('raw', C.c_void_p), ('res', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('relu', C.c_int), ('f16', C.c_int)]


def head_feat(s):
    assert not s.pool and s.off == (0, 0) and s.C == 64
    f = HeadFeat()
    f.raw, f.res = s.x.data_ptr(), (None if s.res is None else s.res.data_ptr())
    f.scale = None if s.scale is None else s.scale.data_ptr()
    f.shift = None if s.shift is None else s.shift.data_ptr()
    f.relu = int(s.relu)
    f.f16 = int(s.f16)
    return f


class ConvLayer:
    """One convolution (optionally followed by BatchNorm) of a network."""

    def __init__(self, name, kind, weight, bias=None, bn=None, cfg_override=None):
        assert kind in ('conv3', 'conv1', 'convT4', 'convT2', 'conv3s2')
        self.name, self.kind, self.weight, self.bias, self.bn = name, kind, weight, bias, bn
        self.cfg_override = cfg_override
        if kind == 'conv3s2':                 # stride-2 3x3 conv over the space-to-depth view of its input (pack mode 6)
            self.Cout, self.Cin = weight.shape[0], 4 * weight.shape[1]
        elif kind in ('conv3', 'conv1'):
            self.Cout, self.Cin = weight.shape[0], weight.shape[1]
        else:
            self.Cin, self.Cout = weight.shape[0], weight.shape[1]
        self.taps = {'conv3': 9, 'conv1': 1, 'convT4': 4, 'convT2': 1, 'conv3s2': 9}[kind]
        self.transposed = kind in ('convT4', 'convT2')
        self.pack_mode = {'conv3': 0, 'conv1': 0, 'convT4': 2, 'convT2': 3, 'conv3s2': 6}[kind]
        self.cfg = None
        self.cfg_f32 = False                   # the precision self.cfg / self.wp / self.wpb were made for
        self.wp = None
        self.wp_version = None
        self.fold = None                       # eval-mode (scale, shift)
        self.fold_version = None
        dev = weight.device
        if bn is not None:
            self.scale = torch.empty((self.Cout,), dtype=torch.float32, device=dev)
            self.shift = torch.empty((self.Cout,), dtype=torch.float32, device=dev)
            self.save_mean = torch.empty((self.Cout,), dtype=torch.float32, device=dev)
            self.save_invstd = torch.empty((self.Cout,), dtype=torch.float32, device=dev)
        self.stats = None
        self.saved = None                      # (srcs, raw, H, W) of the last training forward
        self.node_relu, self.node_res = True, None   # how consumers see this layer's output (set in forward / by the RU)
        self.needs_input_grad = True
        self.bias_grad_from = None
        self.wpb, self.wpb_version, self.cfg_bwd = None, None, None
        self.wp_padded = False                 # forward pack zero-extends Cin (RGB stem): not batchable
        LAYERS.add(self)

    # -- weights ------------------------------------------------------------------------------------
    def _version(self):
        v = self.weight._version
        if self.bias is not None:
            v = (v, self.bias._version)
        return v

    def prepare(self, src_channels, H, W, N=16):
        cin_total = sum(src_channels)
        f32 = PRECISION == 'fp32'
        if self.cfg is None or self.cfg_f32 != f32:
            self.cfg = engine.choose_cfg(src_channels, self.Cout, H, W, self.cfg_override, taps=self.taps, transposed=self.transposed, N=N,
                                         f32=f32)
            self.cfg_f32, self.wp, self.cfg_bwd, self.wpb = f32, None, None, None
        ver = (self.weight._version, WEIGHTS_EPOCH[0])
        if self.wp is None or self.wp_version != ver:
            pad = cin_total if cin_total != self.Cin else None       # RGB stem: 3 -> 16
            self.wp_padded = pad is not None
            self.wp = engine.pack_weights(self.weight.detach(), self.cfg, self.pack_mode, Cin_pad=pad, out=self.wp, split=f32)
            self.wp_version = ver

    def backward_pack(self, cin_total, H, W):
        """(packed weights, cfg) of the backward-data convolution: a forward convolution with flipped / transposed
        weights (Conv2d), or a 3x3 / 1x1 convolution over the space-to-depth view of the gradient (ConvTranspose2d)."""
        ver = (self.weight._version, WEIGHTS_EPOCH[0])
        f32 = self.cfg_f32
        if self.cfg_bwd is None:
            if self.transposed:
                self.cfg_bwd = engine.choose_cfg([2 * self.Cout, 2 * self.Cout], self.Cin, H, W, taps=(9 if self.kind == 'convT4' else 1), f32=f32)
            else:
                self.cfg_bwd = engine.choose_cfg([self.Cout], cin_total, H, W, taps=self.taps, f32=f32)
        if self.wpb is None or self.wpb_version != ver:
            if self.transposed:
                mode = 4 if self.kind == 'convT4' else 5
                self.wpb = engine.pack_weights(self.weight.detach(), self.cfg_bwd, mode, out=self.wpb, split=f32)
            else:
                assert cin_total == self.Cin, 'the RGB stem never needs an input gradient'
                self.wpb = engine.pack_weights(self.weight.detach(), self.cfg_bwd, 7 if self.kind == 'conv3s2' else 1, out=self.wpb, split=f32)
            self.wpb_version = ver
        return self.wpb, self.cfg_bwd

    def eval_fold(self):
        bn = self.bn
        ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               None if self.bias is None else self.bias._version, WEIGHTS_EPOCH[0])
        if self.fold is None or self.fold_version != ver:
            if self.fold is None:
                self.fold = (torch.empty_like(self.scale), torch.empty_like(self.shift))
            _lib.call('cdnet_bn_fold_eval', _lib.ptr(bn.weight.detach()), _lib.ptr(bn.bias.detach()),
                      _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var),
                      None if self.bias is None else _lib.ptr(self.bias.detach()), BN_EPS, self.Cout,
                      _lib.ptr(self.fold[0]), _lib.ptr(self.fold[1]), _lib.stream_ptr())
            self.fold_version = ver
        return self.fold

    def eval_pack(self, src_channels, allow_f32=False):
        """eval mode, 16-bit path, 3x3 / 1x1 Conv2d + BatchNorm: the packed weights with the BatchNorm scale folded in
        (bf16(w[cout] * scale[cout])) and the shift that is left for the epilogue - what conv_ws16_kernel takes as the accumulators'
        initial value (csrc/conv16ws.hip).  The RGB stem's zero-extended pack (3 -> 16 reduction channels) is folded like any other.  Returns
        (packed, shift) or None when the layer keeps the epilogue affine (transposed and stride-2 layers, layers without BatchNorm, fp32 mode
        unless `allow_f32`)."""
        if not EVAL_FOLD_WEIGHTS or not getattr(self, 'fold_eval', True) or self.kind not in ('conv3', 'conv1') or self.bn is None:
            return None
        f32 = PRECISION == 'fp32'
        if f32 and not allow_f32:
            return None                                      # (fp32 mode keeps the epilogue affine: conv_ws32_kernel applies it per lane at no cost;
                                                             #  only the residual units' two-launch form folds the scale - into the hi | lo split pack)
        cin_total = sum(src_channels)
        pad = cin_total if cin_total != self.Cin else None       # RGB stem: 3 -> 16 zero-extended reduction channels
        sc, sh = self.eval_fold()
        ver = (self.weight._version, self.fold_version, WEIGHTS_EPOCH[0], tuple(self.cfg), f32)
        if getattr(self, 'wpe', None) is None or self.wpe_version != ver:
            self.wpe = engine.pack_weights(self.weight.detach(), self.cfg, 0, Cin_pad=pad, out=getattr(self, 'wpe', None), cout_scale=sc, split=f32)
            self.wpe_version = ver
        return self.wpe, sh

    def forward_eval_swapped(self, srcs, relu=True, H=None, W=None):
        """eval mode, 16-bit path, torch.cat([x, skip]) -> conv3x3 -> BatchNorm -> ReLU of a decoder block (model_unet_rev1.py:133-141) whose first
        source has an ODD number of 16-channel chunks (the last block: 16 + 64 channels = five chunks, conv_ws16_kernel's consumers-store form:
        399 us per 64 tiles for 0.8 GB).  The convolution does not care in which order its input channels arrive: the sources are passed as
        [skip, x] with the weight's input channels permuted to match, and one padding chunk of zero weights behind x makes the count even - the
        out-image form with pair requests.  Returns the output Src, or None when the launch is not that kernel's (the caller takes forward())."""
        if PRECISION != 'bf16' or self.kind != 'conv3' or self.bn is None or len(srcs) != 2 or not EVAL_FOLD_WEIGHTS or not getattr(self, 'fold_eval', True):
            return None
        a, b = srcs
        if (a.C // 16) % 2 == 0 or (b.C // 16) % 2 or a.C % 16 or b.C % 16 or a.C + b.C != self.Cin:
            return None
        if any(s.scale is not None or s.relu or s.res is not None or s.f16 or s.pool for s in srcs):
            return None
        if H is None:
            H, W = b.logical_hw()
        if H % 16 or W % 16:
            return None
        self.prepare([s.C for s in srcs], H, W, a.N)
        if tuple(self.cfg[:2]) != (16, 16):
            return None
        sc, sh = self.eval_fold()
        ver = (self.weight._version, self.fold_version, WEIGHTS_EPOCH[0], tuple(self.cfg), a.C)       # (a.C: the split point of the channel permutation)
        if getattr(self, 'wps', None) is None or self.wps_version != ver:
            w = self.weight.detach()
            wz = torch.zeros((w.shape[0], 16, w.shape[2], w.shape[3]), dtype=w.dtype, device=w.device)
            wperm = torch.cat([w[:, a.C:], w[:, :a.C], wz], 1).contiguous()          # input channels: [skip | x | padding chunk]
            self.wps = engine.pack_weights(wperm, self.cfg, 0, cout_scale=sc)
            self.wps_version = ver
        kw = dict(oshift=sh, orelu=relu and not DEBUG_NORELU, H=H, W=W, pad_chunks=1)
        key = (a.N, H, W, a.C, b.C, a.off, b.off, engine.CONV_DEBUG, PRECISION)      # (the tests flip CONV_DEBUG to force the one-tile kernel: another answer)
        elig = self.__dict__.setdefault('_swapped_eligible', {})
        if key not in elig:
            elig[key] = engine.conv_forward([b, a], self.wps, self.Cout, self.cfg, 9, query_ws=True, **kw) == 2
        if not elig[key]:
            return None
        out, _ = engine.conv_forward([b, a], self.wps, self.Cout, self.cfg, 9, **kw)
        return Src(out)

    def forward_eval_pool(self, srcs):
        """eval mode: relu(bn(conv(.))) AND its nn.MaxPool2d(2, 2) from one launch (the 'M' layers of the VGG16-BN encoder,
        model_unet_rev1.py:40-41: conv_ws16_kernel's movers pool the out image beside their stores, conv_ws32_kernel's consumers the windows
        they hold - no cdnet_src_materialize pass).
        Returns (Src, pooled Src), or (Src, None) when the launch is not that kernel's (the caller pools as before)."""
        H, W = srcs[0].logical_hw()
        if PRECISION == 'bf16' and H % 2 == 0 and W % 2 == 0:
            self.prepare([s.C for s in srcs], H, W, srcs[0].N)
            ep = self.eval_pack([s.C for s in srcs])
            if ep is not None:
                N = srcs[0].N
                out = torch.empty((N, H, W, self.Cout), dtype=torch.bfloat16, device=srcs[0].x.device)
                pout = torch.empty((N, H // 2, W // 2, self.Cout), dtype=torch.bfloat16, device=srcs[0].x.device)
                kw = dict(oshift=ep[1], orelu=True, H=H, W=W, out=out, pool_out=pout)
                if engine.conv_forward(srcs, ep[0], self.Cout, self.cfg, self.taps, self.transposed, query_ws=True, **kw) == 2:
                    engine.conv_forward(srcs, ep[0], self.Cout, self.cfg, self.taps, self.transposed, **kw)
                    return Src(out), Src(pout)
        if PRECISION == 'fp32' and H % 2 == 0 and W % 2 == 0 and self.bn is not None and not DEBUG_NORELU:
            # fp32 mode: conv_ws32_kernel's consumers hold whole 2x2 windows - the pooled value leaves beside the stores
            self.prepare([s.C for s in srcs], H, W, srcs[0].N)
            sc, sh = self.eval_fold()
            N = srcs[0].N
            out = torch.empty((N, H, W, self.Cout), dtype=torch.float32, device=srcs[0].x.device)
            pout = torch.empty((N, H // 2, W // 2, self.Cout), dtype=torch.float32, device=srcs[0].x.device)
            kw = dict(oscale=sc, oshift=sh, orelu=True, H=H, W=W, out=out, pool_out=pout)
            if engine.conv_forward(srcs, self.wp, self.Cout, self.cfg, self.taps, self.transposed, query_ws=True, **kw) == 1:
                engine.conv_forward(srcs, self.wp, self.Cout, self.cfg, self.taps, self.transposed, **kw)
                return Src(out), Src(pout)
        return self.forward(srcs, False), None

    # -- forward ------------------------------------------------------------------------------------
    def forward(self, srcs, training, relu=True, H=None, W=None, out_dtype=None, eres=None, debug_or=0):
        """Returns the output as a Src (lazy transform attached in train mode).  Raw (pre-BatchNorm) outputs of the
        training path are stored as fp16: the consumer's affine needs more than bf16's 8 significant bits.  In the fp32
        precision mode every stored tensor is fp32 (out_dtype is ignored)."""
        if PRECISION == 'fp32':
            out_dtype = torch.float32
        elif out_dtype is None:
            out_dtype = torch.bfloat16
        if H is None:
            H, W = srcs[0].logical_hw()
        if DEBUG_NORELU:
            relu = False
        self.prepare([s.C for s in srcs], H, W, srcs[0].N)
        bias = None if self.bias is None else self.bias.detach()
        if self.bn is None:
            # eres: fused residual epilogue (the other branch of a ResidualUnit as a Src): out = bf16([relu](eres' + this conv))
            out, _ = engine.conv_forward(srcs, self.wp, self.Cout, self.cfg, self.taps, self.transposed, bias=bias, H=H, W=W,
                                         out_dtype=out_dtype, eres=eres)
            if training:
                self.saved = (srcs, out, H, W)
                self.node_relu, self.node_res = False, None
                if TAPE is not None:
                    TAPE.append(self)
            return Src(out)
        if not training:
            ep = self.eval_pack([s.C for s in srcs])
            if ep is not None:
                out, _ = engine.conv_forward(srcs, ep[0], self.Cout, self.cfg, self.taps, self.transposed, oshift=ep[1],
                                             orelu=relu and eres is None, H=H, W=W, out_dtype=out_dtype, eres=eres, debug_or=debug_or)
                return Src(out)
            sc, sh = self.eval_fold()
            out, _ = engine.conv_forward(srcs, self.wp, self.Cout, self.cfg, self.taps, self.transposed, oscale=sc,
                                         oshift=sh, orelu=relu and eres is None, H=H, W=W, out_dtype=out_dtype, eres=eres, debug_or=debug_or)
            return Src(out)
        tile = self.cfg[0]
        N = srcs[0].N
        npar = 4 if self.transposed else 1
        T = N * npar * ((H + tile - 1) // tile) * ((W + tile - 1) // tile)
        if self.stats is None or self.stats.shape[0] != T:
            self.stats = torch.empty((T, 2, self.Cout), dtype=torch.float32, device=self.weight.device)
        raw, _ = engine.conv_forward(srcs, self.wp, self.Cout, self.cfg, self.taps, self.transposed, stats=self.stats,
                                     H=H, W=W, out_dtype=raw_dtype())
        count = float(N * H * W * npar)
        bn = self.bn
        _lib.call('cdnet_bn_finalize_train', _lib.ptr(self.stats), T, self.Cout, count, _lib.ptr(bn.weight.detach()),
                  _lib.ptr(bn.bias.detach()), None if bias is None else _lib.ptr(bias), BN_EPS,
                  BN_MOMENTUM if getattr(bn, 'momentum', None) is None else float(bn.momentum),
                  _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), _lib.ptr(self.scale), _lib.ptr(self.shift),
                  _lib.ptr(self.save_mean), _lib.ptr(self.save_invstd), _lib.stream_ptr())
        self.saved = (srcs, raw, H, W)
        self.node_relu, self.node_res = relu, None
        self.fold_version = None               # the running statistics just moved: an eval-mode fold is stale
        if TAPE is not None:
            TAPE.append(self)
        return Src(raw, self.scale, self.shift, relu=relu)


class FuseTerm(C.Structure):                   # cdnet_fuse_term (include/cdnet_hip.h)
    _fields_ = [('x', C.c_void_p), ('Hs', C.c_int), ('Ws', C.c_int), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('f16', C.c_int), ('pad_', C.c_int)]


class GradTerm(C.Structure):                   # cdnet_grad_term
    _fields_ = [('g', C.c_void_p), ('cstride', C.c_int), ('coff', C.c_int)]


class FuseNode:
    """out = [relu](sum of 1..4 terms) through cdnet_fuse_sum: the residual adds and multi-resolution fuse sums of HRNet.
    Terms are Src objects: plain bf16 tensors or raw (fp16) BatchNorm-pending convolution outputs (their scale / shift is
    applied on the fly, without ReLU); lower-resolution terms are up-sampled bilinearly.  With `out` the result lands in
    the channel slice [out_coff, out_coff + C) of a wider tensor (torch.cat of the branches).  In a training forward the
    node goes onto the tape; backward() routes the output gradient to every term."""

    def __init__(self, name):
        self.name = name
        self.saved = None

    def forward(self, terms, relu, out=None, out_coff=0, training=False):
        """`out` (optional) is the wider [N,H,W,Ctot] tensor whose channel slice receives the sum; it also fixes H, W"""
        for t in terms:
            assert t.res is None and not t.pool and not t.relu and t.row_stride == 0 and t.x.is_contiguous()
        N, Cc = terms[0].N, terms[0].C
        H, W = (max(t.Hs for t in terms), max(t.Ws for t in terms)) if out is None else (out.shape[1], out.shape[2])
        arr = (FuseTerm * len(terms))()
        for k, t in enumerate(terms):
            assert t.C == Cc
            arr[k].x, arr[k].Hs, arr[k].Ws = t.x.data_ptr(), t.Hs, t.Ws
            arr[k].scale, arr[k].shift = (None if t.scale is None else t.scale.data_ptr()), (None if t.shift is None else t.shift.data_ptr())
            arr[k].f16 = int(t.f16)
        f32 = terms[0].f32
        assert all(t.f32 == f32 for t in terms)
        if out is None:
            o, cs = torch.empty((N, H, W, Cc), dtype=torch.float32 if f32 else torch.bfloat16, device=terms[0].x.device), Cc
        else:
            o, cs = out, out.shape[3]
        assert (o.dtype == torch.float32) == f32
        if DEBUG_NORELU:
            relu = False
        _lib.call('cdnet_fuse_sum_f32' if f32 else 'cdnet_fuse_sum', C.byref(arr), len(terms), N, H, W, Cc, int(relu), _lib.ptr(o), cs, out_coff,
                  _lib.stream_ptr())
        if training:
            self.saved = (list(terms), o, relu, H, W, Cc, out_coff)
            if TAPE is not None:
                TAPE.append(self)
        return Src(o) if out is None else None

    def backward(self, tr, grads, add):
        """tr: the Trainer (buffer pool).  Gradient of the (slice of the) output -> one contribution per term."""
        terms, o, relu, H, W, Cc, coff = self.saved
        sliced = o.shape[3] != Cc
        if sliced:
            d, dcs, dco = tr.cat_grad(o, grads), o.shape[3], coff      # the whole concatenation's gradient, summed once
            if d is None:
                return
        else:
            gl = tr.take(grads, id(o))
            if gl is None:
                return
            if relu and self._residual_tail(tr, grads, gl, terms, o, H, W, Cc):
                return
            if relu or len(gl) > 1 or gl[0].coff or gl[0].cstride not in (0, Cc):
                d = tr.buf(('dfuse', self.name), o.shape, o.dtype)
                tr.grad_sum(gl, o if relu else None, o.shape[0] * H * W, Cc, d)
            else:
                d = gl[0].t
            dcs, dco = 0, 0
        N = o.shape[0]
        for k, t in enumerate(terms):
            if t.Hs == H and t.Ws == W:
                add(t.x, tr.G(d, H, W, coff=dco, cstride=dcs))
            else:
                din = tr.buf(('dup', self.name, k), (N, t.Hs, t.Ws, Cc), d.dtype)
                _lib.call('cdnet_upsample_bilinear_backward_f32' if d.dtype == torch.float32 else 'cdnet_upsample_bilinear_backward', _lib.ptr(d),
                          N, H, W, Cc, dcs, dco, t.Hs, t.Ws, _lib.ptr(din), _lib.stream_ptr())
                add(t.x, tr.G(din, t.Hs, t.Ws))


def _residual_tail(self, tr, grads, gl, terms, o, H, W, Cc):
    """relu(bn(y) + x) with x stored and y read by this node alone (BasicBlock / Bottleneck tails, seg_hrnet_rev1.py:76-92, :113-133): the
    masked sum of the consumers' gradients is the first thing y's BatchNorm backward computes anyway (the kernels' residual form: mask read
    from the stored output, dz stored for the other branch - what the DAM head's residual units use), so the separate grad_sum pass - one
    more read and write of the tensor - is skipped and the gradient list goes to y's layer as it is.  Returns False when the shape is not that."""
    if DEBUG_NORELU:
        return False
    if len(terms) == 1:
        # relu(bn(y)) stored for a stride-2 reader (`_plain`, the down-sampling chains): y's BatchNorm backward takes the ReLU mask from its
        # own forward values - no masked sum at all
        y = terms[0]
        P = tr._producer.get(id(y.x)) if y.scale is not None else None
        if P is None or P.bn is None or getattr(P, 'node_res', None) is not None or tr._readers.get(id(y.x), 0) != 1 or id(y.x) in grads \
                or tuple(y.x.shape) != tuple(o.shape) or getattr(P, 'fused_res_of', None) is not None or (y.Hs, y.Ws) != (H, W):
            return False
        P.node_relu, P.node_res = True, None
        grads[id(y.x)] = gl
        return True
    if len(terms) != 2 or len(gl) > 3:
        return False
    bn_t = [t for t in terms if t.scale is not None]
    if not bn_t or any((t.Hs, t.Ws) != (H, W) for t in terms):
        return False
    # (two BatchNorm terms - a Bottleneck with its downsample branch: the one whose layer comes first in backward takes the list, the other
    #  one reads the dz that pass stores)
    y = max(bn_t, key=lambda t: tr._tape_pos.get(id(tr._producer.get(id(t.x))), -1))
    x = terms[1] if terms[0] is y else terms[0]
    P = tr._producer.get(id(y.x))
    if P is None or P.bn is None or getattr(P, 'node_res', None) is not None or tr._readers.get(id(y.x), 0) != 1 or id(y.x) in grads \
            or tuple(y.x.shape) != tuple(o.shape) or getattr(P, 'fused_res_of', None) is not None:
        return False
    if any(g.pooled or g.oy or g.ox or (g.Hg, g.Wg) != (H, W) for g in gl):
        return False
    P.node_relu, P.node_res, P.node_res_grad_to = 2, o, x.x
    grads[id(y.x)] = gl
    return True


FuseNode._residual_tail = _residual_tail


def input_pack(x):
    """f32 NCHW [N,C<=16,H,W] cuda -> NHWC [N,H,W,16] in the activation dtype of the current precision (bf16 / fp32)"""
    assert x.dtype == torch.float32 and x.is_cuda and x.dim() == 4
    x = x.contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N, H, W, 16), dtype=act_dtype(), device=x.device)
    _lib.call('cdnet_input_pack_f32' if PRECISION == 'fp32' else 'cdnet_input_pack', _lib.ptr(x), N, Cc, H, W, _lib.ptr(out),
              _lib.stream_ptr())
    return out


POOL_MATERIALIZE = True      # max-pooled sources are stored (measured faster than four loads + max per staged element)
RU_MATERIALIZE = True        # (CDNET_RU_FUSE=0 only) the residual units' outputs are stored once for their three consumers
RU_FUSE = True       # ResidualUnit: add + ReLU in the epilogue of its conv_1x1
# eval mode, 16-bit path: BatchNorm scale folded into the packed weights (ConvLayer.eval_pack), and a residual unit's 1x1 branch as extra K
# steps of its second 3x3 convolution (cdnet_conv_args.taps1 = 1, conv_ws16_kernel)
EVAL_FOLD_WEIGHTS = True
RU_EVAL_ONE_LAUNCH = True
RU_EVAL_POINT_DOT = True       # eval: the point feature's only reader (point_conv) rides in the unit's launch, the feature is not stored


class PointLogit:
    """what a residual unit leaves instead of its output when its only reader is a 1x1 classifier (residual_unit_eval(dot=...)): the
    classifier's logits, f32 [N,1,H,W]"""

    def __init__(self, point):
        self.point = point


def residual_unit_eval(c1, c2, cr, x, relu2=True, dot=None):
    """relu2(bn2(conv2(relu1(bn1(conv1(x))))) + conv_1x1(x)) (model_unet_rev1.py:161-170) in eval mode on the 16-bit path with TWO launches:
    conv1 (BatchNorm folded, ReLU), then conv2 with the 1x1 branch as one-tap chunks of a second source - no stored conv2 output, no
    separate 1x1 pass.  Returns the unit's output as a plain Src, or None when the shape is not conv_ws16_kernel's (the caller then takes
    the three-launch form).  `dot` = (weights f32 [Cout], bias f32 [1]) of a 1x1 classifier that is the unit's ONLY reader (the DAM head's
    point_conv over the point feature, model_unet_rev1.py:252-253): its logits leave with the launch (cdnet_conv_args.dot_out), the
    64-channel output is never stored and a PointLogit comes back - when the launch is not eligible for that, the plain Src as usual."""
    if not (RU_EVAL_ONE_LAUNCH and EVAL_FOLD_WEIGHTS) or not getattr(c2, 'fold_eval', True) or x.pool or x.C % 16 or c2.Cout % 16:
        return None
    H, W = x.logical_hw()
    if H % 16 or W % 16 or x.C != cr.Cin:
        return None
    c2.prepare([c2.Cin], H, W, x.N)
    if tuple(c2.cfg[:2]) != (16, 16):
        return None
    ep = c2.eval_pack([c1.Cout], allow_f32=True)         # (every shape question is asked before conv1 runs: an ineligible unit computes nothing twice)
    if ep is None:
        return None
    f32 = PRECISION == 'fp32'
    # the 1x1 branch's pack in c2's configuration, in a buffer of this form's own (`cr` and its packs - which the trainer's batched re-pack
    # holds raw pointers to - are left alone): the two packs lie end to end per output-channel tile
    rver = (cr.weight._version, WEIGHTS_EPOCH[0], tuple(c2.cfg), f32)
    if getattr(c2, 'ru_wr', None) is None or c2.ru_wr_version != rver:
        c2.ru_wr = engine.pack_weights(cr.weight.detach(), c2.cfg, 0, out=getattr(c2, 'ru_wr', None), split=f32)
        c2.ru_wr_version = rver
    # 16-bit path, an odd number of one-tap chunks (the first unit: 16 input channels): one more one-tap chunk of ZERO weights makes the chunk
    # count even - conv_ws16_kernel's out-image form (whole-line stores by the movers, pair requests) instead of the consumers' direct stores:
    # 548 -> ~440 us per 64 tiles on that launch (engine.conv_forward(pad_chunks=))
    pad = 1 if (not f32 and (x.C // 16) % 2 == 1 and c2.cfg[1] == 16) else 0
    ver = (c2.wpe_version, rver, None if cr.bias is None else cr.bias._version, PRECISION, pad)
    if getattr(c2, 'ru_pack', None) is None or c2.ru_pack_version != ver:
        BN = c2.cfg[2]
        ntile = -(-c2.Cout // BN)
        a, b = ep[0].view(ntile, -1), c2.ru_wr.view(ntile, -1)   # per output-channel tile: the nine-tap chunks, then the one-tap chunks
        parts = [a, b]
        if pad:
            parts.append(torch.zeros((ntile, 16 * BN), dtype=a.dtype, device=a.device))      # (a one-tap chunk: CK x BN bf16 patterns)
        c2.ru_pack = torch.cat(parts, 1).contiguous().view(-1)
        c2.ru_shift = ep[1] if cr.bias is None else ep[1] + cr.bias.detach()
        c2.ru_pack_version = ver
    hC = c1.Cout
    keep = engine.CONV_DEBUG
    # (the persistent kernel also on small launches - nothing else computes this form: a per-launch request, cdnet_conv_args.debug bit 64
    #  through engine.conv_forward(debug_or=); no module state changes, the path is re-entrant across streams)
    # which kernel serves the launch depends on its shape and flags only: asked once per shape, on a placeholder for conv1's output
    elig = c2.__dict__.setdefault('_ru_eligible', {})
    key = (x.N, H, W, x.C, hC, PRECISION, keep, bool(relu2))
    want_dot = dot is not None and RU_EVAL_POINT_DOT and relu2
    kw = dict(oshift=c2.ru_shift, H=H, W=W, taps1=1, pad_chunks=pad, debug_or=64)
    if key not in elig or (want_dot and key + ('dot',) not in elig):
        hp = Src(torch.empty((x.N, H, W, hC), dtype=act_dtype(), device=x.x.device))
        elig[key] = bool(not (keep & 32) and engine.conv_forward([hp, x], c2.ru_pack, c2.Cout, c2.cfg, 9, orelu=relu2, query_ws=True, **kw))
        if want_dot:
            d0 = (dot[0], dot[1], torch.empty((x.N, 1, H, W), dtype=torch.float32, device=x.x.device))
            elig[key + ('dot',)] = elig[key] and engine.conv_forward([hp, x], c2.ru_pack, c2.Cout, c2.cfg, 9, orelu=True, dot=d0, query_ws=True, **kw) == 2
    if not elig[key]:
        return None
    h = c1.forward([x], False, relu=True, debug_or=64)
    assert h.C == hC and not h.pool and h.scale is None
    if want_dot and elig[key + ('dot',)]:
        point = torch.empty((x.N, 1, H, W), dtype=torch.float32, device=x.x.device)
        engine.conv_forward([h, x], c2.ru_pack, c2.Cout, c2.cfg, 9, orelu=True, dot=(dot[0], dot[1], point), **kw)
        return PointLogit(point)
    out, _ = engine.conv_forward([h, x], c2.ru_pack, c2.Cout, c2.cfg, 9, orelu=relu2, **kw)
    return Src(out)


def materialize(s, H=None, W=None):
    """The Src with its pending transform applied, as a stored bf16 tensor (cdnet_src_materialize).  Returns a plain Src."""
    if H is None:
        H, W = s.logical_hw()
    out = torch.empty((s.N, H, W, s.C), dtype=(torch.float32 if s.f32 else torch.bfloat16), device=s.x.device)
    cs = engine.ConvSrc()
    s.fill(cs)
    _lib.call('cdnet_src_materialize', C.byref(cs), s.N, H, W, _lib.ptr(out), _lib.stream_ptr())
    return Src(out)


def pooled(s, ceil_mode=False, materialized=None):
    """nn.MaxPool2d(2,2[,ceil_mode]) applied after the Src's own transform.  Either a lazy view (the consumer's staging
    code takes the max of the four transformed values) or - the default, measured faster - a stored copy; `grad_to` tells
    backward that the gradient of the copy belongs to the un-pooled producer, routed through the max-pool."""
    assert not s.pool and s.res is None
    mode = 2 if ceil_mode else 1
    lazy = Src(s.x, s.scale, s.shift, relu=s.relu, pool=mode)
    if not (POOL_MATERIALIZE if materialized is None else materialized) and not s.f32:      # (the fp32 kernels read stored pools only)
        return lazy
    out = materialize(lazy)
    out.grad_to = (s.x, mode)
    return out


def pad_offsets(small_hw, big_hw):
    """F.pad(x, (dx//2, dx-dx//2, dy//2, dy-dy//2)) of model_unet_rev1.py:128-131 / unet.py:42-46 as (off_y, off_x)"""
    dy, dx = big_hw[0] - small_hw[0], big_hw[1] - small_hw[1]
    # negative differences (ceil-mode pools on odd sizes) make F.pad CROP: floor division gives the same offsets
    return (dy // 2, dx // 2)
