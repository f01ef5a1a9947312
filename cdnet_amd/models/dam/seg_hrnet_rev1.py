"""`HighResolutionNet` - HRNet-W18 + DAM head (reference: models/dam/seg_hrnet_rev1.py:289-548, `HRNet18_rev1` of
utils.chooseModel :880-882, BASELINE config 5), host-side mirror for INFERENCE.

Same constructor argument (`config` with `config.model['out_c']`), same `forward(x) -> (mask, point, direction)`, same
state_dict keys (conv1/bn1/conv2/bn2, layer1.*, transition1-3.*, stage2-4.*.{branches,fuse_layers}.*, the never-used
last_layer.*, the three ResidualUnits and the head).  The nn modules are parameter containers; the forward runs on the
HIP kernels of the UNet path plus two additions:
  * stride-2 3x3 convolutions (transitions :436-443, down-sampling fuse layers :228-247) = a 3x3 convolution over the
    space-to-depth view of the input (weight pack mode 6, include/cdnet_hip.h) - no strided kernel;
  * `cdnet_fuse_sum`: the residual adds of BasicBlock / Bottleneck (:76-92, :113-133), the fuse sums with bilinear
    up-sampling (:256-283) and the final F.upsample + torch.cat (:528-533).
Branch widths 18 / 36 / 72 are carried zero-padded to 32 / 48 / 80 channels (padded weights and BatchNorm rows are
zero / identity, so the padding stays exactly zero).  Training of this model is not wired (raises)."""
import ctypes as C
import types

import torch
import torch.nn as nn

from ... import _lib, runtime
from ...runtime import ConvLayer, Src
from .model_unet_rev1 import ResidualUnit, revAttention, _RU, Unet as _DamUnet

BN_MOMENTUM = 0.01                                   # seg_hrnet_rev1.py:19


def _conv3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


class BasicBlock(nn.Module):                          # :63-92
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv3(inplanes, planes, stride), _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _conv3(planes, planes), _bn(planes)
        self.downsample, self.stride = downsample, stride


class Bottleneck(nn.Module):                          # :95-133
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False), _bn(planes)
        self.conv2, self.bn2 = _conv3(planes, planes, stride), _bn(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False), _bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride


class HighResolutionModule(nn.Module):                # :136-283 (BASIC blocks, FUSE_METHOD 'SUM', multi-scale output)
    def __init__(self, num_branches, num_blocks, channels):
        super().__init__()
        self.num_branches, self.num_inchannels = num_branches, list(channels)
        self.branches = nn.ModuleList([nn.Sequential(*[BasicBlock(c, c) for _ in range(nb)]) for c, nb in zip(channels, num_blocks)])
        fuse = []
        for i in range(num_branches if num_branches > 1 else 0):
            row = []
            for j in range(num_branches):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(channels[j], channels[i], 1, 1, 0, bias=False), _bn(channels[i])))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        cout = channels[i] if last else channels[j]
                        mods = [_conv3(channels[j], cout, 2), _bn(cout)] + ([] if last else [nn.ReLU(inplace=True)])
                        steps.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*steps))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse) if num_branches > 1 else None
        self.relu = nn.ReLU(inplace=True)


def _pad16(c):
    return (c + 15) // 16 * 16


class HighResolutionNet(nn.Module):
    STAGES = (dict(modules=1, blocks=(2, 2), channels=(18, 36)),
              dict(modules=3, blocks=(2, 2, 2), channels=(18, 36, 72)),
              dict(modules=2, blocks=(2, 2, 2, 2), channels=(18, 36, 72, 144)))       # :298-327

    def __init__(self, config, **kwargs):
        super().__init__()
        out_c = config.model['out_c']
        self.conv1, self.bn1 = _conv3(3, 64), _bn(64)
        self.conv2, self.bn2 = _conv3(64, 64), _bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(
            Bottleneck(64, 64, downsample=nn.Sequential(nn.Conv2d(64, 256, kernel_size=1, stride=1, bias=False), _bn(256))),
            Bottleneck(256, 64))                                                       # _make_layer :448-463
        pre = [256]
        for si, st in enumerate(self.STAGES):
            cur = list(st['channels'])
            setattr(self, 'transition%d' % (si + 1), self._make_transition_layer(pre, cur))
            setattr(self, 'stage%d' % (si + 2),
                    nn.Sequential(*[HighResolutionModule(len(cur), st['blocks'], cur) for _ in range(st['modules'])]))
            pre = cur
        last = sum(pre)                                                                # 270
        self.last_layer = nn.Sequential(nn.Conv2d(last, last, 1, 1, 0), _bn(last), nn.ReLU(inplace=True),
                                        nn.Conv2d(last, out_c, 1, 1, 0))               # never used by forward (:535-536)
        self.mask_feature = ResidualUnit(last, 64)
        self.direction_feature = ResidualUnit(64, 64)
        self.point_feature = ResidualUnit(64, 64)
        self.point_conv = nn.Conv2d(64, 1, kernel_size=1)
        self.directionAtt = revAttention(1)
        self.direction_conv = nn.Conv2d(64, 9, kernel_size=1)
        self.maskAtt = revAttention(9)
        self.mask_conv = nn.Conv2d(64, 3, kernel_size=1)
        self._rt, self._rt_ver = None, None
        self._head_w, self._head_ver, self._head_flat = None, None, None

    UNUSED_PREFIXES = ('last_layer.',)

    @staticmethod
    def _make_transition_layer(pre, cur):             # :412-446
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(nn.Sequential(_conv3(pre[i], cur[i]), _bn(cur[i]), nn.ReLU(inplace=True)) if cur[i] != pre[i] else None)
            else:
                steps = []
                for j in range(i + 1 - len(pre)):
                    cin, cout = pre[-1], (cur[i] if j == i - len(pre) else pre[-1])
                    steps.append(nn.Sequential(_conv3(cin, cout, 2), _bn(cout), nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    # ---------------------------------------------------------------------------------------------------
    # runtime (eval): padded parameter copies + ConvLayers
    # ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _padded(conv, bn, cin_segments=None, stride=1):
        """ConvLayer over zero-padded copies of conv / bn.  cin_segments: [(real_start, count, padded_start)] placing the
        real input channels in the padded layout (default: one segment at 0)."""
        w = conv.weight.detach()
        cout, cin, k, _ = w.shape
        cin_p = _pad16(cin) if cin_segments is None else _pad16(max(p + n for _, n, p in cin_segments))
        segs = cin_segments or [(0, cin, 0)]
        cout_p = _pad16(cout)
        wp = torch.zeros((cout_p, cin_p, k, k), dtype=torch.float32, device=w.device)
        for r0, n, p0 in segs:
            wp[:cout, p0:p0 + n] = w[:, r0:r0 + n]
        bnp = None
        if bn is not None:
            bnp = nn.BatchNorm2d(cout_p).to(w.device).eval()
            with torch.no_grad():
                bnp.weight[:cout] = bn.weight
                bnp.bias.zero_()
                bnp.bias[:cout] = bn.bias
                bnp.running_mean.zero_()
                bnp.running_mean[:cout] = bn.running_mean
                bnp.running_var.fill_(1.0)
                bnp.running_var[:cout] = bn.running_var
                bnp.eps = bn.eps
        bias = None
        if conv.bias is not None:
            bias = torch.zeros((cout_p,), dtype=torch.float32, device=w.device)
            bias[:cout] = conv.bias.detach()
        kind = 'conv1' if k == 1 else ('conv3s2' if stride == 2 else 'conv3')
        return ConvLayer('hrnet', kind, wp.contiguous(), bias, bnp)

    def _build_runtime(self):
        P = self._padded
        rt = {'stem': [P(self.conv1, self.bn1), P(self.conv2, self.bn2)]}
        rt['layer1'] = []
        for b in self.layer1:
            ds = None if b.downsample is None else P(b.downsample[0], b.downsample[1])
            rt['layer1'].append((P(b.conv1, b.bn1), P(b.conv2, b.bn2), P(b.conv3, b.bn3), ds))

        def chain(seq):                                   # Sequential of Sequential(conv s2, bn[, relu])
            return [(P(st[0], st[1], stride=2), len(st) == 3) for st in seq]
        for si in range(3):
            tr = []
            for t in getattr(self, 'transition%d' % (si + 1)):
                if t is None:
                    tr.append(None)
                elif isinstance(t[0], nn.Conv2d):
                    tr.append(('s1', P(t[0], t[1])))
                else:
                    tr.append(('s2', chain(t)))
            rt['transition%d' % (si + 1)] = tr
            mods = []
            for m in getattr(self, 'stage%d' % (si + 2)):
                br = [[(P(b.conv1, b.bn1), P(b.conv2, b.bn2)) for b in seq] for seq in m.branches]
                fu = []
                for i in range(m.num_branches):
                    row = []
                    for j in range(m.num_branches):
                        f = m.fuse_layers[i][j]
                        row.append(None if f is None else (('up', P(f[0], f[1])) if j > i else ('down', chain(f))))
                    fu.append(row)
                mods.append((br, fu))
            rt['stage%d' % (si + 2)] = mods
        # mask_feature reads the concatenation of the four branches in their padded layout
        ch = self.STAGES[-1]['channels']
        segs, r0, p0 = [], 0, 0
        for c in ch:
            segs.append((r0, c, p0))
            r0 += c
            p0 += _pad16(c)
        self._cat_layout = [(p, _pad16(c)) for (_, c, p) in segs]
        mf = self.mask_feature
        shim = types.SimpleNamespace(conv1=types.SimpleNamespace(weight=P(mf.conv1, None, segs).weight), bn1=mf.bn1, conv2=mf.conv2,
                                     bn2=mf.bn2, conv_1x1=types.SimpleNamespace(weight=P(mf.conv_1x1, None, segs).weight,
                                                                                bias=mf.conv_1x1.bias))
        rt['ru'] = [_RU('mask_feature', shim), _RU('direction_feature', self.direction_feature),
                    _RU('point_feature', self.point_feature)]
        self._rt = rt
        self._rt_ver = self._param_version()

    def _param_version(self):
        return tuple(t._version for t in list(self.parameters()) + list(self.buffers()))

    # ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _fuse(terms, relu, out=None, out_coff=0):
        """terms: list of Src (plain bf16 NHWC); the first full-size one fixes H, W.  Returns Src of the sum."""
        xs = [t.x for t in terms]
        for t in terms:
            assert t.scale is None and t.res is None and not t.pool and not t.relu and t.x.dtype == torch.bfloat16
        N, _, _, Cc = xs[0].shape
        H, W = max(x.shape[1] for x in xs), max(x.shape[2] for x in xs)
        arr = (FuseTerm * len(xs))()
        for k, x in enumerate(xs):
            assert x.shape[3] == Cc and x.is_contiguous()
            arr[k].x, arr[k].Hs, arr[k].Ws = x.data_ptr(), x.shape[1], x.shape[2]
        if out is None:
            o = torch.empty((N, H, W, Cc), dtype=torch.bfloat16, device=xs[0].device)
            cs = Cc
        else:
            o, cs = out, out.shape[3]
        _lib.call('cdnet_fuse_sum', C.byref(arr), len(xs), N, H, W, Cc, int(relu), _lib.ptr(o), cs, out_coff, _lib.stream_ptr())
        return Src(o)

    @staticmethod
    def _s2(layer, s, relu):
        """3x3 stride-2 convolution: the two row-parity views of the (materialised, plain) input"""
        x = s.x
        N, H, W, Cc = x.shape
        assert H % 2 == 0 and W % 2 == 0, 'stride-2 layers need even sizes (the reference feeds 512x512 / 256x256 tiles)'
        views = [Src(x, view=(a * W * Cc, H // 2, W // 2, 2 * Cc, 2 * W * Cc)) for a in (0, 1)]
        return layer.forward(views, False, relu=relu, H=H // 2, W=W // 2)

    def _basic(self, pair, x):                            # BasicBlock.forward (:76-92)
        c1, c2 = pair
        y = c2.forward([c1.forward([x], False, relu=True)], False, relu=False)
        return self._fuse([y, x], relu=True)

    def _bottleneck(self, quad, x):                       # Bottleneck.forward (:113-133)
        c1, c2, c3, ds = quad
        y = c3.forward([c2.forward([c1.forward([x], False, relu=True)], False, relu=True)], False, relu=False)
        r = x if ds is None else ds.forward([x], False, relu=False)
        return self._fuse([y, r], relu=True)

    def _module(self, mod, xs):                           # HighResolutionModule.forward (:256-283)
        br, fu = mod
        xs = list(xs)
        for i, blocks in enumerate(br):
            for pair in blocks:
                xs[i] = self._basic(pair, xs[i])
        outs = []
        for i in range(len(xs)):
            terms = []
            for j in range(len(xs)):
                f = fu[i][j]
                if f is None:
                    terms.append(xs[j])
                elif f[0] == 'up':
                    terms.append(f[1].forward([xs[j]], False, relu=False))      # 1x1 conv + BN at branch j's size; up-sampled in _fuse
                else:
                    t = xs[j]
                    for layer, relu in f[1]:
                        t = self._s2(layer, t, relu)
                    terms.append(t)
            terms.sort(key=lambda t: -t.x.shape[1])       # a full-size term first (any order sums the same set)
            outs.append(self._fuse(terms, relu=True))
        return outs

    def forward(self, x):
        if self.training:
            raise NotImplementedError('HighResolutionNet: only the inference path is built (DESIGN.md section 8)')
        if not x.is_cuda:
            raise RuntimeError('cdnet_amd HighResolutionNet runs on the MI355X only (no CPU fallback)')
        if self._rt is None or self._rt_ver != self._param_version():
            self._build_runtime()
        rt = self._rt
        t = Src(runtime.input_pack(x.float()))
        for L in rt['stem']:                              # conv1-bn1-relu, conv2-bn2-relu (:495-500)
            t = L.forward([t], False, relu=True)
        for quad in rt['layer1']:
            t = self._bottleneck(quad, t)
        ys = [t]
        for si in range(3):
            tr = rt['transition%d' % (si + 1)]
            xs = []
            for i, e in enumerate(tr):                    # :503-527
                if e is None:
                    xs.append(ys[i])
                elif e[0] == 's1':
                    xs.append(e[1].forward([ys[i]], False, relu=True))
                else:                                     # a new, lower-resolution branch from the last one
                    u = ys[-1]
                    for layer, relu in e[1]:
                        u = self._s2(layer, u, relu)
                    xs.append(u)
            for mod in rt['stage%d' % (si + 2)]:
                xs = self._module(mod, xs)
            ys = xs
        # F.upsample + torch.cat (:528-533): every branch written (up-sampled) into its slice of one padded buffer
        N, H, W, _ = ys[0].x.shape
        ctot = sum(w for _, w in self._cat_layout)
        cat = torch.empty((N, H, W, ctot), dtype=torch.bfloat16, device=x.device)
        for y, (p0, _) in zip(ys, self._cat_layout):
            self._fuse([y], relu=False, out=cat, out_coff=p0) if y.x.shape[1] == H else self._fuse_up(y, cat, p0, H, W)
        f1 = rt['ru'][0].forward(Src(cat), False)
        f2 = rt['ru'][1].forward(f1, False)
        f3 = rt['ru'][2].forward(f2, False)
        return _DamUnet._head(self, (f1, f2, f3))

    def _fuse_up(self, y, cat, p0, H, W):
        """one lower-resolution branch, bilinearly up-sampled into its channel slice"""
        x = y.x
        N, _, _, Cc = x.shape
        arr = (FuseTerm * 1)()
        arr[0].x, arr[0].Hs, arr[0].Ws = x.data_ptr(), x.shape[1], x.shape[2]
        _lib.call('cdnet_fuse_sum', C.byref(arr), 1, N, H, W, Cc, 0, _lib.ptr(cat), cat.shape[3], p0, _lib.stream_ptr())

    head_weight_block = _DamUnet.head_weight_block


class FuseTerm(C.Structure):
    _fields_ = [('x', C.c_void_p), ('Hs', C.c_int), ('Ws', C.c_int), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('f16', C.c_int), ('pad_', C.c_int)]
