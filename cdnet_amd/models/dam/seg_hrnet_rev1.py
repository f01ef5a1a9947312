"""`HighResolutionNet` - HRNet-W18 + DAM head (reference: models/dam/seg_hrnet_rev1.py:289-548, `HRNet18_rev1` of
utils.chooseModel :880-882, BASELINE config 5), host-side mirror (inference and training).

Same constructor argument (`config` with `config.model['out_c']`), same `forward(x) -> (mask, point, direction)`, same
state_dict keys (conv1/bn1/conv2/bn2, layer1.*, transition1-3.*, stage2-4.*.{branches,fuse_layers}.*, the never-used
last_layer.*, the three ResidualUnits and the head).  The nn modules are parameter containers; the forward runs on the
HIP kernels of the UNet path plus two additions:
  * stride-2 3x3 convolutions (transitions :436-443, down-sampling fuse layers :228-247) = a 3x3 convolution over the
    space-to-depth view of the input (weight pack mode 6, include/cdnet_hip.h) - no strided kernel;
  * `cdnet_fuse_sum`: the residual adds of BasicBlock / Bottleneck (:76-92, :113-133), the fuse sums with bilinear
    up-sampling (:256-283) and the final F.upsample + torch.cat (:528-533).
Branch widths 18 / 36 / 72 are carried zero-padded to 32 / 48 / 80 channels (padded weights and BatchNorm rows are
zero, so the padding stays exactly zero through forward, backward and Adam).  Training: cdnet_amd.trainer.Trainer."""
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
    # runtime: zero-padded parameter storage + ConvLayers
    #
    # The kernels want channel counts in multiples of 16, so every parameter with an 18 / 36 / 72 (or RGB 3) dimension is
    # held zero-padded (`compute parameter`); the module's own nn.Parameter / BatchNorm buffer becomes a strided VIEW of the
    # leading corner of that storage, so state_dict() / load_state_dict() keep working on the reference's shapes with no
    # copies.  Two exceptions: mask_feature.conv1 / conv_1x1 read the concatenation of the four padded branches (their
    # input channels are scattered over four segments) - those are synchronised by copy (sync_real_parameters()).
    # ---------------------------------------------------------------------------------------------------
    def _compute_param(self, real, pshape, segs=None):
        """the parameter the kernels use for `real`: `real` itself when no padding is needed, else a zero-padded copy"""
        if tuple(real.shape) == tuple(pshape):
            return real
        pp = nn.Parameter(torch.zeros(pshape, dtype=torch.float32, device=real.device), requires_grad=False)
        with torch.no_grad():
            if segs is None:
                pp[tuple(slice(0, n) for n in real.shape)] = real.detach()
            else:
                for r0, n, p0 in segs:
                    pp[:, p0:p0 + n] = real.detach()[:, r0:r0 + n]
        self._slots.append((real, pp, segs))
        self._pmap[id(real)] = pp
        return pp

    def _padded(self, conv, bn, cin_segments=None, stride=1, name='hrnet'):
        """ConvLayer over the compute parameters of conv / bn.  cin_segments: [(real_start, count, padded_start)] placing
        the real input channels in the padded layout (default: one segment at 0)."""
        cout, cin, k, _ = conv.weight.shape
        cin_p = _pad16(cin) if cin_segments is None else _pad16(max(p + n for _, n, p in cin_segments))
        cout_p = _pad16(cout)
        w = self._compute_param(conv.weight, (cout_p, cin_p, k, k), cin_segments)
        bnp = None
        if bn is not None:
            bnp = bn
            if cout_p != cout:
                # shell module over the padded rows: gamma = beta = mean = 0, var = 1 there, so padded channels stay exactly 0
                bnp = nn.BatchNorm2d(cout_p, momentum=bn.momentum, eps=bn.eps).to(conv.weight.device)
                bnp.weight = self._compute_param(bn.weight, (cout_p,))
                bnp.bias = self._compute_param(bn.bias, (cout_p,))
                with torch.no_grad():
                    bnp.running_mean[:cout] = bn.running_mean
                    bnp.running_var[:cout] = bn.running_var
                bn._buffers['running_mean'] = bnp.running_mean[:cout]
                bn._buffers['running_var'] = bnp.running_var[:cout]
                bnp.train(bn.training)
        bias = None if conv.bias is None else self._compute_param(conv.bias, (cout_p,))
        kind = 'conv1' if k == 1 else ('conv3s2' if stride == 2 else 'conv3')
        L = ConvLayer(name, kind, w, bias, bnp)
        L.fold_eval = False      # eval-mode BatchNorm stays an epilogue affine here: the golden tolerances of tests/test_gpu_hrnet.py were set on it
        return L

    def _build_runtime(self):
        self._slots, self._pmap, self._nodes = [], {}, {}
        names = {id(m): n for n, m in self.named_modules()}

        def P(conv, bn, segs=None, stride=1):
            return self._padded(conv, bn, segs, stride, name=names[id(conv)])
        rt = {'stem': [P(self.conv1, self.bn1), P(self.conv2, self.bn2)]}
        rt['stem'][0].needs_input_grad = False
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
        # 18 + 36 + 72 + 144 carried as 32 + 48 + 80 + 144 = 304 channels = 19 chunks of 16: one more chunk of zeros (zero weights on the other
        # side) makes the concatenation 320 channels wide - 640-byte pixels, whole 128-byte lines per 64-channel block, and an even chunk count,
        # which the persistent convolution kernels' chunk-pair requests need (mask_feature.conv1's training forward ran on the one-tile kernel)
        ctot = (p0 + 31) // 32 * 32
        self._cat_width = ctot
        shim = types.SimpleNamespace(conv1=types.SimpleNamespace(weight=self._compute_param(mf.conv1.weight, (64, ctot, 3, 3), segs)),
                                     bn1=mf.bn1, conv2=mf.conv2, bn2=mf.bn2,
                                     conv_1x1=types.SimpleNamespace(weight=self._compute_param(mf.conv_1x1.weight, (64, ctot, 1, 1), segs),
                                                                    bias=mf.conv_1x1.bias))
        rt['ru'] = [_RU('mask_feature', shim), _RU('direction_feature', self.direction_feature),
                    _RU('point_feature', self.point_feature)]
        for ru in rt['ru']:
            for L in ru.layers():
                L.fold_eval = False
        self._rt = rt
        self.rebind_views()
        self._rt_ver = self._param_version()

    def rebind_views(self):
        """point the module's own parameters at the leading corner of their padded compute storage (and gradient)"""
        for real, pp, segs in self._slots:
            if segs is None:
                idx = tuple(slice(0, n) for n in real.shape)
                real.data = pp.data[idx]
                real.grad = None if pp.grad is None else pp.grad[idx]

    def sync_real_parameters(self):
        """copy the two concatenation-reading weights from their padded compute storage back into the module's parameters
        (the trained values live in the compute storage; every other parameter is a view and needs no copy)"""
        if self._rt is None:
            return
        with torch.no_grad():
            for real, pp, segs in self._slots:
                if segs is not None:
                    for r0, n, p0 in segs:
                        real[:, r0:r0 + n] = pp.detach()[:, p0:p0 + n]
        self._rt_ver = self._param_version()

    def state_dict(self, *args, **kwargs):
        self.sync_real_parameters()
        return super().state_dict(*args, **kwargs)

    def trainer_named_parameters(self):
        """name -> compute parameter, in forward order where it matters to the trainer (cdnet_amd.trainer.FlatState)"""
        self._ensure_runtime()
        return {n: self._pmap.get(id(p), p) for n, p in self.named_parameters()}

    def _param_version(self):
        return tuple(t._version for t in list(self.parameters()) + list(self.buffers()))

    def _ensure_runtime(self):
        dev_moved = False
        if self._rt is not None:
            views = [(real, pp) for real, pp, segs in self._slots if segs is None]
            real, pp = views[0]
            dev_moved = real.untyped_storage().data_ptr() != pp.untyped_storage().data_ptr()     # .to() / .cuda() re-made the parameters
        if self._rt is None or dev_moved:
            self._build_runtime()
        elif self._rt_ver != self._param_version():
            # load_state_dict / in-place edits went through the views; the segment-mapped weights need their scatter
            with torch.no_grad():
                for real, pp, segs in self._slots:
                    if segs is not None:
                        for r0, n, p0 in segs:
                            pp[:, p0:p0 + n] = real.detach()[:, r0:r0 + n]
            runtime.WEIGHTS_EPOCH[0] += 1                  # every packed weight / BatchNorm fold is stale
            self._rt_ver = self._param_version()

    # ---------------------------------------------------------------------------------------------------
    def _node(self, key):
        n = self._nodes.get(key)
        if n is None:
            n = self._nodes[key] = runtime.FuseNode('fuse.' + '.'.join(str(k) for k in key))
        return n

    def _fuse(self, key, terms, relu, training, out=None, out_coff=0):
        """[relu](sum of terms) - see runtime.FuseNode; terms with a pending BatchNorm are passed with their affine"""
        return self._node(key).forward(terms, relu, out=out, out_coff=out_coff, training=training)

    def _plain(self, key, s, training):
        """a conv -> BN -> ReLU output as a stored bf16 tensor (eval: it already is; train: apply the pending BatchNorm + ReLU)"""
        return s if not training else self._fuse(key, [s], True, True)

    @staticmethod
    def _s2(layer, s, relu, training):
        """3x3 stride-2 convolution: the two row-parity views of the (stored, plain) input"""
        x = s.x
        N, H, W, Cc = x.shape
        assert s.scale is None and H % 2 == 0 and W % 2 == 0, 'stride-2 layers need even sizes (the reference feeds 512x512 / 256x256 tiles)'
        views = [Src(x, view=(a * W * Cc, H // 2, W // 2, 2 * Cc, 2 * W * Cc)) for a in (0, 1)]
        return layer.forward(views, training, relu=relu and not training, H=H // 2, W=W // 2)

    def _chain(self, key, steps, t, training):
        """Sequential of (stride-2 conv, BN[, ReLU]); the last element's BatchNorm (+ReLU) is left to the consumer in training"""
        for k, (layer, relu) in enumerate(steps):
            t = self._s2(layer, t, relu, training)
            if training and relu:
                t = self._fuse(key + ('c', k), [t], True, True)
        return t

    def _basic(self, key, pair, x, training):             # BasicBlock.forward (:76-92)
        c1, c2 = pair
        if not training and runtime.RU_FUSE and x.scale is None:
            # eval: BatchNorm is folded, so the shortcut add + ReLU ride in conv2's epilogue (cdnet_conv_args.eres)
            return c2.forward([c1.forward([x], False, relu=True)], False, relu=False, eres=Src(x.x, relu=True))
        y = c2.forward([c1.forward([x], training, relu=True)], training, relu=False)
        return self._fuse(key, [y, x], True, training)

    def _bottleneck(self, key, quad, x, xres, training):  # Bottleneck.forward (:113-133); x may carry a pending BatchNorm, xres is plain
        c1, c2, c3, ds = quad
        if not training and runtime.RU_FUSE:
            r = xres if ds is None else ds.forward([x], False, relu=False)
            return c3.forward([c2.forward([c1.forward([x], False, relu=True)], False, relu=True)], False, relu=False, eres=Src(r.x, relu=True))
        y = c3.forward([c2.forward([c1.forward([x], training, relu=True)], training, relu=True)], training, relu=False)
        r = xres if ds is None else ds.forward([x], training, relu=False)
        return self._fuse(key, [y, r], True, training)

    def _module(self, key, mod, xs, training):            # HighResolutionModule.forward (:256-283)
        br, fu = mod
        xs = list(xs)
        for i, blocks in enumerate(br):
            for k, pair in enumerate(blocks):
                xs[i] = self._basic(key + ('b', i, k), pair, xs[i], training)
        outs = []
        for i in range(len(xs)):
            terms = []
            for j in range(len(xs)):
                f = fu[i][j]
                if f is None:
                    terms.append(xs[j])
                elif f[0] == 'up':
                    terms.append(f[1].forward([xs[j]], training, relu=False))   # 1x1 conv + BN at branch j's size; up-sampled in the sum
                else:
                    terms.append(self._chain(key + ('d', i, j), f[1], xs[j], training))
            terms.sort(key=lambda t: -t.Hs)               # a full-size term first (any order sums the same set)
            outs.append(self._fuse(key + ('f', i), terms, True, training))
        return outs

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('cdnet_amd HighResolutionNet runs on the MI355X only (no CPU fallback)')
        return self.forward_packed(runtime.input_pack(x.float()))

    def forward_packed(self, x16):
        """forward on packed NHWC input [N,H,W,16] (3 real channels; bf16, or fp32 in the fp32 precision mode): what cdnet_input_pack /
        cdnet_window_pack produce (sliding-window / TTA inference, cdnet_amd.utils.split_forward_views)"""
        training = self.training
        self._ensure_runtime()
        rt = self._rt
        x = x16
        assert x.shape[1] % 8 == 0 and x.shape[2] % 8 == 0, 'HRNet18_rev1 needs tile sizes divisible by 8 (three stride-2 stages)'
        t = Src(x16)
        t.is_input = True
        for L in rt['stem']:                              # conv1-bn1-relu, conv2-bn2-relu (:495-500)
            t = L.forward([t], training, relu=True)
        for k, quad in enumerate(rt['layer1']):
            t = self._bottleneck(('l1', k), quad, t, t, training)
        ys = [t]
        for si in range(3):
            tr = rt['transition%d' % (si + 1)]
            xs = []
            for i, e in enumerate(tr):                    # :503-527
                if e is None:
                    xs.append(ys[i])
                elif e[0] == 's1':
                    xs.append(self._plain(('t', si, i), e[1].forward([ys[i]], training, relu=not training), training))
                else:                                     # a new, lower-resolution branch from the last one
                    xs.append(self._chain(('t', si, i), e[1], ys[-1], training))
            for k, mod in enumerate(rt['stage%d' % (si + 2)]):
                xs = self._module(('s', si, k), mod, xs, training)
            ys = xs
        # F.upsample + torch.cat (:528-533): every branch written (up-sampled) into its slice of one padded buffer
        N, H, W, _ = ys[0].x.shape
        ctot, used = self._cat_width, sum(w for _, w in self._cat_layout)
        # one buffer per (shape, mode), kept: its padding channels are zeroed once, every forward rewrites the branch slices only.  (A training
        # forward's buffer is what the tape holds until its backward has run: an eval forward in between takes the other one.)
        key = (N, H, W, runtime.act_dtype(), x.device, bool(training))
        bufs = self.__dict__.setdefault('_cat_bufs', {})
        cat = bufs.get(key)
        if cat is None:
            if len(bufs) >= 4:
                bufs.clear()
            cat = bufs[key] = torch.empty((N, H, W, ctot), dtype=runtime.act_dtype(), device=x.device)
            if ctot > used:
                cat[..., used:].zero_()
        for k, (y, (p0, _)) in enumerate(zip(ys, self._cat_layout)):
            self._node(('cat', k)).forward([y], False, out=cat, out_coff=p0, training=training)
        f1 = rt['ru'][0].forward(Src(cat), training, store=True)
        f2 = rt['ru'][1].forward(f1, training, store=True)
        f3 = rt['ru'][2].forward(f2, training)
        return _DamUnet._head(self, (f1, f2, f3))

    head_weight_block = _DamUnet.head_weight_block


FuseTerm = runtime.FuseTerm
