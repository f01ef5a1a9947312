"""`Unet` - the default CDNet model `UNet2RevA1_vgg16` (VGG16-BN encoder, transposed-conv decoder, direction-aware
mask head), host-side mirror of the reference's models/dam/model_unet_rev1.py:180-320.

Same constructor arguments, same `forward(x) -> (mask, point, direction)` (float32 NCHW), same state_dict key
names (backbone.N.*, upsample_blocks.i.{up,bn1,conv2,bn2}.*, {mask,direction,point}_feature.*, point_conv,
directionAtt.Conv1x1, direction_conv, maskAtt.Conv1x1, mask_conv and the reference's never-used final_conv /
child0 / child_conv1) - so checkpoints interchange.  The torch.nn modules below are PARAMETER CONTAINERS ONLY:
forward() never calls them; every convolution / BatchNorm / ReLU / pool / concat / head op runs in the HIP
kernels of libcdnet_hip.so through cdnet_amd.runtime (NHWC bf16 activations, fp32 accumulation).  There is no
CPU or eager fallback: forward() on a non-CUDA tensor raises.
"""
import torch
import torch.nn as nn

from ... import _lib, runtime
from ...runtime import ConvLayer, Src, pooled, pad_offsets

VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def vgg16_bn_features():
    """torchvision.models.vgg16_bn().features layout (children '0'..'43'), as parameter containers."""
    layers, c = [], 3
    for v in VGG16_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c = v
    return nn.Sequential(*layers)


class revAttention(nn.Module):                # model_unet_rev1.py:8-17 (weights only; fused into the head kernel)
    def __init__(self, in_channels):
        super().__init__()
        self.Conv1x1 = nn.Conv2d(in_channels, 1, kernel_size=1, bias=False)


def get_backbone(name, pretrained=True):
    """model_unet_rev1.py:22-83 for the one backbone the CDNet path uses.  There is no network access for the
    ImageNet weights: load them through load_state_dict (keys backbone.N.*)."""
    if name != 'vgg16_bn':
        raise NotImplementedError('{} backbone model is not implemented so far.'.format(name))
    return vgg16_bn_features(), ['5', '12', '22', '32', '42'], '43'


class UpsampleBlock(nn.Module):               # model_unet_rev1.py:86-143, parametric branch
    def __init__(self, ch_in, ch_out=None, skip_in=0, use_bn=True, parametric=True):
        super().__init__()
        assert parametric and use_bn, 'only the parametric (transposed-conv) + BatchNorm decoder is on the CDNet path'
        self.up = nn.ConvTranspose2d(ch_in, ch_out, kernel_size=(4, 4), stride=2, padding=1, output_padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(ch_out)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(ch_out + skip_in, ch_out, kernel_size=(3, 3), stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(ch_out)


class ResidualUnit(nn.Module):                # model_unet_rev1.py:150-170
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv_1x1 = nn.Conv2d(in_channels, out_channels, kernel_size=1)


class _RU:
    """runtime view of a ResidualUnit: three ConvLayers"""

    def __init__(self, name, m):
        self.c1 = ConvLayer(name + '.conv1', 'conv3', m.conv1.weight, None, m.bn1)
        self.c2 = ConvLayer(name + '.conv2', 'conv3', m.conv2.weight, None, m.bn2)
        self.cr = ConvLayer(name + '.conv_1x1', 'conv1', m.conv_1x1.weight, m.conv_1x1.bias, None)
        self.cr.bias_grad_from = self.c2          # d conv_1x1.bias = sum of dz = d bn2.bias

    def layers(self):
        return [self.c1, self.c2, self.cr]

    def forward(self, x, training, store=False, dot=None):
        """dot: (weights, bias) of a 1x1 classifier that is this unit's only reader (eval mode: runtime.residual_unit_eval)"""
        relu2 = not runtime.DEBUG_NORELU
        if not training:
            f = runtime.residual_unit_eval(self.c1, self.c2, self.cr, x, relu2, dot=dot)      # eval, 16-bit path: conv2 + 1x1 branch in one launch
            if f is not None:
                return f
        if runtime.RU_FUSE:
            # out = relu2(bn2(conv2(relu1(bn1(conv1(x))))) + conv_1x1(x))   (:161-170).  conv_1x1 runs LAST, with the other
            # branch in its epilogue (cdnet_conv_args.eres): the unit's output leaves as one stored bf16 tensor - no separate
            # residual tensor, no add+ReLU pass, and every consumer (next unit, head) reads a plain source
            h = self.c1.forward([x], training, relu=True)
            y = self.c2.forward([h], training, relu=False, out_dtype=runtime.raw_dtype())       # raw fp16; BatchNorm pending in training
            f = self.cr.forward([x], training, eres=Src(y.x, y.scale, y.shift, relu=relu2))
            # backward: the consumers' gradients of f go through bn2 first (ReLU mask read from f), its dz is conv_1x1's gradient
            self.c2.node_relu, self.c2.node_res = (2 if relu2 else 0), f.x
            self.cr.fused_res_of = self.c2
            return f
        self.cr.fused_res_of = None
        # the pre-activation pair (bn2 output, residual) is kept in fp16: it is only ever read through the consumer's
        # add+ReLU transform, never as an MFMA operand
        r = self.cr.forward([x], training, out_dtype=runtime.raw_dtype())               # residual = conv_1x1(x)   (:162)
        h = self.c1.forward([x], training, relu=True)                             # relu1(bn1(conv1(x)))     (:163-165)
        y = self.c2.forward([h], training, relu=False, out_dtype=runtime.raw_dtype())   # bn2(conv2(.))            (:166-167)
        self.c2.node_relu, self.c2.node_res = relu2, r.x         # how consumers (and backward) see the unit's output
        out = Src(y.x, y.scale, y.shift, relu=relu2, res=r.x)    # relu2(out + residual)             (:168-169)
        if store and runtime.RU_MATERIALIZE:
            # three consumers (the next unit's two convolutions and the head) would each redo BatchNorm + add + ReLU over
            # two fp16 tensors: one stored bf16 copy is cheaper (measured); its gradient belongs to bn2's raw output
            lazy, out = out, runtime.materialize(out)
            out.grad_to = (lazy.x, 0)
        return out


class Unet(nn.Module):
    # head wiring: 'rev1' = the direction-aware-mask head of CDNet; the ablation models of models/dam/model_unet_MandD*.py reuse
    # this class with VARIANT 'MandD' (mask + direction, no gates, no point branch) / 'MandDandP' (plus the point branch)
    VARIANT = 'rev1'
    DIRECTION_OUT = 9

    def __init__(self, backbone_name='vgg16_bn', pretrained=True, encoder_freeze=False, classes=21,
                 decoder_filters=(256, 128, 64, 32, 16), parametric_upsampling=True, shortcut_features='default',
                 decoder_use_batchnorm=True):
        super().__init__()
        self.backbone_name = backbone_name
        self.backbone, self.shortcut_features, self.bb_out_name = get_backbone(backbone_name, pretrained=pretrained)
        shortcut_chs, bb_out_chs = [64, 128, 256, 512, 512], 512       # infer_skip_channels (:289-305) for vgg16_bn
        decoder_filters = decoder_filters[:len(self.shortcut_features)]
        decoder_filters_in = [bb_out_chs] + list(decoder_filters[:-1])
        num_blocks = len(self.shortcut_features)
        self.upsample_blocks = nn.ModuleList()
        for i, (fin, fout) in enumerate(zip(decoder_filters_in, decoder_filters)):
            self.upsample_blocks.append(UpsampleBlock(fin, fout, skip_in=shortcut_chs[num_blocks - i - 1],
                                                      parametric=parametric_upsampling, use_bn=decoder_use_batchnorm))
        self.final_conv = nn.Conv2d(decoder_filters[-1], classes, kernel_size=(1, 1))          # never used (:213)
        self.replaced_conv1 = False
        self.child0 = nn.Conv2d(1, 64, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1))       # never used (:220)
        self.child_conv1 = nn.Conv2d(1, 64, kernel_size=(7, 7), stride=(2, 2), padding=(3, 3), bias=False)
        self.mask_feature = ResidualUnit(decoder_filters[-1], 64)
        self.direction_feature = ResidualUnit(64, 64)
        self.point_feature = ResidualUnit(64, 64)
        self.point_conv = nn.Conv2d(64, 1, kernel_size=1)
        self.directionAtt = revAttention(1)
        self.direction_conv = nn.Conv2d(64, self.DIRECTION_OUT, kernel_size=1)
        self.maskAtt = revAttention(self.DIRECTION_OUT)
        self.mask_conv = nn.Conv2d(64, 3, kernel_size=1)
        if self.VARIANT != 'rev1':
            self.residual = ResidualUnit(64, 64)            # model_unet_MandD.py:234
        if encoder_freeze:
            self.freeze_encoder()
        self._rt = None
        self._head_w = None
        self._head_ver = None

    def freeze_encoder(self):
        for p in self.backbone.parameters():
            p.requires_grad = False

    # parameters that never take part in forward for 3-channel input: no gradient, excluded from DDP
    UNUSED_PREFIXES = ('final_conv.', 'child0.', 'child_conv1.')

    # ---------------------------------------------------------------------------------------------------
    def _build_runtime(self):
        enc, i = [], 0
        mods = list(self.backbone.children())
        while i < len(mods):
            if isinstance(mods[i], nn.Conv2d):
                enc.append(('conv', ConvLayer('backbone.%d' % i, 'conv3', mods[i].weight, mods[i].bias, mods[i + 1]), str(i + 2)))
                i += 3
            else:
                enc.append(('pool', None, str(i)))
                i += 1
        dec = []
        for k, b in enumerate(self.upsample_blocks):
            dec.append((ConvLayer('upsample_blocks.%d.up' % k, 'convT4', b.up.weight, None, b.bn1),
                        ConvLayer('upsample_blocks.%d.conv2' % k, 'conv3', b.conv2.weight, None, b.bn2)))
        self._rt = dict(enc=enc, dec=dec, ru=[_RU('mask_feature', self.mask_feature),
                                               _RU('direction_feature', self.direction_feature),
                                               _RU('point_feature', self.point_feature)])
        if self.VARIANT != 'rev1':
            self._rt['ru_res'] = _RU('residual', self.residual)

    def conv_layers(self):
        if self._rt is None:
            self._build_runtime()
        out = [l for kind, l, _ in self._rt['enc'] if kind == 'conv']
        for up, c2 in self._rt['dec']:
            out += [up, c2]
        for ru in self._rt['ru']:
            out += ru.layers()
        if 'ru_res' in self._rt:
            out += self._rt['ru_res'].layers()
        return out

    def head_weight_block(self):
        """f32 device block in the CDNET_HEAD_WEIGHT_FLOATS layout (include/cdnet_hip.h)"""
        ps = [self.point_conv.weight, self.direction_conv.weight, self.mask_conv.weight, self.point_conv.bias,
              self.direction_conv.bias, self.mask_conv.bias, self.directionAtt.Conv1x1.weight, self.maskAtt.Conv1x1.weight]
        if getattr(self, '_head_flat', None) is not None:
            return self._head_flat                      # the trainer keeps these parameters contiguous in this layout
        ver = tuple(p._version for p in ps)
        if self._head_w is None or self._head_ver != ver:
            with torch.no_grad():
                self._head_w = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
            assert self._head_w.numel() == 855
            self._head_ver = ver
        return self._head_w

    def forward_features(self, x, training):
        """runs the encoder/decoder/residual units; returns the three head features as Src (eval mode, 16-bit path: the third one is a
        runtime.PointLogit - the point feature's only reader is point_conv, its logits come with the unit's launch)"""
        if self._rt is None:
            self._build_runtime()
        if not x.is_cuda:
            raise RuntimeError('cdnet_amd.models.dam.model_unet_rev1.Unet runs on the MI355X only (no CPU fallback)')
        assert x.shape[1] == 3, 'the CDNet path feeds 3-channel tiles (child0/child_conv1 branches are never taken)'
        return self.forward_features_packed(runtime.input_pack(x.float()), training)

    def forward_features_packed(self, x16, training):
        """x16: bf16 NHWC [N,H,W,16] (3 real channels) - the form cdnet_input_pack / cdnet_window_pack produce"""
        if self._rt is None:
            self._build_runtime()
        t = Src(x16)
        t.is_input = True
        self._rt['enc'][0][1].needs_input_grad = False
        feats = {}
        enc, tp = self._rt['enc'], None
        for idx, (kind, layer, out_name) in enumerate(enc):         # forward_backbone (:268-287)
            if kind == 'conv':
                if not training and idx + 1 < len(enc) and enc[idx + 1][0] == 'pool' and not getattr(runtime, 'DEBUG_NORELU', False):
                    t, tp = layer.forward_eval_pool([t])           # eval: the max-pool that follows leaves with the convolution's stores
                else:
                    t = layer.forward([t], training)
            else:
                t, tp = (tp if tp is not None else pooled(t)), None
            if out_name in self.shortcut_features:
                feats[out_name] = t
            if out_name == self.bb_out_name:
                break
        for skip_name, (up, conv2) in zip(self.shortcut_features[::-1], self._rt['dec']):   # :250-252
            skip = feats[skip_name]
            u = up.forward([t], training)                           # relu(bn1(up(x)))   (:119-123)
            sh, sw = skip.logical_hw()
            uh, uw = u.logical_hw()
            u.off = pad_offsets((uh, uw), (sh, sw))                # F.pad (:126-131)
            tt = None if training else conv2.forward_eval_swapped([u, skip], H=sh, W=sw)      # (eval, 16-bit: an odd chunk count made even - runtime.py)
            t = tt if tt is not None else conv2.forward([u, skip], training, H=sh, W=sw)      # cat([x, skip]) -> conv2 -> bn2 -> relu (:133-141)
        f1 = self._rt['ru'][0].forward(t, training, store=True)
        f2 = self._rt['ru'][1].forward(f1, training, store=True)
        if self.VARIANT == 'rev1':
            dot = None
            if not training and runtime.PRECISION != 'fp32':
                # the point feature is read by point_conv alone (:252-253): its logits leave with the unit's launch
                hw = self.head_weight_block()
                dot = (hw[0:64], hw[832:833])
            f3 = self._rt['ru'][2].forward(f2, training, dot=dot)
            return f1, f2, f3
        # ablation heads (model_unet_MandD.py:254-266, model_unet_MandDandP.py:254-268)
        f3 = self._rt['ru'][2].forward(f2, training) if self.VARIANT == 'MandDandP' else None
        f1m = self._rt['ru_res'].forward(f1, training)
        return f1m, f2, f3

    def forward(self, *input):
        x = input[0]
        return self._head(self.forward_features(x, self.training))

    def forward_packed(self, x16):
        """forward on already packed bf16 NHWC windows (sliding-window / TTA inference)"""
        return self._head(self.forward_features_packed(x16, self.training))

    def _plain_head(self, feats):
        """ablation heads: plain 1x1 classifiers on the features, no attention gates (cdnet_final_conv1x1)"""
        import ctypes as C
        f1m, f2, f3 = feats
        self._last_feats = feats
        N, H, W, _ = f1m.x.shape
        dev = f1m.x.device

        def cls(f, conv):
            K = conv.out_channels
            out = torch.empty((N, K, H, W), dtype=torch.float32, device=dev)
            hf = runtime.head_feat(f)
            w = conv.weight.detach().reshape(K, 64).contiguous()
            _lib.call('cdnet_final_conv1x1', C.byref(hf), _lib.ptr(w), _lib.ptr(conv.bias.detach()), K, N, H, W, _lib.ptr(out), _lib.stream_ptr())
            return out
        mask, direction = cls(f1m, self.mask_conv), cls(f2, self.direction_conv)
        if self.VARIANT == 'MandDandP':
            return mask, cls(f3, self.point_conv), direction
        return mask, direction

    def _head(self, feats):
        if getattr(self, 'VARIANT', 'rev1') != 'rev1':
            return self._plain_head(feats)
        f1, f2, f3 = feats
        N, H, W, _ = f1.x.shape
        dev = f1.x.device
        mask = torch.empty((N, 3, H, W), dtype=torch.float32, device=dev)
        direction = torch.empty((N, 9, H, W), dtype=torch.float32, device=dev)
        import ctypes as C
        if isinstance(f3, runtime.PointLogit):
            # the point logits came with the point feature's launch: the head reads them (f3.raw = NULL) and writes mask / direction
            point = f3.point
            hf = [runtime.head_feat(f1), runtime.head_feat(f2), runtime.HeadFeat()]
        else:
            point = torch.empty((N, 1, H, W), dtype=torch.float32, device=dev)
            hf = [runtime.head_feat(f) for f in (f1, f2, f3)]
        _lib.call('cdnet_dam_head_forward', C.byref(hf[0]), C.byref(hf[1]), C.byref(hf[2]), _lib.ptr(self.head_weight_block()),
                  N, H, W, _lib.ptr(mask), _lib.ptr(point), _lib.ptr(direction), _lib.stream_ptr())
        self._last_feats = (f1, f2, f3)
        return mask, point, direction
