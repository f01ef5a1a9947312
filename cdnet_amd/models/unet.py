"""`UNet` - the plain 4-level U-Net (64 -> 1024) of the reference (models/unet.py:53-106), host-side mirror.

Same constructor (`UNet(num_classes, in_channels=3, freeze_bn=False)`), same `forward(x) -> logits [B,K,H,W]` float32,
same state_dict keys (down1..4.down_conv.{0,1,3,4}, middle_conv.*, up1..4.{up,up_conv.*}, final_conv and the reference's
never-used `up` / `beforefinal2_conv`), Kaiming-normal initialisation (:80-88).  The nn modules are parameter containers;
forward runs on the HIP kernels: encoder convs with BatchNorm+ReLU, `MaxPool2d(2, ceil_mode=True)` fused into the next
convolution's staging, `ConvTranspose2d(k2,s2)` as four sub-pixel 1x1 convolutions, `F.pad` + `cat([skip, up])` as a
virtual two-source convolution, the 64 -> K classifier per pixel.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib, runtime
from ..runtime import ConvLayer, Src, pooled, pad_offsets


class encoder(nn.Module):                  # models/unet.py:8-24
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.down_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))
        self.pool = nn.MaxPool2d(kernel_size=2, ceil_mode=True)


class decoder(nn.Module):                  # models/unet.py:27-50
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        self.up_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))


class UNet(nn.Module):
    def __init__(self, num_classes, in_channels=3, freeze_bn=False, **_):
        super().__init__()
        self.down1, self.down2 = encoder(in_channels, 64), encoder(64, 128)
        self.down3, self.down4 = encoder(128, 256), encoder(256, 512)
        self.middle_conv = nn.Sequential(
            nn.Conv2d(512, 1024, kernel_size=3, padding=1), nn.BatchNorm2d(1024), nn.ReLU(inplace=True),
            nn.Conv2d(1024, 1024, kernel_size=3, padding=1), nn.BatchNorm2d(1024), nn.ReLU(inplace=True))
        self.up1, self.up2, self.up3, self.up4 = decoder(1024, 512), decoder(512, 256), decoder(256, 128), decoder(128, 64)
        self.up = nn.ConvTranspose2d(128, 128, kernel_size=2, stride=2)            # never used (:72)
        self.beforefinal2_conv = nn.Conv2d(128, num_classes, kernel_size=1)        # never used (:73)
        self.final_conv = nn.Conv2d(64, num_classes, kernel_size=1)
        self.num_classes = num_classes
        self._initialize_weights()
        self._rt = None
        if freeze_bn:
            self.freeze_bn()

    UNUSED_PREFIXES = ('up.', 'beforefinal2_conv.')

    def _initialize_weights(self):
        for module in self.modules():
            if isinstance(module, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(module.weight)
                if module.bias is not None:
                    module.bias.data.zero_()
            elif isinstance(module, nn.BatchNorm2d):
                module.weight.data.fill_(1)
                module.bias.data.zero_()

    def freeze_bn(self):
        for module in self.modules():
            if isinstance(module, nn.BatchNorm2d):
                module.eval()

    def _build_runtime(self):
        def pair(name, seq):
            return [ConvLayer('%s.0' % name, 'conv3', seq[0].weight, seq[0].bias, seq[1]),
                    ConvLayer('%s.3' % name, 'conv3', seq[3].weight, seq[3].bias, seq[4])]
        rt = {'down': [pair('down%d.down_conv' % (i + 1), getattr(self, 'down%d' % (i + 1)).down_conv) for i in range(4)],
              'mid': pair('middle_conv', self.middle_conv), 'up': []}
        for i in range(4):
            d = getattr(self, 'up%d' % (i + 1))
            rt['up'].append((ConvLayer('up%d.up' % (i + 1), 'convT2', d.up.weight, d.up.bias, None),
                             pair('up%d.up_conv' % (i + 1), d.up_conv)))
        self._rt = rt

    def forward(self, x):
        if self._rt is None:
            self._build_runtime()
        if not x.is_cuda:
            raise RuntimeError('cdnet_amd.models.unet.UNet runs on the MI355X only (no CPU fallback)')
        training = self.training
        t = Src(runtime.input_pack(x.float()))
        t.is_input = True
        self._rt['down'][0][0].needs_input_grad = False
        skips = []
        for c1, c2 in self._rt['down']:                               # x_i, x = down_i(x)   (:91-94)
            t = c2.forward([c1.forward([t], training)], training)
            skips.append(t)
            t = pooled(t, ceil_mode=True)
        c1, c2 = self._rt['mid']
        t = c2.forward([c1.forward([t], training)], training)         # middle_conv (:95)
        for (up, (c1, c2)), skip in zip(self._rt['up'], skips[::-1]):  # up_i(x_copy, x) (:40-50)
            u = up.forward([t], training)                              # ConvTranspose2d k2 s2 (+bias), no BN
            sh, sw = skip.logical_hw()
            u.off = pad_offsets(u.logical_hw(), (sh, sw))
            t = c2.forward([c1.forward([skip, u], training, H=sh, W=sw)], training)   # cat([x_copy, x]) (:48)
        N, H, W, _ = t.x.shape
        out = torch.empty((N, self.num_classes, H, W), dtype=torch.float32, device=x.device)
        hf = runtime.head_feat(t)
        _lib.call('cdnet_final_conv1x1', C.byref(hf), _lib.ptr(self.final_conv.weight.detach().reshape(self.num_classes, 64).contiguous()),
                  _lib.ptr(self.final_conv.bias.detach()), self.num_classes, N, H, W, _lib.ptr(out), _lib.stream_ptr())
        self._last_feat = t
        return out
