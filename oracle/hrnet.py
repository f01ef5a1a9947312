"""ORACLE (test infrastructure): fp32 PyTorch restatement of HRNet18_rev1 (models/dam/seg_hrnet_rev1.py:63-548) with the
reference's module / parameter names, so that a reference state_dict loads unchanged.  Pinned by tests/golden/hrnet_fwd.npz
(eval outputs of the reference) and tests/golden/hrnet_train.npz (train-mode loss and gradient norms of the reference)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .models import ResidualUnit, revAttention

BN_MOMENTUM = 0.01                                        # :19


def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


class BasicBlock(nn.Module):                              # :63-92
    expansion = 1

    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1, self.bn1, self.conv2, self.bn2 = _c3(inplanes, planes), _bn(planes), _c3(planes, planes), _bn(planes)
        self.relu = nn.ReLU(inplace=False)

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + x)


class Bottleneck(nn.Module):                              # :95-133
    def __init__(self, inplanes, planes, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(inplanes, planes, 1, bias=False), _bn(planes)
        self.conv2, self.bn2 = _c3(planes, planes), _bn(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), _bn(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class HighResolutionModule(nn.Module):                    # :136-283
    def __init__(self, num_blocks, channels):
        super().__init__()
        nb = len(channels)
        self.branches = nn.ModuleList([nn.Sequential(*[BasicBlock(c, c) for _ in range(k)]) for c, k in zip(channels, num_blocks)])
        rows = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(channels[j], channels[i], 1, 1, 0, bias=False), _bn(channels[i])))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        cout = channels[i] if last else channels[j]
                        steps.append(nn.Sequential(*([_c3(channels[j], cout, 2), _bn(cout)] + ([] if last else [nn.ReLU(inplace=False)]))))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)
        self.relu = nn.ReLU(inplace=False)

    def forward(self, xs):
        xs = [b(x) for b, x in zip(self.branches, xs)]
        out = []
        for i in range(len(xs)):
            y = xs[0] if i == 0 else self.fuse_layers[i][0](xs[0])
            for j in range(1, len(xs)):
                if i == j:
                    y = y + xs[j]
                elif j > i:
                    y = y + F.interpolate(self.fuse_layers[i][j](xs[j]), size=xs[i].shape[-2:], mode='bilinear', align_corners=False)
                else:
                    y = y + self.fuse_layers[i][j](xs[j])
            out.append(self.relu(y))
        return out


class HighResolutionNet(nn.Module):                       # :289-548
    STAGES = (((2, 2), (18, 36), 1), ((2, 2, 2), (18, 36, 72), 3), ((2, 2, 2, 2), (18, 36, 72, 144), 2))

    def __init__(self, out_c=3):
        super().__init__()
        self.conv1, self.bn1, self.conv2, self.bn2 = _c3(3, 64), _bn(64), _c3(64, 64), _bn(64)
        self.relu = nn.ReLU(inplace=False)
        self.layer1 = nn.Sequential(Bottleneck(64, 64, nn.Sequential(nn.Conv2d(64, 256, 1, 1, bias=False), _bn(256))), Bottleneck(256, 64))
        pre = [256]
        for si, (blocks, ch, nmod) in enumerate(self.STAGES):
            tr = []
            for i in range(len(ch)):
                if i < len(pre):
                    tr.append(nn.Sequential(_c3(pre[i], ch[i]), _bn(ch[i]), nn.ReLU(inplace=False)) if ch[i] != pre[i] else None)
                else:
                    steps = []
                    for j in range(i + 1 - len(pre)):
                        cout = ch[i] if j == i - len(pre) else pre[-1]
                        steps.append(nn.Sequential(_c3(pre[-1], cout, 2), _bn(cout), nn.ReLU(inplace=False)))
                    tr.append(nn.Sequential(*steps))
            setattr(self, 'transition%d' % (si + 1), nn.ModuleList(tr))
            setattr(self, 'stage%d' % (si + 2), nn.Sequential(*[HighResolutionModule(blocks, ch) for _ in range(nmod)]))
            pre = list(ch)
        last = sum(pre)
        self.last_layer = nn.Sequential(nn.Conv2d(last, last, 1), _bn(last), nn.ReLU(inplace=False), nn.Conv2d(last, out_c, 1))   # unused
        self.mask_feature, self.direction_feature, self.point_feature = ResidualUnit(last, 64), ResidualUnit(64, 64), ResidualUnit(64, 64)
        self.point_conv = nn.Conv2d(64, 1, kernel_size=1)
        self.directionAtt = revAttention(1)
        self.direction_conv = nn.Conv2d(64, 9, kernel_size=1)
        self.maskAtt = revAttention(9)
        self.mask_conv = nn.Conv2d(64, 3, kernel_size=1)

    def forward(self, x):
        x = self.relu(self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x))))))
        x = self.layer1(x)
        ys = [x]
        for si in range(3):
            tr = getattr(self, 'transition%d' % (si + 1))
            xs = []
            for i, t in enumerate(tr):
                if t is None:
                    xs.append(ys[i])
                elif i < len(ys):
                    xs.append(t(ys[i]))
                else:
                    xs.append(t(ys[-1]))
            for m in getattr(self, 'stage%d' % (si + 2)):
                xs = m(xs)
            ys = xs
        size = ys[0].shape[-2:]
        x = torch.cat([ys[0]] + [F.interpolate(y, size=size, mode='bilinear', align_corners=False) for y in ys[1:]], 1)
        f1 = self.mask_feature(x)
        f2 = self.direction_feature(f1)
        f3 = self.point_feature(f2)
        point = self.point_conv(f3)
        direction = self.direction_conv(self.directionAtt(f2, point))
        mask = self.mask_conv(self.maskAtt(f1, direction))
        return mask, point, direction
