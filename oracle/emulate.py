"""ORACLE (test infrastructure): the fp32 oracle network evaluated with the HIP path's ROUNDING POINTS emulated
(straight-through): weights and activated tensors rounded to bf16, raw pre-BatchNorm outputs and residual branches to
fp16, BatchNorm statistics from the unrounded fp32 accumulators.  Everything else (autograd, fp32 accumulation) is plain
PyTorch.  Purpose: in a ReLU network a 1e-2 forward perturbation flips ~1 % of the ReLU / max-pool decisions, which
alone moves gradients by ~10 %; comparing the HIP training step against THIS model isolates orchestration / kernel
errors from that inherent low-precision effect.  The unrounded oracle (oracle/models.py) stays the accuracy reference.
"""
import os
import torch
import torch.nn.functional as F

NORELU = bool(int(os.environ.get('CDNET_DEBUG_NORELU', '0')))      # debug aid: linearised network (tools/debug_train.py)


QUANT = True       # False: no rounding points at all (the fp32-precision path's counterpart; NORELU still applies)


def q_bf(x):
    if not QUANT:
        return x
    return x + (x.detach().to(torch.bfloat16).float() - x.detach())


def q_h(x):
    if not QUANT:
        return x
    return x + (x.detach().to(torch.float16).float() - x.detach())


def _bn_train(raw, bn, relu=True, res=None, eps=1e-5):
    """y = affine(q_h(raw)) with batch statistics of the unrounded raw; optional residual (already fp16-rounded)."""
    mean = raw.mean((0, 2, 3), keepdim=True)
    var = raw.var((0, 2, 3), unbiased=False, keepdim=True)
    invstd = 1.0 / torch.sqrt(var + eps)
    scale = bn.weight.view(1, -1, 1, 1) * invstd
    shift = bn.bias.view(1, -1, 1, 1) - mean * scale
    y = q_h(raw) * scale + shift
    if res is not None:
        y = y + res
    if relu and not NORELU:
        y = F.relu(y)
    return q_bf(y)


def conv_bn(x, conv, bn, relu=True, res=None):
    raw = F.conv2d(x, q_bf(conv.weight), None, padding=conv.padding)     # bias cancels under batch-stat BN
    return _bn_train(raw, bn, relu, res)


def residual_unit(x, ru):
    r = q_h(F.conv2d(x, q_bf(ru.conv_1x1.weight), ru.conv_1x1.bias))
    h = conv_bn(x, ru.conv1, ru.bn1)
    return conv_bn(h, ru.conv2, ru.bn2, relu=True, res=r)


def dam_unet_forward(net, x):
    """training-mode forward of oracle.models.Unet with the HIP rounding points"""
    x = q_bf(x)
    feats = {}
    mods = list(net.backbone.named_children())
    i = 0
    while i < len(mods):
        name, m = mods[i]
        if isinstance(m, torch.nn.Conv2d):
            x = conv_bn(x, m, mods[i + 1][1])
            name = mods[i + 2][0]
            i += 3
        else:
            x = F.max_pool2d(x, 2)
            i += 1
        if name in net.SKIPS:
            feats[name] = x
        if name == net.BB_OUT:
            break
    for skip_name, blk in zip(net.SKIPS[::-1], net.upsample_blocks):
        raw = F.conv_transpose2d(x, q_bf(blk.up.weight), None, stride=2, padding=1)
        u = _bn_train(raw, blk.bn1)
        skip = feats[skip_name]
        dy, dx = skip.size(2) - u.size(2), skip.size(3) - u.size(3)
        u = F.pad(u, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        x = conv_bn(torch.cat([u, skip], 1), blk.conv2, blk.bn2)
    f1 = residual_unit(x, net.mask_feature)
    f2 = residual_unit(f1, net.direction_feature)
    if getattr(net, 'variant', 'rev1') != 'rev1':               # ablation heads: plain classifiers, no gates
        direction = net.direction_conv(f2)
        mask = net.mask_conv(residual_unit(f1, net.residual))
        if net.variant == 'MandDandP':
            return mask, net.point_conv(residual_unit(f2, net.point_feature)), direction
        return mask, direction
    f3 = residual_unit(f2, net.point_feature)
    point = net.point_conv(f3)
    direction = net.direction_conv(net.directionAtt(f2, point))
    mask = net.mask_conv(net.maskAtt(f1, direction))
    return mask, point, direction
