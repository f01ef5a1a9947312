"""Layer-by-layer comparison of the HIP DAM-Unet against the fp32 oracle (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cdnet_amd import synth
from cdnet_amd.models.dam.model_unet_rev1 import Unet
from oracle import models as om

mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
B, S = int(sys.argv[2]) if len(sys.argv) > 2 else 2, int(sys.argv[3]) if len(sys.argv) > 3 else 64
init = sys.argv[4] if len(sys.argv) > 4 else 'det'
torch.manual_seed(0)
ref = om.Unet()
if init == 'det':
    om.det_fill(ref)
else:
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0, 0.2)
m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
m.load_state_dict(ref.state_dict())
m = m.cuda()
if mode == 'train':
    m.train(); ref.train()
else:
    m.eval(); ref.eval()
x = torch.from_numpy(synth.det_input((B, 3, S, S), 1))
acts = {}
def hook(name):
    def f(mod, inp, out):
        acts[name] = out.detach().clone()
    return f
for n, mod in ref.named_modules():
    if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        mod.register_forward_hook(hook(n))
with torch.no_grad():
    want = ref(x)
    got = m(x.cuda())
def act_of(layer):
    srcs, raw, H, W = layer.saved if layer.saved is not None else (None, None, None, None)
    return raw
print('%-34s %10s %10s %10s' % ('layer (pre-activation BN output)', 'scale', 'maxerr', 'meanerr'))
for L in m.conv_layers():
    if mode != 'train':
        break
    raw = L.saved[1].float().cpu().permute(0, 3, 1, 2)
    if L.bn is not None:
        bn_name = L.name.replace('.up', '.bn1').replace('.conv2', '.bn2').replace('.conv1', '.bn1')
        if L.name.startswith('backbone.'):
            bn_name = 'backbone.%d' % (int(L.name.split('.')[1]) + 1)
        y = raw * L.scale.cpu().view(1, -1, 1, 1) + L.shift.cpu().view(1, -1, 1, 1)
        w = acts[bn_name]
    else:
        y = raw
        w = acts[L.name]
    e = (y - w).abs()
    print('%-34s %10.4f %10.4f %10.5f   minvar-ish invstd max %s' % (L.name, w.abs().max(), e.max(), e.mean(),
          ('%.1f' % float(L.save_invstd.max())) if L.bn is not None else '-'))
for n, g, w in zip(('mask', 'point', 'direction'), got, want):
    e = (g.cpu() - w).abs()
    print(n, 'scale %.3f max err %.4f mean err %.5f' % (w.abs().max(), e.max(), e.mean()))
