"""End-to-end check of one training step (HIP) against the fp32 oracle autograd step (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cdnet_amd import synth, trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet
from oracle import models as om
from oracle import train as ot

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
ref = om.Unet()
for mod in ref.modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0, 0.2)
m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
m.load_state_dict(ref.state_dict())
m = m.cuda()
lab, dirn, point, weight = synth.train_targets(B, S, S, 21)
x = torch.from_numpy(synth.det_input((B, 3, S, S), 9))
tl, td, tp, tw = [torch.from_numpy(a) for a in (lab, dirn, point, weight)]

# oracle: losses + grads (no optimizer step yet)
ref.train()
conv_out = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); conv_out[name] = out
    return f
for n_, mod_ in ref.named_modules():
    if isinstance(mod_, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and '-e' not in sys.argv:
        mod_.register_forward_hook(hook(n_))
from oracle import emulate
mask, pt, dr = emulate.dam_unet_forward(ref, x) if '-e' in sys.argv else ref(x)
L = ot.dam_losses(mask, pt, dr, tl, td, tp, tw)
L['total'].backward()
print('oracle losses', {k: round(float(v), 5) for k, v in L.items()})

tr = trainer.Trainer(m)
dev = torch.device('cuda:0')
o = tr.forward(x.to(dev))
dm, dp, dd = tr.loss_and_grads(o[0], o[1], o[2], tl.to(dev), td.to(dev), tp.to(dev), tw[:, 0].contiguous().to(dev))
torch.cuda.synchronize()
print('hip losses [total,dce,wdice,mse,ce,dice]', [round(float(v), 5) for v in tr.losses.cpu()])
tr.backward(dm, dp, dd)
torch.cuda.synchronize()
print('%-40s %10s' % ('d(raw conv output)', 'rel.err'))
for L in reversed(tr.tape):
    key = ('draw', L.name)
    if key in tr._bufs and L.name in conv_out and conv_out[L.name].grad is not None:
        g = tr._bufs[key].float().cpu().permute(0, 3, 1, 2)
        rg = conv_out[L.name].grad
        print('%-40s %10.4f  |ref| %.4g |hip| %.4g' % (L.name, (g - rg).norm().item() / (rg.norm().item() + 1e-12), rg.norm().item(), g.norm().item()))
named = dict(m.named_parameters())
rnamed = dict(ref.named_parameters())
rows = []
for n in tr.flat.order:
    if n.startswith(m.UNUSED_PREFIXES):
        continue
    g = named[n].grad.detach().float().cpu()
    rg = rnamed[n].grad
    if rg is None:
        rows.append((n, float('nan'), 0, 0)); continue
    den = rg.norm().item() + 1e-12
    rows.append((n, (g - rg).norm().item() / den, den, g.norm().item()))
worst = sorted(rows, key=lambda r: -r[1] if r[1] == r[1] else 0)
print('%-44s %10s %12s %12s' % ('param', 'rel.err', '|ref grad|', '|hip grad|'))
show = rows if '-a' in sys.argv else worst[:25]
for r in show:
    print('%-44s %10.4f %12.5g %12.5g' % r)
rel = np.array([r[1] for r in rows if r[1] == r[1]])
print('median rel err %.4f  max %.4f  n=%d' % (np.median(rel), rel.max(), len(rel)))
