"""Which convolution launches of one training step run on the producer / consumer kernel (conv_ws_kernel)?  Prints one line per launch:
sources' channels, Cout, HxW, taps, transposed, cfg, eligible."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cdnet_amd import engine, trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet

dev = torch.device('cuda:0')
torch.manual_seed(0)
m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
step = trainer.make_bench_step(m, 16, dev, 0, 1)
step = step[0] if isinstance(step, tuple) else step
for _ in range(2):
    step()
orig = engine.conv_forward
log = []


def spy(srcs, wpacked, Cout, cfg, taps=9, transposed=False, **kw):
    kw2 = dict(kw)
    kw2.pop('query_ws', None)
    el = orig(srcs, wpacked, Cout, cfg, taps=taps, transposed=transposed, query_ws=True, **kw2)
    H, W = (kw.get('H'), kw.get('W')) if kw.get('H') is not None else srcs[0].logical_hw()
    log.append(([s.C for s in srcs], Cout, H, W, taps, transposed, tuple(cfg[:3]), el, srcs[0].N))
    return orig(srcs, wpacked, Cout, cfg, taps=taps, transposed=transposed, **kw)


engine.conv_forward = spy
import cdnet_amd.runtime as rt
for mod in (rt, trainer):
    if hasattr(mod, 'conv_forward'):
        mod.conv_forward = spy
step()
torch.cuda.synchronize()
for r in log:
    print(r)
print(len(log), 'launches,', sum(1 for r in log if r[7]), 'on conv_ws')
