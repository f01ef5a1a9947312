"""Ablation timing of conv_ws32_kernel on the dominant fp32 layer (3x3 64->64 @256x256 x B tiles): which half of the workgroup
bounds an interval.  debug bits: 1 consumers skip reads + MFMAs, 4 no weight DMA, 8 no out stores (an ablation branch around the
movers' halo requests or commits makes the compiler drain the loads in flight - vmcnt(0) - and is not offered).  usage: python tools/bench_conv_ws32.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda')
x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3)
w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
cfg = (16, 16, 64)
wp = engine.pack_weights(w, cfg, 0, split=True)
out = torch.empty((B, 256, 256, 64), dtype=torch.float32, device=dev)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
stats = torch.empty((B * 256, 2, 64), dtype=torch.float32, device=dev)


def run(dbg, train=False, steps=30):
    engine.CONV_DEBUG = dbg
    src = engine.Src(x, sc, sh, relu=True) if train else engine.Src(x)
    kw = dict(stats=stats) if train else {}
    for _ in range(5):
        engine.conv_forward([src], wp, 64, cfg, out=out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        engine.conv_forward([src], wp, 64, cfg, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    engine.CONV_DEBUG = 0
    return e0.elapsed_time(e1) / steps * 1e3


# warm the clocks
for _ in range(200):
    engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
torch.cuda.synchronize()
for name, dbg in (('conv_f32_kernel', 32), ('ws32', 64), ('ws32 no MFMA (movers alone)', 65), ('ws32 no DMA', 68),
                  ('ws32 no stores', 72), ('ws32 no DMA no stores', 64 | 12), ('ws32 no MFMA no DMA', 64 | 5), ('ws32 no MFMA no stores', 64 | 9)):
    print('%-50s %7.1f us   (training-mode source + statistics: %7.1f us)' % (name, run(dbg), run(dbg, True)))
