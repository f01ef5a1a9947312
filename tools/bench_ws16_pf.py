"""dominant layer (3x3 64->64 @256x256) on conv_ws16_kernel, 16 and 64 tiles, with the ablations (8 no stores, 2 no halo requests, 1 no MFMAs);
the round-4 A/B of the mover variants (DMA / 4, 8, 12 chunks in flight: profiles/HISTORY.md) was taken with this script under CDNET_WS16_PF"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine
dev = torch.device('cuda:0')
for B in (16, 64):
    x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
    w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
    cfg = (16, 16, 64)
    wp = engine.pack_weights(w, cfg, 0)
    out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
    res = []
    for dbg in (64, 64 | 8, 64 | 2, 64 | 2 | 8, 64 | 1):
        engine.CONV_DEBUG = dbg
        run = lambda: engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
        t0, k = time.perf_counter(), 0
        while k < 3 or time.perf_counter() - t0 < 0.7:
            run(); k += 1
            if k % 16 == 0: torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    engine.CONV_DEBUG = 0
    print('PF=%s B=%d: ws16 %.1f  nostore %.1f  noloads %.1f  noloads-nostore %.1f  nomfma %.1f us' % (os.environ.get('CDNET_WS16_PF', 'default'), B, *res), flush=True)
