"""is the dominant layer power / clock bound?  the same launches on random and on all-zero operands (MI355X_MICROARCH.md, DVFS give-back 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine
dev = torch.device('cuda:0')
for B in (16, 64):
    for zero in (False, True):
        x = torch.zeros((B, 256, 256, 64), device=dev, dtype=torch.bfloat16) if zero else (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
        w = torch.zeros((64, 64, 3, 3), device=dev) if zero else torch.randn((64, 64, 3, 3), device=dev) * 0.06
        cfg = (16, 16, 64)
        wp = engine.pack_weights(w, cfg, 0)
        out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
        res = []
        for dbg in (64 | 128, 64, 64 | 8, 64 | 1, 64 | 2, 64 | 2 | 8, 64 | 16):
            engine.CONV_DEBUG = dbg
            run = lambda: engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
            t0, k = time.perf_counter(), 0
            while k < 3 or time.perf_counter() - t0 < 1.0:
                run(); k += 1
                if k % 16 == 0: torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1e3)
        engine.CONV_DEBUG = 0
        print('B=%d %s: ws %.1f  ws16 %.1f  ws16-nostore %.1f  ws16-nomfma %.1f  ws16-noloads %.1f  ws16-noloads-nostore %.1f  ws16-prio %.1f us' % (B, 'ZERO  ' if zero else 'random', *res), flush=True)
