"""conv_ws16_kernel against conv_ws_kernel / conv_fwd_kernel on the 16-bit path's layer shapes (cdnet_conv_args.debug: 32 = one-tile
kernel, 64 | 128 = conv_ws_kernel where it applies, 64 = conv_ws16_kernel; + ablation bits 1 = no MFMAs / fragment reads, 8 = no stores).
usage: python tools/bench_conv_ws16.py [B] [settle_s]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SETTLE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
dev = torch.device('cuda:0')


def layer(cins, Cout, H, W, relu=True, N=B):
    srcs = [engine.Src((torch.rand((N, H, W, c), device=dev) - 0.3).to(torch.bfloat16)) for c in cins]
    w = torch.randn((Cout, sum(cins), 3, 3), device=dev) * 0.06
    cfg = (16, 16, 64 if Cout > 32 else 32)
    wp = engine.pack_weights(w, cfg, 0)
    out = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device=dev)
    sh = torch.randn((Cout,), device=dev) * 0.1
    nbytes = N * H * W * (sum(cins) + Cout) * 2

    def t(dbg, n=20):
        engine.CONV_DEBUG = dbg
        run = lambda: engine.conv_forward(srcs, wp, Cout, cfg, oshift=sh, orelu=relu, out=out, H=H, W=W)
        t0, k = time.perf_counter(), 0
        while k < 3 or time.perf_counter() - t0 < SETTLE:
            run()
            k += 1
            if k % 16 == 0:
                torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        engine.CONV_DEBUG = 0
        return e0.elapsed_time(e1) / n * 1e3
    return t, nbytes


shapes = [('64->64 @256', (64,), 64, 256, 256), ('16->64 @256', (16,), 64, 256, 256), ('64+16->16 @256', (64, 16), 16, 256, 256),
          ('128+32->32 @128', (128, 32), 32, 128, 128), ('128->128 @128', (128,), 128, 128, 128), ('256->256 @64', (256,), 256, 64, 64),
          ('512->512 @32', (512,), 512, 32, 32), ('256+64->64 @64', (256, 64), 64, 64, 64)]
for name, cins, Cout, H, W in shapes:
    t, nb = layer(cins, Cout, H, W)
    row = []
    for lab, d in (('one-tile', 32), ('ws', 64 | 128), ('ws16', 64), ('ws16 no-mfma', 64 | 1), ('ws16 no-store', 64 | 8), ('ws16', 64), ('ws', 64 | 128)):
        us = t(d)
        row.append('%s %6.1f' % (lab, us))
    us = t(64)
    print('%-18s %s | ws16 %.2f TB/s' % (name, '  '.join(row), nb / us / 1e6), flush=True)
