"""Same-box A/B of conv_ws16_kernel builds on the dominant 16-bit layers (3x3 64 -> 64 @256x256 at 16 and 64 tiles, 256 -> 256 @64x64 at 16):
   bash tools/build_variant.sh nopair "-DCDNET_WS16_PAIR_DEFAULT=0" conv16ws.hip
   for L in nopair "" nopair ""; do CDNET_LIB_PATH=${L:+$PWD/cdnet_amd/libcdnet_hip_$L.so} python tools/ab_ws16.py; done
(alternate the builds: the boxes of the pool and the chip's clock state differ by more than most changes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine
dev = torch.device('cuda:0')
print('lib=%s' % (os.path.basename(os.environ.get('CDNET_LIB_PATH', 'libcdnet_hip.so'))))
for B in (16, 64):
    for (cin, cout, taps1) in ((64, 64, 0), (16, 64, 0), (256, 256, 0)):
        if cin == 256 and B == 64:
            continue
        HW = 64 if cin == 256 else 256
        x = (torch.rand((B, HW, HW, cin), device=dev) - 0.3).to(torch.bfloat16)
        w = torch.randn((cout, cin, 3, 3), device=dev) * 0.06
        cfg = (16, 16, 64)
        wp = engine.pack_weights(w, cfg, 0)
        out = torch.empty((B, HW, HW, cout), dtype=torch.bfloat16, device=dev)
        engine.CONV_DEBUG = 64
        run = lambda: engine.conv_forward([engine.Src(x)], wp, cout, cfg, out=out)
        t0, k = time.perf_counter(), 0
        while k < 3 or time.perf_counter() - t0 < 1.0:
            run(); k += 1
            if k % 16 == 0: torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        print('  %d tiles 3x3 %d->%d @%d: %.1f us' % (B, cin, cout, HW, e0.elapsed_time(e1) / 30 * 1e3), flush=True)
engine.CONV_DEBUG = 0
