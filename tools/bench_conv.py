"""Micro-benchmark of the MFMA convolution kernel on the DAM-Unet layer shapes (B tiles of 256x256)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdnet_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
LAYERS = [  # name, Cin, Cout, H, cfgs
    ('enc1_2 64->64@256', 64, 64, 256, [(16, 16, 64)]),
    ('enc2_1 64->128@128', 64, 128, 128, [(16, 16, 64)]),
    ('enc2_2 128->128@128', 128, 128, 128, [(16, 16, 64), (16, 16, 32), (16, 32, 128)]),
    ('enc3_2 256->256@64', 256, 256, 64, [(16, 32, 64), (16, 16, 64), (16, 16, 32), (16, 32, 32), (16, 32, 128)]),
    ('enc4_2 512->512@32', 512, 512, 32, [(16, 32, 64), (16, 16, 64), (16, 16, 32), (16, 32, 32), (8, 32, 64), (8, 64, 64)]),
    ('enc5_2 512->512@16', 512, 512, 16, [(8, 32, 128), (8, 32, 64), (16, 16, 64), (16, 32, 64)]),
    ('dec3 160->32@128', 160, 32, 128, [(16, 32, 32), (16, 16, 32)]),
]
print('B =', B)
LAYERS.append(('1x1 64->64@256', 64, 64, 256, [(16, 16, 64), (16, 32, 64), (16, 64, 64)]))
for name, Cin, Cout, H, cfgs in LAYERS:
    K = 1 if name.startswith('1x1') else 3
    x = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
    w = torch.randn((Cout, Cin, K, K), device=dev) * 0.05
    flops = 2.0 * B * H * H * Cin * Cout * K * K
    byts = B * H * H * (Cin + Cout) * 2
    for cfg in cfgs:
        wp = engine.pack_weights(w, cfg, 0)
        out = torch.empty((B, H, H, Cout), dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            engine.conv_forward([engine.Src(x)], wp, Cout, cfg, taps=K * K, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            engine.conv_forward([engine.Src(x)], wp, Cout, cfg, taps=K * K, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%-22s cfg=%-14s %8.3f ms  %7.1f TFLOP/s  %6.2f TB/s(alg)' % (name, cfg, ms, flops / ms / 1e9, byts / ms / 1e9))
