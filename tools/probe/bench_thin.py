"""Micro-benchmark of the thin layers at 256x256 that stay on conv_fwd_kernel: 16->64 (first convolutions), [16,64]->16 (last decoder
convolution), 16->80 (its backward-data), with the training-mode epilogues.  usage: python tools/probe/bench_thin.py [B]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cdnet_amd import engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
H = 256


def run(name, cins, cout, cfg, stats, f16out, byts):
    srcs = [engine.Src((torch.rand((B, H, H, c), device=dev) - 0.3).to(torch.bfloat16)) for c in cins]
    w = torch.randn((cout, sum(cins), 3, 3), device=dev) * 0.05
    wp = engine.pack_weights(w, cfg, 0)
    out = torch.empty((B, H, H, cout), dtype=torch.float16 if f16out else torch.bfloat16, device=dev)
    f = lambda: engine.conv_forward(srcs, wp, cout, cfg, taps=9, out=out, stats=True if stats else None)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('%-28s cfg=%-14s %7.1f us   alg %5.0f MB -> %4.2f TB/s' % (name, cfg, us, byts / 1e6, byts / us / 1e6))


px = B * H * H
for cfg in ((16, 16, 64), (16, 16, 32)):
    run('16->64 stats f16', [16], 64, cfg, True, True, px * (16 + 64) * 2)
for cfg in ((16, 16, 32), (16, 16, 64)):
    run('[16,64]->16 stats f16', [16, 64], 16, cfg, True, True, px * (80 + 16) * 2)
for cfg in ((16, 16, 64), (16, 16, 32)):
    run('16->80 plain', [16], 80, cfg, False, False, px * (16 + 80) * 2)
run('64->64 stats f16 (ws)', [64], 64, (16, 16, 64), True, True, px * 128 * 2)
