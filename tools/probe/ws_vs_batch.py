"""conv_ws_kernel time on the dominant layer (3x3 64->64 @256x256) against the number of tiles per launch: fixed cost vs per-tile cost.
usage: python tools/probe/ws_vs_batch.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cdnet_amd import engine
dev = torch.device('cuda:0')
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
cfg = (16, 16, 64)
wp = engine.pack_weights(w, cfg, 0)
for B in (4, 8, 16, 32, 64):
    x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
    raw = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.float16)
    out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
    outh = torch.empty((B, 256, 256, 64), dtype=torch.float16, device=dev)
    stats = torch.empty((B * 256, 2, 64), dtype=torch.float32, device=dev)
    res = []
    for train in (False, True):
        def run():
            if train:
                engine.conv_forward([engine.Src(raw, sc, sh, relu=True)], wp, 64, cfg, out=outh, stats=stats)
            else:
                engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print('B %3d (%3d tiles per workgroup): plain %7.1f us (%5.2f us/tile/wg)   train-mode %7.1f us (%5.2f)' % (B, B, res[0], res[0] / B, res[1], res[1] / B))
