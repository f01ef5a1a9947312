"""conv_ws_kernel with streamed weights (128+ input channels) against conv_fwd_kernel on the shapes of the training / inference step.
usage: python tools/bench_conv_stream.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')


def run(name, cins, cout, H, train=False):
    srcs = []
    for c in cins:
        if train:
            raw = (torch.rand((B, H, H, c), device=dev) - 0.3).to(torch.float16)
            srcs.append(engine.Src(raw, torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, relu=True))
        else:
            srcs.append(engine.Src((torch.rand((B, H, H, c), device=dev) - 0.3).to(torch.bfloat16)))
    w = torch.randn((cout, sum(cins), 3, 3), device=dev) * 0.03
    cfg = engine.choose_cfg(list(cins), cout, H, H)
    wp = engine.pack_weights(w, cfg, 0)
    out = torch.empty((B, H, H, cout), dtype=torch.float16 if train else torch.bfloat16, device=dev)
    stats = torch.empty((B * (H // 16) ** 2, 2, cout), dtype=torch.float32, device=dev) if train else None
    res = []
    for dbg in (32, 0, 32, 0):
        engine.CONV_DEBUG = dbg
        f = lambda: engine.conv_forward(srcs, wp, cout, cfg, out=out, stats=stats) if train else engine.conv_forward(srcs, wp, cout, cfg, out=out)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    engine.CONV_DEBUG = 0
    fl = 2.0 * B * H * H * sum(cins) * cout * 9
    print('%-34s cfg %-14s conv_fwd %7.1f / %7.1f us   default %7.1f / %7.1f us   (%.0f TFLOP/s)' % (name, cfg, res[0], res[2], res[1], res[3], fl / min(res[1], res[3]) / 1e6))


for tr in (False, True):
    print('training-mode sources + statistics' if tr else 'plain sources')
    run('64->64 @256', (64,), 64, 256, tr)
    run('64+64->64 @256 (decoder)', (64, 64), 64, 256, tr)
    run('64->128 @128', (64,), 128, 128, tr)
    run('128->128 @128', (128,), 128, 128, tr)
    run('128+128->128 @128 (decoder)', (128, 128), 128, 128, tr)
    run('256->256 @64', (256,), 256, 64, tr)
    run('512->512 @32', (512,), 512, 32, tr)
    run('512->512 @16', (512,), 512, 16, tr)
