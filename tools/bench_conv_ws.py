"""Phase ablation of the wave-specialised persistent convolution (conv_ws_kernel) on the dominant layer (3x3 64->64 @256x256, 16
tiles): cdnet_conv_args.debug bits 64 = force the kernel, 32 = conv_fwd_kernel for comparison (CDNET_CONV_WS_DEFER=1 selects the
deferred-epilogue form).   usage: python tools/bench_conv_ws.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
raw = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.float16)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
cfg = (16, 16, 64)
wp = engine.pack_weights(w, cfg, 0)
out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
outh = torch.empty((B, 256, 256, 64), dtype=torch.float16, device=dev)
stats = torch.empty((B * 256, 2, 64), dtype=torch.float32, device=dev)


def t(dbg, train=False, n=20):
    engine.CONV_DEBUG = dbg
    def run():
        if train:
            engine.conv_forward([engine.Src(raw, sc, sh, relu=True)], wp, 64, cfg, out=outh, stats=stats)
        else:
            engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    engine.CONV_DEBUG = 0
    return e0.elapsed_time(e1) / n * 1e3


for name, d in (('conv_fwd_kernel', 32), ('conv_ws_kernel', 64), ('conv_ws_kernel padded halo image', 64 | 128), ('conv_fwd_kernel again', 32), ('conv_ws_kernel again', 64), ('conv_ws_kernel padded again', 64 | 128)):
    print('%-36s plain %7.1f us   train-mode source+stats %7.1f us' % (name, t(d), t(d, True)))
