import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cdnet_amd import engine
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
y = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.float16)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
w = torch.randn((64, 64, 1, 1), device=dev) * 0.1
bias = torch.randn(64, device=dev)
for cfg in ((16, 16, 64), (16, 32, 64), (16, 64, 64), (16, 16, 64)):
    try:
        wp = engine.pack_weights(w, cfg, 0)
        out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
        for name, f in (('plain', lambda: engine.conv_forward([engine.Src(x)], wp, 64, cfg, taps=1, out=out)),
                        ('eres', lambda: engine.conv_forward([engine.Src(x)], wp, 64, cfg, taps=1, bias=bias, out=out, eres=engine.Src(y, sc, sh, relu=True)))):
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            print(cfg, name, '%.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
    except Exception as e:
        print(cfg, 'failed:', str(e)[:120])
