"""conv_ws_kernel (wave-specialised) vs conv_fwd_kernel: bit-identical outputs / statistics on every source and epilogue mode,
then timing of both on the DAM-Unet layer shapes.  usage: python tools/check_conv_ws.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def run(ws, fn):
    os.environ['CDNET_CONV_WS'] = str(ws)
    try:
        return fn()
    finally:
        os.environ['CDNET_CONV_WS'] = '0'


def case(name, srcs_fn, Cout, H, W, taps=9, transposed=False, stats=False, fold=False, bias=False, out_dtype=torch.bfloat16):
    Cin = sum(s.C for s in srcs_fn())
    if transposed:
        w = torch.randn((Cin, Cout, 4, 4) if taps == 4 else (Cin, Cout, 2, 2), generator=g).to(dev) * 0.05
        mode = 2 if taps == 4 else 3
    else:
        k = 3 if taps == 9 else 1
        w = torch.randn((Cout, Cin, k, k), generator=g).to(dev) * 0.05
        mode = 0
    cfg = engine.choose_cfg([s.C for s in srcs_fn()], Cout, H, W, taps=taps, transposed=transposed)
    if cfg[0] != 16 or cfg[1] != 16:
        print('%-34s skipped (cfg %s)' % (name, cfg))
        return
    wp = engine.pack_weights(w, cfg, mode)
    kw = {}
    if fold:
        kw = dict(oscale=torch.rand((Cout,), generator=g).to(dev) + 0.5, oshift=torch.randn((Cout,), generator=g).to(dev), orelu=True)
    if bias:
        kw['bias'] = torch.randn((Cout,), generator=g).to(dev)

    def go():
        out, st = engine.conv_forward(srcs_fn(), wp, Cout, cfg, taps=taps, transposed=transposed, stats=True if stats else None, H=H, W=W,
                                      out_dtype=out_dtype, **kw)
        torch.cuda.synchronize()
        return out, st
    o0, s0 = run(0, go)
    o1, s1 = run(1, go)
    same = torch.equal(o0.view(torch.int16), o1.view(torch.int16)) and (s0 is None or torch.equal(s0, s1))
    print('%-34s cfg %s  %s' % (name, cfg, 'identical' if same else 'DIFFERENT  max|d| %.3g' % float((o0.float() - o1.float()).abs().max())))
    assert same, name


N = 3
x64 = (torch.randn((N, 40, 56, 64), generator=g)).to(dev).to(torch.bfloat16)
raw64 = (torch.randn((N, 40, 56, 64), generator=g) * 2).to(dev).half()
res64 = (torch.randn((N, 40, 56, 64), generator=g)).to(dev).half()
sc, sh = (torch.rand((64,), generator=g) + 0.5).to(dev), (torch.randn((64,), generator=g) * 0.3).to(dev)
x32 = torch.randn((N, 40, 56, 32), generator=g).to(dev).to(torch.bfloat16)
x128 = torch.randn((N, 37, 51, 128), generator=g).to(dev).to(torch.bfloat16)
S = engine.Src
case('plain 64->64', lambda: [S(x64)], 64, 40, 56)
case('plain 64->64 folded BN+ReLU', lambda: [S(x64)], 64, 40, 56, fold=True)
case('plain 64->64 stats f16 out', lambda: [S(x64)], 64, 40, 56, stats=True, out_dtype=torch.float16)
case('bn+relu f16 source -> 64 stats', lambda: [S(raw64, sc, sh, relu=True)], 64, 40, 56, stats=True, out_dtype=torch.float16)
case('bn+relu+res source -> 64', lambda: [S(raw64, sc, sh, relu=True, res=res64)], 64, 40, 56)
case('two sources 32 + 64 -> 32', lambda: [S(x32), S(raw64, sc, sh, relu=True)], 32, 40, 56, stats=True, out_dtype=torch.float16)
case('ragged 128 -> 80 bias', lambda: [S(x128)], 80, 37, 51, bias=True)
case('1x1 64 -> 64 bias f16', lambda: [S(x64)], 64, 40, 56, taps=1, bias=True, out_dtype=torch.float16)
case('pooled bn+relu 64 -> 128', lambda: [S(raw64, sc, sh, relu=True, pool=1)], 128, 20, 28, stats=True, out_dtype=torch.float16)
case('convT 4x4 s2 64 -> 32', lambda: [S(x64)], 32, 40, 56, taps=4, transposed=True, stats=True, out_dtype=torch.float16)

print('--- timing (16 tiles)')
B = 16
for name, Cin, Cout, H, K in (('64->64@256', 64, 64, 256, 3), ('64->128@128', 64, 128, 128, 3), ('128->128@128', 128, 128, 128, 3),
                              ('256->256@64', 256, 256, 64, 3), ('512->512@32', 512, 512, 32, 3), ('160->32@128', 160, 32, 128, 3),
                              ('1x1 64->64@256', 64, 64, 256, 1)):
    x = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
    w = torch.randn((Cout, Cin, K, K), device=dev) * 0.05
    cfg = (16, 16, 64 if Cout > 32 else 32)
    wp = engine.pack_weights(w, cfg, 0)
    out = torch.empty((B, H, H, Cout), dtype=torch.bfloat16, device=dev)
    res = []
    for ws in (0, 1, 0, 1):
        os.environ['CDNET_CONV_WS'] = str(ws)
        for _ in range(3):
            engine.conv_forward([S(x)], wp, Cout, cfg, taps=K * K, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            engine.conv_forward([S(x)], wp, Cout, cfg, taps=K * K, out=out)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    os.environ['CDNET_CONV_WS'] = '0'
    print('%-18s  fwd kernel %.3f / %.3f ms   ws kernel %.3f / %.3f ms' % (name, res[0], res[2], res[1], res[3]))
