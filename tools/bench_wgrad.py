"""Micro-benchmark of cdnet_conv_backward_weight (wgrad + split-K reduce) on training-step shapes.
usage: python tools/bench_wgrad.py [B] [reps]"""
import ctypes as C
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdnet_amd import _lib, engine, trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load()
dev = 'cuda'


def run(name, Cin, Cout, H, fused=False, res=False, pool=False, ksplit=None):
    N = B
    Hs = H * 2 if pool else H
    x = torch.randn(N, Hs, Hs, Cin, device=dev).to(torch.float16 if fused else torch.bfloat16)
    g = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16)
    kw = {}
    if fused:
        kw = dict(scale=torch.rand(Cin, device=dev) + 0.5, shift=torch.rand(Cin, device=dev) - 0.5, relu=True)
    if res:
        kw['res'] = torch.randn(N, Hs, Hs, Cin, device=dev).to(torch.float16)
    if pool:
        kw['pool'] = 1
    s = engine.Src(x, **kw)
    dw = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float32, device=dev)
    ci_t = trainer._choose_ci_tiles(Cin, Cout)
    CI, CO = ci_t * 32, (4 // ci_t) * 32
    other = -(-Cin // CI) * -(-Cout // CO)
    ntiles = N * (-(-H // 8)) * (-(-H // 16))
    ks = ksplit or max(1, min(ntiles, 256 // other if other < 256 else 1))
    slab = torch.empty((lib.cdnet_conv_wgrad_slab_floats(Cin, Cout, 9, 1, ci_t, ks),), dtype=torch.float32, device=dev)
    cs = engine.ConvSrc()
    s.fill(cs)
    call = lambda: _lib.call('cdnet_conv_backward_weight', C.byref(cs), 0, Cin, Cin, _lib.ptr(g), Cout, N, H, H, 9, 1, 1, ci_t, ks,
                             _lib.ptr(slab), _lib.ptr(dw), 0, _lib.stream_ptr())
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f'{name:34s} ksplit={ks:4d} {ms*1e3:8.1f} us   {fl / ms / 1e9:7.1f} TFLOP/s')


run('64->64@256 plain', 64, 64, 256)
if os.environ.get('WG_STAMPS'):
    import ctypes, numpy as np, collections
    buf = np.zeros(2048, dtype=np.uint64)
    f = lib.cdnet_debug_wgrad_stamps
    f.argtypes = [ctypes.c_void_p]
    assert f(buf.ctypes.data) == 0
    for role, off in (('consumer', 0), ('producer', 1024)):
        v = buf[off:off + 1024]
        n = int(np.argmax(v == 0))
        ids = (v[:n] & np.uint64(255)).astype(int)
        ts = (v[:n] >> np.uint64(8)).astype(np.int64)
        ts = (ts - ts[0]) / 100.0
        print(role, n, 'stamps, span %.1f us' % ts[-1])
        d = collections.defaultdict(list)
        for i in range(1, n):
            d[(ids[i - 1], ids[i])].append(ts[i] - ts[i - 1])
        for k in sorted(d):
            a = np.array(d[k])
            print('  %2d -> %2d : n %3d  mean %6.2f  med %6.2f  min %6.2f  max %6.2f  total %7.1f' % (k[0], k[1], len(a), a.mean(), np.median(a), a.min(), a.max(), a.sum()))
        print('  first 30:', ' '.join('%d@%.2f' % (ids[i], ts[i]) for i in range(min(n, 30))))
    sys.exit(0)
run('64->64@256 plain', 64, 64, 256, ksplit=512)
run('64->64@256 plain', 64, 64, 256, ksplit=128)
run('64->64@256 bn+relu f16', 64, 64, 256, fused=True)
run('64->64@256 bn+relu+res', 64, 64, 256, fused=True, res=True)
run('64->128@128 pooled bn+relu', 64, 128, 128, fused=True, pool=True)
run('128->128@128 bn+relu', 128, 128, 128, fused=True)
run('256->256@64 bn+relu', 256, 256, 64, fused=True)
run('512->512@32 bn+relu', 512, 512, 32, fused=True)
run('512->512@16 bn+relu', 512, 512, 16, fused=True)
