"""Micro-benchmark of the fp32-precision weight gradient (wgrad_ws32_kernel / wgrad_f32_kernel + split-K reduce) on training-step shapes.
CDNET_WGRAD_WS32=0 selects the older kernel, CDNET_WGRAD_DEBUG=1 drops the MFMA loop.  usage: python tools/bench_wgrad32.py [B] [reps]"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cdnet_amd
from cdnet_amd import _lib, engine, trainer

cdnet_amd.set_precision('fp32')          # (the block shape the trainer picks depends on the precision mode)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = _lib.load()
dev = 'cuda'


def run(name, Cin, Cout, H, fused=False, cap=256, taps=9):
    N = B
    x = torch.randn(N, H, H, Cin, device=dev)
    g = torch.randn(N, H, H, Cout, device=dev)
    kw = dict(scale=torch.rand(Cin, device=dev) + 0.5, shift=torch.rand(Cin, device=dev) - 0.5, relu=True) if fused else {}
    s = engine.Src(x, **kw)
    dw = torch.zeros((Cout, Cin, 3, 3) if taps == 9 else (Cout, Cin, 1, 1), dtype=torch.float32, device=dev)
    ci_t = trainer._choose_ci_tiles(Cin, Cout)
    CI, CO = ci_t * 32, (4 // ci_t) * 32
    other = -(-Cin // CI) * -(-Cout // CO)
    ntiles = N * (-(-H // 8)) * (-(-H // 16))
    ks = max(1, min(ntiles, cap // other if other < cap else 1))
    slab = torch.empty((lib.cdnet_conv_wgrad_slab_floats(Cin, Cout, taps, 1, ci_t, ks),), dtype=torch.float32, device=dev)
    cs = engine.ConvSrc()
    s.fill(cs)
    call = lambda: _lib.call('cdnet_conv_backward_weight', C.byref(cs), 0, Cin, Cin, _lib.ptr(g), Cout, N, H, H, taps, 1, 1, ci_t, ks,
                             _lib.ptr(slab), _lib.ptr(dw), 0, _lib.stream_ptr())
    for _ in range(20):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    fl = 3 * 2.0 * N * H * H * Cin * Cout * taps
    gbs = 4.0 * N * H * H * (Cin + Cout) / ms / 1e6
    print(f'{name:34s} ksplit={ks:4d} {ms*1e3:8.1f} us   {fl / ms / 1e9:7.1f} TFLOP/s of bf16 MFMA work, {gbs:6.0f} GB/s of operand bytes')


run('64->64@256 plain', 64, 64, 256)
if os.environ.get('WG_STAMPS'):
    # stamped build (CDNET_HIPCC_FLAGS=-DCDNET_WS_STAMPS): per-interval timeline of one consumer and one mover wave of workgroup 17
    import ctypes, numpy as np, collections
    buf = np.zeros(2048, dtype=np.uint64)
    f = lib.cdnet_debug_wgrad_stamps
    f.argtypes = [ctypes.c_void_p]
    assert f(buf.ctypes.data) == 0
    for role, off in (('consumer', 0), ('mover', 1024)):
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
        print('  first 24:', ' '.join('%d@%.2f' % (ids[i], ts[i]) for i in range(min(n, 24))))
    sys.exit(0)
if os.environ.get('WG32_ONLY') == '1':
    sys.exit(0)
run('64->64@256 BN+ReLU source', 64, 64, 256, fused=True)
run('128->128@128 BN+ReLU source', 128, 128, 128, fused=True)
run('256->256@64 BN+ReLU source', 256, 256, 64, fused=True)
run('512->512@32 BN+ReLU source', 512, 512, 32, fused=True)
# layers with at most 32 channels on one side (the decoder's 32- / 16-channel blocks, the first residual unit's 16-channel input): quadrants of zero
# padding are not multiplied (wgrad_ws32_kernel<XF, TAPS, QM>); CDNET_WGRAD_DEBUG=16 multiplies all four again
run('16->16@256 BN+ReLU source', 16, 16, 256, fused=True, cap=160)
run('64->16@256 plain', 64, 16, 256, cap=160)
run('16->64@256 plain', 16, 64, 256, cap=160)
run('32->32@128 BN+ReLU source', 32, 32, 128, fused=True, cap=160)
# the residual units' 1x1 convolutions (TAPS = 1: HBM-bound - 537 MB of operands per launch at 16 tiles)
run('1x1 64->64@256 plain', 64, 64, 256, cap=160, taps=1)
run('1x1 64->64@256 plain, 256 workgroups', 64, 64, 256, cap=256, taps=1)
run('1x1 16->64@256 plain', 16, 64, 256, cap=160, taps=1)
