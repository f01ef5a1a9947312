"""Timeline of one consumer wave and one producer wave of conv_ws_kernel (debug build: CDNET_HIPCC_FLAGS=-DCDNET_WS_STAMPS python -m
cdnet_amd.csrc.build --force).  Prints per-phase durations in microseconds (100 MHz wall clock)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cdnet_amd import engine, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BNS = os.environ.get('WS_BNS') == '1'          # backward-data launch that also accumulates the BatchNorm-backward channel sums
EXTRA = int(sys.argv[3]) if len(sys.argv) > 3 else 0
train = len(sys.argv) > 2 and sys.argv[2] == 'train'
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
bns = (raw, sc, sh, sc * 0.1, sc, torch.zeros((1024, 2, 64), dtype=torch.float32, device=dev)) if BNS else None
engine.CONV_DEBUG = 64
for _ in range(5):
    if train:
        engine.conv_forward([engine.Src(raw, sc, sh, relu=True)], wp, 64, cfg, out=outh, stats=stats)
    else:
        engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out, bns=bns)
torch.cuda.synchronize()
for dbg, nm in ((32, 'conv_fwd_kernel'), (64 | EXTRA, 'conv_ws_kernel (stamped build, debug %d)' % EXTRA))[(1 if BNS else 0):]:
    engine.CONV_DEBUG = dbg
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        engine.conv_forward([engine.Src(raw, sc, sh, relu=True)], wp, 64, cfg, out=outh, stats=stats) if train else engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out, bns=bns)
    e0.record()
    for _ in range(20):
        engine.conv_forward([engine.Src(raw, sc, sh, relu=True)], wp, 64, cfg, out=outh, stats=stats) if train else engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out, bns=bns)
    e1.record()
    torch.cuda.synchronize()
    print('%-32s %.1f us per launch' % (nm, e0.elapsed_time(e1) / 20 * 1e3))
engine.CONV_DEBUG = 64 | EXTRA
for _ in range(3):
    engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out, bns=bns)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(2048, dtype=np.uint64)
f = lib.cdnet_debug_ws_stamps
f.argtypes = [ctypes.c_void_p]
assert f(buf.ctypes.data) == 0
for role, off in (('consumer', 0), ('producer', 1024)):
    v = buf[off:off + 1024]
    print(role, 'raw', v[:4], int((v != 0).sum()))
    n = int(np.argmax(v == 0))
    ids = (v[:n] & np.uint64(255)).astype(int)
    ts = (v[:n] >> np.uint64(8)).astype(np.int64)
    cyc = ids >= 128
    if cyc.any():
        c, ci = ts[cyc], ids[cyc]
        steps = [(c[i + 1] - c[i]) for i in range(len(c) - 1) if ci[i] == 130 and ci[i + 1] == 131]
        rt = ts[~cyc]
        print(role, 'shader cycles per MFMA step: mean %.0f min %d max %d; cycles first->last %d over %.2f us = %.2f GHz'
              % (np.mean(steps), min(steps), max(steps), c[-1] - c[0], (rt[-1] - rt[0]) / 100.0, (c[-1] - c[0]) / ((rt[-1] - rt[0]) * 10.0)))
    ids, ts = ids[~cyc], ts[~cyc]
    n = len(ids)
    ts = (ts - ts[0]) / 100.0
    print(role, n, 'stamps, span %.1f us' % (ts[-1] if n else 0))
    # durations by (from id -> to id)
    import collections
    d = collections.defaultdict(list)
    for i in range(1, n):
        d[(ids[i - 1], ids[i])].append(ts[i] - ts[i - 1])
    for k in sorted(d):
        a = np.array(d[k])
        print('  %2d -> %2d : n %3d  mean %6.2f  med %6.2f  min %6.2f  max %6.2f  total %7.1f' % (k[0], k[1], len(a), a.mean(), np.median(a), a.min(), a.max(), a.sum()))
    print('  first 40:', ' '.join('%d@%.2f' % (ids[i], ts[i]) for i in range(min(n, 40))))
