"""Barrier-interval timeline of conv_ws16_kernel on the dominant layer (3x3 64 -> 64 @256x256), stamped build:

  bash tools/build_variant.sh stamps "-DCDNET_WS_STAMPS" conv16ws.hip
  CDNET_LIB_PATH=cdnet_amd/libcdnet_hip_stamps.so python tools/ws16_stamps.py [tiles = 64]

One consumer wave (wave 0) and one mover wave (wave 4) of one workgroup stamp the 100 MHz wall clock over ~45 barrier intervals in the
middle of the workgroup's run.  Printed per variant (random / all-zero operands x full / no stores / no halo requests): the launch time and,
per interval, where each role spends it -
  mover   : 1 -> 2 wait for halo chunk A + its LDS writes | 2 -> 3 request A, commit B (wait + writes), request B | 3 -> 4 out-image reads +
            global stores (issue) | 4 -> 5 lgkmcnt(0) + barrier (waiting for the consumers) | 5 -> 1 loop overhead
  consumer: 12 -> 13 first chunk (36 MFMAs + fragment reads + epilogue units) | 13 -> 11 second chunk | 11 -> 12 barrier (waiting for the movers)
"""
import collections
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cdnet_amd import engine, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
lib = _lib.load()
f = lib.cdnet_debug_ws16_stamps
f.argtypes = [ctypes.c_void_p]
cfg = (16, 16, 64)
NAMES = {(1, 2): 'mover: wait halo A + LDS writes', (2, 3): 'mover: request A, commit B, request B', (3, 4): 'mover: out-image reads + global stores',
         (4, 5): 'mover: lgkmcnt(0) + barrier', (5, 1): 'mover: loop', (12, 13): 'consumer: chunk 1 (36 MFMAs)', (13, 11): 'consumer: chunk 2 (36 MFMAs)',
         (11, 12): 'consumer: barrier',
         (21, 22): 'storer: out-image reads + global stores', (22, 23): 'storer: lgkmcnt(0) + barrier', (23, 21): 'storer: loop'}


def report(tag):
    buf = np.zeros(3072, dtype=np.uint64)
    assert f(buf.ctypes.data) == 0
    for role, off in (('consumer', 0), ('mover', 1024), ('storer', 2048)):
        v = buf[off:off + 1024]
        n = int(np.argmax(v == 0))
        ids = (v[:n] & np.uint64(255)).astype(int)
        ts = (v[:n] >> np.uint64(8)).astype(np.int64)
        if n < 8:
            if role != 'storer':
                print('  %s: no stamps' % role)
            continue
        ts = (ts - ts[0]) / 100.0
        d = collections.defaultdict(list)
        for i in range(1, n):
            d[(ids[i - 1], ids[i])].append(ts[i] - ts[i - 1])
        first = {'consumer': 12, 'mover': 1, 'storer': 21}[role]
        marks = [ts[i] for i in range(n) if ids[i] == first]
        per = (marks[-1] - marks[0]) / (len(marks) - 1) if len(marks) > 1 else float('nan')
        print('  %s: %d stamps over %.1f us, %.3f us per interval' % (role, n, ts[-1], per))
        for k in sorted(d):
            a = np.array(d[k])
            print('    %-44s n %3d  mean %5.2f  med %5.2f  min %5.2f  max %5.2f us' % (NAMES.get(k, '%d -> %d' % k), len(a), a.mean(), np.median(a), a.min(), a.max()))


for zero in (False, True):
    x = torch.zeros((B, 256, 256, 64), device=dev, dtype=torch.bfloat16) if zero else (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
    w = torch.zeros((64, 64, 3, 3), device=dev) if zero else torch.randn((64, 64, 3, 3), device=dev) * 0.06
    wp = engine.pack_weights(w, cfg, 0)
    out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
    for dbg, nm in ((64, 'full'), (64 | 8, 'no stores'), (64 | 2, 'no halo requests'), (64 | 1, 'no MFMAs (memory alone)'), (64 | 2 | 8, 'consumers alone')):
        engine.CONV_DEBUG = dbg
        run = lambda: engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
        import time
        t0, k = time.perf_counter(), 0
        while k < 3 or time.perf_counter() - t0 < 0.7:
            run(); k += 1
            if k % 16 == 0:
                torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print('%d tiles, %s operands, %s: %.1f us per launch' % (B, 'ALL-ZERO' if zero else 'random', nm, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
        report(nm)
engine.CONV_DEBUG = 0
