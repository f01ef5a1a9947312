"""Phase timeline of tile_chain_kernel (csrc/postproc_tile.hip) from s_memtime stamps at every workgroup barrier - a DIAGNOSTIC build:
    bash tools/build_variant.sh tstamps "-DCDNET_TILE_STAMPS" postproc_tile.hip
    CDNET_LIB_PATH=$PWD/cdnet_amd/libcdnet_hip_tstamps.so python3 tools/tile_stamps.py [tiles]
Prints, per phase, the median over the first 64 workgroups of the interval in shader cycles and its share of the kernel."""
import ctypes as C
import os
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import torch

from cdnet_amd import _lib, postproc
import bench_postproc

NAMES = ['load foreground words', 'init background', 'merge background', 'mark border', 'fill', 'init A (+ zero areas)', 'merge A',
         'flatten + areas', 'survivors', 'diagonal unions', 'flatten kept', 'root bits + scan', 'ranks', 'labels at the heads',
         'labels -> every pixel', 'labels out + dilate']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device('cuda:0')
    m, p, d = bench_postproc.logits(B, dev)
    for _ in range(3):
        r = postproc.tile_postproc(m, d, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        r = postproc.tile_postproc(m, d, p)
    e1.record()
    torch.cuda.synchronize()
    print('three launches: %.1f us per batch of %d tiles' % (e0.elapsed_time(e1) * 100, B))
    lib = _lib.load()
    fn = lib.cdnet_debug_tile_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
    buf = np.zeros((64, 32), np.uint64)
    assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
    n = min(B, 64)
    st = buf[:n, :len(NAMES) + 1].astype(np.int64)
    iv = np.diff(st, axis=1)
    tot = np.median(st[:, -1] - st[:, 0])
    print('kernel body: %d cycles (median over %d workgroups)' % (tot, n))
    print('   merge background = unions %d + halving walk %d + flatten %d cycles' % (np.median(buf[:n, 17].astype(np.int64) - st[:, 2]), np.median(buf[:n, 18].astype(np.int64) - buf[:n, 17].astype(np.int64)), np.median(st[:, 3] - buf[:n, 18].astype(np.int64))))
    for k, name in enumerate(NAMES):
        print('%-34s %9d cycles  %5.1f %%' % (name, np.median(iv[:, k]), 100.0 * np.median(iv[:, k]) / tot))


if __name__ == '__main__':
    main()
