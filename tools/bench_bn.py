"""Micro-benchmark of cdnet_bn_backward (reduce + finalize + apply) on the shapes the training step uses.
usage: python tools/bench_bn.py [B]"""
import ctypes as C
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdnet_amd import _lib
from cdnet_amd.trainer import BnBwdArgs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = 'cuda'


def run(name, Cc, H, ngin, pooled, res):
    N = B
    raw = torch.randn(N, H, H, Cc, device=dev).to(torch.float16)
    r = torch.randn(N, H, H, Cc, device=dev).to(torch.float16) if res else None
    g = [torch.randn(N, H, H, Cc, device=dev).to(torch.bfloat16) for _ in range(ngin)]
    gp = torch.randn(N, H // 2, H // 2, Cc, device=dev).to(torch.bfloat16) if pooled else None
    f = lambda: torch.rand(Cc, device=dev) + 0.5
    scale, shift, mean, invstd, gamma = f(), f() - 1, f() - 1, f(), f()
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    a = BnBwdArgs()
    a.raw, a.res = raw.data_ptr(), (r.data_ptr() if res else None)
    a.scale, a.shift, a.mean, a.invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    k = 0
    for t in g:
        a.gin[k].g, a.gin[k].Hg, a.gin[k].Wg, a.gin[k].cstride = t.data_ptr(), H, H, Cc
        k += 1
    if pooled:
        a.gin[k].g, a.gin[k].Hg, a.gin[k].Wg, a.gin[k].cstride, a.gin[k].pooled = gp.data_ptr(), H // 2, H // 2, Cc, 1
        k += 1
    a.ngin, a.f16, a.relu, a.N, a.H, a.W, a.C = k, 1, 1, N, H, H, Cc
    ws = torch.empty((_lib.load().cdnet_bn_backward_workspace_floats(Cc),), dtype=torch.float32, device=dev)
    draw = torch.empty(N, H, H, Cc, device=dev, dtype=torch.bfloat16)
    dz = torch.empty_like(draw) if res else None
    call = lambda: _lib.call('cdnet_bn_backward', C.byref(a), _lib.ptr(gamma), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(ws), ws.numel(),
                             _lib.ptr(draw), _lib.ptr(dz), _lib.stream_ptr())
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    elems = N * H * H * Cc
    # both passes read raw(+res)+grads; apply writes draw (+dz)
    rd = 2 * (1 + int(res) + ngin) * elems * 2 + (2 * elems // 4 * 2 if pooled else 0)
    wr = (1 + int(res)) * elems * 2
    print(f'{name:34s} {ms*1e3:8.1f} us   {(rd + wr) / ms / 1e9:6.2f} TB/s (alg)')


run('64ch@256 1 grad', 64, 256, 1, False, False)
run('64ch@256 1 grad + residual', 64, 256, 1, False, True)
run('64ch@256 3 grads', 64, 256, 3, False, False)
run('64ch@256 skip + pooled', 64, 256, 1, True, False)
run('128ch@128 skip + pooled', 128, 128, 1, True, False)
run('512ch@32 skip + pooled', 512, 32, 1, True, False)
run('32ch@256 1 grad', 32, 256, 1, False, False)
run('256ch@64 1 grad', 256, 64, 1, False, False)
