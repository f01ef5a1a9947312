"""CU-partitioned concurrency: does a matrix-bound persistent convolution that is held down by power lose less than its
share of CUs when it runs on fewer of them - and does an HBM-bound kernel on the freed CUs, launched from a second stream, finish inside the
time the pair of them would take one after the other?

  python3 tools/cu_partition.py [tiles = 16]
      1. the dominant fp32 layer (conv_ws32_kernel, 3x3 64 -> 64 @256x256 x tiles) alone at 256 / 224 / 192 / 160 / 128 workgroups
         (cdnet_conv_args.debug = 64 | grid << 8: one persistent workgroup per CU);
      2. three HBM-bound partners alone: a float4 copy (cdnet_box_copy) sized to take about as long as the convolution, the residual units'
         1x1 convolution with fused residual epilogue (conv_f32_kernel<16,16,64,4,1,1>: three tensors), the BatchNorm backward of a 64-channel
         layer (bn_bwd_flat32 reduce + apply);
      3. every (convolution at G workgroups, partner) pair started together on two streams: time until both are done, against the sum of the
         two alone and against max(.).
  The clocks come from a PMC pass over the same script:
      rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d <dir> -o t -- python3 tools/cu_partition.py 16 solo
      python3 tools/cu_partition.py --clocks <dir>          (per grid size: duration, GRBM_GUI_ACTIVE / 8 / duration)
"""
import csv
import ctypes as C
import glob
import os
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GRIDS = (256, 224, 192, 160, 128)


def clocks(d):
    dur = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_ws32_kernel' in r['Kernel_Name']:
                dur[r['Dispatch_Id']] = (int(r['Workgroup_Size_X']) and int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']),
                                         float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    acc = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r['Dispatch_Id'] in dur:
                acc[r['Dispatch_Id']] = acc.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    per = {}
    for k, v in acc.items():
        g, ns = dur[k]
        per.setdefault(g, []).append((ns, v / 8.0 / ns * 1e3))
    print('workgroups   launches   duration us (under the counter pass)   clock MHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)')
    for g in sorted(per, reverse=True):
        rows = per[g][len(per[g]) // 3:]                 # (the first third of a size's launches: settling)
        print('%10d %10d %14.1f %28.0f' % (g, len(rows), sum(a for a, _ in rows) / len(rows) / 1e3, sum(b for _, b in rows) / len(rows)))


def main():
    import torch
    import cdnet_amd
    from cdnet_amd import _lib, engine, streams, trainer
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    solo = len(sys.argv) > 2 and sys.argv[2] == 'solo'
    cdnet_amd.set_precision('fp32')
    dev = torch.device('cuda:0')
    x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3)
    w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
    cfg = (16, 16, 64)
    wp = engine.pack_weights(w, cfg, 0, split=True)
    out = torch.empty_like(x)
    side = streams.side_stream(dev)
    main_s = torch.cuda.current_stream()

    def conv(G):
        engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out, debug_or=64 | (G << 8))

    # partners
    nbytes = (B * 256 * 256 * 64 * 4 * 2) // 16 * 16              # copy: read + write of one activation tensor pair (~1.07 GB of traffic at 16 tiles)
    ca, cb = torch.empty((nbytes // 4,), device=dev), torch.empty((nbytes // 4,), device=dev)
    w1 = torch.randn((64, 64, 1, 1), device=dev) * 0.1
    wp1 = engine.pack_weights(w1, cfg, 0, split=True)
    x1, r1, o1 = torch.rand_like(x), torch.rand_like(x), torch.empty_like(x)

    def p_copy():
        _lib.call('cdnet_box_copy', _lib.ptr(ca), _lib.ptr(cb), nbytes, _lib.stream_ptr())

    def p_1x1():
        engine.conv_forward([engine.Src(x1)], wp1, 64, cfg, taps=1, out=o1, eres=engine.Src(r1, relu=False))

    # BatchNorm backward of a 64-channel fp32 layer through the trainer's own call (reduce + finalize + apply)
    raw, g = torch.randn_like(x), torch.randn_like(x)
    a = trainer.BnBwdArgs()
    sc, sh, mu, inv = [torch.rand(64, device=dev) + 0.5 for _ in range(4)]
    gam, dgam, dbet = torch.rand(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    a.raw, a.res, a.scale, a.shift, a.mean, a.invstd = raw.data_ptr(), None, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), inv.data_ptr()
    a.ngin = 1
    a.gin[0].g, a.gin[0].Hg, a.gin[0].Wg, a.gin[0].cstride = g.data_ptr(), 256, 256, 64
    a.f16, a.relu, a.N, a.H, a.W, a.C = 2, 1, B, 256, 256, 64
    ws = torch.empty((_lib.load().cdnet_bn_backward_workspace_floats(64),), dtype=torch.float32, device=dev)
    draw = torch.empty_like(x)

    def p_bn():
        _lib.call('cdnet_bn_backward', C.byref(a), _lib.ptr(gam), _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(ws), ws.numel(), _lib.ptr(draw), None,
                  _lib.stream_ptr())

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    for _ in range(300):                                            # settle the clock under the matrix load
        conv(256)
    torch.cuda.synchronize()
    alone = {}
    print('conv_ws32_kernel 3x3 64->64 @256x256 x %d tiles alone:' % B)
    for G in GRIDS:
        alone[G] = timed(lambda: conv(G))
        print('  %3d workgroups: %7.1f us   (x %.3f of 256; a proportional loss would be x %.3f)' % (G, alone[G], alone[G] / alone[256], 256.0 / G))
    if solo:
        return
    partners = (('float4 copy, %.2f GB of traffic' % (2 * nbytes / 1e9), p_copy), ('1x1 64->64 + residual epilogue (conv_f32_kernel, 3 tensors)', p_1x1),
                ('BatchNorm backward 64 channels (bn_bwd_flat32: reduce + apply)', p_bn))
    palone = {}
    for name, fn in partners:
        palone[name] = timed(fn)
        print('partner alone: %-66s %7.1f us' % (name, palone[name]))

    def pair(G, fn, n=30):
        def once():
            ev = torch.cuda.Event()
            ev.record(main_s)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                fn()
                ev2 = torch.cuda.Event()
                ev2.record(side)
            conv(G)
            main_s.wait_event(ev2)
        return timed(once, n)
    print('pairs (both started together, time until both are done):')
    print('  %-66s %5s %9s %9s %9s %7s' % ('partner', 'wgs', 'pair us', 'sum us', 'max us', 'pair/sum'))
    for name, fn in partners:
        for G in GRIDS:
            t = pair(G, fn)
            s = alone[256] + palone[name]
            print('  %-66s %5d %9.1f %9.1f %9.1f %7.3f' % (name, G, t, s, max(alone[256], palone[name]), t / s))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--clocks':
        clocks(sys.argv[2])
    else:
        main()
