#!/usr/bin/env python3
"""bench.py - throughput of the CDNet hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--mode train|infer|image|roofline] [--dtype bf16|fp32] [--batch B]

One rank per GPU (the driver launches N>1 through torch.distributed.run); RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* come
from the environment.  W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; the step time
is the max over ranks; rank 0 prints ONE JSON line (see DESIGN.md "Measurement").

The line's `value` is the training rate of BASELINE config 2 in the arithmetic named by `dtype`.  The default is `fp32` - the
like-for-like mode: fp32 storage and accumulation as in the reference, every product a*b as three bf16 MFMAs over split operands
("bf16x3", <= 2^-16 relative error per product; gfx950 has no TF32 path and its fp32 MFMA runs at 1/16 of the bf16 rate).  Beside it
(same protocol, fewer steps): `inference` (tiles/s incl. post-processing), `bf16` = the same two rates on the 16-bit fast path
(bf16 / fp16 activations, bf16 MFMA operands, fp32 accumulation; legitimate only behind the label-level gate
tests/test_gpu_label_gate.py), `image` = BASELINE config 3 (one 1000x1000 image, 8 TTA views x 25 windows), `roofline` (dominant
kernel of the headline arithmetic), `roofline_path` (whole-step fractions in SURVEY 8d's units) and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime starts: see cdnet_amd/__init__.py (stream -> hardware queue aliasing)

DENSE_BF16_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16/f16 MFMA
FP32_MATRIX_PEAK_TFLOPS = 157.3      # ibid.: v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate (no TF32/xf32 on gfx950)
HBM_PEAK_GBS = 8000.0                # ibid.: 8 TB/s spec (6.3 TB/s measured float4 copy)

# SURVEY.md 8d / BASELINE.md section 3 (forward hooks on the reference modules), per 256x256 tile
INFER_GFLOP_PER_TILE = 75.06
TRAIN_GFLOP_PER_TILE = 225.2
INFER_MB_PER_TILE = {'bf16': 256.2, 'fp32': 512.4}          # activations r+w, conv+BN+ReLU fused per layer
TRAIN_MB_PER_TILE = {'bf16': 770.0, 'fp32': 1540.0}         # ~3x forward (fwd write, bwd re-read, grad r/w)
WEIGHT_MB = {'bf16': 41.0, 'fp32': 81.9}                    # read once per batch (inference)
OPT_MB_PER_STEP = 4 * 81.9                                  # parameter / gradient / Adam-moment traffic per step (fp32 masters in both modes)
POSTPROC_MB_PER_TILE = 1.835                                # 28 B/pixel
# plain UNet (cfg 1) 96.35 / 289.1 GFLOP per tile (inference / train), 328.2 MB fp32 forward activations, 124.2 MB of weights;
# HRNet18_rev1 (cfg 5) per 512x512 tile: 455.8 GFLOP and 6.68 GB fp32 / 3.34 GB bf16 forward; training = 3 x forward (as for the DAM-Unet)
UNET_TRAIN_GFLOP_PER_TILE = 289.1
UNET_FWD_MB_PER_TILE = {'bf16': 164.1, 'fp32': 328.2}
UNET_WEIGHT_MB = 124.2
HRNET_TRAIN_GFLOP_PER_TILE = 3 * 455.8
HRNET_FWD_MB_PER_TILE = {'bf16': 3340.0, 'fp32': 6680.0}
HRNET_WEIGHT_MB = 38.5                                      # 9.64 M parameters in fp32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default=os.environ.get('CDNET_BENCH_MODE', 'train'), choices=['auto', 'train', 'infer', 'image', 'roofline', 'forced_allreduce'],
                    help="'roofline': only the dominant-kernel measurement of the roofline object (the command profiles/<round>/dominant_conv_* are taken with); "
                         "'forced_allreduce': only the 1-rank step with the all-reduce forced (the child process of the default run's dp1_forced_allreduce leg)")
    ap.add_argument('--dtype', default=os.environ.get('CDNET_BENCH_DTYPE', 'fp32'), choices=['bf16', 'fp32'],
                    help='arithmetic of the headline value; the other precision is reported beside it')
    ap.add_argument('--batch', type=int, default=None, help='tiles per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='only the headline measurement (+ roofline)')
    ap.add_argument('--no-infer-extra', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-forced-allreduce', action='store_true', help='skip the 1-rank forced all-reduce leg (it creates a one-rank RCCL group at the end of the run)')
    ap.add_argument('--no-live-pmc', action='store_true', help='quote the dominant kernel\'s HBM traffic from the committed summary instead of two rocprofv3 --pmc child passes')
    return ap.parse_args()


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _median_time(fn, warm=2, reps=5, budget_s=60.0):
    """BASELINE.md section 4 protocol: `warm` warm-ups, then >= `reps` timed repetitions (fewer only if the time budget is
    exhausted - reported), median wall-clock"""
    for _ in range(warm):
        fn()
    ts, t_all = [], time.time()
    for _ in range(reps):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
        if time.time() - t_all > budget_s and len(ts) >= 3:
            break
    ts.sort()
    return ts[len(ts) // 2], len(ts)


def cpu_baseline_train(n_tiles=4):
    """The fp32 PyTorch-CPU oracle train iteration (oracle/train.py, pinned to the reference's train_util_dam.train) on a
    bounded sample of the same synthetic workload."""
    import torch
    from cdnet_amd.trainer import synthetic_batch
    from oracle import models as om
    from oracle import train as ot
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = om.Unet()
    opt = ot.make_adam(net)
    x, lab, dirn, point, weight = [t.cpu() for t in synthetic_batch(n_tiles, torch.device('cpu'))]
    med, reps = _median_time(lambda: ot.train_iteration(net, opt, x, lab, dirn, point, weight), 2, 5, 75.0)
    return dict(value=n_tiles / med, unit='tiles/s', cores=cores, kind='port', cpu=cpu_model(), seconds_per_iteration=med,
                sample='%d synthetic 256x256 tiles per iteration (the workload of the GPU run at batch %d): oracle fp32 PyTorch-CPU train '
                       'iteration (forward, 5 losses, autograd backward, Adam; %d threads), median of %d repetitions after 2 warm-ups'
                       % (n_tiles, n_tiles, cores, reps))


def cpu_baseline_infer(n_tiles=4):
    """The oracle (PyTorch fp32 CPU network + plain-C post-processing) timed on the host: a bounded sample."""
    import numpy as np
    import torch
    from cdnet_amd import synth
    from oracle import models as om
    from oracle import postproc as orc
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = om.Unet().eval()
    x = torch.from_numpy(synth.tiles_u8(n_tiles).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous()

    def run():
        with torch.no_grad():
            mask, point, direction = net(x)
        for i in range(n_tiles):
            prob, dcm = orc.probmaps(mask[i].numpy(), direction[i].numpy())
            ddm = orc.generate_dd_map(dcm, 9)
            with np.errstate(all='ignore'):
                r = orc.fuse_boost_argmax(prob[None], point[i].numpy()[None], ddm[None])
            orc.cc_chain(r['pred'] == 1, 20, 2)
    med, reps = _median_time(run, 2, 5, 45.0)
    return dict(value=n_tiles / med, unit='tiles/s', cores=cores, kind='port', cpu=cpu_model(), seconds_per_iteration=med,
                sample='%d synthetic 256x256 tiles: oracle fp32 PyTorch-CPU UNet2RevA1_vgg16 forward (%d threads) + plain-C '
                       'probmaps/DDM/boost/CC chain (1 thread), median of %d repetitions after 2 warm-ups' % (n_tiles, cores, reps))


def cpu_baseline_cdm(n_tiles=16):
    """oracle/cdm_oracle.c (the plain-C restatement of LabelEncoding + get_centerpoint2, my_transforms_direction.py:650-885) on the label
    images of the GPU leg, one thread (the reference runs it per sample inside a DataLoader worker)"""
    from oracle import cdm as oc
    labs = cdm_labels(n_tiles)

    def run():
        for b in range(n_tiles):
            oc.label_encoding(labs[b])
    med, reps = _median_time(run, 1, 5, 30.0)
    return dict(value=n_tiles / med, unit='tiles/s', cores=1, kind='port', cpu=cpu_model(), seconds_per_iteration=med,
                sample='%d synthetic 256x256 label images (60 ellipse nuclei each, SURVEY 8d recipe): oracle/cdm_oracle.c, 1 thread, median of '
                       '%d repetitions after 1 warm-up (a C port: the reference\'s numba / per-nucleus Python loop is slower)' % (n_tiles, reps))


def cpu_baseline_image_postproc():
    """the direction-difference maps of the 8 views + mean / boost / arg-max + CC chain of ONE 1000x1000 image
    (getDirectionDiffMap.py:44, test_dam.py:455-563) by the plain-C oracle, 1 thread"""
    from cdnet_amd import synth
    from oracle import postproc as orc
    probs, points, dcms = synth.postproc_case(1000, 1000, 500, 5)
    med, reps = _median_time(lambda: orc.postprocess_views(probs, points, dcms), 1, 5, 30.0)
    return dict(value=1.0 / med, unit='images/s', cores=1, kind='port', cpu=cpu_model(), seconds_per_iteration=med,
                sample='one synthetic 1000x1000 image, 8 views (500 nuclei): oracle/postproc_oracle.c generate_dd_map x 8 + fuse / boost / argmax + '
                       'fill holes / remove small / label / dilate, 1 thread, median of %d repetitions after 1 warm-up (the network forward is '
                       'not part of this leg)' % reps)


def cpu_baseline_unet(n_tiles=4):
    """BASELINE config 1 as the reference runs it: plain UNet (models/unet.py:53-106) train step on 4 x 256x256x3 tiles, PyTorch fp32 CPU
    (oracle/models.py UNet + oracle/train.py unet_train_iteration, pinned to train_util.train by tests/golden/unet_train_iter.npz)"""
    import torch
    from oracle import models as om
    from oracle import train as ot
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = om.UNet(3)
    opt = ot.make_adam(net)
    x, lab, weight = [t.cpu() for t in unet_batch(n_tiles, torch.device('cpu'))]
    med, reps = _median_time(lambda: ot.unet_train_iteration(net, opt, x, lab, weight), 1, 3, 40.0)
    return dict(value=n_tiles / med, unit='tiles/s', cores=cores, kind='port', cpu=cpu_model(), seconds_per_iteration=med,
                sample='%d synthetic 256x256 tiles per iteration = BASELINE config 1 at its own size: oracle fp32 PyTorch-CPU UNet train iteration '
                       '(%d threads), median of %d repetitions after 1 warm-up' % (n_tiles, cores, reps))


def cdm_labels(n_tiles, seed=2022):
    """channel 0 of the 3-class label PNG (> 127 = inside) for `n_tiles` synthetic 256x256 tiles with 60 ellipse nuclei each"""
    import numpy as np
    from cdnet_amd import synth
    rs = np.random.RandomState(seed)
    return np.stack([(synth.ellipse_instances(256, 256, 60, rs, 5, 12, 10) > 0).astype(np.uint8) * 255 for _ in range(n_tiles)])


def unet_batch(B, dev, seed=2022):
    """config 1's batch: uniform RGB tiles, {0,1,2} labels from ellipse nuclei, constant weight map 20"""
    import torch
    from cdnet_amd.trainer import synthetic_batch
    x, lab, _, _, weight = synthetic_batch(B, dev, seed=seed)
    return x, lab, weight


def child_json(args, timeout_s):
    """one more `bench.py` as a fresh child process in its own process group (killed and reaped as a group on a timeout); returns the dict of
    its JSON line or a dict(error=...).  Used for the leg that creates an RCCL process group: a hang there must not take the main line with it,
    and nothing is ever re-executed in a process that holds the GPU."""
    import signal
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), cwd=ROOT, env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, start_new_session=True, text=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.communicate()
        return dict(error='child exceeded %d s (its process group was killed and reaped)' % timeout_s)
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    if proc.returncode != 0 or not lines:
        return dict(error='child failed (rc %s)' % proc.returncode)
    return json.loads(lines[-1])


def box_calibration(torch, dev, copy_bytes=1 << 30):
    """What THIS box grants, measured before anything else (boxes of the pool differ by 10-25 %): a float4 copy over 1 GiB
    (read + write bytes / time) and a v_mfma_f32_32x32x16_bf16 loop on random LDS-fed operands, one 8-wave workgroup per CU, settled for
    ~1.5 s, with the clock the chip holds inside that loop (s_memtime / s_memrealtime stamps, median over workgroups) - cdnet_box_copy /
    cdnet_box_mfma (csrc/box.hip)."""
    import numpy as np
    from cdnet_amd import _lib
    n = copy_bytes // 4
    src = torch.empty((n,), dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    st = _lib.stream_ptr

    def copy():
        _lib.call('cdnet_box_copy', _lib.ptr(src), _lib.ptr(dst), copy_bytes, st())
    for _ in range(5):
        copy()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        copy()
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 2.0 * copy_bytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    rs = np.random.RandomState(1)
    bf = lambda k: ((rs.randint(0, 2, k) << 15) | ((120 + rs.randint(0, 7, k)) << 7) | rs.randint(0, 128, k)).astype(np.uint32)    # random bf16 in [-1, 1)
    seed = torch.from_numpy((bf(16384) | (bf(16384) << 16)).view(np.int32).copy()).to(dev)
    sink = torch.zeros((4,), dtype=torch.float32, device=dev)
    WG, WAVES, ITERS = 256, 8, 20000
    stamps = torch.zeros((2 * WG,), dtype=torch.int64, device=dev)

    def mfma(stamp):
        _lib.call('cdnet_box_mfma', _lib.ptr(seed), _lib.ptr(sink), _lib.ptr(stamps) if stamp else None, WG, WAVES, ITERS, st())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:                 # settle: the clock under matrix load is reached after about a second
        mfma(False)
        torch.cuda.synchronize()
    reps = 10
    e0.record()
    for _ in range(reps):
        mfma(False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tflops = WG * WAVES * ITERS * 4 * 32768.0 / (ms * 1e-3) / 1e12
    mfma(True)
    torch.cuda.synchronize()
    sc = stamps.cpu().numpy().reshape(WG, 2).astype(np.float64)
    clk = np.median(sc[:, 0] / np.maximum(sc[:, 1], 1.0)) * 100.0           # MHz: shader cycles per tick of the constant 100 MHz counter
    return dict(copy_GBs=copy_gbs, copy_bytes=copy_bytes, mfma_TFLOPs=tflops, mfma_clock_mhz=float(clk), mfma_ms_per_launch=ms,
                mfma_loop='v_mfma_f32_32x32x16_bf16, operands re-read from LDS (ds_read_b128), random bf16, %d workgroups x %d waves x %d iterations '
                          '(4 MFMAs each), settled 1.5 s' % (WG, WAVES, ITERS),
                note='measured in this run before any other leg; `value_per_box_mfma` / `inference_per_box_copy` divide the rates by these')


def dominant_roofline(ms, B, precision, traffic=None, traffic_src=None, mfma_busy=None, clock=None, kernel=None):
    """The `roofline` object of the dominant layer (3x3 conv 64->64 @256x256 x B tiles) from its measured launch duration `ms` (pure
    arithmetic - tests/test_bench_contract.py calls it on the CPU).  `frac` is the ALGORITHMIC fraction of SURVEY 8d:
    max(algorithmic flops / dense bf16 MFMA peak, algorithmic bytes / HBM peak) / measured time - 2*MACs of the layer and one read of
    its input + one write of its output, whatever the kernel does internally.  The fp32 mode's three bf16 MFMAs per product are the
    strategy's own overhead, not useful work: that view (3 x flops against the same peak) stays under `mfma_work_frac`."""
    f32 = precision == 'fp32'
    esz = 4 if f32 else 2
    flops = 2.0 * B * 256 * 256 * 64 * 64 * 9             # algorithmic: 2*MACs of this layer (forward hooks)
    alg_bytes = B * 256 * 256 * (64 + 64) * esz           # algorithmic: input read once + output written once
    t_mfma = flops / 1e12 / DENSE_BF16_PEAK_TFLOPS * 1e3  # ms at the dense bf16 peak
    t_hbm = alg_bytes / 1e9 / HBM_PEAK_GBS * 1e3          # ms at the HBM peak
    gbs = alg_bytes / ms / 1e6
    tfl = flops / ms / 1e9
    # bf16: 30.9 us of MFMA vs 33.6 us of HBM; fp32: 30.9 us vs 67.1 us -> the HBM term bounds this layer in both modes
    if t_mfma > t_hbm:
        head = dict(bound='mfma', achieved=tfl, peak=DENSE_BF16_PEAK_TFLOPS, unit='TFLOP/s', frac=t_mfma / ms)
    else:
        head = dict(bound='hbm', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s', frac=t_hbm / ms)
    work = 3 if f32 else 1                                # bf16 MFMAs issued per algorithmic product
    return dict(**head, hbm_GBs=gbs, hbm_frac=gbs / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src, kernel=kernel, dtype=precision,
                mfma_busy_frac=mfma_busy, clock_mhz=clock, ms_per_launch=ms, algorithmic_bytes=alg_bytes, algorithmic_flops=flops,
                mfma_tflops=tfl, mfma_frac=tfl / DENSE_BF16_PEAK_TFLOPS,
                mfma_work_frac=work * tfl / DENSE_BF16_PEAK_TFLOPS, mfma_work_per_product=work,
                vs_fp32_mfma_peak=(tfl / FP32_MATRIX_PEAK_TFLOPS if f32 else None))


def committed_pmc(precision, B=16):
    """(traffic bytes per launch, source note, matrix-pipe busy fraction, clock MHz) of the dominant kernel at B tiles per launch from the newest
    committed PMC summary (tools/prof_roofline_pmc.sh -> profiles/<round>/dominant_conv_<dtype>[_<B>tiles]_pmc.json)"""
    name = 'dominant_conv_%s%s_pmc.json' % (precision, '' if B == 16 else '_%dtiles' % B)
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        tj = os.path.join(ROOT, 'profiles', rnd, name)
        if os.path.exists(tj):
            with open(tj) as f:
                pj = json.load(f)
            return (pj.get('hbm_bytes_per_launch'),
                    'profiles/%s/%s (rocprofv3 --pmc passes of `bench.py --mode roofline --batch %d`, not re-measured in this run)' % (rnd, name, B),
                    pj.get('mfma_busy_frac'), pj.get('clock_mhz'))
    return None, None, None, None


def live_pmc_traffic(precision, kernel_names, timeout_s=60, batch=16):
    """HBM bytes per launch of the dominant kernel measured IN THIS RUN when rocprofv3 is on the box: two child processes
    `rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 bench.py --mode roofline` (separate passes, the program right
    after `--`, from /tmp: MI355X_MICROARCH.md's HBM / rocprofv3 section), FETCH_SIZE x 2 on gfx950 (a 16-B/lane read stream counts
    half), both in KiB.  Returns (bytes, note) or (None, reason).  Children, not exec: this process keeps the GPU."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get('CDNET_BENCH_LIVE_PMC', '1') == '0':
        return None, 'disabled (CDNET_BENCH_LIVE_PMC=0)'
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on the box'
    if any('rocprof' in (os.environ.get(k) or '').lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'ROCPROFILER_REGISTER_ROOT')):
        return None, 'this process already runs under a profiler'
    env = dict(os.environ)
    env['TMPDIR'] = '/tmp'
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    vals = {}
    out = tempfile.mkdtemp(prefix='cdnet_pmc_', dir='/tmp')
    try:
        for ctrs in (('FETCH_SIZE',), ('WRITE_SIZE',), ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE')):
            d = os.path.join(out, ctrs[0])
            cmd = [exe, '--kernel-trace', '--pmc'] + list(ctrs) + ['--output-format', 'csv', '-d', d, '-o', 't', '--',
                   sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'roofline', '--dtype', precision, '--steps', '20', '--batch', str(batch)]
            # the profiler and the program it starts form their own process group: on a timeout the WHOLE group is killed and reaped
            # before anything else is timed (killing only the launcher would leave `bench.py --mode roofline` running on this GPU)
            proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
                    break                                   # (the traffic is there: the matrix-pipe pass is optional)
                return None, 'rocprofv3 --pmc %s pass exceeded %d s (its process group was killed and reaped)' % (ctrs[0], timeout_s)
            if rc != 0:
                if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
                    break
                return None, 'rocprofv3 --pmc %s failed (rc %d)' % (ctrs[0], rc)
            for ctr in ctrs:
                per = {}
                for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if row.get('Counter_Name') == ctr and any(w in row.get('Kernel_Name', '') for w in kernel_names):
                                per[row['Dispatch_Id']] = per.get(row['Dispatch_Id'], 0.0) + float(row['Counter_Value'])
                if len(per) < 10:
                    if ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
                        return None, 'no %s rows for the dominant kernel' % ctr
                    continue
                vals[ctr] = sum(per.values()) / len(per)
            if ctrs[0] == 'SQ_VALU_MFMA_BUSY_CYCLES':
                durs = []
                for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if any(w in row.get('Kernel_Name', '') for w in kernel_names):
                                durs.append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
                if len(durs) >= 10:
                    vals['duration_ns_under_pmc'] = sum(durs) / len(durs)
        traffic = vals['FETCH_SIZE'] * 1024 * 2 + vals['WRITE_SIZE'] * 1024
        extra = {}
        if 'GRBM_GUI_ACTIVE' in vals and 'duration_ns_under_pmc' in vals:
            # (tools/make_roofline_pmc.py's formulas: GRBM_GUI_ACTIVE sums the 8 XCDs; busy cycles sum the chip's 1024 SIMDs)
            extra['clock_mhz'] = vals['GRBM_GUI_ACTIVE'] / 8.0 / vals['duration_ns_under_pmc'] * 1e3
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in vals:
                extra['mfma_busy_frac'] = vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * vals['GRBM_GUI_ACTIVE'] / 8.0)
        LIVE_EXTRA.clear()
        LIVE_EXTRA.update(extra)
        return traffic, ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child passes of `bench.py --mode roofline '
                         '--dtype %s --batch %d`), per-launch mean, FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB units%s' %
                         (precision, batch, '; mfma_busy_frac / clock_mhz from a third live pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE)' if extra else ''))
    except Exception as e:                               # (timeouts, parse errors: the committed summary is the fallback)
        return None, 'live PMC pass failed: %r' % (e,)
    finally:
        shutil.rmtree(out, ignore_errors=True)


LIVE_EXTRA = {}          # mfma_busy_frac / clock_mhz of the last successful live_pmc_traffic call (its third pass)


def time_dominant_conv(torch, B, steps=20, precision='bf16', settle_s=0.5, live_pmc=False):
    """Average launch duration of the dominant kernel (3x3 conv 64->64 @256x256, the head/stem shape), measured live
    with HIP events on the stream the kernel is launched on (torch's current stream == the ABI stream argument)."""
    from cdnet_amd import engine
    f32 = precision == 'fp32'
    dev = torch.device('cuda', torch.cuda.current_device())
    x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.float32 if f32 else torch.bfloat16)
    w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
    cfg = engine.choose_cfg([64], 64, 256, 256, f32=f32)
    wp = engine.pack_weights(w, cfg, 0, split=f32)
    out = torch.empty((B, 256, 256, 64), dtype=x.dtype, device=dev)
    # settle: the chip lowers its clock under load over about a second; time the launches it would see inside a step, not a cold burst
    t_s, n_s = time.perf_counter(), 0
    while n_s < 3 or (time.perf_counter() - t_s < settle_s and n_s < 1500):
        engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
        n_s += 1
        if n_s % 16 == 0:
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    name = engine.dominant_kernel_name(precision, B)
    # HBM traffic of this kernel: measured in this run by PMC child passes when asked and possible, else quoted from the committed
    # summary; matrix-pipe busy fraction and clock always from the committed summary (six more passes: tools/prof_roofline_pmc.sh)
    traffic, traffic_src, mfma_busy, clock = committed_pmc(precision, B)
    if live_pmc:
        del x, out
        torch.cuda.synchronize()
        live, note = live_pmc_traffic(precision, (name.split('<')[0],), batch=B)
        if live is not None:
            traffic, traffic_src = live, note
            mfma_busy, clock = LIVE_EXTRA.get('mfma_busy_frac', mfma_busy), LIVE_EXTRA.get('clock_mhz', clock)
        elif traffic_src is not None:
            traffic_src += '; live pass: ' + note
    return dominant_roofline(ms, B, precision, traffic, traffic_src, mfma_busy, clock,
                             kernel='%s 3x3 64->64 @256x256 x%d tiles' % (name, B))


def path_roofline(kind, precision, tiles_per_step, ms_per_step):
    """Whole-step fraction of roofline in SURVEY 8d's units: roofline_time = max(FLOPs / peak_MFMA, bytes / peak_HBM) with the
    algorithmic per-tile figures (every convolution reads its input once and writes its output once; 2*MACs) against the dense bf16
    MFMA peak in BOTH modes.  `mfma_work_frac`: the matrix work the fp32 mode actually issues (three bf16 MFMAs per product)."""
    if kind == 'train':
        gflop = TRAIN_GFLOP_PER_TILE * tiles_per_step
        mb = TRAIN_MB_PER_TILE[precision] * tiles_per_step + OPT_MB_PER_STEP
    else:
        gflop = INFER_GFLOP_PER_TILE * tiles_per_step
        mb = (INFER_MB_PER_TILE[precision] + POSTPROC_MB_PER_TILE) * tiles_per_step + WEIGHT_MB[precision]
    peak_tf = DENSE_BF16_PEAK_TFLOPS
    t_flop = gflop / peak_tf                      # ms  (GFLOP / (TFLOP/s) = ms)
    t_hbm = mb / HBM_PEAK_GBS                     # ms  (MB / (GB/s) = ms)
    t_roof = max(t_flop, t_hbm)
    work = 3 if precision == 'fp32' else 1
    out = dict(frac=t_roof / ms_per_step, bound='mfma' if t_flop > t_hbm else 'hbm', roofline_ms=t_roof, hbm_ms=t_hbm, mfma_ms=t_flop,
               hbm_frac=t_hbm / ms_per_step, mfma_frac=t_flop / ms_per_step, mfma_work_frac=work * t_flop / ms_per_step,
               achieved_GBs=mb / ms_per_step, achieved_TFLOPs=gflop / ms_per_step,
               algorithmic_MB_per_step=mb, algorithmic_GFLOP_per_step=gflop, peak_GBs=HBM_PEAK_GBS, peak_TFLOPs=peak_tf)
    if precision == 'fp32':
        out['vs_fp32_mfma_peak'] = gflop / ms_per_step / FP32_MATRIX_PEAK_TFLOPS
    return out


def step_roofline(gflop, mb, ms_per_step, precision):
    """whole-step fraction of roofline from algorithmic GFLOP and MB per step (SURVEY 8d's units), against the dense bf16 MFMA peak and 8 TB/s"""
    t_flop, t_hbm = gflop / DENSE_BF16_PEAK_TFLOPS, mb / HBM_PEAK_GBS
    t_roof = max(t_flop, t_hbm)
    return dict(frac=t_roof / ms_per_step, bound='mfma' if t_flop > t_hbm else 'hbm', roofline_ms=t_roof, hbm_ms=t_hbm, mfma_ms=t_flop,
                hbm_frac=t_hbm / ms_per_step, mfma_frac=t_flop / ms_per_step, achieved_GBs=mb / ms_per_step, achieved_TFLOPs=gflop / ms_per_step,
                algorithmic_MB_per_step=mb, algorithmic_GFLOP_per_step=gflop, peak_GBs=HBM_PEAK_GBS, peak_TFLOPs=DENSE_BF16_PEAK_TFLOPS,
                mfma_work_per_product=3 if precision == 'fp32' else 1)


def cpu_baseline_hrnet(n_tiles=1):
    """BASELINE config 5's network as the reference runs it: HRNet18_rev1 (models/dam/seg_hrnet_rev1.py:289-548) train iteration on one
    512x512 tile, PyTorch fp32 CPU (oracle/hrnet.py + oracle/train.py train_iteration, pinned by tests/golden/hrnet_train.npz)"""
    import torch
    from cdnet_amd.trainer import synthetic_batch
    from oracle import hrnet as oh
    from oracle import train as ot
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = oh.HighResolutionNet(3)
    opt = ot.make_adam(net)
    x, lab, dirn, point, weight = [t.cpu() for t in synthetic_batch(n_tiles, torch.device('cpu'), H=512, W=512)]
    med, reps = _median_time(lambda: ot.train_iteration(net, opt, x, lab, dirn, point, weight), 1, 3, 30.0)
    return dict(value=n_tiles / med, unit='tiles/s', cores=cores, kind='port', cpu=cpu_model(), seconds_per_iteration=med,
                sample='%d synthetic 512x512 tile per iteration (the GPU leg runs 4): oracle fp32 PyTorch-CPU HRNet18_rev1 train iteration '
                       '(forward, 5 losses, autograd backward, Adam; %d threads), median of %d repetitions after 1 warm-up' % (n_tiles, cores, reps))


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (RCCL's version banner at communicator creation)
    # goes to stderr - file descriptor 1 points at stderr until the line is printed
    sys.stdout.flush()
    fd_out = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):       # launched by torch.distributed.run (also with one rank)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)
    import cdnet_amd
    from cdnet_amd import synth, pipeline, trainer, streams
    from cdnet_amd.models.dam.model_unet_rev1 import Unet

    mode = 'train' if a.mode == 'auto' else a.mode
    if mode == 'roofline':
        # exactly the measurement that fills the `roofline` object of the normal run, alone in the process, so that
        # `rocprofv3 --kernel-trace --stats -- python3 bench.py --mode roofline` averages this kernel and nothing else
        sys.stdout.flush()
        os.dup2(fd_out, 1)
        print(json.dumps({'roofline': time_dominant_conv(torch, a.batch or 16, steps=a.steps, precision=a.dtype)}), flush=True)
        return

    def timed(step, steps, warmup, settle_s=1.0):
        """(settling: >= settle_s seconds of the same load first - the chip lowers its clock under load over about a second, and a
        timed region that starts cold reads high) then W warm-ups, then exactly K steps between barrier + synchronize on both sides;
        max over ranks"""
        step()                                      # (allocations, kernel attributes, weight packs)
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        step()
        torch.cuda.synchronize()
        t1 = torch.tensor([time.perf_counter() - t_s], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t1, op=dist.ReduceOp.MAX)      # every rank settles for the same number of steps (collectives inside)
        for _ in range(min(2000, int(settle_s / max(float(t1.item()), 1e-4)))):
            step()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def new_model():
        torch.manual_seed(2022)
        return Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)

    def run_train(precision, B, steps, warmup):
        cdnet_amd.set_precision(precision)
        step, metric, workload = trainer.make_bench_step(new_model(), B, dev, rank, world)
        dt = timed(step, steps, warmup)
        return dict(metric=metric, workload=workload, value=world * B * steps / dt, ms_per_step=dt / steps * 1e3, tiles=B, steps=steps)

    def run_infer(precision, B, steps, warmup):
        cdnet_amd.set_precision(precision)
        model = new_model().eval()
        x = torch.from_numpy(synth.tiles_u8(B, seed=2022 + rank).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous().to(dev)
        # batch i's post-processing chain is queued on a second stream and runs beside batch i + 1's forward (pipeline.infer_tiles);
        # every batch's chain has finished when the timed region's closing synchronize returns
        post = streams.side_stream(dev)             # (a stream on another hardware queue than the compute stream's)
        last = {}

        def step():
            last['r'] = pipeline.infer_tiles(model, x, post_stream=post)
        dt = timed(step, steps, warmup)
        pipeline.check_tiles(last['r'])             # the reference's constant-DDM assertion (test_dam.py:535), from the device flag, outside the timed region
        return dict(metric='tiles/sec inference incl. post-proc, 256x256',
                    workload='CDNet UNet2RevA1_vgg16 (UNet+DAM) inference + direction-diff/CC post-processing, 256x256x3 synthetic tiles '
                             '(post-processing of batch i on a second stream beside the forward of batch i+1)',
                    value=world * B * steps / dt, ms_per_step=dt / steps * 1e3, tiles=B, steps=steps)

    def run_image(precision, steps, warmup):
        """BASELINE config 3: one 1000x1000 image per rank and step, 8 TTA views x 25 sliding windows (256/40), per-view DDM, CC chain"""
        cdnet_amd.set_precision(precision)
        model = new_model().eval()
        img = torch.from_numpy(np.random.RandomState(2022 + rank).randint(0, 256, size=(3, 1000, 1000)).astype(np.float32) / 255.0).to(dev)
        dt = timed(lambda: pipeline.infer_image(model, img, tta=True, all_img_test=0, patch_size=256, overlap=40), steps, warmup)
        return dict(metric='images/sec, 1000x1000 image, 8-view TTA x 25 sliding windows + post-proc',
                    workload='CDNet inference of 1000x1000 images (BASELINE config 3): 8 TTA views x 25 windows of 256/40, per-view '
                             'direction-difference maps, boost, CC chain',
                    value=world * steps / dt, ms_per_step=dt / steps * 1e3, window_evaluations_per_s=world * 200 * steps / dt, steps=steps)

    def run_cdm(B, steps, warmup):
        """centripetal-direction-map generation (`LabelEncoding` + `get_centerpoint2`, my_transforms_direction.py:650-885) of
        one batch of label images on the device: label image u8 -> 3-class label u8, centre-point map f16, direction classes u8"""
        from cdnet_amd.my_transforms_direction import label_encoding_batch
        lab0 = torch.from_numpy(cdm_labels(B, 2022 + rank)).to(dev)
        dt = timed(lambda: label_encoding_batch(lab0), steps, warmup, settle_s=0.3)
        ms = dt / steps * 1e3
        alg = B * 256 * 256 * 5                    # algorithmic bytes: 1 B/px read (label) + 1 + 2 + 1 B/px written (label3, point f16, direction)
        return dict(metric='tiles/sec, centripetal-direction-map generation (LabelEncoding), 256x256 labels with 60 nuclei', value=world * B * steps / dt,
                    unit='tiles/s', ms_per_batch=ms, tiles_per_gpu_per_step=B, steps=steps, warmup=warmup, dtype='u8/f64',
                    roofline=dict(bound='hbm', achieved=alg / ms / 1e6, peak=HBM_PEAK_GBS, unit='GB/s', frac=alg / ms / 1e6 / HBM_PEAK_GBS,
                                  algorithmic_bytes=alg, traffic=None,
                                  note='5 B/pixel algorithmic (u8 label in; u8 label3 + f16 point + u8 direction out).  The chain is 14 launches; '
                                       'its time is the centre search (8 rays x 30 dependent fp64 bisection gathers per nucleus pixel) and the '
                                       'labelling passes - latency-bound work, priced against HBM only because the path has no other roof'))

    def run_train_e2e(precision, B, steps, warmup, prefetch):
        """the training step WITH its input pipeline's device part: target generation of the batch (label_encoding_batch, as
        cdnet_amd/data_folder.py runs it per batch) + train_step, per iteration.  prefetch=False: both on the compute stream, one after the
        other.  prefetch=True: the targets of batch i + 1 are generated on the probed side stream while the step of batch i runs (double
        buffering, what a prefetching loader does); every iteration still generates one batch of targets and the closing synchronize covers
        both streams"""
        from cdnet_amd.my_transforms_direction import label_encoding_batch
        cdnet_amd.set_precision(precision)
        tr = trainer.Trainer(new_model(), world_size=world)
        x, _, _, _, weight = trainer.synthetic_batch(B, dev, seed=2022 + rank)
        lab0 = torch.from_numpy(cdm_labels(B, 2022 + rank)).to(dev)

        def targets():
            l3, point, dirn = label_encoding_batch(lab0)
            return torch.div(l3, 127, rounding_mode='floor'), dirn, point         # {0,127,255} -> {0,1,2} (train_util_dam.py:107-108)
        if not prefetch:
            def step():
                lab, dirn, point = targets()
                tr.train_step(x, lab, dirn, point, weight)
        else:
            side = streams.side_stream(dev, 1)       # (a stream of the loader's own: not the trainer's weight-gradient stream, not its queue)
            state = {}

            def produce():
                main = torch.cuda.current_stream()
                side.wait_stream(main)                            # (lab0 and the allocator's blocks are the compute stream's)
                with torch.cuda.stream(side):
                    t = targets()
                    ev = torch.cuda.Event()
                    ev.record(side)
                for q in t:
                    q.record_stream(main)                         # (made on the side stream, read by the step on the compute stream)
                state['next'] = (t, ev)
            produce()

            def step():
                (lab, dirn, point), ev = state['next']
                torch.cuda.current_stream().wait_event(ev)
                produce()                                         # batch i + 1's targets: queued before step i, run beside it
                tr.train_step(x, lab, dirn, point, weight)
        dt = timed(step, steps, warmup)
        return dict(value=world * B * steps / dt, unit='tiles/s', ms_per_step=dt / steps * 1e3, tiles_per_gpu_per_step=B, steps=steps, warmup=warmup,
                    dtype=precision, workload='label_encoding_batch (CDM generation of one batch on the device) + train_step per iteration; ' +
                    ('targets of batch i + 1 generated on a third stream (the loader\'s own: neither the compute nor the weight-gradient stream) beside step i (prefetching loader)' if prefetch else 'same stream, one after the other'))

    def run_unet_cfg1(precision, B, steps, warmup):
        """BASELINE config 1 at its own size: plain UNet (models/unet.py:53-106) train step, 4 x 256x256x3 tiles"""
        from cdnet_amd.models.unet import UNet
        cdnet_amd.set_precision(precision)
        torch.manual_seed(2022)
        tr = trainer.UNetTrainer(UNet(num_classes=3).to(dev), world_size=world)
        x, lab, weight = unet_batch(B, dev, seed=2022 + rank)
        dt = timed(lambda: tr.train_step(x, lab, weight), steps, warmup, settle_s=0.5)
        return dict(metric='tiles/sec (train fwd+bwd+Adam), plain UNet 3-class, 256x256', value=world * B * steps / dt, unit='tiles/s',
                    ms_per_step=dt / steps * 1e3, tiles_per_gpu_per_step=B, steps=steps, warmup=warmup, dtype=precision,
                    workload='BASELINE config 1: UNet 3-class (31.04 M parameters), %d x 256x256x3 synthetic tiles, forward, CE x weight + dice, backward, Adam' % B)

    def run_hrnet_cfg5(precision, B, steps, warmup):
        """BASELINE config 5's per-rank shape: HRNet18_rev1 + DAM head (models/dam/seg_hrnet_rev1.py:289-548), B x 512x512x3 tiles, train step"""
        from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet

        class _Cfg:
            model = {'out_c': 3}
        cdnet_amd.set_precision(precision)
        torch.manual_seed(2022)
        tr = trainer.Trainer(HighResolutionNet(_Cfg()).to(dev), world_size=world)
        batch = trainer.synthetic_batch(B, dev, seed=2022 + rank, H=512, W=512)
        dt = timed(lambda: tr.train_step(*batch), steps, warmup, settle_s=0.5)
        ms = dt / steps * 1e3
        out = dict(metric='tiles/sec (train fwd+bwd+Adam), HRNet18_rev1 + DAM head, 512x512', value=world * B * steps / dt, unit='tiles/s',
                   ms_per_step=ms, tiles_per_gpu_per_step=B, steps=steps, warmup=warmup, dtype=precision,
                   workload='BASELINE config 5 (per-rank shape): HRNet18_rev1 (9.64 M parameters) + DAM head, %d x 512x512x3 synthetic tiles, forward, '
                            '5-term loss, backward, fused Adam' % B,
                   roofline_path=step_roofline(HRNET_TRAIN_GFLOP_PER_TILE * B, 3 * HRNET_FWD_MB_PER_TILE[precision] * B + 4 * HRNET_WEIGHT_MB, ms, precision))
        del tr, batch
        torch.cuda.empty_cache()
        return out

    def run_image_postproc(steps, warmup):
        """BASELINE.md section 4 (d) on the device: everything after get_probmaps for one 1000x1000 image with 8 views (per-view
        direction-difference maps, mean, boost, arg-max, CC chain)"""
        from cdnet_amd import postproc
        probs, points, dcms = synth.postproc_case(1000, 1000, 500, 5)
        t = lambda a_: torch.from_numpy(a_).to(dev)[None]
        pr, po, dc = t(probs), t(points), t(dcms)
        dt = timed(lambda: postproc.postprocess_views(pr, po, dc), steps, warmup, settle_s=0.3)
        return dict(metric='images/sec, DDM x 8 views + boost/argmax + CC chain of one 1000x1000 image', value=world * steps / dt, unit='images/s',
                    ms_per_image=dt / steps * 1e3, steps=steps, warmup=warmup, dtype='u8/i32')

    def run_forced_allreduce(precision, B, steps, warmup):
        """the 1-rank step with the bucketed all-reduce path forced (CDNET_FORCE_ALLREDUCE=1) on a one-rank RCCL group: the same-box baseline
        for the overlap machinery's cost before the first multi-GPU run.  Runs LAST: RCCL's streams change the stream -> hardware-queue deal,
        the side stream is probed again after the group exists (cdnet_amd.streams)"""
        import socket
        made = False
        if not dist.is_initialized():
            with socket.socket() as so:
                so.bind(('127.0.0.1', 0))
                port = so.getsockname()[1]
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
            made = True
        os.environ['CDNET_FORCE_ALLREDUCE'] = '1'
        try:
            cdnet_amd.set_precision(precision)
            tr = trainer.Trainer(new_model(), world_size=1)
            batch = trainer.synthetic_batch(B, dev, seed=2022 + rank)
            dt = timed(lambda: tr.train_step(*batch), steps, warmup)
            st = tr.ar_stats or {}
            probe = next((p_ for p_ in reversed(streams.PROBES) if p_.get('k', 0) == 0), None)
            return dict(ms_per_step=dt / steps * 1e3, value=B * steps / dt, unit='tiles/s', steps=steps, warmup=warmup, dtype=precision,
                        buckets=st.get('buckets'), buckets_released_during_backward=st.get('released_during_backward'),
                        bucket_mb=tr.bucket * 4 / (1 << 20), process_group='%s, %d rank(s)' % (dist.get_backend(), dist.get_world_size()),
                        side_stream_probe=probe)
        finally:
            os.environ.pop('CDNET_FORCE_ALLREDUCE', None)
            if made:
                torch.cuda.synchronize()
                dist.destroy_process_group()

    if mode == 'forced_allreduce':
        fa = run_forced_allreduce(a.dtype, a.batch or 16, a.steps, a.warmup)
        sys.stdout.flush()
        os.dup2(fd_out, 1)
        print(json.dumps(fa), flush=True)
        return
    other = 'fp32' if a.dtype == 'bf16' else 'bf16'
    extras = not a.no_extras
    box = box_calibration(torch, dev)               # before anything else is timed: what this box grants (copy GB/s, MFMA TFLOP/s, clock)
    if mode == 'train':
        B = a.batch or 16
        head = run_train(a.dtype, B, a.steps, a.warmup)
    elif mode == 'infer':
        B = a.batch or 64
        head = run_infer(a.dtype, B, a.steps, a.warmup)
    else:
        B = 1
        head = run_image(a.dtype, a.steps, a.warmup)
    kind = 'train' if mode == 'train' else 'infer'
    line = {
        'metric': head['metric'], 'value': head['value'], 'unit': 'images/s' if mode == 'image' else 'tiles/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': head['workload'], 'mode': mode, 'tiles_per_gpu_per_step': B, 'global_batch': world * B,
                   'tile': '1000x1000x3' if mode == 'image' else '256x256x3', 'parallelism': 'dp%d' % world,
                   'precision': ('bf16/fp16 NHWC activations, bf16 MFMA operands, fp32 accumulation; label-level parity gate vs the fp32 oracle: '
                                 'tests/test_gpu_label_gate.py' if a.dtype == 'bf16' else
                                 'fp32 storage + accumulation (NHWC fp32 activations, gradients, master weights), products = 3 x bf16 MFMA over '
                                 'split operands (hi*hi + hi*lo + lo*hi), <= 2^-16 relative error per product ("bf16x3"; not IEEE-fp32 '
                                 'multiplication: gfx950 has no TF32 and its fp32 MFMA runs at 1/16 of the bf16 rate)')},
    }
    line['box'] = box
    # rates beside what the box itself grants: the fp32 / bf16 training steps are matrix-bound (per TFLOP/s of the box's MFMA loop), the
    # inference steps lean on the memory side (per GB/s of the box's copy) - the columns of DESIGN section 0's per-round table
    line['value_per_box_mfma'] = head['value'] / box['mfma_TFLOPs']
    line['config']['process_group'] = ('%s, %d rank(s)' % (dist.get_backend(), dist.get_world_size())) if dist.is_initialized() else None
    rp = {}
    if mode != 'image':
        rp[kind + '_' + a.dtype] = path_roofline(kind, a.dtype, B, head['ms_per_step'])
    if extras and mode == 'train' and not a.no_infer_extra:
        inf = run_infer(a.dtype, 64, 10, 2)
        line['inference'] = {'metric': inf['metric'], 'value': inf['value'], 'unit': 'tiles/s', 'ms_per_step': inf['ms_per_step'],
                             'tiles_per_gpu_per_step': 64, 'steps': inf['steps'], 'dtype': a.dtype, 'workload': inf['workload']}
        rp['infer_' + a.dtype] = path_roofline('infer', a.dtype, 64, inf['ms_per_step'])
        line['inference_per_box_copy'] = inf['value'] / box['copy_GBs']
    if extras and mode == 'train' and world == 1:
        # the same two rates in the other arithmetic, and BASELINE config 3, same protocol with fewer steps
        ksteps = max(5, a.steps // 2)
        t2 = run_train(other, B, ksteps, 2)
        i2 = run_infer(other, 64, 10, 2)
        line[other] = {'value': t2['value'], 'unit': 'tiles/s', 'ms_per_step': t2['ms_per_step'], 'steps': ksteps, 'warmup': 2,
                       'tiles_per_gpu_per_step': B, 'dtype': other,
                       'inference': {'value': i2['value'], 'unit': 'tiles/s', 'ms_per_step': i2['ms_per_step'], 'tiles_per_gpu_per_step': 64, 'steps': i2['steps']}}
        line[other]['value_per_box_mfma'] = t2['value'] / box['mfma_TFLOPs']
        line[other]['inference_per_box_copy'] = i2['value'] / box['copy_GBs']
        rp['train_' + other] = path_roofline('train', other, B, t2['ms_per_step'])
        rp['infer_' + other] = path_roofline('infer', other, 64, i2['ms_per_step'])
        IMG_STEPS, IMG_WARMUP = 8, 2
        im = run_image(a.dtype, IMG_STEPS, IMG_WARMUP)
        im2 = run_image(other, IMG_STEPS, IMG_WARMUP)
        line['image'] = {'metric': im['metric'], 'value': im['value'], 'unit': 'images/s', 'ms_per_image': im['ms_per_step'],
                         'window_evaluations_per_s': im['window_evaluations_per_s'], 'steps': im['steps'], 'warmup': IMG_WARMUP, 'dtype': a.dtype,
                         other: {'value': im2['value'], 'ms_per_image': im2['ms_per_step'], 'steps': im2['steps'], 'warmup': IMG_WARMUP}}
        line['image_postproc'] = run_image_postproc(10, 2)
        # SURVEY 8a-9 / VERDICT r04 #5: the target generation timed alone and inside the step
        line['cdm'] = run_cdm(B, 20, 3)
        e2e = run_train_e2e(a.dtype, B, max(5, a.steps // 2), 2, prefetch=True)
        e2e['vs_value'] = e2e['value'] / head['value']
        e2s = run_train_e2e(a.dtype, B, max(5, a.steps // 2), 2, prefetch=False)
        e2e['serial'] = {'value': e2s['value'], 'ms_per_step': e2s['ms_per_step'], 'vs_value': e2s['value'] / head['value'], 'workload': e2s['workload']}
        line['train_e2e'] = e2e
        # BASELINE config 1 at its own size
        line['unet_cfg1'] = run_unet_cfg1(a.dtype, 4, 10, 2)
        line['unet_cfg1']['roofline_path'] = step_roofline(UNET_TRAIN_GFLOP_PER_TILE * 4, 3 * UNET_FWD_MB_PER_TILE[a.dtype] * 4 + 4 * UNET_WEIGHT_MB,
                                                           line['unet_cfg1']['ms_per_step'], a.dtype)
        # BASELINE config 5 at its per-rank size (VERDICT r05 missing #3): the 16-bit step (the one profiles/<round>/hrnet_train_b4_512_* describe)
        # and the headline arithmetic's
        line['hrnet_cfg5'] = run_hrnet_cfg5('bf16', 4, 8, 2)
        if a.dtype != 'bf16':
            line['hrnet_cfg5'][a.dtype] = run_hrnet_cfg5(a.dtype, 4, 5, 2)
    if rp:
        line['roofline_path'] = rp
    line['config']['side_stream_probe'] = next((p_ for p_ in reversed(streams.PROBES) if p_.get('k', 0) == 0), None)       # (the trainer's / pipeline's second stream)
    if rank == 0:
        cdnet_amd.set_precision(a.dtype)
        line['roofline'] = time_dominant_conv(torch, 16, precision=a.dtype, live_pmc=(world == 1 and extras and not a.no_live_pmc))
        if extras and world == 1:
            # the 16-bit path's dominant layer at the INFERENCE batch (64 tiles: 1.07 GB of tensors, beyond the 256 MB Infinity Cache - at 16 tiles
            # the 268 MB working set is re-used across launches out of that cache and the figure flatters the kernel); 16 tiles beside it
            if other == 'bf16':
                line['roofline_bf16'] = time_dominant_conv(torch, 64, precision='bf16', live_pmc=not a.no_live_pmc)
                line['roofline_bf16_16tiles'] = time_dominant_conv(torch, 16, precision='bf16')
            else:
                line['roofline_' + other] = time_dominant_conv(torch, 16, precision=other)
        if extras and mode == 'train' and world == 1 and not a.no_forced_allreduce:
            # (a child process: the leg creates a one-rank RCCL group - a hang or a crash there costs this key, not the line)
            torch.cuda.synchronize()
            fa = child_json(['--mode', 'forced_allreduce', '--dtype', a.dtype, '--steps', str(max(5, a.steps // 2)), '--warmup', '2',
                             '--batch', str(B)], 240)
            if 'value' in fa:
                fa['vs_value'] = fa['value'] / head['value']
            line['dp1_forced_allreduce'] = fa
        if not a.no_cpu_baseline and world == 1:
            # every reported rate has its CPU leg beside it: the headline's under `cpu_baseline`, the others under
            # their own keys; all are bounded samples (a few repetitions of a few tiles / one image), ~40 s of host time in total
            line['cpu_baseline'] = cpu_baseline_infer() if mode == 'infer' else cpu_baseline_train()
            if extras and mode == 'train':
                line['cpu_baseline_infer'] = cpu_baseline_infer()
                line['cpu_baseline_cdm'] = cpu_baseline_cdm()
                line['cpu_baseline_image_postproc'] = cpu_baseline_image_postproc()
                line['cpu_baseline_unet_cfg1'] = cpu_baseline_unet()
                line['cpu_baseline_hrnet'] = cpu_baseline_hrnet()
        sys.stdout.flush()
        os.dup2(fd_out, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.barrier()                    # rank 0 times the roofline kernel after the timed region: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
