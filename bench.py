#!/usr/bin/env python3
"""bench.py - throughput of the CDNet hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--mode train|infer] [--batch B]

One rank per GPU (the driver launches N>1 through torch.distributed.run); RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* come
from the environment.  W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; the step time
is the max over ranks; rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DENSE_BF16_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16/f16 MFMA
HBM_PEAK_GBS = 8000.0                # ibid.: 8 TB/s spec (6.3 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default=os.environ.get('CDNET_BENCH_MODE', 'auto'), choices=['auto', 'train', 'infer', 'roofline'],
                    help="'roofline': only the dominant-kernel measurement of the roofline object (the command profiles/<round>/dominant_conv_* are taken with)")
    ap.add_argument('--batch', type=int, default=None, help='tiles per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-infer-extra', action='store_true', help='train mode: skip the additional inference timing')
    return ap.parse_args()


def cpu_baseline_infer(n_tiles=2):
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
    run()
    t0 = time.time()
    reps = 2
    for _ in range(reps):
        run()
    dt = (time.time() - t0) / reps
    return dict(value=n_tiles / dt, unit='tiles/s', cores=cores, kind='port',
                sample='%d synthetic 256x256 tiles: oracle fp32 PyTorch-CPU UNet2RevA1_vgg16 forward (%d threads) + '
                       'plain-C probmaps/DDM/boost/CC chain (1 thread), %d repetitions' % (n_tiles, cores, reps))


def time_dominant_conv(torch, B, steps=20):
    """Average launch duration of the dominant kernel (3x3 conv 64->64 @256x256, the head/stem shape), measured live
    with HIP events on the stream the kernel is launched on (torch's current stream == the ABI stream argument)."""
    from cdnet_amd import engine
    dev = torch.device('cuda', torch.cuda.current_device())
    x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
    w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
    cfg = engine.choose_cfg([64], 64, 256, 256)
    wp = engine.pack_weights(w, cfg, 0)
    out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    flops = 2.0 * B * 256 * 256 * 64 * 64 * 9             # algorithmic: 2*MACs of this layer (forward hooks)
    alg_bytes = B * 256 * 256 * (64 + 64) * 2             # algorithmic: input read once + output written once, bf16
    # roofline time = max(flops / MFMA peak, bytes / HBM peak): 30.9 us vs 33.6 us at B=16 -> the HBM term bounds this layer
    traffic = None
    tj = os.path.join(ROOT, 'profiles', 'r01', 'dominant_conv_traffic.json')
    if os.path.exists(tj) and B == 16:
        with open(tj) as f:
            traffic = json.load(f).get('hbm_bytes_per_launch')       # PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes
    gbs = alg_bytes / ms / 1e6
    return dict(bound='hbm', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s', frac=gbs / HBM_PEAK_GBS, traffic=traffic,
                kernel='conv_fwd_kernel<%s> 3x3 64->64 @256x256 x%d tiles' % (','.join(str(c) for c in cfg), B),
                ms_per_launch=ms, algorithmic_bytes=alg_bytes, algorithmic_flops=flops,
                mfma_tflops=flops / ms / 1e9, mfma_frac=flops / ms / 1e9 / DENSE_BF16_PEAK_TFLOPS)


def cpu_baseline_train(n_tiles=2):
    """The fp32 PyTorch-CPU oracle train iteration (oracle/train.py, pinned to the reference) on a bounded sample."""
    import numpy as np
    import torch
    from cdnet_amd import synth
    from cdnet_amd.trainer import synthetic_batch
    from oracle import models as om
    from oracle import train as ot
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = om.Unet()
    opt = ot.make_adam(net)
    x, lab, dirn, point, weight = [t.cpu() for t in synthetic_batch(n_tiles, torch.device('cpu'))]
    ot.train_iteration(net, opt, x, lab, dirn, point, weight)
    t0 = time.time()
    reps = 2
    for _ in range(reps):
        ot.train_iteration(net, opt, x, lab, dirn, point, weight)
    dt = (time.time() - t0) / reps
    return dict(value=n_tiles / dt, unit='tiles/s', cores=cores, kind='port',
                sample='%d synthetic 256x256 tiles per iteration: oracle fp32 PyTorch-CPU train iteration (forward, 5 losses, '
                       'autograd backward, Adam; %d threads), %d repetitions after 1 warm-up' % (n_tiles, cores, reps))


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)
    from cdnet_amd import synth, pipeline
    from cdnet_amd.models.dam.model_unet_rev1 import Unet

    mode = a.mode
    if mode == 'roofline':
        # exactly the measurement that fills the `roofline` object of the normal run, alone in the process, so that
        # `rocprofv3 --kernel-trace --stats -- python3 bench.py --mode roofline` averages this kernel and nothing else
        print(json.dumps({'roofline': time_dominant_conv(torch, 16, steps=a.steps)}))
        return
    if mode == 'auto':
        try:
            from cdnet_amd import trainer  # noqa: F401
            mode = 'train'
        except ImportError:
            mode = 'infer'
    torch.manual_seed(2022)
    model = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)

    if mode == 'infer':
        B = a.batch or 64
        model.eval()
        x = torch.from_numpy(synth.tiles_u8(B, seed=2022 + rank).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous().to(dev)

        def step():
            return pipeline.infer_tiles(model, x)
        metric = 'tiles/sec inference incl. post-proc, 256x256'
        workload = 'CDNet UNet2RevA1_vgg16 (UNet+DAM) inference + direction-diff/CC post-processing, 256x256x3 synthetic tiles'
    else:
        from cdnet_amd import trainer
        B = a.batch or 16
        step, metric, workload = trainer.make_bench_step(model, B, dev, rank, world)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
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
    # the metric names both rates: in the default (train) run also time the inference + post-processing path, same
    # protocol (warm-up, barrier + synchronize on both sides, max over ranks); reported beside the headline value
    infer_extra = None
    if mode == 'train' and not a.no_infer_extra:
        model.eval()
        Bi = 64
        xi = torch.from_numpy(synth.tiles_u8(Bi, seed=4044 + rank).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous().to(dev)
        for _ in range(2):
            pipeline.infer_tiles(model, xi)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ti = time.perf_counter()
        ni = 5
        for _ in range(ni):
            pipeline.infer_tiles(model, xi)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dti = time.perf_counter() - ti
        if world > 1:
            t = torch.tensor([dti], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dti = float(t.item())
        infer_extra = {'metric': 'tiles/sec inference incl. post-proc, 256x256', 'value': world * Bi * ni / dti, 'unit': 'tiles/s',
                       'ms_per_step': dti / ni * 1e3, 'tiles_per_gpu_per_step': Bi, 'steps': ni}
    if rank == 0:
        roof = time_dominant_conv(torch, 16)
        line = {
            'metric': metric, 'value': world * B * a.steps / dt, 'unit': 'tiles/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': workload, 'mode': mode, 'tiles_per_gpu_per_step': B, 'global_batch': world * B,
                       'tile': '256x256x3', 'parallelism': 'dp%d' % world},
            'roofline': roof,
        }
        if infer_extra is not None:
            line['inference'] = infer_extra
        if not a.no_cpu_baseline and world == 1:
            if mode == 'infer':
                line['cpu_baseline'] = cpu_baseline_infer()
            else:
                line['cpu_baseline'] = cpu_baseline_train()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()                    # rank 0 times the roofline kernel after the timed region: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
