"""The whole `Trainer` at world size 2 (train.py:185-186 nn.DataParallel -> one process per GPU) on the ONE GPU of the
test box: two child processes, gloo over the CUDA flat buffers.  After three steps

  * both ranks hold bit-identical parameters and Adam moments although they were built from different seeds (`sync_from_rank0`) and
    trained on different tiles (the exchange step), their BatchNorm running statistics differ (per-rank statistics, as the
    reference's DataParallel replicas), buckets were released while backward was still running;
  * they equal a SINGLE-process run of two trainers that starts from rank 0's weights, runs each rank's batch, sums the two flat
    gradient buffers by hand and lets every trainer take its Adam step on that sum with grad_scale 1/2 - no collective anywhere.

Covers tape order, side-stream bucket release, bucket-wise Adam, the broadcast and the scalar mean at N = 2 - everything but xGMI."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
STEPS = 3


def _run_ranks(tmp_path, precision):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_trainer_world2_worker.py')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), precision, str(STEPS)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, o[-2000:] + '\n' + e[-4000:]
    import torch
    return [torch.load(str(tmp_path / ('rank%d.pt' % r)), weights_only=True) for r in range(2)]


def _reference(p_start, precision):
    """one process, no collective: two trainers from rank 0's weights, each rank's batch, gradients summed by hand"""
    import torch
    import cdnet_amd
    from cdnet_amd import _lib, trainer
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    cdnet_amd.set_precision(precision)
    dev = torch.device('cuda', 0)
    trs, batches = [], []
    for r in range(2):
        torch.manual_seed(1000 + r)                   # (the same constructor draws as the worker: BatchNorm buffers start alike anyway)
        m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
        tr = trainer.Trainer(m, world_size=1, bucket_mb=4)
        tr.flat.P.copy_(p_start.to(dev))
        tr.refresh_parameters()
        trs.append(tr)
        batches.append(trainer.synthetic_batch(2, dev, seed=50 + r, H=64, W=64))
    for _ in range(STEPS):
        for tr, b in zip(trs, batches):
            mask, point, direction = tr.forward(b[0])
            tr.backward(*tr.loss_and_grads(mask, point, direction, *b[1:]))
        torch.cuda.synchronize()
        n = trs[0].flat.n_used
        gsum = trs[0].flat.G[:n] + trs[1].flat.G[:n]
        for tr in trs:
            f = tr.flat
            f.G[:n].copy_(gsum)
            f.step_count += 1
            _lib.call('cdnet_adam_step', _lib.ptr(f.P), _lib.ptr(f.G), _lib.ptr(f.M), _lib.ptr(f.V), n, tr.lr, tr.betas[0], tr.betas[1], tr.eps,
                      tr.wd, f.step_count, 0.5, _lib.stream_ptr())
            tr.refresh_parameters()
        torch.cuda.synchronize()
    return trs


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_two_ranks_train_identically_and_match_hand_averaged_gradients(tmp_path, precision):
    import torch
    import cdnet_amd
    before = cdnet_amd.get_precision()
    try:
        r0, r1 = _run_ranks(tmp_path, precision)
        assert torch.equal(r0['P_start'], r1['P_start']), 'sync_from_rank0 did not make the replicas identical'
        for k in ('P', 'M', 'V'):
            assert torch.equal(r0[k], r1[k]), 'ranks diverged in %s' % k
        assert not torch.equal(r0['P'], r0['P_start'])
        assert r0['early'] > 0 and r1['early'] > 0, 'no bucket was released during backward'
        assert not np.allclose(r0['losses'].numpy(), r1['losses'].numpy()), 'the ranks saw the same tiles'
        want = 0.5 * (r0['losses'].double() + r1['losses'].double())
        assert torch.allclose(r0['reduced'].double(), want, rtol=1e-6, atol=1e-9) and torch.equal(r0['reduced'], r1['reduced'])
        diff = [k for k in r0['buffers'] if k.endswith('running_mean') and not torch.equal(r0['buffers'][k], r1['buffers'][k])]
        assert diff, 'BatchNorm statistics are per rank (unsynchronised), yet every running mean is equal'
        trs = _reference(r0['P_start'], precision)
        n = trs[0].flat.n_used
        for k, attr in (('P', 'P'), ('M', 'M'), ('V', 'V')):
            got, ref = r0[k][:n], getattr(trs[0].flat, attr)[:n].cpu()
            assert torch.equal(got, ref), '%s differs from the hand-averaged single-process run: max |d| %g' % (k, float((got - ref).abs().max()))
        for r, tr in zip((r0, r1), trs):
            for name, b in tr.model.named_buffers():
                if b.is_floating_point():
                    assert torch.equal(r['buffers'][name], b.detach().cpu()), name
    finally:
        cdnet_amd.set_precision(before)
