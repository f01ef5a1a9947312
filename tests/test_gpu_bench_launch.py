"""bench.py under the launcher the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P bench.py --gpus N ...`), with one rank - what a 1-GPU box can run of the 8-GPU contract: the RCCL process
group comes up from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, the Trainer takes its world size from it, rank 0 prints the one JSON
line.  Both runs are fresh child processes (the launcher starts before anything in that child touches the GPU); the reference's
counterpart is nn.DataParallel in train.py:185-186."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(cmd, timeout=600):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_under_torch_distributed_run_with_one_rank():
    args = ['bench.py', '--gpus', '1', '--steps', '4', '--warmup', '1', '--no-extras', '--no-cpu-baseline']
    plain = _run([sys.executable] + args)
    launched = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                     '--master-port', str(_free_port())] + args)
    for d in (plain, launched):
        assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['dtype'] == 'fp32'
        assert d['config']['parallelism'] == 'dp1' and d['config']['global_batch'] == 16
        assert abs(d['value'] - 16 / d['ms_per_step'] * 1e3) < 1e-6 * d['value']
        assert d['roofline']['bound'] == 'hbm' and 0.05 < d['roofline']['frac'] < 1.0
    assert plain['config']['process_group'] is None
    assert launched['config']['process_group'] == 'nccl, 1 rank(s)'           # RCCL group initialised from the launcher's environment
    # the launched run pays the (one-rank) bucketed all-reduce path's bookkeeping and nothing else; two 4-step runs in separate processes on a
    # shared, clock-managed GPU: the comparison is printed, the bound is a wide one
    print('plain %.1f tiles/s, launched %.1f tiles/s' % (plain['value'], launched['value']))
    assert abs(launched['value'] - plain['value']) <= 0.30 * plain['value'], (launched['value'], plain['value'])
    # the side stream was probed AFTER the process group existed (RCCL's streams change the stream -> hardware-queue deal)
    assert launched['config']['side_stream_probe']['group'] is True and launched['config']['side_stream_probe']['probed']
    assert plain['config']['side_stream_probe']['group'] is False
