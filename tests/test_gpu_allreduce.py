"""The exchange step of the training path (RCCL sum-all-reduce of the flat fp32 gradient buffer in buckets, overlapped with
backward; parameter broadcast; scalar mean) exercised on ONE GPU through a real 1-rank 'nccl' process group
(train.py:185-186 nn.DataParallel -> one process per GPU).  Runs in a child process so that the process group does
not leak into the other tests."""
import os
import socket
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu


def test_overlapped_allreduce_is_bit_identical_on_one_rank():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_allreduce_worker.py')
    r = subprocess.run([sys.executable, worker], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + '\n' + r.stderr[-4000:]
    assert 'allreduce ok' in r.stdout
