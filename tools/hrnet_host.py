"""Host-side issue time of one HRNet18_rev1 training step (4 x 512x512) against its GPU time: how long Python needs to QUEUE the step's ~1 300
launches (time until train_step returns, nothing synchronised) and how long the GPU needs to run them.  python3 tools/hrnet_host.py [bf16|fp32]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import runtime, trainer
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet


class O:
    model = {'out_c': 3}


runtime.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().train()
tr = trainer.Trainer(m)
batch = trainer.synthetic_batch(4, dev, seed=5, H=512, W=512)
for _ in range(5):
    tr.train_step(*batch)
torch.cuda.synchronize()
issue, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
issue.sort(); total.sort()
print('one step from an idle GPU: host issue %.2f ms, until the GPU is done %.2f ms (medians of 8)' % (issue[4], total[4]))
# forward / backward split of the host time
t0 = time.perf_counter(); out = tr.forward(batch[0]); t1 = time.perf_counter()
g = tr.loss_and_grads(out[0], out[1], out[2], *batch[1:]); t2 = time.perf_counter()
tr.backward(*g); t3 = time.perf_counter()
tr.allreduce_and_step(); t4 = time.perf_counter()
torch.cuda.synchronize()
print('host: forward %.2f ms, loss %.2f, backward %.2f, Adam + re-pack %.2f' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
