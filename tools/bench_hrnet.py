"""HRNet18_rev1 inference forward (BASELINE config 5 shape: 512x512 tiles) timing.  usage: python tools/bench_hrnet.py [B]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4


class O:
    model = {'out_c': 3}


torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().eval()
x = torch.rand((B, 3, 512, 512), device='cuda')
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
print('HRNet18_rev1 eval forward, %d tiles 512x512: %.2f ms = %.1f tiles/s' % (B, dt * 1e3, B / dt))

# training step (forward with batch-statistics BatchNorm, loss, backward, Adam) at the same shape
from cdnet_amd import trainer
m.train()
tr = trainer.Trainer(m)
batch = trainer.synthetic_batch(B, torch.device('cuda:0'), seed=5, H=512, W=512)
for _ in range(3):
    loss = tr.train_step(*batch)
torch.cuda.synchronize()
t = time.perf_counter()
K = 8
for _ in range(K):
    loss = tr.train_step(*batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print('HRNet18_rev1 train step, %d tiles 512x512: %.2f ms = %.1f tiles/s  (loss %.4f)' % (B, dt * 1e3, B / dt, float(loss[0])))

# the same step with forward + loss + backward replayed from a HIP graph (cdnet_amd.graphs)
from cdnet_amd.graphs import GraphedTrainStep, GraphedCallable
g = GraphedTrainStep(tr, batch, warmup=1)
for _ in range(2):
    g(*batch)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(K):
    loss = g(*batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print('HRNet18_rev1 train step (HIP graph), %d tiles 512x512: %.2f ms = %.1f tiles/s  (loss %.4f)' % (B, dt * 1e3, B / dt, float(loss[0])))
m.eval()
with torch.no_grad():
    gi = GraphedCallable(lambda t_: m(t_), x)
    for _ in range(2):
        gi(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        gi(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
print('HRNet18_rev1 eval forward (HIP graph), %d tiles 512x512: %.2f ms = %.1f tiles/s' % (B, dt * 1e3, B / dt))
