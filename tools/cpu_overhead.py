"""How far ahead of the GPU does the host run?  Enqueue time of a training step (no synchronisation inside) vs its GPU time."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet

dev = torch.device('cuda:0')
torch.manual_seed(2022)
m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
step, _, _ = trainer.make_bench_step(m, 16, dev, 0, 1)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
