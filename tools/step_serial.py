"""The benchmark's training step with everything on ONE stream (trainer.WGRAD_STREAM = False): under rocprofv3 --kernel-trace every kernel's
duration is then its own, not inflated by the other stream's workgroups.  python3 tools/step_serial.py [fp32|bf16] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cdnet_amd
from cdnet_amd import trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cdnet_amd.set_precision(prec)
trainer.WGRAD_STREAM = False
trainer._RU_1X1_SIDE = False
dev = torch.device('cuda:0')
torch.manual_seed(2022)
step, _, _ = trainer.make_bench_step(Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev), 16, dev, 0, 1)
for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print('one stream: %.3f ms per step' % ((time.perf_counter() - t) / steps * 1e3))
