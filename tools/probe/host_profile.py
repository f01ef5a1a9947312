"""cProfile of the host side of the training step (which Python calls cost the launch path its time)"""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cdnet_amd import trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
step = trainer.make_bench_step(m, 16, dev, 0, 1)[0]
for _ in range(5):
    step()
torch.cuda.synchronize()
# host-only time of a step: issue while the GPU is far behind (no sync inside)
t = time.perf_counter()
for _ in range(10):
    step()
t_issue = (time.perf_counter() - t) / 10
torch.cuda.synchronize()
print('host issue time per step (GPU-bound overall, so this includes back-pressure): %.2f ms' % (t_issue * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
