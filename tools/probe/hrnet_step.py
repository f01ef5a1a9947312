import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
from cdnet_amd import trainer
class O:
    model = {'out_c': 3}
torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().train()
tr = trainer.Trainer(m)
batch = trainer.synthetic_batch(4, torch.device('cuda:0'), seed=5, H=512, W=512)
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    tr.train_step(*batch)
torch.cuda.synchronize()
print('HRNet ms per step %.2f' % ((time.perf_counter() - t) * 100))
