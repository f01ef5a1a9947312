"""HRNet18_rev1 training step (4 x 512x512, bf16) with trainer._WGRAD_WGS_KQ alternating in one process: python3 tools/ab_hrnet_wgs.py 128 192 256"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
from cdnet_amd import trainer


class O:
    model = {'out_c': 3}


torch.manual_seed(0)
m = HighResolutionNet(O()).cuda().train()
tr = trainer.Trainer(m)
batch = trainer.synthetic_batch(4, torch.device('cuda:0'), seed=5, H=512, W=512)
caps = [int(a) for a in sys.argv[1:]] or [128, 256]
for _ in range(3):
    tr.train_step(*batch)
for rep in range(3):
    for c in caps:
        trainer._WGRAD_WGS_KQ = c
        for _ in range(2):
            tr.train_step(*batch)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            tr.train_step(*batch)
        torch.cuda.synchronize()
        print('cap %3d: %.2f ms per step' % (c, (time.perf_counter() - t) * 100), flush=True)
