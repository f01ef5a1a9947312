"""DAM-Unet bf16 training step with / without the 32-input-channel-block rule for layers of at most 32 output channels (wgrad_ws_kernel<1, 1>),
alternating in one process: python3 tools/ab_wgrad_kq.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import cdnet_amd
from cdnet_amd import trainer, utils
cdnet_amd.set_precision('bf16')
dev = torch.device('cuda:0')
new_rule = trainer._choose_ci_tiles
def old_rule(C_src, Cout):
    best, best_cost = 2, None
    for ci_t in (2, 1, 4):
        CI, CO = ci_t * 32, (4 // ci_t) * 32
        cost = -(-C_src // CI) * CI * -(-Cout // CO) * CO
        if best_cost is None or cost < best_cost:
            best, best_cost = ci_t, cost
    return best
from cdnet_amd.models.dam.model_unet_rev1 import Unet
torch.manual_seed(0)
model = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
step, metric, workload = trainer.make_bench_step(model, 16, dev, 0, 1)
for _ in range(5): step()
for rep in range(3):
    for name, rule in (('old', old_rule), ('new', new_rule)):
        trainer._choose_ci_tiles = rule
        for _ in range(3): step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        print('%s: %.3f ms per step' % (name, (time.perf_counter() - t) * 50), flush=True)
