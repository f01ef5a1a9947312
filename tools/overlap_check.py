"""One-GPU sanity check of the overlapped gradient all-reduce: a 1-rank RCCL group, CDNET_FORCE_ALLREDUCE=1; the
parameters after 3 steps must be bit-identical with the overlap on and off (sum over one rank = identity).
usage: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 tools/overlap_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ['CDNET_FORCE_ALLREDUCE'] = '1'
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
dist.init_process_group('nccl', device_id=dev)
from cdnet_amd import trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet


def run(overlap):
    os.environ['CDNET_ALLREDUCE_OVERLAP'] = '1' if overlap else '0'
    torch.manual_seed(5)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
    tr = trainer.Trainer(m, world_size=1, bucket_mb=4)
    batch = trainer.synthetic_batch(4, dev, seed=3, H=128, W=128)
    launched = 0
    for _ in range(3):
        out = tr.train_step(*batch)
    torch.cuda.synchronize()
    return tr.flat.P.clone(), float(out[0])


p1, l1 = run(True)
p0, l0 = run(False)
print('loss', l1, l0, 'max |dP|', float((p1 - p0).abs().max()))
assert torch.equal(p1, p0), 'overlapped all-reduce changed the result'
print('overlap ok')
dist.barrier()
dist.destroy_process_group()
