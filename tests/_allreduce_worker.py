"""Worker of tests/test_gpu_allreduce.py: a 1-rank RCCL ('nccl') process group on one GPU with the exchange step forced on
(CDNET_FORCE_ALLREDUCE=1).  The all-reduce over one rank is the identity, so the parameters after 3 training steps must be
bit-identical (a) with the bucket releases overlapped with backward, (b) with the single post-backward pass and (c) without any
collective; the parameter broadcast and the scalar reduction run through RCCL as well."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)
from cdnet_amd import trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet


def run(force, overlap):
    os.environ['CDNET_FORCE_ALLREDUCE'] = '1' if force else '0'
    os.environ['CDNET_ALLREDUCE_OVERLAP'] = '1' if overlap else '0'
    torch.manual_seed(5)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
    tr = trainer.Trainer(m, world_size=1, bucket_mb=4)
    tr.world = 1
    batch = trainer.synthetic_batch(4, dev, seed=3, H=128, W=128)
    early = 0
    for _ in range(3):
        mask, point, direction = tr.forward(batch[0])
        g = tr.loss_and_grads(mask, point, direction, *batch[1:])
        tr.backward(*g)
        if tr._ar is not None:
            early += tr._ar.early
        tr.allreduce_and_step()
    torch.cuda.synchronize()
    return tr, tr.flat.P.clone(), float(tr.losses[0]), early


tr1, p1, l1, early = run(True, True)
_, p0, l0, _ = run(True, False)
_, pn, ln, _ = run(False, False)
assert early > 0, 'no bucket was released during backward: the overlap never happened'
assert torch.equal(p1, p0), 'overlapped all-reduce changed the result'
assert torch.equal(p1, pn), 'the exchange step changed a single-rank result'
# broadcast of parameters / moments / BatchNorm buffers and the scalar mean through RCCL (identity on one rank)
before = {k: v.clone() for k, v in tr1.model.state_dict().items()}
tr1.world = 2                                        # force the collective paths (the group still has one rank)
try:
    tr1.sync_from_rank0()
    r = tr1.reduce_scalars([1.0, 2.5, -3.0])
finally:
    tr1.world = 1
after = tr1.model.state_dict()
assert all(torch.equal(before[k], after[k]) for k in before)
assert [float(v) for v in r] == [1.0, 2.5, -3.0]
print('allreduce ok: loss %.6f, %d buckets released during backward' % (l1, early))
dist.barrier()
dist.destroy_process_group()
