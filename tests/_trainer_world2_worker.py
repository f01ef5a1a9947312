"""Worker of tests/test_gpu_trainer_world2.py: one of TWO ranks that share the one GPU of the box (gloo process group over the CUDA
flat buffers - RCCL cannot put two ranks on one device).  Everything of the distributed training path except xGMI runs: replicas
built from different seeds, `Trainer.sync_from_rank0`, per-rank batches, tape backward with the side stream, buckets released during
backward, the optimiser following the collectives bucket by bucket, `reduce_scalars`.  Rank r writes its flat parameter buffer,
BatchNorm buffers and the reduced scalars to <out>/rank<r>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
out_dir, precision, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('gloo', rank=rank, world_size=world)
import cdnet_amd
from cdnet_amd import trainer
from cdnet_amd.models.dam.model_unet_rev1 import Unet

cdnet_amd.set_precision(precision)
torch.manual_seed(1000 + rank)                        # replicas start DIFFERENT: the broadcast must make them identical
m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
tr = trainer.Trainer(m, world_size=world, bucket_mb=4)
torch.cuda.synchronize()
p_start = tr.flat.P.clone()
batch = trainer.synthetic_batch(2, dev, seed=50 + rank, H=64, W=64)      # every rank its own tiles
early, rows = 0, []
for _ in range(steps):
    mask, point, direction = tr.forward(batch[0])
    g = tr.loss_and_grads(mask, point, direction, *batch[1:])
    tr.backward(*g)
    if tr._ar is not None:
        early += tr._ar.early
    tr.allreduce_and_step()
    rows.append(tr.losses.cpu().numpy().copy())
torch.cuda.synchronize()
red = tr.reduce_scalars(rows[-1])
torch.save({'P': tr.flat.P.cpu(), 'P_start': p_start.cpu(), 'M': tr.flat.M.cpu(), 'V': tr.flat.V.cpu(), 'early': early,
            'losses': torch.tensor(rows[-1]), 'reduced': torch.as_tensor(red),
            'buffers': {k: v.detach().cpu() for k, v in m.named_buffers()}}, os.path.join(out_dir, 'rank%d.pt' % rank))
dist.barrier()
dist.destroy_process_group()
print('world2 worker %d ok: %d buckets released during backward' % (rank, early))
