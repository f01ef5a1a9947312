"""`train(train_loader, model, optimizer, criterion, epoch, opt, logger, ...) -> ndarray[11]` with the reference's
signature (train_util_dam.py:45-339).

`optimizer` is the object `cdnet_amd.utils.get_optimizer(opt, model)` returns: the `Trainer` that owns the flat parameter /
gradient / Adam buffers and runs forward, the fused 5-term loss, backward and the Adam step on the HIP kernels.
`criterion` is accepted and ignored (the NLL / dice / MSE terms live in `cdnet_dam_loss`).  A sample is the reference's
tuple (input f32 [B,3,H,W], weight_map u8 [B,1,H,W], target0 [B,1,H,W] with values {0,127/128,255} or [B,3,H,W] one-hot
colours, target_point0 f16 [B,H,W], target_direction0 [B,H,W]) as `get_transforms` emits it (:71-142).
Returns `results.avg`: [loss, loss_direction_CE, loss_direction_dice, loss_mse, loss_CE, loss_var (= -1, alpha = 0),
pixel_accu, pixel_iou, pixel_recall, pixel_precision, pixel_F1] (:294-299, :339)."""
import numpy as np
import torch

from . import utils


def _label3(target0, boundary=2):
    """{0,1,2} class map from the loader's label tensor (train_util_dam.py:73-116)"""
    t = target0
    if t.shape[1] == 3:                                   # colour-coded three-channel label
        mx = t.max()
        lab = torch.zeros((t.shape[0], t.shape[-2], t.shape[-1]), dtype=torch.uint8, device=t.device)
        lab[t[:, 1] == mx] = 1
        lab[t[:, 2] == mx] = boundary
        return lab
    t = t[:, 0] if t.dim() == 4 else t
    mx = int(t.max())
    if mx == 255:
        t = t // int(255 / 2)                             # :107-108
    elif mx > 2:
        # e.g. a {0, 127} batch: the reference divides only when the batch maximum is 255 (:107) and its NLLLoss then raises on
        # class 127 ("Target 127 is out of bounds"); fail as loudly here instead of handing the value to the loss kernel
        raise ValueError('label values %s are neither {0,1,2} nor the {0,127/128,255} encoding (batch maximum %d)'
                         % (torch.unique(t).tolist()[:8], mx))
    return t.to(torch.uint8)


def train(train_loader, model, optimizer, criterion, epoch, opt, logger, get_process_worktime=1, get_process_detail=1,
          accuracy_tensor=0):
    trainer = optimizer
    assert opt.model['direction'] == 1 and opt.model['mseloss'] == 1 and opt.train['alpha'] == 0, \
        'the fused loss implements the default configuration (direction + point branches, no variance term)'
    results = utils.AverageMeter(11)
    dev = trainer.dev
    for i, sample in enumerate(train_loader):
        input, weight_map, target0, target_point0, target_direction0 = sample
        label = _label3(target0.to(dev), 2 if opt.model['multi_class'] else 1)
        w = weight_map.to(dev)
        w = (w[:, 0] if w.dim() == 4 else w).to(torch.uint8).contiguous()        # /20 on the device (:102)
        losses = trainer.train_step(input.to(dev).float(), label.contiguous(), target_direction0.to(dev).to(torch.uint8).contiguous(),
                                    target_point0.to(dev).to(torch.float16).contiguous(), w)
        r = losses.detach().cpu().numpy().astype(np.float64)
        r[5] = -1.0                                        # loss_var = torch.ones(1) * -1 when alpha == 0 (:249-251)
        results.update(r, input.size(0))
        if i % opt.train['log_interval'] == 0 and logger is not None:
            logger.info('\tIteration: [{:d}/{:d}]\tLoss {r[0]:.4f}\tloss_direction_CE {r[1]:.4f}\tloss_direction_dice {r[2]:.4f}'
                        '\tloss_mse {r[3]:.4f}\tLoss_CE {r[4]:.4f}\tPixel_Accu {r[6]:.4f}\tpixel_IoU {r[7]:.4f}'
                        .format(i, len(train_loader), r=results.avg))
    if getattr(trainer, 'world', 1) > 1:
        results.avg = trainer.reduce_scalars(results.avg)         # global-batch means, as DataParallel's gathered loss gives
    if logger is not None:
        logger.info('\t=> Train Avg: Loss {r[0]:.4f}\tloss_CE {r[4]:.4f}\tPixel_Accu {r[6]:.4f}\tIoU {r[7]:.4f}'.format(r=results.avg))
    return results.avg
