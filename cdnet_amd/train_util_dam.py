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


def _check_branches(opt, model):
    """the option combinations the reference's loops accept for the model at hand (train_util_dam.py:152-166): three outputs ->
    direction = 1 and mseloss = 1 (the defaults); the two-output ablation models (mask + direction) -> direction = 1 and mseloss = 0
    (with mseloss = 1 the reference would read the direction logits as the point map).  alpha = 0: no variance term."""
    two = getattr(model, 'VARIANT', 'rev1') == 'MandD'
    assert opt.model['direction'] == 1 and opt.train['alpha'] == 0, 'the fused loss implements direction = 1, alpha = 0'
    if two:
        assert opt.model['mseloss'] in (0, 1), opt.model['mseloss']       # 0 is the reference's setting; 1 is tolerated (no point term either way)
    else:
        assert opt.model['mseloss'] == 1, 'three-output models train with the point branch (mseloss = 1)'


def train(train_loader, model, optimizer, criterion, epoch, opt, logger, get_process_worktime=1, get_process_detail=1,
          accuracy_tensor=0):
    trainer = optimizer
    _check_branches(opt, trainer.model)
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


def validate(val_loader, model, criterion, opt, logger, get_process_worktime=1, get_process_detail=1, all_img_test=1, accuracy_tensor=0,
             do_object_metric=0):
    """train_util_dam.validate (train_util_dam.py:367-636) -> ndarray[16]: [loss, loss_direction_CE, loss_direction_dice, loss_mse,
    pixel_accu, pixel_iou, pixel_recall, pixel_precision, pixel_F1, obj_recall, obj_precision, obj_F1, obj_dice, obj_iou, obj_haus,
    obj_AJI].  Eval-mode forward of the whole tile (all_img_test == 1) or through `utils.split_forward_dam` with
    opt.train['input_size'] / opt.train['val_overlap'] (:474); validate's OWN loss mix - unweighted mask CE + multi-class dice +
    weighted direction CE + plain dice on the background-gated direction probabilities + MSE against point / 255 - from one pass of
    `cdnet_dam_val_sums` over the logits, combined here in float64; the pixel metrics of the mask arg-max (:585-590).  With
    do_object_metric = 0 (the reference's call, train.py:348) the object slots are 0 and obj_iou = pixel_iou (:617-619); with 1 they
    are utils.nuclei_accuracy_object_level of sample 0's post-processed inside class (:588-604)."""
    import ctypes as C
    from . import _lib
    _check_branches(opt, model)
    assert opt.model['dice'] == 1, 'validate implements the default configuration (dice = 1)'
    results = utils.AverageMeter(16)
    model.eval()
    dev = next(model.parameters()).device
    lib = _lib.load()
    ws = None
    for i, sample in enumerate(val_loader):
        input, weight_map, target0, target_point0, target_direction0 = sample
        label = _label3(target0.to(dev), 2 if opt.model['multi_class'] else 1).contiguous()
        w = weight_map.to(dev)
        w = (w[:, 0] if w.dim() == 4 else w).to(torch.uint8).contiguous()
        dirlab = target_direction0.to(dev).to(torch.uint8).contiguous()
        point_t = target_point0.to(dev).to(torch.float16).contiguous()
        x = input.to(dev).float()
        with torch.no_grad():
            if all_img_test == 1:
                out = model(x)
                if len(out) == 2:                    # two-output ablation head (train_util_dam.py:481-487, direction = 1, mseloss = 0): no point term
                    mask, direction = out
                    point = torch.zeros((mask.shape[0], 1) + tuple(mask.shape[2:]), dtype=torch.float32, device=dev)
                    point_t = torch.zeros_like(point_t)
                else:
                    mask, point, direction = out
            else:
                outs = [utils.split_forward_dam(model, x[b:b + 1], opt.train['input_size'], opt.train['val_overlap'], opt) for b in range(x.shape[0])]
                mask, point, direction = [torch.cat([o[k] for o in outs], 0).contiguous() for k in range(3)]
        B, _, H, W = mask.shape
        # channel of a direction class = its rank among the batch's unique values (:462-468); a class the batch lacks has none
        uniq = torch.unique(dirlab).cpu().tolist()
        if len(uniq) < opt.direction_classes:
            raise IndexError('validate: the batch holds %d of the %d direction classes (the reference indexes unique_number[k] for '
                             'every k, train_util_dam.py:467)' % (len(uniq), opt.direction_classes))
        ND = direction.shape[1]                     # 9, or 5 / 17 for the 4- / 16-direction ablation models
        rank = (C.c_int * ND)(*[(uniq.index(v) if v in uniq else -1) for v in range(ND)])
        need = lib.cdnet_dam_val_sums_classes_workspace_floats(B, H * W, ND)
        if ws is None or ws.numel() < need:
            ws = torch.empty((need,), dtype=torch.float32, device=dev)
        sums = torch.empty((B, 15 + 3 * ND), dtype=torch.float32, device=dev)
        _lib.call('cdnet_dam_val_sums_classes', _lib.ptr(mask.contiguous()), _lib.ptr(point.contiguous()), _lib.ptr(direction.contiguous()),
                  _lib.ptr(label), _lib.ptr(dirlab), _lib.ptr(point_t), _lib.ptr(w), C.cast(rank, C.c_void_p), ND, B, H, W, _lib.ptr(ws),
                  ws.numel(), _lib.ptr(sums), _lib.stream_ptr())
        S = sums.cpu().numpy().astype(np.float64)
        n = float(B * H * W)
        iq, pq, tq, vs = 10, 10 + ND, 10 + 2 * ND, 10 + 3 * ND
        ce, dce, mse = S[:, 9].sum() / n, S[:, vs].sum() / n, S[:, vs + 1].sum() / n
        dice = sum(1.0 - np.mean(2.0 * (S[:, c] + 1.0) / (S[:, 3 + c] + S[:, 6 + c] + 1.0)) for c in range(3))          # loss.py:135-176
        ddice = sum(1.0 - np.mean(2.0 * (S[:, iq + c] + 1.0) / (S[:, pq + c] + S[:, tq + c] + 1.0)) for c in range(ND))
        loss = ce + dice + dce + ddice + mse
        tp, fp, fn = S[:, vs + 2], S[:, vs + 3], S[:, vs + 4]
        tn = H * W - tp - fp - fn
        precision, recall = tp / (tp + fp + 1e-10), tp / (tp + fn + 1e-10)
        m = [np.mean((tp + tn) / (tp + fp + tn + fn + 1e-10)), np.mean(tp / (tp + fp + fn + 1e-10)), np.mean(recall), np.mean(precision),
             np.mean(2 * precision * recall / (precision + recall + 1e-10))]
        obj = [0, 0, 0, 0, m[1], 0, 0]                       # :606-609: without the object metrics obj_iou = pixel_iou
        if do_object_metric == 1:
            # :588-604: SAMPLE 0's inside class through fill holes / remove small / label / dilate (the device chain of the inference
            # post-processing), scored against the labelled inside class of its target
            from . import postproc
            pred0 = (mask[:1].argmax(1) == 1).to(torch.uint8)
            labeled = postproc.cc_chain(pred0, fg_value=1, min_area=opt.post['min_area'], radius=opt.post['radius'])['final'][0]
            gt0 = (label[0] == 1).to(torch.uint8) * 255
            obj = list(utils.nuclei_accuracy_object_level(labeled.cpu().numpy(), gt0.cpu().numpy()))
        results.update([loss, dce, ddice, mse, m[0], m[1], m[2], m[3], m[4]] + obj)
    if logger is not None:
        logger.info('\t=> Val Avg:   \tLoss {r[0]:.4f}\tloss_direction_CE {r[1]:.4f}\tloss_direction_dice {r[2]:.4f}\tloss_mse {r[3]:.4f}'
                    '\tPixel_Acc {r[4]:.4f}\tPixel_IoU {r[5]:.4f}\tpixel_Recall {r[6]:.4f}\tpixel_Precision {r[7]:.4f}\tpixel_F1 {r[8]:.4f}'
                    .format(r=results.avg))
    return results.avg
