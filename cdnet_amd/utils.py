"""Model factory, sliding-window forward and training helpers with the reference's names (utils.py).

  chooseModel(opt)                                     utils.py:816-886
  split_forward_dam(model, input, size, overlap, opt)  utils.py:658-726
  get_optimizer(args, model)                           utils.py:907-962   (Adam path; returns the fused device optimiser)
  AverageMeter                                         utils.py:755-774
  adjust_learning_rate                                 utils.py:965-977
"""

import numpy as np
import torch

from . import _lib


def chooseModel(opt):
    name = opt.model['modelName']
    if name == 'UNet':
        from .models.unet import UNet
        return UNet(num_classes=opt.model['out_c'], in_channels=opt.model['in_c'])
    if name == 'UNet2RevA1_vgg16':
        from .models.dam.model_unet_rev1 import Unet
        # the reference hard-codes pretrained=True (ImageNet download); weights are supplied via load_state_dict here
        return Unet(backbone_name='vgg16_bn', pretrained=False, encoder_freeze=False, classes=opt.model['out_c'])
    if name in ('model_unet_MandD', 'model_unet_MandD4', 'model_unet_MandD16', 'model_unet_MandDandP'):      # utils.py:857-874
        import importlib
        mod = importlib.import_module('.models.dam.' + name, __package__)
        return mod.Unet(backbone_name='vgg16_bn', pretrained=False, encoder_freeze=False, classes=opt.model['out_c'])
    if name == 'HRNet18_rev1':                       # utils.py:880-882
        from .models.dam.seg_hrnet_rev1 import HighResolutionNet
        return HighResolutionNet(opt)
    raise NotImplementedError('model {} is outside the CDNet hot path (SURVEY.md section 8: UNet, UNet2RevA1_vgg16, model_unet_MandD*, HRNet18_rev1)'.format(name))


def window_grid(h0, w0, size, overlap):
    """tile geometry of split_forward_dam (utils.py:664-683): padded extent, tile size, counts"""
    stride = size - overlap
    h = h0 + (stride - (h0 - size) % stride if h0 - size > 0 else 0)
    w = w0 + (stride - (w0 - size) % stride if w0 - size > 0 else 0)
    ny, nx = len(range(0, h - overlap, stride)), len(range(0, w - overlap, stride))
    return stride, min(size, h), min(size, w), ny, nx


def split_forward_views(model, image, size, overlap, xforms=(0,), direction_classes=9, max_batch=128, out=None):
    """Sliding-window forward of one image [3,H,W] (cuda float32) for several TTA views at once.
    Returns a list (one entry per view) of stitched logits (mask [3,hv,wv], point [1,hv,wv], direction [9,hv,wv]).
    `out` = (mask f32 [V,3,H*W], point f32 [V,1,H*W], direction f32 [V,K,H*W]): the views are stitched straight into these buffers (a rotated
    view as [K][W][H]: the same element count) - pipeline.infer_image's one get_probmaps launch over all views reads them in place."""
    assert image.dim() == 3 and image.is_cuda and image.dtype == torch.float32
    Cc, H0, W0 = image.shape
    image = image.contiguous()
    from . import runtime
    f32 = runtime.PRECISION == 'fp32'
    geo = []
    for xf in xforms:
        hv, wv = (W0, H0) if xf & 4 else (H0, W0)
        stride, th, tw, ny, nx = window_grid(hv, wv, size, overlap)
        geo.append((hv, wv, stride, th, tw, ny, nx))

    def pack(i, dst):
        hv, wv, stride, th, tw, ny, nx = geo[i]
        _lib.call('cdnet_window_pack_f32' if f32 else 'cdnet_window_pack', _lib.ptr(image), Cc, H0, W0, int(xforms[i]), th, tw, stride, ny, nx,
                  _lib.ptr(dst), _lib.stream_ptr())

    def stitch(i, logits, off):
        hv, wv, stride, th, tw, ny, nx = geo[i]
        n = ny * nx
        st = []
        for j_, t_ in enumerate(logits):
            K = t_.shape[1]
            o = torch.empty((K, hv, wv), dtype=torch.float32, device=image.device) if out is None else out[j_][i].view(K, hv, wv)
            src = t_[off:off + n]                          # (a batch slice of a contiguous tensor is contiguous)
            _lib.call('cdnet_window_stitch', _lib.ptr(src), K, th, tw, stride, overlap, ny, nx, hv, wv, _lib.ptr(o), _lib.stream_ptr())
            st.append(o)
        return tuple(st)

    # windows of equal shape go through the network together, in batches of WHOLE views whenever a view's windows fit one batch: the
    # window tensors are then packed straight into the batch and stitched straight out of the network's outputs (the first version
    # concatenated all windows, cut 64-window batches across views and concatenated the outputs again: 1.1 GB of copies per 1000 x 1000
    # image with 8 views, and a last batch of 8 windows)
    outs = [None] * len(xforms)
    by_shape = {}
    for i, g_ in enumerate(geo):
        by_shape.setdefault((g_[3], g_[4], g_[5] * g_[6]), []).append(i)
    # (128: two batches of four views for a 1000 x 1000 image.  All eight views in one batch is another 3 % in steady state, but its multi-GB
    #  tensors make the caching allocator's behaviour - and the time - depend on what the process ran before: 24 to 65 ms per image)
    for (th, tw, n), idxs in by_shape.items():
        max_batch = max(1, min(max_batch, (1 << 24) // (th * tw)))      # at most 256 windows of 256 x 256 (64 of 512 x 512) per network batch
        if n <= max_batch:
            per = max(1, max_batch // n)
            per = -(-len(idxs) // -(-len(idxs) // per))    # equal-sized batches (8 views x 25 windows: 4 x 50, not 2 x 64 + ...)
            for g0 in range(0, len(idxs), per):
                grp = idxs[g0:g0 + per]
                tiles = torch.empty((len(grp) * n, th, tw, 16), dtype=runtime.act_dtype(), device=image.device)
                for k, i in enumerate(grp):
                    pack(i, tiles[k * n:(k + 1) * n])
                logits = model.forward_packed(tiles)
                for k, i in enumerate(grp):
                    outs[i] = stitch(i, logits, k * n)
            continue
        # a view with more windows than one batch holds: batches cut across the view, outputs gathered before the stitch
        for i in idxs:
            tiles = torch.empty((n, th, tw, 16), dtype=runtime.act_dtype(), device=image.device)
            pack(i, tiles)
            res = [model.forward_packed(tiles[s_:s_ + max_batch]) for s_ in range(0, n, max_batch)]
            logits = [torch.cat([r[k] for r in res], 0) for k in range(len(res[0]))]
            outs[i] = stitch(i, logits, 0)
    return outs


def split_forward_dam(model, input, size, overlap, opt=None):
    """split the input image for forward process (reference signature).  input [1,C,H,W]; returns
    (output [1,out_c,H,W], output_point [1,1,H,W], output_direction [1,classes,H,W]) on the GPU."""
    assert input.shape[0] == 1, 'the reference calls this with one image at a time'
    with torch.no_grad():
        (m, p, d), = split_forward_views(model, input[0].cuda().float(), size, overlap, (0,))
    return m[None], p[None], d[None]


class AverageMeter(object):
    """ Computes and stores the average and current value (utils.py:755-774) """

    def __init__(self, shape=1):
        self.shape = shape
        self.reset()

    def reset(self):
        self.val = np.zeros(self.shape)
        self.avg = np.zeros(self.shape)
        self.sum = np.zeros(self.shape)
        self.count = 0

    def update(self, val, n=1):
        val = np.array(val)
        assert val.shape == self.val.shape
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def get_optimizer(args, model, world_size=1):
    """utils.py:907-962 for the default 'adam' (lr, betas=(0.9, 0.99), weight_decay): returns (Trainer, None) - the
    fused device optimiser lives inside cdnet_amd.trainer.Trainer."""
    if args.train['optimizer'].lower() != 'adam':
        raise NotImplementedError("only the reference's default optimizer 'adam' is on the hot path")
    from .trainer import Trainer, AblationTrainer
    # the ablation heads (model_unet_MandD / model_unet_MandDandP) train with plain classifiers instead of the gated head
    cls = AblationTrainer if getattr(model, 'VARIANT', 'rev1') != 'rev1' else Trainer
    return cls(model, lr=args.train['lr'], weight_decay=args.train['weight_decay'], world_size=world_size), None


def adjust_learning_rate(args, trainer, epoch):
    """utils.py:965-977: scheduler 'None' keeps the learning rate constant"""
    return trainer.lr


class EarlyStopping:
    """Early stops the training if the monitored value does not improve after `patience` epochs - and, as in the reference, never
    before epoch 100 (utils.py:992-1034)."""

    def __init__(self, patience=7, verbose=False, delta=0):
        self.patience, self.verbose, self.delta = patience, verbose, delta
        self.counter, self.best_score, self.early_stop = 0, None, False
        self.val_loss_min = np.inf

    def __call__(self, val_loss, epoch):
        score = -val_loss
        if self.best_score is None:
            self.best_score = score
            self.val_loss_min = val_loss
        elif score < self.best_score + self.delta:
            self.counter += 1
            print('===================== EarlyStopping counter: {} out of {} ====================='.format(self.counter, self.patience))
            if self.counter >= self.patience and epoch >= 100:
                self.early_stop = True
        else:
            self.best_score = score
            self.val_loss_min = val_loss
            self.counter = 0


def measure_label(a):
    """skimage.measure.label semantics (:248-249): 8-connected regions of EQUAL value, background 0, ids 1..K in raster order of
    each region's first pixel.  Instance maps from this package already are such regions (one lookup table); a binary image
    (validate's 0/255 target, train_util_dam.py:600-602) or a value that occurs in several separate regions is split here."""
    from scipy import ndimage as ndi
    a = np.ascontiguousarray(a).astype(np.int64)
    comp, _ = ndi.label(a != 0, structure=np.ones((3, 3), dtype=int))
    key = comp * (int(a.max()) + 1 if a.size else 1) + a              # (foreground component, value)
    key[a == 0] = 0
    vals, inv = np.unique(key.ravel(), return_inverse=True)
    lab = inv.reshape(a.shape).astype(np.int64)                         # 0 = background (key 0 sorts first), 1..K otherwise
    if vals[0] != 0:
        lab += 1
    # two regions of one value inside one foreground component touch only through other values: split them
    nxt = int(lab.max())
    for k, sl in enumerate(ndi.find_objects(lab), start=1):
        if sl is None:
            continue
        sub, n = ndi.label(lab[sl] == k, structure=np.ones((3, 3), dtype=int))
        for j in range(2, n + 1):
            nxt += 1
            lab[sl][sub == j] = nxt
    ids, first = np.unique(lab.ravel(), return_index=True)
    keep = ids != 0
    ids, first = ids[keep], first[keep]
    lut = np.zeros(nxt + 1, np.int32)
    lut[ids[np.argsort(first, kind='stable')]] = np.arange(1, len(ids) + 1, dtype=np.int32)
    return lut[lab].astype(np.int32)


def nuclei_accuracy_object_level(pred, gt):
    """(recall, precision, F1, dice, iou, haus, AJI) of utils.nuclei_accuracy_object_level (utils.py:245-330): ground-truth objects in
    id order, each greedily paired with the not-yet-used predicted object of largest IoU (first maximum in id order), used objects
    removed (:312).  The pixel pass (areas + sparse pairwise intersections) runs on the device (`stats_utils.pair_table`); the
    per-pair arithmetic is the reference's float arithmetic; the Hausdorff distance of a matched pair is scipy's
    directed_hausdorff on both sides, exactly the call the reference makes (:6, :303).
    pred / gt: integer label or binary images; both are re-labelled like the reference does with skimage.measure.label (8-connected
    regions of equal value, raster-order ids)."""
    from scipy.spatial.distance import directed_hausdorff
    from . import stats_utils

    p, g = measure_label(pred), measure_label(gt)
    ag, ap, pairs = stats_utils.pair_table(g, p)
    Ng, Ns = int((ag[1:] > 0).sum()), int((ap[1:] > 0).sum())
    by_gt = {}
    for ti, pi, inter in pairs:
        by_gt.setdefault(int(ti), []).append((int(pi), float(inter)))
    used = set()
    TP = FN = 0.0
    dice = iou = haus = C = U = count = 0.0
    for i in range(1, Ng + 1):
        cands = [(k, it) for k, it in sorted(by_gt.get(i, [])) if k not in used]
        if not cands:
            FN += 1
            U += float(ag[i])
            continue
        max_iou, best, overlap_area = 0.0, None, 0.0
        for k, it in cands:
            tmp_iou = it / (float(ap[k]) + float(ag[i]) - it)
            if tmp_iou > max_iou:
                max_iou, best, overlap_area = tmp_iou, k, it
        TP += 1
        count += 1
        a_p, a_g = float(ap[best]), float(ag[i])
        dice += 2 * overlap_area / (a_p + a_g)
        iou += overlap_area / (a_p + a_g - overlap_area)
        seg_ind, gt_ind = np.argwhere(p == best), np.argwhere(g == i)
        haus += max(directed_hausdorff(seg_ind, gt_ind)[0], directed_hausdorff(gt_ind, seg_ind)[0])
        C += overlap_area
        U += a_p + a_g - overlap_area
        used.add(best)
    FP = Ns - TP
    recall = TP / (TP + FN + 1e-10)
    precision = TP / (TP + FP + 1e-10)
    F1 = 2 * TP / (2 * TP + FP + FN + 1e-10)
    if count == 0:
        count = 1
    dice, iou, haus = dice / count, iou / count, haus / count
    U += float(sum(int(ap[k]) for k in range(1, len(ap)) if ap[k] > 0 and k not in used))
    AJI = float(C) / U if U > 0 else float('nan')
    return recall, precision, F1, dice, iou, haus, AJI
