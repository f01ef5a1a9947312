#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE ITSELF.

CONTAINER-ONLY: imports /root/reference (read-only, never copied) through tests/golden/_ref_shims.py.
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [names...]

Every fixture is data only (inputs + the reference's outputs); the generating recipe is this file.
Weights are filled by the closed-form, name-keyed function `det_fill` (also implemented, independently,
in oracle/detfill.py) so that no weight blobs need to be stored.

Fixtures:
  ddm.npz          generate_dd_map (data_prepare/getDirectionDiffMap.py:44-108), 9/5/17 classes
  unet_fwd.npz     models/unet.py UNet forward+backward, det_fill weights
  dam_fwd.npz      models/dam/model_unet_rev1.py Unet forward (train-mode BN and eval-mode BN) + grads
  losses.npz       loss.py dice / weighted-dice, log-softmax NLL, MSE on fixed logits
  losses_classes.npz  the direction terms on 5- and 17-class direction maps
  train_iter.npz   train_util_dam.train: two iterations on a 1-batch loader (losses + params after Adam)
  ablation.npz     models/dam/model_unet_MandD{,4,16,andP}.py eval forward (heads without attention gates)
  validate.npz     train_util_dam.validate: the 16-value result vector, whole-tile and sliding-window forward
  cdm.npz          my_transforms_direction.LabelEncoding (direction branch) on synthetic ellipse labels
                   [skimage-semantics restated: scipy stand-ins for dilation/erosion/label]
  cdm_inst.npz     the same transform on instance-level labels (:752-760) [watershed stand-in]
  split_fwd.npz    utils.split_forward_dam stitching with a position-coding toy model
  probmaps.npz     test_dam.get_probmaps epilogue (softmax / gated argmax), re-assembled from :982-1015
  postproc.npz     test_dam.py:445-450,479-491,529-563 re-assembled (TTA mean, DDM fuse, boost, argmax,
                   fill holes, remove small, label, dilation) [skimage-semantics restated]
  aji.npz          stats_utils.get_fast_aji / get_dice_1 / get_fast_pq on label fixtures
"""
import os
import sys
import math
import zlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from scipy import ndimage as ndi  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------------------
# deterministic inputs / parameter fill: cdnet_amd/synth.py (own code, pure numpy, no reference)
# ----------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cdnet_amd import synth  # noqa: E402


def det_fill(model):
    import torch.nn as nn
    bn = {n for n, m in model.named_modules() if isinstance(m, nn.BatchNorm2d)}
    with torch.no_grad():
        synth.det_fill_state_dict(model.state_dict(), bn)
    return model


def det_input(shape, seed, f16_exact=False, bf16_exact=False):
    return torch.from_numpy(synth.det_input(shape, seed, f16_exact, bf16_exact))


# ----------------------------------------------------------------------------------------------
def gen_ddm():
    from data_prepare.getDirectionDiffMap import generate_dd_map
    rs = np.random.RandomState(7)
    out = {}
    cases = []

    def blobs(h, w, classes, nblob, rs):
        lab = np.zeros((h, w), np.uint8)
        for _ in range(nblob):
            cy, cx = rs.randint(0, h), rs.randint(0, w)
            ry, rx = rs.randint(3, 12), rs.randint(3, 12)
            yy, xx = np.ogrid[:h, :w]
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
            ang = np.degrees(np.arctan2(cy - yy, cx - xx)) * np.ones((h, w))
            step = 360.0 / (classes - 1)
            cls = (np.floor((ang + 180 + step / 2) / step).astype(int) % (classes - 1)) + 1
            lab[m] = cls[m]
        return lab

    # dense random, sparse random, blobs, borders, single class + bg
    for classes in (9, 5, 17):
        cases.append(('rand_%d_a' % classes, rs.randint(0, classes, size=(64, 64)).astype(np.uint8), classes))
        x = rs.randint(0, classes, size=(37, 53)).astype(np.uint8)
        x[rs.rand(37, 53) < 0.7] = 0
        cases.append(('sparse_%d' % classes, x, classes))
        cases.append(('blobs_%d' % classes, blobs(96, 80, classes, 14, rs), classes))
    x = np.zeros((32, 32), np.uint8); x[8:20, 5:25] = 3
    cases.append(('single_9', x, 9))
    x = rs.randint(1, 9, size=(24, 40)).astype(np.uint8)   # no background at all
    cases.append(('nobg_9', x, 9))
    x = np.zeros((16, 16), np.uint8)                         # constant -> 0/0 -> NaN contract
    cases.append(('allbg_9', x, 9))
    x = np.full((16, 16), 4, np.uint8)                       # all one direction, full image
    cases.append(('allone_9', x, 9))
    cases.append(('big_9', blobs(256, 256, 9, 60, rs), 9))
    x = rs.randint(0, 9, size=(1, 17)).astype(np.uint8)
    cases.append(('row_9', x, 9))
    x = rs.randint(0, 9, size=(19, 1)).astype(np.uint8)
    cases.append(('col_9', x, 9))
    names = []
    for name, x, classes in cases:
        with np.errstate(all='ignore'):
            y = generate_dd_map(x.copy(), classes)
        assert y.dtype == np.float32, y.dtype
        out['in_' + name] = x
        out['out_' + name] = y
        out['cls_' + name] = np.int32(classes)
        names.append(name)
    out['names'] = np.array(names)
    save('ddm', **out)


# ----------------------------------------------------------------------------------------------
def _loss_sum(outs):
    if isinstance(outs, torch.Tensor):
        outs = (outs,)
    tot = 0
    for k, o in enumerate(outs):
        c = torch.cos(torch.arange(o.numel(), dtype=torch.float32) * 0.013 * (k + 1)).view_as(o)
        tot = tot + (o * c).mean()
    return tot


def _grad_summary(model, picks=64):
    d = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().reshape(-1)
        d['gn_' + n] = np.float64(g.double().norm().item())
        d['gs_' + n] = g[:picks].numpy().copy()
    return d


def gen_unet():
    from models.unet import UNet
    m = UNet(num_classes=3, in_channels=3)
    det_fill(m)
    x = det_input((2, 3, 64, 64), 1)
    m.train()
    y = m(x)
    loss = _loss_sum(y)
    loss.backward()
    out = {'x_cfg': np.array([2, 3, 64, 64, 1]), 'y_train': y.detach().numpy(), 'loss': np.float64(loss.item())}
    out.update(_grad_summary(m))
    rm = {('rm_' + k): v.numpy().copy() for k, v in m.state_dict().items() if 'running' in k}
    out.update(rm)
    m2 = det_fill(UNet(num_classes=3, in_channels=3)).eval()
    with torch.no_grad():
        out['y_eval'] = m2(x).numpy()
        x2 = det_input((1, 3, 50, 70), 2)     # ragged size: ceil-mode pools + F.pad path
        out['x_ragged_cfg'] = np.array([1, 3, 50, 70, 2])
        out['y_eval_ragged'] = m2(x2).numpy()
    save('unet_fwd', **out)


def _dam_model():
    from models.dam.model_unet_rev1 import Unet
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        m = Unet(backbone_name='vgg16_bn', pretrained=False, encoder_freeze=False, classes=3)
    return m


def gen_dam():
    m = det_fill(_dam_model())
    x = det_input((2, 3, 64, 64), 1)
    m.train()
    outs = m(x)
    loss = _loss_sum(outs)
    loss.backward()
    out = {'x_cfg': np.array([2, 3, 64, 64, 1]), 'loss': np.float64(loss.item())}
    for n, o in zip(('mask', 'point', 'direction'), outs):
        out['train_' + n] = o.detach().numpy()
    out.update(_grad_summary(m))
    out.update({('rm_' + k): v.numpy().copy() for k, v in m.state_dict().items() if 'running' in k})
    m2 = det_fill(_dam_model()).eval()
    with torch.no_grad():
        for n, o in zip(('mask', 'point', 'direction'), m2(x)):
            out['eval_' + n] = o.numpy()
        x2 = det_input((1, 3, 256, 256), 3, f16_exact=True)
        for n, o in zip(('mask', 'point', 'direction'), m2(x2)):
            out['eval256_' + n] = o.numpy().astype(np.float16)
        x3 = det_input((1, 3, 72, 104), 4)
        out['x_ragged_cfg'] = np.array([1, 3, 72, 104, 4])
        out['x256_cfg'] = np.array([1, 3, 256, 256, 3])
        for n, o in zip(('mask', 'point', 'direction'), m2(x3)):
            out['evalragged_' + n] = o.numpy()
    out['param_names'] = np.array([k for k in m.state_dict().keys()])
    out['param_count'] = np.int64(sum(p.numel() for p in m.parameters()))
    save('dam_fwd', **out)


# ----------------------------------------------------------------------------------------------
def _synthetic_targets(B, H, W, seed):
    lab, dirn, point, weight = synth.train_targets(B, H, W, seed)
    return lab.astype(np.int64), dirn.astype(np.int64), point, weight.astype(np.int64)


def gen_losses():
    from loss import MulticlassDiceLoss, WeightMulticlassDiceLoss
    B, H, W = 3, 40, 48
    lab, dirn, point, weight = _synthetic_targets(B, H, W, 11)
    rs = np.random.RandomState(5)
    lo_mask = torch.from_numpy((rs.randn(B, 3, H, W) * 2).astype(np.float32))
    lo_dir = torch.from_numpy((rs.randn(B, 9, H, W) * 2).astype(np.float32))
    lo_pt = torch.from_numpy(rs.randn(B, 1, H, W).astype(np.float32))
    for t in (lo_mask, lo_dir, lo_pt):
        t.requires_grad_(True)
    w = torch.from_numpy(weight).float().div(20).squeeze(1)
    tgt = torch.from_numpy(lab)
    crit = torch.nn.NLLLoss(reduction='none')
    ce = (crit(F.log_softmax(lo_mask, 1), tgt) * w).mean()
    onehot = F.one_hot(tgt, 3).permute(0, 3, 1, 2).float()
    dice = MulticlassDiceLoss()(F.softmax(lo_mask, 1), onehot)
    dce = (crit(F.log_softmax(lo_dir, 1), torch.from_numpy(dirn)) * w).mean()
    onehot9 = F.one_hot(torch.from_numpy(dirn), 9).permute(0, 3, 1, 2).float()
    wdice = WeightMulticlassDiceLoss()(F.softmax(lo_dir, 1), onehot9, w)
    mse = torch.nn.MSELoss()(lo_pt, torch.from_numpy(point).float().unsqueeze(1))
    total = ce + dice + dce + wdice + mse
    total.backward()
    save('losses', cfg=np.array([B, H, W, 11, 5]),      # targets: synth.train_targets(B,H,W,11); logits: RandomState(5)
         ce=np.float64(ce.item()), dice=np.float64(dice.item()), dce=np.float64(dce.item()),
         wdice=np.float64(wdice.item()), mse=np.float64(mse.item()), total=np.float64(total.item()),
         g_mask=lo_mask.grad.numpy(), g_dir=lo_dir.grad.numpy(), g_pt=lo_pt.grad.numpy())


def gen_losses_classes():
    """the direction terms (weighted NLL + loss.py WeightMulticlassDiceLoss) on 4+1- and 16+1-class direction maps
    (options.py:45; models/dam/model_unet_MandD4.py / model_unet_MandD16.py)"""
    from loss import WeightMulticlassDiceLoss
    B, H, W = 3, 40, 48
    lab, dirn, point, weight = _synthetic_targets(B, H, W, 11)
    w = torch.from_numpy(weight).float().div(20).squeeze(1)
    crit = torch.nn.NLLLoss(reduction='none')
    out = {'cfg': np.array([B, H, W, 11, 5])}       # targets: synth.remap_direction(train_targets(B,H,W,11)); logits: RandomState(5 + C)
    for C in (5, 17):
        d = torch.from_numpy(synth.remap_direction(dirn, C).astype(np.int64))
        lo_dir = torch.from_numpy((np.random.RandomState(5 + C).randn(B, C, H, W) * 2).astype(np.float32)).requires_grad_(True)
        dce = (crit(F.log_softmax(lo_dir, 1), d) * w).mean()
        onehot = F.one_hot(d, C).permute(0, 3, 1, 2).float()
        wdice = WeightMulticlassDiceLoss()(F.softmax(lo_dir, 1), onehot, w)
        (dce + wdice).backward()
        out['c%d_dce' % C], out['c%d_wdice' % C] = np.float64(dce.item()), np.float64(wdice.item())
        out['c%d_g_dir' % C] = lo_dir.grad.numpy()
    save('losses_classes', **out)


# ----------------------------------------------------------------------------------------------
class _Opt:
    """the option fields train_util_dam.train reads (defaults from options.py:50-60,88-95)"""
    def __init__(self):
        self.model = dict(direction=1, mseloss=1, multi_class=True, add_weightMap=True, boundary_loss=0,
                          dice=1, out_c=3, modelName='UNet2RevA1_vgg16')
        self.train = dict(alpha=0, log_interval=1000, lr=0.001, weight_decay=1e-4, optimizer='adam',
                          scheduler='None')
        self.direction_classes = 9


class _Logger:
    def info(self, *a, **k):
        pass


def gen_train_iter():
    import train_util_dam
    import utils as ref_utils
    B, H, W = 2, 64, 64
    lab, dirn, point, weight = _synthetic_targets(B, H, W, 21)
    x = det_input((B, 3, H, W), 9)
    target0 = torch.from_numpy(lab * 127 + (lab == 2)).long().unsqueeze(1)   # {0,127,255} as ToTensor emits
    sample = (x, torch.from_numpy(weight), target0, torch.from_numpy(point), torch.from_numpy(dirn))
    m = det_fill(_dam_model())
    opt = _Opt()
    optimizer, _ = ref_utils.get_optimizer(opt, m)
    crit = torch.nn.NLLLoss(reduction='none')
    res = []
    snaps = []
    pick = ['backbone.0.weight', 'backbone.1.weight', 'backbone.40.weight', 'upsample_blocks.0.up.weight',
            'upsample_blocks.4.conv2.weight', 'upsample_blocks.4.bn2.bias', 'mask_feature.conv1.weight',
            'mask_feature.conv_1x1.bias', 'point_conv.weight', 'directionAtt.Conv1x1.weight',
            'direction_conv.weight', 'maskAtt.Conv1x1.weight', 'mask_conv.bias', 'point_feature.bn2.weight']
    sd = dict(m.named_parameters())
    for it in range(2):
        r = train_util_dam.train([sample], m, optimizer, crit, it, opt, _Logger())
        res.append(np.array(r, dtype=np.float64))
        snaps.append({k: sd[k].detach().reshape(-1)[:96].numpy().copy() for k in pick})
    out = {'x_cfg': np.array([B, 3, H, W, 9]), 'tgt_cfg': np.array([B, H, W, 21]),
           'results': np.stack(res), 'pick': np.array(pick)}
    for it in range(2):
        for k in pick:
            out['p%d_%s' % (it, k)] = snaps[it][k]
    out['rm_backbone.1.running_mean'] = m.state_dict()['backbone.1.running_mean'].numpy().copy()
    out['rm_backbone.1.running_var'] = m.state_dict()['backbone.1.running_var'].numpy().copy()
    save('train_iter', **out)


def gen_ablation():
    """eval forward of the four ablation models of utils.py:857-874 (models/dam/model_unet_MandD*.py), closed-form weights; outputs
    stored as float16 slices (the networks share encoder / decoder with model_unet_rev1: the heads are what is pinned)"""
    import importlib, io, contextlib
    out = {}
    x = det_input((1, 3, 48, 64), 12)
    for name in ('model_unet_MandD', 'model_unet_MandD4', 'model_unet_MandD16', 'model_unet_MandDandP'):
        mod = importlib.import_module('models.dam.' + name)
        with contextlib.redirect_stdout(io.StringIO()):
            m = mod.Unet(backbone_name='vgg16_bn', pretrained=False, encoder_freeze=False, classes=3)
        m = det_fill(m).eval()
        with torch.no_grad():
            res = m(x)
        out['n_' + name] = np.int64(len(res))
        out['keys_' + name] = np.array(list(m.state_dict().keys()))
        for k, o in enumerate(res):
            out['%s_%d' % (name, k)] = o.numpy().astype(np.float16)
    out['x_cfg'] = np.array([1, 3, 48, 64, 12])
    save('ablation', **out)


def gen_validate():
    """train_util_dam.validate (train_util_dam.py:367-636) on a 1-batch loader, default options: eval-mode forward of the whole
    tile (all_img_test = 1) and through utils.split_forward_dam (all_img_test = 0, 64 / 16 windows), its own loss mix (unweighted
    mask CE + multi-class dice + weighted direction CE + plain dice on the background-gated direction probabilities + MSE against
    point / 255) and the pixel metrics of the mask arg-max -> the 16-value result vector"""
    import train_util_dam
    B, H, W = 2, 96, 96
    lab, dirn, point, weight = _synthetic_targets(B, H, W, 23)
    x = det_input((B, 3, H, W), 10)
    target0 = torch.from_numpy(lab * 127 + (lab == 2)).long().unsqueeze(1)
    sample = (x, torch.from_numpy(weight), target0, torch.from_numpy(point), torch.from_numpy(dirn))
    m = det_fill(_dam_model())
    opt = _Opt()
    opt.train.update(input_size=64, val_overlap=16)
    opt.post = dict(min_area=20, radius=2)
    crit = torch.nn.NLLLoss(reduction='none')
    whole = train_util_dam.validate([sample], m, crit, opt, _Logger(), all_img_test=1)
    one = (x[:1], torch.from_numpy(weight[:1]), target0[:1], torch.from_numpy(point[:1]), torch.from_numpy(dirn[:1]))
    split = train_util_dam.validate([one], m, crit, opt, _Logger(), all_img_test=0)
    save('validate', x_cfg=np.array([B, 3, H, W, 10]), tgt_cfg=np.array([B, H, W, 23]), win_cfg=np.array([64, 16]),
         whole=np.array(whole, dtype=np.float64), split=np.array(split, dtype=np.float64))


class _StubModel(torch.nn.Module):
    """returns prescribed outputs (validate()'s metric branches need object-like predictions, which no closed-form weight fill gives)"""

    def __init__(self, outs):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))
        self.outs = [torch.from_numpy(o) for o in outs]

    def forward(self, x):
        return tuple(o.to(x.device) for o in self.outs)


def gen_validate_obj():
    """train_util_dam.validate with do_object_metric = 1 (:588-604): sample 0's mask arg-max through fill holes / remove small /
    label / dilate, then utils.nuclei_accuracy_object_level against the labelled inside class of its target.  The model is a stub
    that returns synth.stub_outputs (the closed-form weights predict no objects at all)."""
    import io, contextlib
    import train_util_dam
    B, H, W = 2, 96, 96
    lab, dirn, point, weight = _synthetic_targets(B, H, W, 23)
    x = det_input((B, 3, H, W), 10)
    target0 = torch.from_numpy(lab * 127 + (lab == 2)).long().unsqueeze(1)
    sample = (x, torch.from_numpy(weight), target0, torch.from_numpy(point), torch.from_numpy(dirn))
    m = _StubModel(synth.stub_outputs(lab, dirn, point, 77))
    opt = _Opt()
    opt.train.update(input_size=64, val_overlap=16)
    opt.post = dict(min_area=20, radius=2)
    crit = torch.nn.NLLLoss(reduction='none')
    with contextlib.redirect_stdout(io.StringIO()):
        row = train_util_dam.validate([sample], m, crit, opt, _Logger(), all_img_test=1, do_object_metric=1)
    save('validate_obj', tgt_cfg=np.array([B, H, W, 23]), stub_seed=np.int64(77), post=np.array([20, 2]), row=np.array(row, dtype=np.float64))


def gen_hrnet():
    """HRNet18_rev1 (seg_hrnet_rev1.HighResolutionNet) eval forward; closed-form weights with every Conv2d scaled by 0.45
    (the un-scaled fill overflows through the 30 residual additions)"""
    from models.dam.seg_hrnet_rev1 import HighResolutionNet

    class _O:
        model = {'out_c': 3}
    m = det_fill(HighResolutionNet(_O())).eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.mul_(0.45)
        out = {'gain': np.float64(0.45)}
        for tag, shape, seed in (('a', (1, 3, 64, 64), 5), ('b', (2, 3, 32, 96), 6)):
            x = det_input(shape, seed, bf16_exact=True)
            for n, o in zip(('mask', 'point', 'direction'), m(x)):
                out['%s_%s' % (n, tag)] = o.numpy()
            out['x_cfg_' + tag] = np.array(list(shape) + [seed])
        # BASELINE config 5 size: one 512x512 tile, every 8th pixel of the outputs kept (float16)
        shape, seed = (1, 3, 512, 512), 7
        x = det_input(shape, seed, bf16_exact=True)
        for n, o in zip(('mask', 'point', 'direction'), m(x)):
            out['%s_c' % n] = o.numpy()[:, :, ::8, ::8].astype(np.float16)
        out['x_cfg_c'] = np.array(list(shape) + [seed])
    out['param_count'] = np.int64(sum(p.numel() for p in m.parameters()))
    out['n_keys'] = np.int64(len(m.state_dict()))
    save('hrnet_fwd', **out)


def gen_hrnet_train():
    """Two iterations of the reference's train() (train_util_dam.py:17-160) on HRNet18_rev1: closed-form weights with the
    convolutions scaled by 0.45, BatchNorm in batch-statistics mode, Adam from get_optimizer"""
    import train_util_dam
    import utils as ref_utils
    from models.dam.seg_hrnet_rev1 import HighResolutionNet

    class _O:
        model = {'out_c': 3}
    m = det_fill(HighResolutionNet(_O()))
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.mul_(0.45)
    B, H, W = 2, 64, 64
    lab, dirn, point, weight = _synthetic_targets(B, H, W, 41)
    x = det_input((B, 3, H, W), 15, bf16_exact=True)
    target0 = torch.from_numpy(lab * 127 + (lab == 2)).long().unsqueeze(1)
    sample = (x, torch.from_numpy(weight), target0, torch.from_numpy(point), torch.from_numpy(dirn))
    opt = _Opt()
    optimizer, _ = ref_utils.get_optimizer(opt, m)
    crit = torch.nn.NLLLoss(reduction='none')
    pick = ['conv1.weight', 'bn1.weight', 'conv2.weight', 'layer1.0.conv1.weight', 'layer1.0.downsample.0.weight', 'layer1.1.conv3.weight',
            'transition1.0.0.weight', 'transition1.1.0.0.weight', 'stage2.0.branches.1.1.conv2.weight', 'stage2.0.fuse_layers.0.1.0.weight',
            'stage2.0.fuse_layers.1.0.0.0.weight', 'stage3.1.fuse_layers.2.0.0.0.weight', 'stage3.2.fuse_layers.2.0.1.0.weight',
            'transition3.3.0.0.weight', 'stage4.0.branches.3.0.bn1.bias', 'stage4.1.fuse_layers.0.3.1.weight',
            'stage4.1.fuse_layers.3.0.2.0.weight', 'stage4.1.branches.0.1.conv1.weight', 'mask_feature.conv1.weight',
            'mask_feature.conv_1x1.weight', 'point_conv.weight', 'directionAtt.Conv1x1.weight', 'mask_conv.bias']
    sd = dict(m.named_parameters())
    p0 = {k: sd[k].detach().reshape(-1)[:96].numpy().copy() for k in pick}
    res, snaps = [], []
    for it in range(2):
        r = train_util_dam.train([sample], m, optimizer, crit, it, opt, _Logger())
        res.append(np.array(r, dtype=np.float64))
        snaps.append({k: sd[k].detach().reshape(-1)[:96].numpy().copy() for k in pick})
    out = {'x_cfg': np.array([B, 3, H, W, 15]), 'tgt_cfg': np.array([B, H, W, 41]), 'gain': np.float64(0.45),
           'results': np.stack(res), 'pick': np.array(pick), 'lr': np.float64(opt.train['lr'])}
    for k in pick:
        out['p_init_' + k] = p0[k]
        for it in range(2):
            out['p%d_%s' % (it, k)] = snaps[it][k]
    out['rm_bn1.running_mean'] = m.state_dict()['bn1.running_mean'].numpy().copy()
    out['rm_stage4.1.branches.3.1.bn2.running_var'] = m.state_dict()['stage4.1.branches.3.1.bn2.running_var'].numpy().copy()
    save('hrnet_train', **out)


def gen_unet_train_iter():
    """two iterations of the reference's plain-UNet train loop (train_util.train, default options) on one batch"""
    import train_util
    import utils as ref_utils
    from models.unet import UNet
    B, H, W = 2, 64, 64
    lab, _, _, weight = _synthetic_targets(B, H, W, 31)
    x = det_input((B, 3, H, W), 12)
    target0 = torch.from_numpy(lab * 127 + (lab == 2)).long().unsqueeze(1)   # {0,127,255} as ToTensor emits
    sample = (x, torch.from_numpy(weight), target0)
    m = det_fill(UNet(num_classes=3, in_channels=3))
    opt = _Opt()
    opt.model['modelName'] = 'UNet'
    opt.train['num_epochs'] = 1
    optimizer, _ = ref_utils.get_optimizer(opt, m)
    crit = torch.nn.NLLLoss(reduction='none')
    pick = ['down1.down_conv.0.weight', 'down1.down_conv.1.weight', 'down4.down_conv.3.weight', 'middle_conv.0.weight',
            'up1.up.weight', 'up1.up.bias', 'up4.up_conv.3.weight', 'up4.up_conv.4.bias', 'final_conv.weight', 'final_conv.bias']
    sd = dict(m.named_parameters())
    res, snaps = [], []
    for it in range(2):
        r = train_util.train([sample], m, optimizer, crit, it, opt, _Logger())
        res.append(np.array(r, dtype=np.float64))
        snaps.append({k: sd[k].detach().reshape(-1)[:96].numpy().copy() for k in pick})
    out = {'x_cfg': np.array([B, 3, H, W, 12]), 'tgt_cfg': np.array([B, H, W, 31]), 'results': np.stack(res), 'pick': np.array(pick)}
    for it in range(2):
        for k in pick:
            out['p%d_%s' % (it, k)] = snaps[it][k]
    save('unet_train_iter', **out)


# ----------------------------------------------------------------------------------------------
_ellipse_instances = synth.ellipse_instances


def gen_cdm():
    from my_transforms_direction import LabelEncoding
    from PIL import Image
    out = {}
    names = []
    for name, (H, W, n, seed) in {'a': (96, 96, 12, 3), 'b': (80, 128, 16, 4), 'c': (64, 64, 3, 5)}.items():
        rs = np.random.RandomState(seed)
        inst = _ellipse_instances(H, W, n, rs)
        # 3-class PNG input branch (my_transforms_direction.py:763-774): channel 0 > 127.5 = inside
        lab_rgb = np.zeros((H, W, 3), np.uint8)
        lab_rgb[..., 0] = (inst > 0) * 255
        img = Image.fromarray(np.zeros((H, W, 3), np.uint8))
        wmap = Image.fromarray(np.full((H, W), 20, np.uint8))
        enc = LabelEncoding(3, 2, 1)
        res = enc((img, wmap, Image.fromarray(lab_rgb)))
        lab3 = np.array(res[2])
        point = res[3]
        direction = res[4]
        assert point.dtype == np.float16 and direction.dtype == np.int64
        out['in_' + name] = lab_rgb[..., 0]
        out['label_' + name] = lab3
        out['point_' + name] = point
        out['direction_' + name] = direction.astype(np.uint8)
        names.append(name)
    out['names'] = np.array(names)
    save('cdm', **out)


def gen_cdm_inst():
    """LabelEncoding on INSTANCE-level labels (my_transforms_direction.py:752-760: boundary from the instance ids, instances
    through postproc_other.process(..., min_size=5) = the watershed branch).  skimage.segmentation.watershed is absent here: the
    reference code runs with oracle/postproc.py's restatement of it patched in [watershed stand-in: equal-priority ties unpinned]."""
    import postproc_other
    from my_transforms_direction import LabelEncoding
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import postproc as orc

    def ws(image, markers, mask=None):
        return orc.watershed(np.asarray(image).astype(np.uint8), markers, mask)
    postproc_other.watershed = ws
    out, names = {}, []
    for name, (H, W, n, seed, touching) in {'a': (96, 96, 12, 3, False), 'b': (80, 128, 18, 4, True), 'c': (64, 64, 2, 5, False)}.items():
        rs = np.random.RandomState(seed)
        inst = _ellipse_instances(H, W, n, rs)
        if touching:                                   # grow the instances until some touch: boundaries BETWEEN ids
            for _ in range(2):
                g = ndi.grey_dilation(inst, footprint=ndi.generate_binary_structure(2, 1))
                inst = np.where(inst == 0, g, inst)
        lab = inst.astype(np.uint8)
        img = Image.fromarray(np.zeros((H, W, 3), np.uint8))
        wmap = Image.fromarray(np.full((H, W), 20, np.uint8))
        res = LabelEncoding(3, 2, 1)((img, wmap, Image.fromarray(lab)))
        assert res[3].dtype == np.float16 and res[4].dtype == np.int64
        out['in_' + name] = lab
        out['label_' + name] = np.array(res[2])
        out['point_' + name] = res[3]
        out['direction_' + name] = res[4].astype(np.uint8)
        names.append(name)
    out['names'] = np.array(names)
    save('cdm_inst', **out)


# ----------------------------------------------------------------------------------------------
def gen_split():
    import utils as ref_utils

    class Toy(torch.nn.Module):
        """outputs depend on absolute content and on position-in-patch, so any stitching error shows"""
        def forward(self, x):
            b, c, h, w = x.shape
            yy = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1).expand(b, 1, h, w)
            xx = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w)
            mask = torch.cat([x[:, :1] * 2 + yy * 1e-3, x[:, 1:2] - xx * 1e-3, x[:, 2:3] + 1], 1)
            point = x.sum(1, keepdim=True) + yy * 1e-2 + xx * 1e-4
            direction = torch.cat([x[:, :1] * (k + 1) + (yy + xx) * 1e-3 for k in range(9)], 1)
            return mask, point, direction

    opt = _Opt()
    out = {}
    cases = {'a': (130, 200, 64, 16),      # ragged both ways
             'b': (64, 64, 64, 16),        # single window, no padding branch (h0 - size == 0)
             'c': (128, 64, 64, 16),       # pad rows only
             'd': (112, 160, 64, 16),      # exact fit still receives one extra stride of zero padding
             'e': (280, 300, 256, 40)}     # the reference's own size/overlap (options.py test patch 256/40)
    for name, (h, w, size, ov) in cases.items():
        seed = 31 + ord(name)
        x = det_input((1, 3, h, w), seed, bf16_exact=True)  # regenerated by synth.det_input(bf16_exact=True) in the tests
        o, p, d = ref_utils.split_forward_dam(Toy(), x, size, ov, opt)
        out['cfg_' + name] = np.array([size, ov, h, w, seed])
        if name == 'e':                                       # big case: keep a strided sample of the outputs
            o, p, d = o[:, :, ::7, ::5], p[:, :, ::7, ::5], d[:, :, ::7, ::5]
        out['mask_' + name] = o.numpy()
        out['point_' + name] = p.numpy()
        out['dir_' + name] = d.numpy()
    out['names'] = np.array(list(cases))
    save('split_fwd', **out)


# ----------------------------------------------------------------------------------------------
def gen_probmaps():
    """test_dam.get_probmaps epilogue, re-assembled from test_dam.py:982-1015 (the function itself needs
    module-level state of test_dam.py whose import does os.chdir to a Windows path)."""
    g = torch.Generator().manual_seed(77)
    out = {}
    for name, (H, W) in {'a': (64, 64), 'b': (50, 70)}.items():
        output = torch.randn((3, H, W), generator=g) * 3
        output_point = torch.randn((1, H, W), generator=g)
        output_direction = torch.randn((9, H, W), generator=g) * 3
        point_maps = output_point.detach().cpu().numpy()                     # :982
        prob_maps = F.softmax(output, dim=0).cpu().numpy()                   # :984
        prob_maps_direction = F.softmax(output_direction[:, :, :], dim=0).cpu().numpy()   # :1011
        prob_maps_direction[0, :, :] = prob_maps_direction[0, :, :] * prob_maps[0, :, :]  # :1012
        pred_direction = np.argmax(prob_maps_direction, axis=0)              # :1013
        pred_direction = pred_direction.reshape(1, H, W)                     # :1015
        srt = np.sort(prob_maps_direction, axis=0)
        out['mask_logits_' + name] = output.numpy()
        out['point_logits_' + name] = output_point.numpy()
        out['dir_logits_' + name] = output_direction.numpy()
        out['prob_' + name] = prob_maps
        out['point_' + name] = point_maps
        out['dcm_' + name] = pred_direction.astype(np.uint8)
        out['margin_' + name] = (srt[-1] - srt[-2]).astype(np.float32)
    out['names'] = np.array(['a', 'b'])
    save('probmaps', **out)


# ----------------------------------------------------------------------------------------------
def _postproc_case(H, W, n, seed):
    probs, points, dcms = synth.postproc_case(H, W, n, seed)
    return list(probs), list(points), list(dcms)


def gen_postproc():
    from data_prepare.getDirectionDiffMap import generate_dd_map
    import skimage.morphology as morph       # shims (scipy stand-ins)
    from skimage import measure
    out = {}
    names = []
    for name, (H, W, n, seed) in {'a': (128, 128, 40, 1), 'b': (96, 160, 50, 2), 'c': (256, 256, 150, 3),
                                  'd': (61, 47, 8, 4)}.items():
        probs, points, dcms = _postproc_case(H, W, n, seed)
        # test_dam.py:445-450  (views are already un-flipped/un-rotated here)
        prob_maps = (probs[0] + probs[1] + probs[2] + probs[3] + probs[4] + probs[5] + probs[6] + probs[7]) / 8
        point_maps = (points[0] + points[1] + points[2] + points[3] + points[4] + points[5] + points[6] + points[7]) / 8
        # :459-491
        prob_dcm_map = np.zeros((H, W, 8), np.uint8)
        for v in range(8):
            prob_dcm_map[:, :, v] = dcms[v][0]
        prob_ddm_map = np.zeros((H, W, 8), np.float64)
        for v in range(8):
            prob_ddm_map[:, :, v] = generate_dd_map(prob_dcm_map[:, :, v], 9)
        pred_direction = np.mean(prob_ddm_map, axis=2)
        prob_direction_maps = pred_direction.reshape(1, H, W)
        # :529-539
        prob_in = prob_maps.copy()
        pred_inside3 = (point_maps[0] / np.max(point_maps) > 0.2) * 1
        pred_inside3 = morph.dilation(pred_inside3, selem=morph.selem.disk(1))
        prob_direction_map_pred_inside = prob_direction_maps[0] * pred_inside3
        enhanced_boundary = prob_direction_maps[0] - prob_direction_map_pred_inside
        enhanced_boundary = 2 * enhanced_boundary
        assert (np.min(enhanced_boundary) >= 0)
        prob_maps[2, :, :] = (prob_maps[2, :, :] + 0.5 * enhanced_boundary) * (1 + enhanced_boundary)
        pred = np.argmax(prob_maps, axis=0)
        pred_inside = pred == 1
        # :546-563
        pred_inside2 = ndi.binary_fill_holes(pred_inside)
        pred2 = morph.remove_small_objects(pred_inside2, 20)
        pred2 = pred2.astype(np.uint8)
        pred_labeled = measure.label(pred2)
        pred_labeled_d = morph.dilation(pred_labeled, selem=morph.selem.disk(2))
        # inputs are regenerated by cdnet_amd.synth.postproc_case(H, W, n, seed); only their CRC is stored
        out['cfg_' + name] = np.array([H, W, n, seed])
        out['crc_' + name] = synth.crc(np.stack(probs), np.stack(points), np.stack(dcms))
        if name == 'd':
            out['probs_' + name] = np.stack(probs).astype(np.float32)
            out['points_' + name] = np.stack(points).astype(np.float32)
            out['dcms_' + name] = np.stack(dcms)
        out['prob_mean_crc_' + name] = synth.crc(prob_in)
        out['point_mean_crc_' + name] = synth.crc(point_maps)
        out['ddm_mean16_' + name] = np.round(pred_direction * 16).astype(np.uint8)   # exact: values are k/16
        assert np.array_equal(out['ddm_mean16_' + name] / 16.0, pred_direction)
        out['inside3_' + name] = pred_inside3.astype(np.uint8)
        out['pred_' + name] = pred.astype(np.uint8)
        out['fill_' + name] = pred_inside2.astype(np.uint8)
        out['small_' + name] = pred2
        out['label_' + name] = pred_labeled.astype(np.int32)
        out['final_' + name] = pred_labeled_d.astype(np.int32)
        names.append(name)
        print('  postproc', name, 'instances', pred_labeled.max(), 'boundary boosted px',
              int((enhanced_boundary > 0).sum()))
    out['names'] = np.array(names)
    save('postproc', **out)


def gen_aji():
    import stats_utils
    z = np.load(os.path.join(HERE, 'postproc.npz'))
    out = {}
    for name in z['names']:
        pred = z['final_' + name]
        rs = np.random.RandomState(5)
        true = ndi.label(ndi.binary_dilation(z['small_' + name] > 0, iterations=1))[0].astype(np.int32)
        true = np.roll(true, 1, axis=0)
        p = stats_utils.remap_label(pred.copy())
        t = stats_utils.remap_label(true.copy())
        aji = stats_utils.get_fast_aji(t, p)
        aji_v = aji[0] if isinstance(aji, (tuple, list)) else aji
        dice = stats_utils.get_dice_1(t, p)
        pq = stats_utils.get_fast_pq(t, p)[0]
        out['true_' + name] = true
        out['aji_' + name] = np.float64(aji_v)
        out['dice_' + name] = np.float64(dice)
        out['pq_' + name] = np.array(pq, dtype=np.float64)
        import utils as ref_utils
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            obj = ref_utils.nuclei_accuracy_object_level(pred.copy(), true.copy())        # utils.py:245-330
        out['obj_' + name] = np.array(obj, dtype=np.float64)
    out['names'] = z['names']
    save('aji', **out)


ALL = {'validate': gen_validate, 'validate_obj': gen_validate_obj, 'ablation': gen_ablation, 'ddm': gen_ddm, 'unet': gen_unet, 'dam': gen_dam, 'losses': gen_losses, 'losses_classes': gen_losses_classes, 'train_iter': gen_train_iter, 'unet_train_iter': gen_unet_train_iter, 'hrnet': gen_hrnet, 'hrnet_train': gen_hrnet_train,
       'cdm': gen_cdm, 'cdm_inst': gen_cdm_inst, 'split': gen_split, 'probmaps': gen_probmaps, 'postproc': gen_postproc, 'aji': gen_aji}

if __name__ == '__main__':
    which = sys.argv[1:] or list(ALL)
    for w in which:
        print('==', w)
        ALL[w]()
