"""ORACLE (test infrastructure): the reference's DAM training iteration restated in plain PyTorch fp32 (CPU),
vectorised.  Pinned by tests/test_oracle_models.py against tests/golden/{losses,train_iter}.npz.

Follows train_util_dam.py:54-311 (paths relative to /root/reference):
  targets        :73-142  (label // 127, one-hot of the 3-class label, direction one-hot masked by SAMPLE 0's
                           foreground - reference quirk :139, kept)
  losses         :167-276 with loss.py:131-147 (DiceLoss), :150-176 (MulticlassDiceLoss), :181-199
                 (Weight_DiceLoss), :202-260 (WeightMulticlassDiceLoss)
  optimiser      utils.py:915-918: Adam(lr, betas=(0.9, 0.99), weight_decay) - L2 added to the gradient
"""
import torch
import torch.nn.functional as F


def dice_loss(p, t):                                   # loss.py:135-147
    N = t.size(0)
    pf, tf = p.reshape(N, -1), t.reshape(N, -1)
    inter = (pf * tf).sum(1)
    loss = 2 * (inter + 1) / (pf.sum(1) + tf.sum(1) + 1)
    return 1 - loss.sum() / N


def wdice_loss(p, t, w):                               # loss.py:185-199
    N = t.size(0)
    pf, tf, wf = p.reshape(N, -1), t.reshape(N, -1), w.reshape(N, -1)
    inter = (pf * tf * wf).sum(1)
    d = 2 * (inter + 1) / ((pf * wf).sum(1) + (tf * wf).sum(1) + 1)
    return 1 - d.sum() / N


def multiclass_dice(prob, onehot):                     # loss.py:160-176
    return sum(dice_loss(prob[:, i], onehot[:, i]) for i in range(onehot.shape[1]))


def weight_multiclass_dice(prob, onehot, w):           # loss.py:216-260
    C = onehot.shape[1]
    total = 0
    for i in range(C):
        if i == 0:
            d = wdice_loss(prob[:, 0], onehot[:, 0], w) * 2
        else:
            prev = C - 1 if i == 1 else i - 1
            nxt = 1 if i == C - 1 else i + 1
            d = wdice_loss(prob[:, i], onehot[:, i], w)
            d = d - (1 - wdice_loss(prob[:, i], onehot[:, prev], w)) - (1 - wdice_loss(prob[:, i], onehot[:, nxt], w))
        total = total + d
    return total / C


def direction_onehot(direction, label, classes=9, quirk_sample0=True):
    """train_util_dam.py:123-142.  direction int [B,H,W], label int [B,H,W] in {0,1,2}."""
    B = direction.shape[0]
    oh = torch.zeros((B, classes) + tuple(direction.shape[1:]), dtype=torch.float32)
    for j in range(B):
        uniq = torch.unique(direction[j])
        if len(uniq) > 1:
            m = (label[0] if quirk_sample0 else label[j]) != 0
            for k in uniq.tolist():
                oh[j, k] = ((direction[j] == k) & m).float()
        else:
            oh[j, 0] = (direction[j] == uniq[0]).float()
    return oh


def dam_losses(mask_logits, point_out, dir_logits, label, direction, point_target, weight_png, quirk_sample0=True):
    """Returns dict of the five terms and the total (train_util_dam.py:167-276, default options)."""
    w = weight_png.float().div(20)
    if w.dim() == 4:
        w = w.squeeze(1)
    label = label.long()
    direction = direction.long()
    ce = (F.nll_loss(F.log_softmax(mask_logits, 1), label, reduction='none') * w).mean()
    onehot3 = F.one_hot(label, 3).permute(0, 3, 1, 2).float()
    dice = multiclass_dice(F.softmax(mask_logits, 1), onehot3)
    dce = (F.nll_loss(F.log_softmax(dir_logits, 1), direction, reduction='none') * w).mean()
    oh9 = direction_onehot(direction, label, dir_logits.shape[1], quirk_sample0)
    wdice = weight_multiclass_dice(F.softmax(dir_logits, 1), oh9, w)
    pt = point_target.float()
    if pt.dim() == 3:
        pt = pt.unsqueeze(1)
    mse = F.mse_loss(point_out, pt)
    total = ce + dice + dce + wdice + mse
    return dict(ce=ce, dice=dice, dce=dce, wdice=wdice, mse=mse, total=total)


def ablation_losses(outputs, label, direction, point_target, weight_png, quirk_sample0=True):
    """the same loss for the ablation heads (train_util_dam.py:152-166 unpacks by the number of outputs): three outputs as the
    rev1 model; two outputs (mask, direction; options direction = 1, mseloss = 0) without the point term"""
    if len(outputs) == 3:
        return dam_losses(outputs[0], outputs[1], outputs[2], label, direction, point_target, weight_png, quirk_sample0)
    zero = torch.zeros((outputs[0].shape[0], 1) + tuple(outputs[0].shape[2:]))
    return dam_losses(outputs[0], zero, outputs[1], label, direction, torch.zeros_like(point_target), weight_png, quirk_sample0)


def make_adam(model, lr=1e-3, weight_decay=1e-4):
    return torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, 0.99), weight_decay=weight_decay)


def pixel_metrics(pred, target):
    """utils.accuracy_pixel_level (utils.py:67-110): per-sample tp/fp/fn/tn of (pred == 1) vs (target == 1), the derived
    [accuracy, IoU, recall, precision, F1] averaged over the batch (float64)."""
    import numpy as np
    pred, target = np.asarray(pred), np.asarray(target)
    res = np.zeros(5)
    for i in range(target.shape[0]):
        p, t = (pred[i] == 1).astype(np.float64), (target[i] == 1).astype(np.float64)
        tp, tn, fp, fn = (p * t).sum(), ((1 - p) * (1 - t)).sum(), (p * (1 - t)).sum(), ((1 - p) * t).sum()
        precision, recall = tp / (tp + fp + 1e-10), tp / (tp + fn + 1e-10)
        res += [(tp + tn) / (tp + fp + tn + fn + 1e-10), tp / (tp + fp + fn + 1e-10), recall, precision,
                2 * precision * recall / (precision + recall + 1e-10)]
    return res / target.shape[0]


def train_iteration(model, optimizer, x, label, direction, point_target, weight_png):
    """One reference iteration; returns the loss dict (floats) + 'metrics' (train_util_dam.py:279-293, default options:
    argmax of the direction branch against the direction target)."""
    model.train()
    mask, point, dirn = model(x)
    L = dam_losses(mask, point, dirn, label, direction, point_target, weight_png)
    metrics = pixel_metrics(dirn.detach().argmax(1).numpy(), direction.numpy())
    optimizer.zero_grad()
    L['total'].backward()
    optimizer.step()
    out = {k: float(v) for k, v in L.items()}
    out['metrics'] = metrics
    return out


# ---------------------------------------------------------------------------------------------------------
# plain UNet (train_util.py:58-260, default options: add_weightMap, dice = 1, alpha = 0, boundary_loss = 0)
# ---------------------------------------------------------------------------------------------------------
def unet_losses(logits, label, weight_png):
    w = weight_png.float().div(20)                                   # train_util.py:109
    if w.dim() == 4:
        w = w.squeeze(1)
    label = label.long()
    ce = (F.nll_loss(F.log_softmax(logits, 1), label, reduction='none') * w).mean()      # :128-136
    onehot3 = F.one_hot(label, 3).permute(0, 3, 1, 2).float()
    dice = multiclass_dice(F.softmax(logits, 1), onehot3)                                 # :183-186
    return dict(ce=ce, dice=dice, total=ce + dice)


def unet_train_iteration(model, optimizer, x, label, weight_png):
    model.train()
    L = unet_losses(model(x), label, weight_png)
    optimizer.zero_grad()
    L['total'].backward()
    optimizer.step()
    return {k: float(v) for k, v in L.items()}


# ---------------------------------------------------------------------------------------------------------
# validate (train_util_dam.py:367-636, default options; do_object_metric = 0)
# ---------------------------------------------------------------------------------------------------------
def validate_losses(mask_logits, point_out, dir_logits, label, direction, point_target, weight_png):
    """The loss mix of validate(): unweighted mask CE (:499-505, the weight-map multiply is commented out), multi-class dice on
    softmax(mask) (:540-543), weighted direction CE (:553-559), PLAIN multi-class dice on the direction probabilities whose
    channel 0 is multiplied by P(background) (:564-568) against the one-hot direction target masked by SAMPLE 0's foreground
    (:463-470, `target[0]`), MSE against point_target / 255 (:575-580)."""
    w = weight_png.float().div(20)
    if w.dim() == 4:
        w = w.squeeze(1)
    label = label.long()
    direction = direction.long()
    ce = F.nll_loss(F.log_softmax(mask_logits, 1), label, reduction='none').mean()
    prob = F.softmax(mask_logits, 1)
    dice = multiclass_dice(prob, F.one_hot(label, 3).permute(0, 3, 1, 2).float())
    dce = (F.nll_loss(F.log_softmax(dir_logits, 1), direction, reduction='none') * w).mean()
    # one-hot: channel k <-> k-th value of torch.unique over the batch (:462-468), zeroed off sample 0's foreground
    uniq = torch.unique(direction)
    C = dir_logits.shape[1]
    oh = torch.zeros((direction.shape[0], C) + tuple(direction.shape[1:]))
    fg0 = (label[0] == 1) | (label[0] == 2)
    for k in range(C):
        oh[:, k] = (direction == uniq[k]).float() * fg0.float()          # IndexError like the reference when a class is absent
    q = F.softmax(dir_logits, 1).clone()
    q[:, 0] = q[:, 0] * prob[:, 0]
    ddice = multiclass_dice(q, oh)
    pt = point_target.float()
    if pt.dim() == 3:
        pt = pt.unsqueeze(1)
    mse = F.mse_loss(point_out, pt / 255)
    return dict(ce=ce, dice=dice, dce=dce, ddice=ddice, mse=mse, total=ce + dice + dce + ddice + mse)


def validate_iteration(model, x, label, direction, point_target, weight_png, split=None):
    """One validate() sample -> the 16-value row [loss, direction CE, direction dice, MSE, pixel accuracy, IoU, recall, precision,
    F1, 0, 0, 0, 0, IoU, 0, 0] (:620-623 with do_object_metric = 0: `iou = pixel_iou`).  split = (size, overlap) evaluates through
    the sliding windows of utils.split_forward_dam (all_img_test = 0)."""
    import numpy as np
    model.eval()
    with torch.no_grad():
        if split is None:
            mask, point, dirn = model(x)
        else:
            from . import infer
            mask, point, dirn = infer.split_forward(model, x, split[0], split[1], dirn_classes(model))
        L = validate_losses(mask, point, dirn, label, direction, point_target, weight_png)
    m = pixel_metrics(mask.argmax(1).numpy(), label.numpy())
    return np.array([float(L['total']), float(L['dce']), float(L['ddice']), float(L['mse']), m[0], m[1], m[2], m[3], m[4],
                     0, 0, 0, 0, m[1], 0, 0], dtype=np.float64)


def dirn_classes(model):
    return model.direction_conv.out_channels
