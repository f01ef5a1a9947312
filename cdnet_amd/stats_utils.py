"""Instance metrics of the reference's stats_utils.py on the GPU: get_fast_aji (:7-106), get_fast_pq (:182-276),
get_dice_1 (:323-335), remap_label (:361-392).

Same names, arguments and return values; `true` / `pred` are numpy HW integer label images (what the reference passes)
or torch CUDA int32 tensors.  The pass over the pixels (per-label areas, sparse pairwise intersections) runs in
csrc/metrics.hip; the per-pair arithmetic is done here in float64 with the reference's formulas, so results agree with
the reference to the last bits (tests/golden/aji.npz).  No CPU fallback: the device pass is required."""
import numpy as np
import torch

from . import _lib


def _dev(a):
    if torch.is_tensor(a):
        assert a.is_cuda
        return a.to(torch.int32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()


def _check(err, who):
    e = int(err.item())
    if e == 1:
        raise ValueError('%s: instance id outside [0, capacity)' % who)
    if e == 2:
        raise RuntimeError('%s: pair table full' % who)


def remap_label(pred, by_size=False):
    """ids -> 1..K in increasing id order (by_size=True is not on the accelerated path)"""
    if by_size:
        raise NotImplementedError('remap_label(by_size=True) is outside the accelerated path')
    is_np = not torch.is_tensor(pred)
    d = _dev(pred)
    cap = int(d.max().item()) + 1 if d.numel() else 1
    if cap <= 1:
        return pred
    scratch = torch.empty((cap,), dtype=torch.int32, device=d.device)
    out = torch.empty_like(d)
    err = torch.empty((1,), dtype=torch.int32, device=d.device)
    _lib.call('cdnet_remap_label', _lib.ptr(d), 1, d.numel(), cap, _lib.ptr(scratch), _lib.ptr(out), _lib.ptr(err), _lib.stream_ptr())
    _check(err, 'remap_label')
    return out.cpu().numpy() if is_np else out


def pair_table(true, pred):
    """(area_true [T+1], area_pred [P+1], pairs: int64 array [K,3] of (true_id, pred_id, intersection) sorted by ids)"""
    t, p = _dev(true), _dev(pred)
    assert t.shape == p.shape
    cap = int(max(t.max().item(), p.max().item())) + 1
    cap = max(cap, 2)
    if cap > 65536:
        raise ValueError('more than 65535 instances: remap_label first')
    slots = 1 << 12
    while slots < 8 * cap:
        slots <<= 1
    while True:
        at = torch.empty((cap,), dtype=torch.int32, device=t.device)
        ap = torch.empty((cap,), dtype=torch.int32, device=t.device)
        hk = torch.empty((slots,), dtype=torch.int32, device=t.device)
        hc = torch.empty((slots,), dtype=torch.int32, device=t.device)
        err = torch.empty((1,), dtype=torch.int32, device=t.device)
        _lib.call('cdnet_label_pair_histogram', _lib.ptr(t), _lib.ptr(p), 1, t.numel(), cap, slots, _lib.ptr(at), _lib.ptr(ap), _lib.ptr(hk),
                  _lib.ptr(hc), _lib.ptr(err), _lib.stream_ptr())
        if int(err.item()) == 2 and slots < (1 << 24):
            slots <<= 2
            continue
        _check(err, 'pair_table')
        break
    keys = hk.cpu().numpy().view(np.uint32)
    cnt = hc.cpu().numpy()
    nz = keys != 0
    keys, cnt = keys[nz].astype(np.int64), cnt[nz].astype(np.int64)
    order = np.argsort(keys, kind='stable')
    keys, cnt = keys[order], cnt[order]
    pairs = np.stack([keys >> 16, keys & 0xffff, cnt], 1) if keys.size else np.zeros((0, 3), np.int64)
    return at.cpu().numpy().astype(np.int64), ap.cpu().numpy().astype(np.int64), pairs


def get_dice_1(true, pred):
    """2 |T & P| / (|T| + |P|) on the binarised label images"""
    at, ap, pairs = pair_table(true, pred)
    return 2.0 * float(pairs[:, 2].sum()) / float(at.sum() + ap.sum())


def get_fast_aji(true, pred):
    """(aji, FP/fm, FN/fm, less/fm, more/fm) exactly as stats_utils.get_fast_aji returns them.  Ids must be contiguous
    (call remap_label first, as the reference requires)."""
    at, ap, pairs = pair_table(true, pred)
    T, P = len(at) - 1, len(ap) - 1
    true_ids = [i for i in range(1, T + 1) if at[i] > 0]
    pred_ids = [i for i in range(1, P + 1) if ap[i] > 0]
    assert true_ids == list(range(1, len(true_ids) + 1)) and pred_ids == list(range(1, len(pred_ids) + 1)), \
        'get_fast_aji needs contiguous instance ids (remap_label)'
    best = {}                                    # true id -> (iou, pred id, inter, union) with np.argmax's first-maximum rule
    for ti, pi, inter in pairs:
        inter = float(inter)
        union = float(at[ti] + ap[pi]) - inter
        iou = inter / (union + 1.0e-6)
        cur = best.get(int(ti))
        if cur is None or iou > cur[0]:
            best[int(ti)] = (iou, int(pi), inter, union)
    overall_inter = overall_union = overall_FP = overall_FN = 0.0
    paired_pred = set()
    for ti in sorted(best):
        iou, pi, inter, union = best[ti]
        if iou > 0.0:
            overall_inter += inter
            overall_union += union
            overall_FP += float(ap[pi]) - inter
            overall_FN += float(at[ti]) - inter
            paired_pred.add(pi)
    less_pred = more_pred = 0
    for ti in true_ids:
        if ti not in best or not best[ti][0] > 0.0:
            less_pred += int(at[ti])
            overall_union += int(at[ti])
    for pi in pred_ids:
        if pi not in paired_pred:
            more_pred += int(ap[pi])
            overall_union += int(ap[pi])
    with np.errstate(divide='ignore', invalid='ignore'):       # numpy float64 semantics as in the reference (nan / inf, no exception)
        f = np.float64
        aji_score = f(overall_inter) / f(overall_union)
        fm = f(overall_union) - f(overall_inter)
        return float(aji_score), float(f(overall_FP) / fm), float(f(overall_FN) / fm), float(f(less_pred) / fm), float(f(more_pred) / fm)


def get_fast_pq(true, pred, match_iou=0.5):
    """[dq, sq, pq], [paired_true, paired_pred, unpaired_true, unpaired_pred] (stats_utils.py:182-276)"""
    assert match_iou >= 0.0, "Cant' be negative"
    at, ap, pairs = pair_table(true, pred)
    true_ids = [i for i in range(1, len(at)) if at[i] > 0]
    pred_ids = [i for i in range(1, len(ap)) if ap[i] > 0]
    iou = {}
    for ti, pi, inter in pairs:
        inter = float(inter)
        total = float(at[ti] + ap[pi])
        iou[(int(ti), int(pi))] = inter / (total - inter)
    if match_iou >= 0.5:
        sel = sorted(k for k, v in iou.items() if v > match_iou)          # np.nonzero order: by true id, then pred id
        paired_true = [k[0] for k in sel]
        paired_pred = [k[1] for k in sel]
        paired_iou = np.array([iou[k] for k in sel], dtype=np.float64)
    else:
        from scipy.optimize import linear_sum_assignment
        m = np.zeros((len(true_ids), len(pred_ids)), np.float64)
        for (ti, pi), v in iou.items():
            m[ti - 1, pi - 1] = v
        r, c = linear_sum_assignment(-m)
        piou = m[r, c]
        paired_true = list(r[piou > match_iou] + 1)
        paired_pred = list(c[piou > match_iou] + 1)
        paired_iou = piou[piou > match_iou]
    unpaired_true = [i for i in true_ids if i not in paired_true]
    unpaired_pred = [i for i in pred_ids if i not in paired_pred]
    tp, fp, fn = len(paired_true), len(unpaired_pred), len(unpaired_true)
    dq = tp / (tp + 0.5 * fp + 0.5 * fn)
    sq = paired_iou.sum() / (tp + 1.0e-6)
    return [dq, sq, dq * sq], [paired_true, paired_pred, unpaired_true, unpaired_pred]
