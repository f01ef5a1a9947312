"""Inference entry points with the reference's names (test_dam.py).

  get_probmaps(input, model, opt, name, times)      test_dam.py:932-1035  -> [prob_maps, point_maps, pred_direction] (numpy)
  process_image(model, image, opt)                  test_dam.py:297-563   -> dict with the instance label map
  main()                                            test_dam.py:90-925    (command line; images from opt.test['img_dir'])
"""
import os
import numpy as np
import torch

from . import checkpoint, pipeline, postproc, stats_utils, utils
from .options import Options


def get_probmaps(input, model, opt, name=None, times=None):
    """input: float tensor [1,C,H,W].  Whole-image forward when opt.all_img_test == 1, else 256/40 sliding windows;
    returns [prob_maps f32 [3,H,W], point_maps f32 [1,H,W], pred_direction i64 [1,H,W]] as numpy arrays (the reference's
    return value for direction == 1 and mseloss == 1)."""
    x = input[0].cuda().float()
    _, H, W = x.shape
    with torch.no_grad():
        if opt.all_img_test == 1:
            (mask, point, direction), = utils.split_forward_views(model, x, max(H, W), 0, (0,), opt.direction_classes)
        else:
            (mask, point, direction), = utils.split_forward_views(model, x, opt.test['patch_size'], opt.test['overlap'], (0,),
                                                                  opt.direction_classes)
        prob, dcm = postproc.probmaps(mask[None], direction[None])
    if times is not None:
        times[0] = times[0] + 1
    return [prob[0].cpu().numpy(), point.cpu().numpy(), dcm.cpu().numpy().astype(np.int64)]


def process_image(model, image, opt):
    """image: float tensor [3,H,W] (after the test transform).  Returns dict(final=np.int32 [H,W], count=int)."""
    r = pipeline.infer_image(model, image.cuda().float(), opt)
    return dict(final=r['final'].cpu().numpy(), count=r['count'], pred=r['pred'].cpu().numpy())


def ground_truth_instances(label_dir, name):
    """instance map of the ground truth as test_dam.py builds it: `<label_dir>_ins/<name>.npy` when present (:237-238), else
    the 8-connected components of channel 0 of `<name>_label.png` grown by disk(1) (:242-246)"""
    from PIL import Image
    from scipy import ndimage as ndi
    ins = '{:s}_ins/{:s}.npy'.format(label_dir.rstrip('/'), name)
    if os.path.exists(ins):
        a = np.load(ins)
        return (a[:, :, 0] if a.ndim == 3 else a).astype(np.int32)
    path = '{:s}/{:s}_label.png'.format(label_dir, name)
    if not os.path.exists(path):
        return None
    lab = np.asarray(Image.open(path))
    inside = (lab if lab.ndim == 2 else lab[:, :, 0]) > 127
    cc, _ = ndi.label(inside, structure=np.ones((3, 3), int))
    disk1 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)
    return ndi.grey_dilation(cc, footprint=disk1).astype(np.int32)


def evaluate_labels(pred_labeled, gt_labeled):
    """the per-image numbers of test_dam.py:591-660: pixel-level accuracy / IoU / recall / precision / F1 of foreground vs
    foreground (utils.compute_pixel_level_metrics) and AJI, Dice, DQ / SQ / PQ of the instance maps (stats_utils, on the GPU)"""
    p, t = (np.asarray(pred_labeled) > 0).astype(np.float64), (np.asarray(gt_labeled) > 0).astype(np.float64)
    tp, tn = float((p * t).sum()), float(((1 - p) * (1 - t)).sum())
    fp, fn = float((p * (1 - t)).sum()), float(((1 - p) * t).sum())
    precision, recall = tp / (tp + fp + 1e-10), tp / (tp + fn + 1e-10)
    out = {'pixel_accu': (tp + tn) / (tp + fp + tn + fn + 1e-10), 'pixel_iou': tp / (tp + fp + fn + 1e-10), 'pixel_recall': recall,
           'pixel_precision': precision, 'pixel_F1': 2 * precision * recall / (precision + recall + 1e-10)}
    gt, pr = stats_utils.remap_label(np.asarray(gt_labeled).astype(np.int32)), stats_utils.remap_label(np.asarray(pred_labeled).astype(np.int32))
    if gt.max() == 0 or pr.max() == 0:
        out.update(AJI=0.0, Dice=0.0, DQ=0.0, SQ=0.0, PQ=0.0)
        return out
    (dq, sq, pq), _ = stats_utils.get_fast_pq(gt, pr, match_iou=0.5)
    out.update(AJI=float(stats_utils.get_fast_aji(gt, pr)[0]), Dice=float(stats_utils.get_dice_1(gt, pr)), DQ=float(dq), SQ=float(sq), PQ=float(pq))
    return out


def shard_names(names, rank, world):
    """inference partitions by image - rank r of `world` takes names[r::world] (no collective on the data path)"""
    return list(names[rank::world])


def gather_results(rows, rank, world):
    """per-image metric rows {name: {metric: value}} of every rank -> the union on rank 0 (None elsewhere): the only exchange of the
    inference path, a few scalars per image (torch.distributed all_gather_object: gloo on CPU, RCCL's group on the GPUs)"""
    if world <= 1:
        return dict(rows)
    import torch.distributed as dist
    assert dist.is_available() and dist.is_initialized(), 'WORLD_SIZE > 1 needs an initialised process group (python -m torch.distributed.run ...)'
    parts = [None] * world
    dist.all_gather_object(parts, dict(rows))
    if rank != 0:
        return None
    merged = {}
    for part in parts:
        for k, v in part.items():
            assert k not in merged, 'image %s was processed by two ranks' % k
            merged[k] = v
    return merged


def write_results(all_results, save_dir):
    """test_results.txt as test_dam.py:714-760 writes it: the average over the images, then one row per image (sorted by name: the file does
    not depend on how many ranks produced the rows)"""
    keys = list(next(iter(all_results.values())).keys())
    names = sorted(all_results)
    avg = {k: float(np.mean([all_results[n][k] for n in names])) for k in keys}
    with open(os.path.join(save_dir, 'test_results.txt'), 'w') as fh:
        fh.write('Average:\t' + '\t'.join('{:.4f}'.format(avg[k]) for k in keys) + '\n')
        for n in names:
            fh.write(n + '\t' + '\t'.join('{:.4f}'.format(all_results[n][k]) for k in keys) + '\n')
    return avg


def _dist_env():
    """(rank, world, made): the launcher's RANK / WORLD_SIZE (python -m torch.distributed.run, one process per GPU); a process group is
    created here when the launcher's rendezvous variables are present and none exists yet"""
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    made = False
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            dist.init_process_group(os.environ.get('CDNET_DIST_BACKEND', 'nccl'))
            made = True
    return rank, world, made


def main(argv=None):
    """The reference's test loop (test_dam.py:158-760) sharded by image over the ranks of one node and pipelined inside a rank:
    rank r takes names[r::world]; image i + 1's forward and post-processing are queued before image i's label map is waited for (its
    device-to-host copy runs on a copy stream into pinned memory; nothing calls .cpu() between two images); the per-image metric rows are
    gathered on rank 0, which alone writes test_results.txt and returns the averages (the other ranks return None)."""
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    trusted = '--trusted-pickle' in argv                    # legacy checkpoints that need full unpickling (trusted source only)
    argv = [a for a in argv if a != '--trusted-pickle']
    opt = Options(isTrain=False).parse(argv)
    rank, world, made_group = _dist_env()
    if torch.cuda.device_count() > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    model = utils.chooseModel(opt).cuda()
    if os.path.exists(opt.test['model_path']):
        checkpoint.load_checkpoint(opt.test['model_path'], model, strict=False, trusted_pickle=trusted)      # DataParallel prefix (test_dam.py:158-167)
    elif os.environ.get('CDNET_ALLOW_RANDOM_WEIGHTS') == '1':
        print("=> no checkpoint at '{}': evaluating RANDOM weights (CDNET_ALLOW_RANDOM_WEIGHTS=1)".format(opt.test['model_path']))
    else:
        # the reference's torch.load raises here (test_dam.py:158); a mistyped path must not produce plausible-looking metrics
        raise FileNotFoundError("checkpoint '{}' not found (set CDNET_ALLOW_RANDOM_WEIGHTS=1 to evaluate an untrained model)"
                                .format(opt.test['model_path']))
    model.eval()
    img_dir, label_dir = opt.test['img_dir'], opt.test['label_dir']
    names = sorted(f for f in os.listdir(img_dir) if f.endswith('.png')) if os.path.isdir(img_dir) else []
    mine = shard_names(names, rank, world)
    os.makedirs(opt.test['save_dir'], exist_ok=True)
    if opt.test.get('groundtruth', 0) == 1:
        # test_dam.py:600-602: object metrics against the XML annotations (utils.nuclei_accuracy_annotation_object_level)
        raise NotImplementedError('--groundtruth 1 (XML annotation files) is outside the accelerated path: supply instance labels')
    from PIL import Image
    dev = torch.device('cuda', torch.cuda.current_device())
    copy_stream = torch.cuda.Stream(device=dev)
    all_results = {}

    def launch(f):
        """queue one image: host decode, upload, the whole device pipeline, the copy-out - returns without waiting for any of it"""
        img = np.asarray(Image.open(os.path.join(img_dir, f)).convert('RGB'), dtype=np.float32) / 255.0
        x = torch.from_numpy(img).permute(2, 0, 1).contiguous()
        if 'normalize' in opt.transform['test']:
            mean, std = opt.transform['test']['normalize']
            x = (x - torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)) / torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
        r = pipeline.infer_image(model, x.pin_memory().to(dev, non_blocking=True), opt, defer=True)
        done = torch.cuda.Event()
        done.record()
        host = {k: torch.empty(r[k].shape, dtype=r[k].dtype).pin_memory() for k in ('final', 'counts', 'minmax')}
        copy_stream.wait_event(done)
        with torch.cuda.stream(copy_stream):
            for k, h in host.items():
                r[k].record_stream(copy_stream)
                h.copy_(r[k], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(copy_stream)
        return f, host, copied

    def finish(item):
        f, host, copied = item
        copied.synchronize()                                               # (this image's copy only: the next image is already queued)
        postproc.check_views(None, host['minmax'])                         # the reference's assertion, test_dam.py:535
        final, count = host['final'].numpy(), int(host['counts'].reshape(-1)[0])
        if opt.test.get('save_flag', True):                                # test_dam.py:670-684 (save_flag)
            Image.fromarray(final.astype(np.uint16)).save(os.path.join(opt.test['save_dir'], f[:-4] + '_seg.tiff'))
        print('{:s}: {:d} nuclei'.format(f, count))
        gt = ground_truth_instances(label_dir, f[:-4]) if label_dir and os.path.isdir(label_dir) else None
        if gt is not None and gt.shape == final.shape:                     # eval_flag (test_dam.py:132, 591-660)
            res = evaluate_labels(final, gt)
            rec, prec, f1, dice_o, iou_o, haus, aji_o = utils.nuclei_accuracy_object_level(final, gt)       # :603-604
            res.update(obj_recall=rec, obj_precision=prec, obj_F1=f1, obj_dice=dice_o, obj_iou=iou_o, obj_haus=haus, obj_AJI=aji_o)
            all_results[f[:-4]] = res
            print('\tpixel_iou = {pixel_iou:.4f}, pixel_F1 = {pixel_F1:.4f}, AJI = {AJI:.4f}, Dice = {Dice:.4f}, DQ = {DQ:.4f}, '
                  'SQ = {SQ:.4f}, PQ = {PQ:.4f}'.format(**res))

    pending = None
    for f in mine:
        item = launch(f)                  # image i + 1 is in flight ...
        if pending is not None:
            finish(pending)               # ... while image i is copied out, checked, saved and scored
        pending = item
    if pending is not None:
        finish(pending)
    merged = gather_results(all_results, rank, world)
    avg = None
    if rank == 0 and merged:
        avg = write_results(merged, opt.test['save_dir'])
        print('Average of {:d} images: '.format(len(merged)) + ', '.join('{:s} = {:.4f}'.format(k, v) for k, v in avg.items()))
    if made_group:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return avg


if __name__ == '__main__':
    main()
