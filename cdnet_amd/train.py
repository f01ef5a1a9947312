"""Training entry point with the reference's shape (train.py:59-530, the part that is the hot path): options -> seeds ->
`chooseModel` -> optimiser -> epochs of `train_util_dam.train` (or the plain-UNet step) -> checkpoint.

    python -m cdnet_amd.train --synthetic 64 --epochs 2              # synthetic tiles (no dataset needed)
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m cdnet_amd.train ...   # one process per GPU, RCCL all-reduce

The reference's augmentation pipeline (albumentations / PIL RNG streams) and its CSV / tensorboard logging are outside
the accelerated path: with a real dataset the loader below applies only random 256x256 crops,
`LabelEncoding` (on the GPU) and `ToTensor`; `nn.DataParallel` (train.py:185) is replaced by one process per GPU."""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

from . import checkpoint, synth, train_util_dam, utils
from .options import Options
from .trainer import synthetic_batch, UNetTrainer


class _SyntheticLoader:
    """`n` batches of the SURVEY 8d recipe in the sample layout of the reference's DataLoader"""

    def __init__(self, n_batches, batch, dev, seed):
        self.items = []
        for k in range(n_batches):
            x, lab, dirn, point, weight = synthetic_batch(batch, dev, seed=seed + k)
            target0 = (lab.to(torch.int64) * 127 + (lab == 2).to(torch.int64)).unsqueeze(1)      # {0,127,255} as ToTensor emits
            self.items.append((x, weight.unsqueeze(1), target0, point, dirn))

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)


def _dataset_layout(opt, x, logger=None):
    """directories and file suffixes of split `x` as the reference lays them out (train.py:216-288): with validation = 1 the targets
    are instance-level label files under <label_dir>/<x>_ins ('label.mat' for CPM2017 / MultiOrgan, 'label.npy' otherwise; CPM2017
    validates on its test split), with validation = 0 three-class 'label.png' under <label_dir>/<x>.  Falls back to the other
    layout when only that one exists on disk."""
    tr = opt.train
    split = 'test' if (opt.dataset == 'CPM2017' and x == 'val' and tr['validation'] == 1) else x
    img, wmap = '{:s}/{:s}'.format(tr['img_dir'], split), '{:s}/{:s}'.format(tr['weight_map_dir'], split)
    ins = ('{:s}/{:s}_ins'.format(tr['label_dir'], split), ['weight.png', 'label.mat' if opt.dataset in ('CPM2017', 'MultiOrgan') else 'label.npy'])
    png = ('{:s}/{:s}'.format(tr['label_dir'], split), ['weight.png', 'label.png'])
    first, second = (ins, png) if tr['validation'] == 1 else (png, ins)
    pick = first if os.path.isdir(first[0]) or not os.path.isdir(second[0]) else second
    if pick is second and logger is not None:
        logger.info('{:s} not found: using {:s} ({:s})'.format(first[0], second[0], second[1][1]))
    return [img, wmap, pick[0]], pick[1]


def main(argv=None):
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--synthetic', type=int, default=0, help='number of synthetic batches per epoch (0 = read the dataset folders)')
    ap.add_argument('--synthetic-val', type=int, default=0, help='synthetic validation batches per epoch (with --synthetic and --validation 1)')
    ap.add_argument('--trusted-pickle', action='store_true', help='resume from a legacy checkpoint that needs full unpickling (trusted source only)')
    own, rest = ap.parse_known_args(argv)
    opt = Options(isTrain=True).parse(rest)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format='%(message)s', stream=sys.stdout)
    logger = logging.getLogger('cdnet_amd.train')
    torch.manual_seed(opt.train['seed'])                   # train.py:75-81 (same seed on every rank = same initial weights)
    np.random.seed(opt.train['seed'])
    model = utils.chooseModel(opt).to(dev)
    plain_unet = opt.model['modelName'] == 'UNet'
    if plain_unet:
        trainer = UNetTrainer(model, lr=opt.train['lr'], weight_decay=opt.train['weight_decay'], world_size=world)
    else:
        trainer, _ = utils.get_optimizer(opt, model, world_size=world)
    best_iou, best_loss = 0.0, float('inf')
    if opt.train['checkpoint']:                            # train.py:293-306: resume (weights, Adam moments, epoch, best values)
        if os.path.isfile(opt.train['checkpoint']):
            ck = checkpoint.load_checkpoint(opt.train['checkpoint'], model, trainer, trusted_pickle=own.trusted_pickle)
            opt.train['start_epoch'] = ck.get('epoch', 0)
            best_iou, best_loss = ck.get('best_iou', best_iou), ck.get('best_loss', best_loss)
            logger.info("=> loaded checkpoint '{}' (epoch {})".format(opt.train['checkpoint'], opt.train['start_epoch']))
            trainer.sync_from_rank0()                      # every replica continues from rank 0's file
        else:
            logger.info("=> no checkpoint found at '{}'".format(opt.train['checkpoint']))
    B = opt.train['batch_size']
    if own.synthetic > 0:
        loader = _SyntheticLoader(own.synthetic, B, dev, seed=opt.train['seed'] + 1000 * rank)
    else:
        # train.py:262-290 without validation split: <img_dir>/train, <weight_map_dir>/train, <label_dir>/train
        from .data_folder import DataFolder, TileBatches
        dir_list, post_fix = _dataset_layout(opt, 'train', logger)
        dset = DataFolder(dir_list, post_fix, [3, 1, 3])
        loader = TileBatches(dset, opt.transform['train'], B, dev, seed=opt.train['seed'] + 1000 * rank, logger=logger)
        logger.info('{:d} training images in {:s}'.format(len(dset), dir_list[0]))
    # validation set (train.py:262-290: <img_dir>/val etc.) for the best-checkpoint / early-stopping logic of train.py:348-447
    val_loader = None
    if opt.train['validation'] == 1 and not plain_unet:
        if own.synthetic > 0:
            val_loader = _SyntheticLoader(max(1, own.synthetic_val), B, dev, seed=opt.train['seed'] + 777 + 1000 * rank)
        else:
            from .data_folder import DataFolder, TileBatches
            vdirs, vfix = _dataset_layout(opt, 'val', logger)
            if all(os.path.isdir(d) for d in vdirs):
                vset = DataFolder(vdirs, vfix, [3, 1, 3])
                # options.py:358: the validation transform is {label_encoding, to_tensor, normalize} - no crop: every epoch scores the same
                # whole images (validate() runs them whole or through split_forward_dam with input_size / val_overlap, train_util_dam.py:474)
                vt = opt.transform.get('val') or {k: v for k, v in opt.transform['train'].items() if k in ('label_encoding', 'to_tensor', 'normalize')}
                val_loader = TileBatches(vset, vt, 1, dev, seed=opt.train['seed'], shuffle=False, logger=logger)
            else:
                logger.info('validation = 1 but {} is missing: the training results stand in for the validation results'.format(vdirs[0]))
    early_stopping = utils.EarlyStopping(patience=opt.train['early_stop']) if opt.train['early_stop'] > 0 else None
    res = None
    for epoch in range(opt.train['start_epoch'], opt.train['num_epochs']):
        t0 = time.time()
        if plain_unet:
            tot = np.zeros(3)
            for x, w, target0, _, _ in loader:
                tot += trainer.train_step(x, train_util_dam._label3(target0), w[:, 0].contiguous()).cpu().numpy()
            res = trainer.reduce_scalars(tot / len(loader))
        else:
            res = train_util_dam.train(loader, model, trainer, None, epoch, opt, logger)
        torch.cuda.synchronize()
        dt = time.time() - t0
        logger.info('epoch {:d}: loss {:.4f}  ({:.1f} tiles/s on {:d} GPU(s))'.format(epoch + 1, float(res[0]), world * B * len(loader) / dt, world))
        # train.py:348-387: validation results (or the training results standing in for them), best model by val_iou
        if val_loader is not None:
            val = trainer.reduce_scalars(train_util_dam.validate(val_loader, model, None, opt, logger, all_img_test=opt.all_img_test))
            val_loss, val_iou, val_F1 = float(val[0]), float(val[5]), float(val[8])
        elif plain_unet:
            val_loss, val_iou, val_F1 = float(res[0]), 0.0, 0.0
        else:
            val_loss, val_iou, val_F1 = float(res[0]), float(res[7]), float(res[10])
        is_best = val_iou > best_iou                       # train.py:385
        best_iou, new_best_loss = max(val_iou, best_iou), min(val_loss, best_loss)
        if rank == 0 and opt.train.get('save_dir'):
            # train.py:406-427: checkpoint.pth.tar every epoch, numbered copies at checkpoint_freq, checkpoint_best on a new best
            if val_loader is None and plain_unet:
                is_best = val_loss < best_loss
            best_loss = new_best_loss
            cp_flag = int((epoch + 1) % opt.train['checkpoint_freq'] == 0 or epoch + 1 == opt.train['num_epochs'])
            os.makedirs(opt.train['save_dir'], exist_ok=True)
            checkpoint.save_checkpoint(checkpoint.make_state(model, trainer, epoch, best_iou, best_loss), epoch, is_best,
                                       opt.train['save_dir'], 'Main', cp_flag)
        best_loss = new_best_loss
        if early_stopping is not None:                     # train.py:442-447: monitored value -F1 - IoU, never before epoch 100
            early_stopping(-val_F1 - val_iou, epoch)
            if early_stopping.early_stop:
                logger.info('epoch = {} Early stopping...'.format(epoch))
                break
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == '__main__':
    main()
