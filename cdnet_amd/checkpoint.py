"""Checkpoint interchange with the reference.

The reference saves {'epoch', 'state_dict', 'best_iou', 'best_loss', 'optimizer'} with `model` wrapped in nn.DataParallel
(every key prefixed 'module.', train.py:183-185, 421-427), names the files checkpoint.pth.tar / checkpoint_<epoch>.pth.tar /
checkpoint_best.pth.tar (save_checkpoint, train.py:461-480) and loads them with load_state_dict (train.py:297-302,
test_dam.py:163-165).  The files written here load in the reference and vice versa: parameter names and shapes are the
reference's, and the optimiser entry is a torch.optim.Adam state_dict (cdnet_amd.trainer.Trainer.state_dict)."""
import os
import shutil

import torch

PREFIX = 'module.'


def model_state(model):
    """state_dict with the DataParallel prefix, on the CPU, contiguous (parameters may be views of flat / padded storage)"""
    return {PREFIX + k: v.detach().cpu().contiguous().clone() for k, v in model.state_dict().items()}


def make_state(model, trainer, epoch, best_iou=0.0, best_loss=float('inf')):
    """the dict train.py:421-427 hands to save_checkpoint"""
    trainer.write_bn_counters()
    return {'epoch': epoch + 1, 'state_dict': model_state(model), 'best_iou': best_iou, 'best_loss': best_loss,
            'optimizer': trainer.state_dict()}


def save_checkpoint(state, epoch, is_best, save_dir, branch_value, cp_flag):
    """train.py:461-480 (same arguments, same file names)"""
    cp_dir = '{:s}/checkpoints'.format(save_dir)
    os.makedirs(cp_dir, exist_ok=True)
    filename = '{:s}/checkpoint.pth.tar'.format(cp_dir)
    torch.save(state, filename)
    branch = '' if branch_value == 'Main' else branch_value
    if cp_flag:
        shutil.copyfile(filename, '{:s}/checkpoint{:s}_{:d}.pth.tar'.format(cp_dir, branch, epoch + 1))
    if is_best:
        shutil.copyfile(filename, '{:s}/checkpoint{:s}_best.pth.tar'.format(cp_dir, branch))
    return filename


def _numpy_safe_globals():
    """what unpickling a numpy scalar needs: numpy._core.multiarray.scalar, numpy.dtype and the dtype classes of the scalars
    the reference writes (float / int AverageMeter values)"""
    import numpy as np
    try:
        from numpy._core.multiarray import scalar
    except ImportError:                                  # numpy < 2
        from numpy.core.multiarray import scalar
    return [scalar, np.dtype] + [type(np.dtype(t)) for t in ('float64', 'float32', 'int64', 'int32', 'bool')]


def load_checkpoint(path, model, trainer=None, strict=True, trusted_pickle=False):
    """Load a checkpoint written by the reference or by save_checkpoint.  Returns the checkpoint dict (epoch, best_iou, ...).
    Keys with or without the DataParallel prefix are accepted (test_dam.py:163-165 loads with strict=False).
    The interchange format holds only tensors, numbers, strings and containers of them, so the file is read with
    `weights_only=True` - with numpy's scalar / dtype reconstructors allow-listed, because the reference stores
    `best_iou` / `best_loss` as numpy.float64 (AverageMeter.avg, train.py:390-427), which a bare `weights_only=True`
    refuses; `trusted_pickle=True` (CDNET_TRUSTED_PICKLE=1, `--trusted-pickle`) opts into full unpickling for legacy files
    from a trusted source."""
    if trusted_pickle or os.environ.get('CDNET_TRUSTED_PICKLE', '0') == '1':
        ck = torch.load(path, map_location='cpu', weights_only=False)
    else:
        with torch.serialization.safe_globals(_numpy_safe_globals()):
            ck = torch.load(path, map_location='cpu', weights_only=True)
    for k in ('best_iou', 'best_loss'):
        if k in ck and not isinstance(ck[k], torch.Tensor):
            ck[k] = float(ck[k])                         # numpy.float64 -> float
    sd = ck['state_dict'] if 'state_dict' in ck else ck
    sd = {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=strict)
    if trainer is not None:
        # the fused optimiser keeps its own fp32 master copy of the parameters: refresh it from the module
        trainer.refresh_parameters()
        nbt = [v for k, v in sd.items() if k.endswith('num_batches_tracked')]
        trainer._bn_base, trainer._forwards = (int(max(int(v) for v in nbt)) if nbt else 0), 0
        if ck.get('optimizer') is not None:
            trainer.load_state_dict(ck['optimizer'])
    return ck
