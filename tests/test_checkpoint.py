"""Checkpoint interchange (cdnet_amd/checkpoint.py): files in the reference's format - 'module.'-prefixed
state_dict + a torch.optim.Adam state_dict - written here load into plain PyTorch (the oracle models carry the reference's
parameter names) and the other way round.  Host logic only: no kernels run."""
import numpy as np
import pytest
import torch

from cdnet_amd import checkpoint, trainer
from oracle import train as ot


class _Opt:
    model = {'out_c': 3}


def _pair(kind):
    if kind == 'dam':
        from cdnet_amd.models.dam.model_unet_rev1 import Unet
        from oracle import models as om
        return Unet(backbone_name='vgg16_bn', pretrained=False, classes=3), om.Unet()
    from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
    from oracle import hrnet as oh
    return HighResolutionNet(_Opt()), oh.HighResolutionNet()


@pytest.mark.parametrize('kind', ['dam', 'hrnet'])
def test_checkpoint_written_here_loads_in_pytorch_and_back(kind, tmp_path):
    torch.manual_seed(1)
    m, ref = _pair(kind)
    tr = trainer.Trainer(m, lr=2e-3, weight_decay=1e-4)
    f = tr.flat
    g = torch.Generator().manual_seed(5)
    f.M[:f.n_used] = torch.randn((f.n_used,), generator=g) * 1e-2                 # as if some steps had run
    f.V[:f.n_used] = torch.rand((f.n_used,), generator=g) * 1e-3
    if hasattr(m, '_slots'):                                                      # padding carries no state
        for real, pp, segs in m._slots:
            off, sz = f.offsets[[n for n, q in m.trainer_named_parameters().items() if q is pp][0]]
            keep = torch.zeros(pp.shape, dtype=torch.bool)
            if segs is None:
                keep[tuple(slice(0, n) for n in real.shape)] = True
            else:
                for r0, n, p0 in segs:
                    keep[:, p0:p0 + n] = True
            f.M[off:off + sz] *= keep.reshape(-1)
            f.V[off:off + sz] *= keep.reshape(-1)
    f.step_count = 7
    tr._forwards = 7                                       # seven training forwards: BatchNorm's num_batches_tracked
    path = checkpoint.save_checkpoint(checkpoint.make_state(m, tr, epoch=4, best_iou=0.5, best_loss=1.25), 4, True, str(tmp_path), 'Main', 1)
    for name in ('checkpoint.pth.tar', 'checkpoint_5.pth.tar', 'checkpoint_best.pth.tar'):       # train.py:461-480
        assert (tmp_path / 'checkpoints' / name).exists()
    # --- the reference's side: DataParallel(model).load_state_dict + Adam.load_state_dict (train.py:297-302)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert ck['epoch'] == 5 and ck['best_iou'] == 0.5 and all(k.startswith('module.') for k in ck['state_dict'])
    nbt = [int(v) for k, v in ck['state_dict'].items() if k.endswith('num_batches_tracked')]
    unused_p = tuple('module.' + u for u in getattr(m, 'UNUSED_PREFIXES', ()))
    for k, v in ck['state_dict'].items():
        if k.endswith('num_batches_tracked'):
            assert int(v) == (0 if unused_p and k.startswith(unused_p) else 7), k          # never-run BatchNorms keep 0, as in PyTorch
    assert nbt
    dp = torch.nn.DataParallel(ref)
    dp.load_state_dict(ck['state_dict'])
    for (n, p), (n2, p2) in zip(ref.named_parameters(), m.named_parameters()):
        assert n == n2 and torch.equal(p.detach(), p2.detach()), n
    opt = ot.make_adam(ref, lr=1e-3)
    opt.load_state_dict(ck['optimizer'])
    assert opt.param_groups[0]['lr'] == 2e-3 and opt.param_groups[0]['betas'] == (0.9, 0.99)
    st = opt.state_dict()['state']
    unused = tuple(getattr(m, 'UNUSED_PREFIXES', ()))
    names = [n for n, _ in ref.named_parameters()]
    assert sorted(st.keys()) == [i for i, n in enumerate(names) if not n.startswith(unused)]
    assert all(int(s['step']) == 7 for s in st.values())
    # one PyTorch Adam step from the loaded state runs (shapes / dtypes are what torch expects)
    for p in ref.parameters():
        p.grad = torch.ones_like(p) * 1e-3
    opt.step()
    # --- and back: a checkpoint written by PyTorch loads here, optimiser moments included
    ck = torch.load(path, map_location='cpu', weights_only=False)       # (Adam.load_state_dict shares tensors with the dict it is given)
    opt2 = ot.make_adam(ref)
    opt2.load_state_dict(ck['optimizer'])
    # the reference stores numpy.float64 here (AverageMeter.avg through max() / min(), train.py:390-427)
    import numpy as np
    ref_ck = {'epoch': 9, 'state_dict': torch.nn.DataParallel(ref).state_dict(), 'best_iou': np.float64(0.1), 'best_loss': np.float64(2.0),
              'optimizer': opt2.state_dict()}
    torch.save(ref_ck, str(tmp_path / 'ref.pth.tar'))
    m2, _ = _pair(kind)
    tr2 = trainer.Trainer(m2)
    got = checkpoint.load_checkpoint(str(tmp_path / 'ref.pth.tar'), m2, tr2)
    assert got['epoch'] == 9 and tr2.flat.step_count == 7 and tr2.lr == 2e-3 and tr2._bn_base == 7
    assert type(got['best_iou']) is float and got['best_iou'] == 0.1 and type(got['best_loss']) is float and got['best_loss'] == 2.0
    for (n, p), (_, p2) in zip(ref.named_parameters(), m2.named_parameters()):
        assert torch.equal(p.detach(), p2.detach()), n
    if hasattr(m2, '_slots'):
        m2._ensure_runtime()                                # the scatter of the concatenation-reading weights
        named, named2 = m.trainer_named_parameters(), m2.trainer_named_parameters()
        pp2 = named2['mask_feature.conv1.weight']
        assert torch.equal(pp2[:, 32:68], ref.mask_feature.conv1.weight.detach()[:, 18:54])
    assert torch.equal(tr2.flat.M[:f.n_used], f.M[:f.n_used]) and torch.equal(tr2.flat.V[:f.n_used], f.V[:f.n_used])
