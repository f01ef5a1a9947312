"""world-size-2 gloo test (CPU) of the multi-GPU exchange step: the bucketed gradient all-reduce of the flat buffer,
followed by the 1/world scaling Adam applies.  The GPU path differs only by the backend (RCCL)."""
import os
import socket
import numpy as np
import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, n, n_used, bucket, out):
    import torch.distributed as dist
    from cdnet_amd.trainer import bucketed_allreduce
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    keep_tail = flat[n_used:].clone()
    bucketed_allreduce(flat, n_used, bucket)
    assert torch.equal(flat[n_used:], keep_tail)          # never-used parameters are not communicated
    if rank == 0:
        out.put(flat.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    n, n_used, bucket = 10007, 9000, 2048                 # several buckets + a ragged last one
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, n_used, bucket, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a = torch.randn(n, generator=torch.Generator().manual_seed(100)).numpy()
    b = torch.randn(n, generator=torch.Generator().manual_seed(101)).numpy()
    np.testing.assert_allclose(got[:n_used], (a + b)[:n_used], rtol=1e-6)
    np.testing.assert_array_equal(got[n_used:], a[n_used:])


def test_flat_state_layout_cpu():
    """parameter flattening: head block first (kernel layout), unused parameters last and outside the stepped range"""
    from cdnet_amd.trainer import FlatState, HEAD_PARAMS
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fs = FlatState(m)
    assert fs.order[:8] == HEAD_PARAMS and fs.n_head == 855
    assert all(n.startswith(m.UNUSED_PREFIXES) for n in fs.order if fs.offsets[n][0] >= fs.n_used)
    assert fs.n_used == sum(p.numel() for n, p in m.named_parameters() if not n.startswith(m.UNUSED_PREFIXES))
    for k, v in m.state_dict().items():                    # values preserved, parameters are views of the flat buffer
        assert torch.equal(v, before[k])
    p = dict(m.named_parameters())['mask_conv.weight']
    off, sz = fs.offsets['mask_conv.weight']
    fs.P[off] = 123.0
    assert float(p.reshape(-1)[0]) == 123.0 and p.grad.data_ptr() == fs.G[off:].data_ptr()


def _overlap_worker(rank, world, port, sizes, n_tail, bucket, out):
    """the schedule of the overlapped all-reduce: tensors complete in reverse (backward) order, buckets are released as
    they become whole"""
    import torch.distributed as dist
    from cdnet_amd.trainer import BucketReducer
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n_used = sum(sizes)
    flat = torch.zeros(n_used + n_tail)
    offs, o = [], 0
    for sz in sizes:
        offs.append(o)
        o += sz
    red = BucketReducer(flat, n_used, bucket, {a: a + sz for a, sz in zip(offs, sizes)})
    g = torch.Generator().manual_seed(200 + rank)
    early = 0
    for a, sz in reversed(list(zip(offs, sizes))):        # "backward": last layer first
        early = red.early                                  # buckets already in flight before this tensor completed
        flat[a:a + sz] = torch.randn(sz, generator=g)
        red.done([a])
    red.finish()
    if rank == 0:
        out.put((flat.numpy().copy(), early, len(red.works)))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_schedule_two_ranks():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    sizes, n_tail, bucket = [855, 1728, 64, 64, 36864, 64, 9000, 5, 300, 2048], 77, 4096
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, sizes, n_tail, bucket, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, early, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = np.zeros(sum(sizes) + n_tail, np.float32)
    for r in range(2):
        g = torch.Generator().manual_seed(200 + r)
        o = sum(sizes)
        for sz in reversed(sizes):
            o -= sz
            want[o:o + sz] += torch.randn(sz, generator=g).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    assert total == -(-sum(sizes) // bucket)               # every bucket reduced exactly once
    assert 0 < early < total                               # most buckets were in flight before the first layer's gradient existed


def test_bucket_boundaries_count_from_the_top():
    """the remainder bucket is the lowest one - the one that completes last (first layers) and whose all-reduce nothing hides"""
    from cdnet_amd.trainer import BucketReducer
    red = BucketReducer(torch.zeros(1200), 1000, 300, {0: 1000})
    assert red.bounds == [1000, 700, 400, 100, 0]
    red = BucketReducer(torch.zeros(900), 900, 300, {0: 900})
    assert red.bounds == [900, 600, 300, 0]
    red = BucketReducer(torch.zeros(10), 10, 300, {0: 10})
    assert red.bounds == [10, 0]


def test_hrnet_parameters_live_in_padded_flat_storage():
    """HRNet18_rev1 computes on zero-padded parameter copies (18/36/72 -> 32/48/80 channels): the module's own parameters are
    strided views of that storage, also after the trainer re-homes it into its flat buffers (host logic only, no kernels)"""
    import torch
    from cdnet_amd import trainer
    from cdnet_amd.models.dam.seg_hrnet_rev1 import HighResolutionNet
    from oracle import hrnet as oh

    class Opt:
        model = {'out_c': 3}
    ref = oh.HighResolutionNet()
    m = HighResolutionNet(Opt())
    m.load_state_dict(ref.state_dict())
    fs = trainer.FlatState(m)
    assert fs.n_head == 855 and fs.P.numel() > sum(p.numel() for p in m.parameters())
    w = m.stage2[0].branches[0][0].conv1.weight
    assert tuple(w.shape) == (18, 18, 3, 3) and w.untyped_storage().data_ptr() == fs.P.untyped_storage().data_ptr()
    assert w.grad is not None and w.grad.untyped_storage().data_ptr() == fs.G.untyped_storage().data_ptr()
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.state_dict().keys())
    assert all(torch.equal(sd[k], v) for k, v in ref.state_dict().items())
    # load_state_dict writes through the views; the two concatenation-reading weights are scattered on the next use
    ref2 = oh.HighResolutionNet()
    m.load_state_dict(ref2.state_dict())
    m._ensure_runtime()
    named = m.trainer_named_parameters()
    pp = named['mask_feature.conv1.weight']
    assert tuple(pp.shape) == (64, 320, 3, 3)          # 32 + 48 + 80 + 144 = 304 carried channels + one chunk of zeros (round 6: 640-byte pixels, an even chunk count)
    assert torch.equal(pp[:, 32:68], ref2.mask_feature.conv1.weight[:, 18:54]) and float(pp[:, 18:32].abs().max()) == 0 and float(pp[:, 304:].abs().max()) == 0
    g = named['stage3.0.branches.2.0.bn1.weight']
    assert tuple(g.shape) == (80,) and torch.equal(g[:72], ref2.stage3[0].branches[2][0].bn1.weight) and float(g[72:].abs().max()) == 0
    # values written into the padded storage (what the fused Adam does) show up in the module's state_dict
    with torch.no_grad():
        pp[:, 32:68] += 1.0
        named['conv1.weight'][:, :3] -= 0.5
    sd = m.state_dict()
    assert torch.allclose(sd['mask_feature.conv1.weight'][:, 18:54], ref2.mask_feature.conv1.weight[:, 18:54] + 1.0)
    assert torch.allclose(sd['conv1.weight'], ref2.conv1.weight - 0.5)


class _TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 8, 3)
        self.bn = torch.nn.BatchNorm2d(8)
        self.unused = torch.nn.Conv2d(1, 2, 1)
    UNUSED_PREFIXES = ('unused.',)


def _sync_worker(rank, world, port, out):
    """replicas built from DIFFERENT seeds (and different BatchNorm statistics / Adam state) are identical after the trainer's
    start-up broadcast; the 11 logging scalars come back as the mean over ranks"""
    import torch.distributed as dist
    from cdnet_amd import trainer
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)
    m = _TinyNet()
    with torch.no_grad():
        m.bn.running_mean.add_(float(rank + 1))
        m.bn.running_var.mul_(2.0 + rank)
    tr = trainer.Trainer(m, world_size=world)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    # a later divergence (e.g. one rank loads a checkpoint) is repaired by an explicit sync
    with torch.no_grad():
        tr.flat.P.add_(float(rank))
        tr.flat.M.add_(float(rank) * 0.5)
    tr.flat.step_count = 7 + rank
    tr.sync_from_rank0()
    means = tr.reduce_scalars(np.arange(11, dtype=np.float64) * (rank + 1))
    # numpy copies travel by value; torch tensors on an mp queue are shared through a descriptor server of this process, which
    # may be gone by the time the parent unpickles them
    out.put((rank, {k: v.numpy().copy() for k, v in sd.items()}, tr.flat.P.numpy().copy(), tr.flat.M.numpy().copy(),
             tr.flat.step_count, np.asarray(means)))
    dist.barrier()
    dist.destroy_process_group()


def test_parameter_broadcast_and_scalar_mean_two_ranks():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r = q.get(timeout=120)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    t = torch.from_numpy
    sd0, P0, M0, step0, mean0 = got[0]
    sd1, P1, M1, step1, mean1 = got[1]
    sd0, sd1 = {k: t(v) for k, v in sd0.items()}, {k: t(v) for k, v in sd1.items()}
    P0, P1, M0, M1 = t(P0), t(P1), t(M0), t(M1)
    torch.manual_seed(1000)
    want = _TinyNet().state_dict()                         # rank 0's initialisation
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
    assert torch.equal(sd0['conv.weight'], want['conv.weight']) and torch.equal(sd0['unused.weight'], want['unused.weight'])
    assert float(sd1['bn.running_mean'][0]) == 1.0 and float(sd1['bn.running_var'][0]) == 2.0       # rank 0's buffers
    assert torch.equal(P0, P1) and torch.equal(M0, M1) and step0 == step1 == 7
    np.testing.assert_allclose(mean0, np.arange(11) * 1.5)
    np.testing.assert_allclose(mean1, mean0)


# ---------------------------------------------------------------------------------------------------------
# inference entry point sharded by image (cdnet_amd/test_dam.py main): the host logic at world size 2
# ---------------------------------------------------------------------------------------------------------
def _fake_rows(names):
    """deterministic per-image metric rows (what evaluate_labels + nuclei_accuracy_object_level leave per image)"""
    rows = {}
    for n in names:
        h = sum(ord(c) * (i + 1) for i, c in enumerate(n))
        rows[n[:-4]] = {'pixel_iou': (h % 97) / 97.0, 'AJI': (h % 89) / 89.0, 'Dice': (h % 83) / 83.0, 'obj_F1': (h % 79) / 79.0}
    return rows


def _shard_worker(rank, world, port, names, save_dir, out):
    import torch.distributed as dist
    from cdnet_amd import test_dam
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = test_dam.shard_names(names, rank, world)
    merged = test_dam.gather_results(_fake_rows(mine), rank, world)
    if rank == 0:
        out.put((mine, test_dam.write_results(merged, save_dir), sorted(merged)))
    else:
        assert merged is None
        out.put((mine, None, None))
    dist.barrier()
    dist.destroy_process_group()


def test_inference_entry_shards_by_image_and_gathers_on_rank0(tmp_path):
    """names[rank::world] partitions the images, the per-image rows of both ranks arrive on rank 0, and test_results.txt (averages + sorted
    rows) is byte for byte the single-process file"""
    from cdnet_amd import test_dam
    names = ['im%02d.png' % i for i in range(7)]
    assert test_dam.shard_names(names, 0, 1) == names
    assert sorted(test_dam.shard_names(names, 0, 2) + test_dam.shard_names(names, 1, 2)) == names
    assert not set(test_dam.shard_names(names, 0, 2)) & set(test_dam.shard_names(names, 1, 2))
    one, two = tmp_path / 'one', tmp_path / 'two'
    one.mkdir(); two.mkdir()
    want = test_dam.write_results(test_dam.gather_results(_fake_rows(names), 0, 1), str(one))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, names, str(two), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    avg = [g[1] for g in got if g[1] is not None]
    assert len(avg) == 1 and avg[0] == want
    assert sorted(n for g in got for n in g[0]) == names
    assert (one / 'test_results.txt').read_bytes() == (two / 'test_results.txt').read_bytes()
