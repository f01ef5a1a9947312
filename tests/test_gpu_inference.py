"""GPU parity of the inference driver: sliding windows (utils.split_forward_dam), TTA views, whole pipeline."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Opt:
    def __init__(self):
        self.model = dict(out_c=3, mseloss=1, direction=1)
        self.direction_classes = 9


class _ToyModel:
    """the position-coding toy network of tests/golden/make_golden.py:gen_split, evaluated on packed bf16 windows
    (torch ops here are test scaffolding, not product code)"""
    training = False

    def forward_packed(self, x16):
        import torch
        x = x16[..., :3].float().permute(0, 3, 1, 2)
        b, c, h, w = x.shape
        yy = torch.arange(h, dtype=torch.float32, device=x.device).view(1, 1, h, 1).expand(b, 1, h, w)
        xx = torch.arange(w, dtype=torch.float32, device=x.device).view(1, 1, 1, w).expand(b, 1, h, w)
        mask = torch.cat([x[:, :1] * 2 + yy * 1e-3, x[:, 1:2] - xx * 1e-3, x[:, 2:3] + 1], 1)
        point = x.sum(1, keepdim=True) + yy * 1e-2 + xx * 1e-4
        direction = torch.cat([x[:, :1] * (k + 1) + (yy + xx) * 1e-3 for k in range(9)], 1)
        return mask.contiguous(), point.contiguous(), direction.contiguous()


def test_split_forward_dam_matches_reference(golden):
    import torch
    from cdnet_amd import synth, utils
    z = golden('split_fwd')
    for name in z['names']:
        size, ov, h, w, seed = [int(v) for v in z['cfg_' + name]]
        x = torch.from_numpy(synth.det_input((1, 3, h, w), seed, bf16_exact=True))
        o, p, d = utils.split_forward_dam(_ToyModel(), x, size, ov, _Opt())
        assert tuple(o.shape) == (1, 3, h, w) and tuple(p.shape) == (1, 1, h, w) and tuple(d.shape) == (1, 9, h, w)
        if name == 'e':
            o, p, d = o[:, :, ::7, ::5], p[:, :, ::7, ::5], d[:, :, ::7, ::5]
        np.testing.assert_allclose(o.cpu().numpy(), z['mask_' + name], rtol=0, atol=2e-6)
        np.testing.assert_allclose(p.cpu().numpy(), z['point_' + name], rtol=0, atol=2e-6)
        np.testing.assert_allclose(d.cpu().numpy(), z['dir_' + name], rtol=0, atol=4e-6)


def test_tta_views_are_pil_transforms():
    """window pack with a view code == the PIL transpose / rotate(90, expand) of test_dam.py:313-385 (= np.flip / rot90)"""
    import torch
    from cdnet_amd import _lib, synth, utils
    H, W = 37, 53
    img = synth.det_input((3, H, W), 4, bf16_exact=True)
    x = torch.from_numpy(img).cuda()
    for xf in range(8):
        v = img
        if xf & 4:
            v = np.rot90(v, k=1, axes=(1, 2))
        if xf & 1:
            v = np.flip(v, 2)
        if xf & 2:
            v = np.flip(v, 1)
        hv, wv = v.shape[1:]
        stride, th, tw, ny, nx = utils.window_grid(hv, wv, 24, 8)
        t = torch.empty((ny * nx, th, tw, 16), dtype=torch.bfloat16, device='cuda')
        _lib.call('cdnet_window_pack', _lib.ptr(x), 3, H, W, xf, th, tw, stride, ny, nx, _lib.ptr(t), _lib.stream_ptr())
        got = t.float().cpu().numpy()
        pad = np.zeros((3, (ny - 1) * stride + th, (nx - 1) * stride + tw), np.float32)
        pad[:, :hv, :wv] = v
        for ky in range(ny):
            for kx in range(nx):
                want = pad[:, ky * stride:ky * stride + th, kx * stride:kx * stride + tw].transpose(1, 2, 0)
                assert np.array_equal(got[ky * nx + kx, :, :, :3], want), (xf, ky, kx)
                assert not got[ky * nx + kx, :, :, 3:].any()


def test_infer_image_pipeline_tta_windows_vs_whole_and_oracle():
    """the device pipeline == per-view network outputs pushed through the CPU oracle post-processing (bit-exact on the
    integer stages given the same logits), for sliding windows + 8-view TTA on a ragged image"""
    import torch
    from cdnet_amd import pipeline, postproc, synth, utils
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    from oracle import postproc as orc
    torch.manual_seed(1)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).cuda().eval()
    H, W = 120, 152
    img = torch.from_numpy(synth.det_input((3, H, W), 8)).cuda()
    r = pipeline.infer_image(m, img, tta=True, all_img_test=0, patch_size=64, overlap=16, want_stages=True)
    # rebuild the reference's arrays on the host from the per-view logits
    views = utils.split_forward_views(m, img, 64, 16, postproc.TTA_XFORMS)
    probs, points, dcms = [], [], []
    for xf, (mask, point, direction) in zip(postproc.TTA_XFORMS, views):
        prob, dcm = postproc.probmaps(mask[None], direction[None])
        p, t, d = prob[0].cpu().numpy(), point.cpu().numpy(), dcm[0].cpu().numpy()
        def unflip(a):                      # test_dam.py:356-441
            if xf & 2: a = np.flip(a, -2)
            if xf & 1: a = np.flip(a, -1)
            if xf & 4: a = np.rot90(a, k=3, axes=(-2, -1))
            return np.ascontiguousarray(a)
        probs.append(unflip(p)); points.append(unflip(t)); dcms.append(unflip(d)[None])
    # (seeded weights and input: no view of this network has a constant direction map - the oracle asserts that, as the reference would
    #  divide by zero in generate_dd_map's normalisation; a failure here is a failure, not a skip)
    want = orc.postprocess_views(np.stack(probs), np.stack(points), np.stack(dcms))
    assert np.array_equal(r['pred'].cpu().numpy(), want['pred'])
    assert np.array_equal(r['final'].cpu().numpy(), want['final'])
    assert r['count'] == want['count']


def test_test_dam_entry_point_with_ground_truth(tmp_path):
    """python -m cdnet_amd.test_dam on a folder of images + labels: train a few steps, save a reference-format checkpoint, run the
    entry point (TTA, whole-image forward, post-processing, instance metrics against the ground truth); identical instance maps
    score 1"""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_data_folder import make_dataset
    from cdnet_amd import checkpoint, test_dam, trainer
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    make_dataset(tmp_path, n=2, size=(96, 112), seed=5, sub='test1')
    torch.manual_seed(0)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).cuda()
    tr = trainer.Trainer(m)
    batch = trainer.synthetic_batch(2, torch.device('cuda:0'), seed=1, H=64, W=64)
    for _ in range(3):
        tr.train_step(*batch)
    ck = checkpoint.save_checkpoint(checkpoint.make_state(m, tr, 0), 0, True, str(tmp_path), 'Main', 0)
    out = str(tmp_path / 'out')
    avg = test_dam.main(['--img-dir', str(tmp_path / 'images' / 'test1'), '--label-dir', str(tmp_path / 'labels' / 'test1'),
                         '--save-dir', out, '--model-path', ck])
    assert os.path.exists(os.path.join(out, 'im0_seg.tiff')) and os.path.exists(os.path.join(out, 'test_results.txt'))
    assert avg is not None and all(np.isfinite(v) and 0.0 <= v <= 1.0 + 1e-9 for v in avg.values()), avg
    gt = test_dam.ground_truth_instances(str(tmp_path / 'labels' / 'test1'), 'im0')
    assert gt is not None and gt.shape == (96, 112) and gt.max() >= 2
    same = test_dam.evaluate_labels(gt, gt)
    for k in ('pixel_iou', 'pixel_F1', 'AJI', 'Dice', 'DQ'):
        assert abs(same[k] - 1.0) < 1e-6, (k, same[k])
    assert abs(same['PQ'] - same['SQ']) < 1e-9 and same['SQ'] > 0.999
    none = test_dam.evaluate_labels(np.zeros_like(gt), gt)
    assert none['AJI'] == 0.0 and none['pixel_recall'] == 0.0


def test_entry_point_sharded_over_two_ranks(tmp_path):
    """SURVEY 8e / test_dam.py:158-760: `cdnet_amd.test_dam.main` as TWO ranks (the one GPU of the box, gloo for the gather) against the
    one-process run on the same five images: every label map bit-identical, one test_results.txt on rank 0 with the same averages, the other
    rank returns None"""
    import json
    import os
    import socket
    import subprocess
    import sys
    import torch
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_data_folder import make_dataset
    from cdnet_amd import checkpoint, test_dam, trainer
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    make_dataset(tmp_path, n=5, size=(96, 112), seed=7, sub='test1')
    torch.manual_seed(0)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).cuda()
    tr = trainer.Trainer(m)
    batch = trainer.synthetic_batch(2, torch.device('cuda:0'), seed=1, H=64, W=64)
    for _ in range(3):
        tr.train_step(*batch)
    ck = checkpoint.save_checkpoint(checkpoint.make_state(m, tr, 0), 0, True, str(tmp_path), 'Main', 0)
    args = ['--img-dir', str(tmp_path / 'images' / 'test1'), '--label-dir', str(tmp_path / 'labels' / 'test1'), '--model-path', ck]
    one = str(tmp_path / 'one')
    want = test_dam.main(args + ['--save-dir', one])
    assert want is not None
    two = str(tmp_path / 'two')
    os.makedirs(two)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_test_dam_world2_worker.py')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, worker, two] + args + ['--save-dir', two], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, o[-2000:] + '\n' + e[-4000:]
    got0, got1 = [json.load(open(os.path.join(two, 'rank%d.json' % r))) for r in range(2)]
    assert got1 is None and got0 is not None
    assert set(got0) == set(want)
    for k in want:
        assert abs(got0[k] - want[k]) < 1e-12, (k, got0[k], want[k])
    for i in range(5):
        a, b = np.asarray(Image.open(os.path.join(one, 'im%d_seg.tiff' % i))), np.asarray(Image.open(os.path.join(two, 'im%d_seg.tiff' % i)))
        assert np.array_equal(a, b), i
    assert open(os.path.join(one, 'test_results.txt')).read() == open(os.path.join(two, 'test_results.txt')).read()


def test_full_size_image_properties():
    """BASELINE config 3 (one 1000x1000 image, 8 TTA views x 25 windows of 256/40, DDM, boost, CC chain): too large for the CPU
    oracle in a test, so size-independent properties - bit-identical repeat runs, instance ids exactly 1..count, no instance
    below min_area (they only grow in the final dilation), windows and views cover every pixel (no NaN / untouched logits), and
    the post-processing chain is idempotent on its own output mask up to the final dilation"""
    import torch
    from cdnet_amd import pipeline, postproc, synth, utils
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    torch.manual_seed(4)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).cuda().eval()
    img = torch.from_numpy(synth.tiles_u8(16, seed=9).astype(np.float32) / 255.0)          # 16 tiles -> one 1000x1000 mosaic
    big = torch.zeros((3, 1024, 1024))
    for k in range(16):
        big[:, (k // 4) * 256:(k // 4 + 1) * 256, (k % 4) * 256:(k % 4 + 1) * 256] = img[k].permute(2, 0, 1)
    image = big[:, :1000, :1000].contiguous().cuda()
    with torch.no_grad():
        r1 = pipeline.infer_image(m, image, tta=True, all_img_test=0, patch_size=256, overlap=40, want_stages=True)
        r2 = pipeline.infer_image(m, image, tta=True, all_img_test=0, patch_size=256, overlap=40)
        (mask, point, direction), = utils.split_forward_views(m, image, 256, 40, (0,))
    final = r1['final']
    assert tuple(final.shape) == (1000, 1000) and torch.equal(final, r2['final']) and r1['count'] == r2['count']
    assert torch.isfinite(mask).all() and torch.isfinite(point).all() and torch.isfinite(direction).all()
    ids = torch.unique(final)
    assert int(ids[0]) == 0 and torch.equal(ids[1:].cpu(), torch.arange(1, r1['count'] + 1, dtype=ids.dtype))
    if r1['count']:
        areas = torch.bincount(final.flatten().long())[1:]
        assert int(areas.min()) >= 20


@pytest.mark.gpu
def test_post_stream_pipelining_is_bit_identical():
    """pipeline.infer_tiles(post_stream=...): the post-processing of batch i runs on a second stream beside the forward of batch i + 1.
    Three different batches queued back to back give exactly the label maps of the serial calls (no buffer of the forward or of the
    post-processing chain is shared between two batches in flight)."""
    import torch
    import cdnet_amd
    from cdnet_amd import pipeline, synth
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    dev = torch.device('cuda:0')
    for prec in ('bf16', 'fp32'):
        cdnet_amd.set_precision(prec)
        try:
            torch.manual_seed(7)
            m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev).eval()
            xs = [torch.from_numpy(synth.tiles_u8(8, seed=40 + k).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous().to(dev)
                  for k in range(3)]
            serial = [pipeline.infer_tiles(m, x, want_prob=True) for x in xs]
            torch.cuda.synchronize()
            post = torch.cuda.Stream()
            for rep in range(3):
                piped = [pipeline.infer_tiles(m, x, post_stream=post, want_prob=True) for x in xs]         # no synchronisation in between
                for r in piped:
                    r['done'].synchronize()
                for a, b in zip(serial, piped):
                    for k in ('final', 'counts', 'pred', 'prob', 'dcm', 'point'):
                        assert torch.equal(a[k], b[k]), (prec, rep, k)
        finally:
            cdnet_amd.set_precision('bf16')


@pytest.mark.gpu
def test_side_stream_is_cached_and_runs_beside_the_compute_stream():
    """cdnet_amd.streams.side_stream: one stream per device (the trainer's weight-gradient stream and the inference pipeline's
    post-processing stream), not the current stream, and - what the probe is for - on another hardware queue: two spin kernels, one per
    stream, take about as long as one."""
    import torch
    from cdnet_amd import _lib, streams
    s = streams.side_stream()
    assert s is streams.side_stream(torch.device('cuda', torch.cuda.current_device()))
    main = torch.cuda.current_stream()
    assert s.cuda_stream != main.cuda_stream

    def timed(both):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if both:
            s.wait_stream(main)
            _lib.call('cdnet_spin', 1000, s.cuda_stream)
        _lib.call('cdnet_spin', 1000, main.cuda_stream)
        if both:
            main.wait_stream(s)
        e1.record(main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    timed(False)
    alone = min(timed(False) for _ in range(3))
    together = min(timed(True) for _ in range(3))
    assert 0.9 < alone < 1.3, alone                       # (the spin kernel waits on the constant 100 MHz counter: 1 000 us asked)
    assert together < 1.5 * alone, (alone, together)
    assert streams.PROBES and streams.PROBES[-1]['probed']
