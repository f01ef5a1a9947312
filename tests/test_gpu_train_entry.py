"""train_util_dam.train (reference signature -> ndarray[11]) and the train.py entry point on synthetic data."""
import numpy as np
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_train_function_matches_oracle_iteration():
    """one call of train() over a one-batch loader == one oracle iteration (losses 3e-3, pixel metrics 2e-2 absolute)"""
    import torch
    from cdnet_amd import synth, train_util_dam, utils
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    from cdnet_amd.options import Options
    from oracle import models as om
    from oracle import train as ot
    torch.manual_seed(0)
    ref = om.Unet()
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.normal_(mod.bias, 0, 0.2)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    m.load_state_dict(ref.state_dict())
    m = m.cuda()
    B, S = 2, 64
    lab, dirn, point, weight = synth.train_targets(B, S, S, 21)
    x = torch.from_numpy(synth.det_input((B, 3, S, S), 9))
    target0 = torch.from_numpy(lab.astype(np.int64) * 127 + (lab == 2)).unsqueeze(1)       # {0,127,255}
    sample = (x, torch.from_numpy(weight), target0, torch.from_numpy(point), torch.from_numpy(dirn))
    opt = Options(isTrain=True)
    trainer, _ = utils.get_optimizer(opt, m)
    got = train_util_dam.train([sample], m, trainer, None, 0, opt, None)
    L = ot.train_iteration(ref, ot.make_adam(ref), x, torch.from_numpy(lab), torch.from_numpy(dirn), torch.from_numpy(point),
                           torch.from_numpy(weight))
    assert got.shape == (11,) and got[5] == -1.0
    np.testing.assert_allclose(got[:5], [L['total'], L['dce'], L['wdice'], L['mse'], L['ce']], rtol=3e-3)
    np.testing.assert_allclose(got[6:], L['metrics'], atol=2e-2)


def test_entry_point_synthetic_epochs(tmp_path):
    import os
    import torch
    from cdnet_amd import train
    d = str(tmp_path / 'dam')
    res = train.main(['--synthetic', '3', '--epochs', '2', '--batch-size', '2', '--save-dir', d])
    assert len(res) == 11 and np.isfinite(res).all()
    # the reference's checkpoint layout (train.py:461-480), resumable: epoch, Adam moments and step count come back
    for f in ('checkpoint.pth.tar', 'checkpoint_2.pth.tar', 'checkpoint_best.pth.tar'):
        assert os.path.exists(os.path.join(d, 'checkpoints', f)), f
    ck = torch.load(os.path.join(d, 'checkpoints', 'checkpoint.pth.tar'), map_location='cpu', weights_only=False)
    assert ck['epoch'] == 2 and all(k.startswith('module.') for k in ck['state_dict'])
    assert all(int(s['step']) == 6 for s in ck['optimizer']['state'].values())
    res2 = train.main(['--synthetic', '3', '--epochs', '3', '--batch-size', '2', '--save-dir', d, '--checkpoint-path',
                       os.path.join(d, 'checkpoints', 'checkpoint.pth.tar')])
    assert np.isfinite(res2).all() and res2[0] < res[0] * 1.05           # one more epoch from where it stopped
    ck2 = torch.load(os.path.join(d, 'checkpoints', 'checkpoint.pth.tar'), map_location='cpu', weights_only=False)
    assert ck2['epoch'] == 3 and all(int(s['step']) == 9 for s in ck2['optimizer']['state'].values())
    res_u = train.main(['--synthetic', '2', '--epochs', '1', '--batch-size', '2', '--model-name', 'UNet', '--save-dir', str(tmp_path / 'u')])
    assert len(res_u) == 3 and np.isfinite(res_u).all()
    res_h = train.main(['--synthetic', '2', '--epochs', '1', '--batch-size', '2', '--model-name', 'HRNet18_rev1', '--save-dir', str(tmp_path / 'h')])
    assert len(res_h) == 11 and np.isfinite(res_h).all()
    ckh = torch.load(str(tmp_path / 'h' / 'checkpoints' / 'checkpoint.pth.tar'), map_location='cpu', weights_only=False)
    assert tuple(ckh['state_dict']['module.stage2.0.branches.0.0.conv1.weight'].shape) == (18, 18, 3, 3)


def test_entry_point_reads_dataset_folders(tmp_path, monkeypatch):
    """python -m cdnet_amd.train without --synthetic: ./data/<dataset>/{images,weight_maps,labels}/train through DataFolder and the
    device-side batch pipeline (crop / flips on the host, one cdnet_label_encoding launch per batch)"""
    import torch
    from test_data_folder import make_dataset
    from cdnet_amd import train
    from cdnet_amd.data_folder import DataFolder, TileBatches
    from cdnet_amd.my_transforms_direction import label_encoding_batch
    from cdnet_amd.options import Options
    root = tmp_path / 'data' / Options(isTrain=True).dataset
    dirs = make_dataset(root, n=5, size=(150, 170), seed=3)
    monkeypatch.chdir(tmp_path)
    # the batches carry exactly what the label-encoding kernel makes of the cropped label
    ds = DataFolder(dirs, ['weight.png', 'label.png'], [3, 1, 3])
    tb = TileBatches(ds, {'horizontal_flip': True, 'vertical_flip': True, 'random_crop': 64, 'label_encoding': [3, 2, 1], 'to_tensor': 1},
                     2, torch.device('cuda:0'), seed=1)
    assert len(tb) == 3
    seen = 0
    for img, weight, label, point, direction in tb:
        B = img.shape[0]
        seen += B
        assert img.shape == (B, 3, 64, 64) and img.dtype == torch.float32 and 0 <= float(img.min()) and float(img.max()) <= 1
        assert weight.shape == (B, 1, 64, 64) and weight.dtype == torch.uint8 and int(weight.max()) == 20
        assert label.shape == (B, 1, 64, 64) and set(np.unique(label.cpu().numpy())) <= {0, 127, 255} and len(torch.unique(label)) > 1
        assert point.shape == (B, 64, 64) and point.dtype == torch.float16 and direction.dtype == torch.uint8 and int(direction.max()) <= 8
        assert bool(((direction > 0) <= (label[:, 0] != 0)).all())              # direction classes only on the nuclei
    assert seen == 5
    res = train.main(['--epochs', '2', '--batch-size', '2', '--input-size', '64', '--save-dir', str(tmp_path / 'exp')])
    assert len(res) == 11 and np.isfinite(res).all()


@pytest.mark.parametrize('precision,rtol', [('bf16', 2.5e-2), ('fp32', 2e-4)])
def test_validate_matches_oracle_and_reference_golden(golden, precision, rtol):
    """cdnet_amd.train_util_dam.validate (eval forward + cdnet_dam_val_sums) vs the reference's own 16-value result vector
    (tests/golden/validate.npz; the oracle restatement is pinned to it on the CPU), whole tile and 64/16 sliding windows.
    fp32 precision: 2e-4 (proves the loss mix); bf16 forward on the closed-form weights: losses within 2.5 %."""
    import torch
    import cdnet_amd
    from cdnet_amd import synth, train_util_dam
    old = cdnet_amd.get_precision()
    cdnet_amd.set_precision(precision)
    try:
        _validate_case(golden, rtol)
    finally:
        cdnet_amd.set_precision(old)


def _validate_case(golden, rtol):
    import torch
    from cdnet_amd import synth, train_util_dam
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    from cdnet_amd.options import Options
    from oracle import models as om
    z = golden('validate')
    B, _, H, W, seed = [int(v) for v in z['x_cfg']]
    lab, dirn, point, weight = synth.train_targets(B, H, W, int(z['tgt_cfg'][3]))
    x = torch.from_numpy(synth.det_input((B, 3, H, W), seed))
    ref = om.det_fill(om.Unet())
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3)
    m.load_state_dict(ref.state_dict())
    m = m.cuda()
    opt = Options(isTrain=True).parse([])
    size, ov = [int(v) for v in z['win_cfg']]
    opt.train['input_size'], opt.train['val_overlap'] = size, ov
    target0 = torch.from_numpy(lab.astype(np.int64) * 127 + (lab == 2)).unsqueeze(1)
    sample = (x, torch.from_numpy(weight), target0, torch.from_numpy(point), torch.from_numpy(dirn))
    got = train_util_dam.validate([sample], m, None, opt, None, all_img_test=1)
    assert got.shape == (16,)
    np.testing.assert_allclose(got[:4], z['whole'][:4], rtol=rtol)
    np.testing.assert_allclose(got[4:], z['whole'][4:], atol=5e-3)
    one = tuple(t[:1] for t in sample)
    got = train_util_dam.validate([one], m, None, opt, None, all_img_test=0)
    np.testing.assert_allclose(got[:4], z['split'][:4], rtol=rtol)
    np.testing.assert_allclose(got[4:], z['split'][4:], atol=5e-3)


def test_validate_object_metrics_match_reference_golden(golden):
    """validate(do_object_metric = 1) (train_util_dam.py:588-604): sample 0's mask arg-max through the device fill-holes / remove-small /
    label / dilate chain, then utils.nuclei_accuracy_object_level - against the reference's own 16-value row for the same prescribed
    model outputs (tests/golden/validate_obj.npz: a stub model returning synth.stub_outputs, because no closed-form weight fill predicts
    objects).  Losses 2e-5 (identical logits), object slots 1e-9."""
    import torch
    from cdnet_amd import synth, train_util_dam
    from cdnet_amd.options import Options
    z = golden('validate_obj')
    B, H, W, tseed = [int(v) for v in z['tgt_cfg']]
    lab, dirn, point, weight = synth.train_targets(B, H, W, tseed)
    outs = [torch.from_numpy(o).cuda() for o in synth.stub_outputs(lab, dirn, point, int(z['stub_seed']))]

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dummy = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return tuple(outs)
    opt = Options(isTrain=True).parse([])
    opt.post['min_area'], opt.post['radius'] = int(z['post'][0]), int(z['post'][1])
    x = torch.zeros((B, 3, H, W))
    target0 = torch.from_numpy(lab.astype(np.int64) * 127 + (lab == 2)).unsqueeze(1)
    sample = (x, torch.from_numpy(weight), target0, torch.from_numpy(point), torch.from_numpy(dirn))
    got = train_util_dam.validate([sample], Stub().cuda(), None, opt, None, all_img_test=1, do_object_metric=1)
    want = z['row']
    assert want[9] < 1 and want[10] < 1 and want[15] > 0.1            # the fixture has misses, false positives and real overlaps
    np.testing.assert_allclose(got[:4], want[:4], rtol=2e-5)
    np.testing.assert_allclose(got[4:9], want[4:9], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(got[9:], want[9:], rtol=1e-9, atol=1e-12)
    # and without the object metrics the slots are zero except obj_iou = pixel_iou (:606-609)
    got0 = train_util_dam.validate([sample], Stub().cuda(), None, opt, None, all_img_test=1, do_object_metric=0)
    assert list(got0[9:13]) == [0, 0, 0, 0] and got0[13] == got0[5] and list(got0[14:]) == [0, 0]
