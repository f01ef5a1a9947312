"""cdnet_amd.data_folder: the paired-file dataset with the reference's pairing / re-draw rules (data_folder.py:45-110).
Host logic only (PIL); the device-side batch pipeline is covered by tests/test_gpu_train_entry.py."""
import numpy as np
import pytest
import torch

from cdnet_amd.data_folder import DataFolder, get_imgs_list, img_loader


def make_dataset(root, n=3, size=(70, 90), seed=0, sub='train', empty_first=False):
    """./images/<sub>/im<k>.png, ./weight_maps/<sub>/im<k>_weight.png, ./labels/<sub>/im<k>_label.png (3-class colour label)"""
    from PIL import Image
    from cdnet_amd import synth
    rs = np.random.RandomState(seed)
    dirs = [root / d / sub for d in ('images', 'weight_maps', 'labels')]
    for d in dirs:
        d.mkdir(parents=True, exist_ok=True)
    H, W = size
    for k in range(n):
        inst = synth.ellipse_instances(H, W, 12, rs, 5, 10, 6)
        inside = inst > 0
        ero = synth.erode8(inside)
        lab = np.zeros((H, W, 3), np.uint8)                      # red = inside, green = boundary, blue = background
        lab[..., 0][ero] = 255
        lab[..., 1][inside & ~ero] = 255
        lab[..., 2][~inside] = 255
        if empty_first and k == 0:
            lab[...] = 0
            lab[..., 2] = 255
        Image.fromarray(rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)).save(dirs[0] / ('im%d.png' % k))
        Image.fromarray(np.full((H, W), 20, np.uint8)).save(dirs[1] / ('im%d_weight.png' % k))
        Image.fromarray(lab).save(dirs[2] / ('im%d_label.png' % k))
    return [str(d) for d in dirs]


def test_pairing_and_loading(tmp_path):
    dirs = make_dataset(tmp_path)
    (tmp_path / 'images' / 'train' / 'notes.txt').write_text('x')                 # not an image: ignored
    from PIL import Image
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(tmp_path / 'images' / 'train' / 'orphan.png')     # no companions: dropped
    items = get_imgs_list(dirs, ['weight.png', 'label.png'])
    assert sorted(i[0].split('/')[-1] for i in items) == ['im0.png', 'im1.png', 'im2.png']
    assert all(i[1].endswith('_weight.png') and i[2].endswith('_label.png') for i in items)
    with pytest.raises(RuntimeError):
        get_imgs_list(dirs, ['weight.png'])
    with pytest.raises(RuntimeError):
        DataFolder(dirs, ['weight.png', 'label.png'], [3, 1])
    with pytest.raises(RuntimeError):
        DataFolder([str(tmp_path / 'images')] * 3, ['weight.png', 'label.png'], [3, 1, 3])      # 'Found 0 image pairs'
    ds = DataFolder(dirs, ['weight.png', 'label.png'], [3, 1, 3])
    assert len(ds) == 3
    img, weight, label = ds[0]
    assert img.mode == 'RGB' and weight.mode == 'L' and label.mode == 'RGB' and img.size == (90, 70)
    np.save(tmp_path / 'a.npy', np.arange(12).reshape(3, 4))
    assert np.array_equal(np.asarray(img_loader(str(tmp_path / 'a.npy'), 1)), np.arange(12).reshape(3, 4).astype(np.uint8))


def test_constant_label_is_redrawn(tmp_path):
    """data_folder.py:103-105: the transform is applied again while the label tensor holds a single value"""
    dirs = make_dataset(tmp_path, n=1)
    calls = []

    def transform(sample):
        calls.append(1)
        lab = torch.from_numpy(np.asarray(sample[2])[:, :, 0].astype(np.int64))
        if len(calls) < 3:
            lab = torch.zeros_like(lab)                                          # a crop that missed every nucleus
        return (torch.zeros(3, 4, 4), torch.zeros(1, 4, 4), lab)
    ds = DataFolder(dirs, ['weight.png', 'label.png'], [3, 1, 3], data_transform=transform)
    out = ds[0]
    assert len(calls) == 3 and len(torch.unique(out[2])) > 1


def test_dataset_layout_follows_the_reference(tmp_path):
    """train.py:216-288: validation = 1 reads instance-level targets from <label_dir>/<split>_ins ('label.mat' for CPM2017 / MultiOrgan,
    'label.npy' otherwise; CPM2017 validates on its test split), validation = 0 three-class 'label.png' from <label_dir>/<split>; when
    only the other layout is on disk it is taken instead"""
    import os
    import types
    from cdnet_amd import train
    d = str(tmp_path)
    for sub in ('images/train', 'weight_maps/train', 'labels/train_ins', 'labels/val', 'labels/test_ins'):
        os.makedirs(os.path.join(d, sub))
    opt = types.SimpleNamespace(dataset='MoNuSeg', train={'img_dir': d + '/images', 'weight_map_dir': d + '/weight_maps', 'label_dir': d + '/labels', 'validation': 1})
    dirs, fix = train._dataset_layout(opt, 'train')
    assert dirs == [d + '/images/train', d + '/weight_maps/train', d + '/labels/train_ins'] and fix == ['weight.png', 'label.npy']
    dirs, fix = train._dataset_layout(opt, 'val')                      # only the png layout exists for this split
    assert dirs[2] == d + '/labels/val' and fix == ['weight.png', 'label.png']
    opt.train['validation'] = 0
    dirs, fix = train._dataset_layout(opt, 'train')                    # only the instance layout exists for this split
    assert dirs[2] == d + '/labels/train_ins' and fix[1] == 'label.npy'
    opt.dataset, opt.train['validation'] = 'CPM2017', 1
    dirs, fix = train._dataset_layout(opt, 'val')
    assert dirs == [d + '/images/test', d + '/weight_maps/test', d + '/labels/test_ins'] and fix == ['weight.png', 'label.mat']


def test_validation_transform_keeps_the_whole_image(tmp_path):
    """options.py:358: the validation transform has no crop - every epoch scores the same, whole, unpadded images (a random crop per
    epoch made val_iou / val_F1, which pick checkpoint_best and drive early stopping, a noisy subset metric)"""
    from cdnet_amd.data_folder import TileBatches
    dirs = make_dataset(tmp_path, n=2, size=(70, 90), empty_first=True)
    ds = DataFolder(dirs, ['weight.png', 'label.png'], [3, 1, 3])
    tb = TileBatches(ds, {'to_tensor': 1}, 1, 'cpu', seed=3, shuffle=False)
    assert len(tb) == 2
    for k in range(2):
        a, b = tb._draw(*tb.items[k]), tb._draw(*tb.items[k])
        assert a[0].shape == (70, 90, 3) and a[2].shape[:2] == (70, 90)               # not cropped, not padded to a square
        assert all(np.array_equal(x, y) for x, y in zip(a, b))                        # the same pixels every epoch
        assert np.array_equal(a[0], np.asarray(ds.load(k)[0]))
    # the training transform still crops (and pads what is smaller than the crop)
    tr = TileBatches(ds, {'random_crop': 64, 'horizontal_flip': True, 'to_tensor': 1}, 1, 'cpu', seed=3)
    assert tr._draw(*tr.items[1])[0].shape == (64, 64, 3)
