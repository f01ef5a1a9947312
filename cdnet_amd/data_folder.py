"""Paired-file dataset of the reference (data_folder.py:20-110) and the device-side batch pipeline that feeds the train step.

  img_loader(path, num_channels), get_imgs_list(dir_list, post_fix), DataFolder(dir_list, post_fix, num_channels,
  data_transform)                     same names / arguments / pairing rule as the reference (host side, PIL)
  TileBatches(dataset, opt, device)   the MI355X input pipeline: crops / flips on the host (cheap), then ONE
                                      cdnet_label_encoding launch per batch makes the 3-class label, the centre-point map and
                                      the centripetal direction classes (the reference does this per sample in DataLoader
                                      workers, seconds per sample: SURVEY 8f.1)

Samples leave TileBatches in the layout train_util_dam.train expects: (input f32 [B,3,H,W], weight u8 [B,1,H,W],
label i64 [B,1,H,W] in {0,127,255}, point f16 [B,H,W], direction u8 [B,H,W]).  The photometric / elastic augmentations of
the reference (albumentations, PIL RNG streams: my_transforms_direction.py:38-630) are not reproduced - their random streams
cannot be pinned ("parity unpinned"); random crop and flips are, with a numpy RandomState."""
import os

import numpy as np
import torch

IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp')


def is_image_file(filename):
    return filename.lower().endswith(IMG_EXTENSIONS)


def img_loader(path, num_channels):
    """PIL image of a png / jpg, or of the 'inst_map' of a .mat / the array of a .npy (data_folder.py:20-41)"""
    from PIL import Image
    if path.endswith('.mat'):
        import scipy.io as scio
        arr = scio.loadmat(path)['inst_map']
        return Image.fromarray(arr.astype(np.uint8)) if num_channels == 1 else arr
    if path.endswith('.npy'):
        return Image.fromarray(np.load(path).astype(np.uint8))
    img = Image.open(path)
    return img if num_channels == 1 else img.convert('RGB')


def get_imgs_list(dir_list, post_fix=None):
    """[(img, img_<post_fix[0]>, img_<post_fix[1]>, ...)] for every image of dir_list[0] that has all its companions
    (data_folder.py:45-73): companion i lives in dir_list[i] and is called '<image stem>_<post_fix[i-1]>'"""
    if len(dir_list) == 0:
        return []
    if len(dir_list) != len(post_fix) + 1:
        raise RuntimeError('Should specify the postfix of each img type except the first input.')
    present = [set(os.listdir(d)) for d in dir_list]
    out = []
    for name in os.listdir(dir_list[0]):
        if not is_image_file(name):
            continue
        stem = os.path.splitext(name)[0]
        item = [os.path.join(dir_list[0], name)]
        for d, names, pf in zip(dir_list[1:], present[1:], post_fix):
            companion = '{:s}_{:s}'.format(stem, pf)
            if companion in names:
                item.append(os.path.join(d, companion))
        if len(item) == len(dir_list):
            out.append(tuple(item))
    return out


class DataFolder(torch.utils.data.Dataset):
    """one input image, one weight map, one target image per item (data_folder.py:78-110).  With a `data_transform` the item
    is transformed, and re-drawn while its label (third tensor) holds a single value (:103-105)."""

    def __init__(self, dir_list, post_fix, num_channels, data_transform=None, loader=img_loader):
        super().__init__()
        if len(dir_list) != len(post_fix) + 1:
            raise RuntimeError('Length of dir_list is different from length of post_fix + 1.')
        if len(dir_list) != len(num_channels):
            raise RuntimeError('Length of dir_list is different from length of num_channels.')
        self.img_list = get_imgs_list(dir_list, post_fix)
        if len(self.img_list) == 0:
            raise RuntimeError('Found 0 image pairs in given directories.')
        self.data_transform, self.num_channels, self.loader = data_transform, num_channels, loader

    def load(self, index):
        return [self.loader(p, c) for p, c in zip(self.img_list[index], self.num_channels)]

    def __getitem__(self, index):
        sample = self.load(index)
        if self.data_transform is None:
            return sample
        out = self.data_transform(sample)
        while len(torch.unique(out[2])) <= 1:
            out = self.data_transform(sample)
        return out

    def __len__(self):
        return len(self.img_list)


class TileBatches:
    """Epoch iterator over a DataFolder for the device train step.  Per item: random crop of `input_size` (zero / 0-weight padded
    when the image is smaller), random horizontal / vertical flip when the transform dict asks for them; per batch: one
    cdnet_label_encoding launch.  A crop whose label is constant is re-drawn (the DataFolder rule).  A transform without crop and
    flips (the validation transform) yields every image whole and unpadded, in a fixed order with shuffle=False (batch size 1
    unless the images share one size)."""

    SKIPPED = ('random_color', 'random_elastic', 'random_chooseAug', 'random_resize', 'random_affine', 'random_rotation')

    def __init__(self, dataset, transform, batch_size, device, seed=0, shuffle=True, drop_last=False, logger=None):
        self.ds, self.B, self.dev = dataset, batch_size, device
        self.rs = np.random.RandomState(seed)
        self.shuffle, self.drop_last = shuffle, drop_last
        self.crop = transform.get('random_crop')
        self.hflip, self.vflip = bool(transform.get('horizontal_flip')), bool(transform.get('vertical_flip'))
        self.normalize = transform.get('normalize')
        skipped = [k for k in transform if k in self.SKIPPED]
        if skipped and logger is not None:
            logger.info('input pipeline: augmentations {} are not reproduced (parity unpinned); crop / flips / label encoding are'.format(skipped))
        unknown = [k for k in transform if k not in self.SKIPPED + ('random_crop', 'horizontal_flip', 'vertical_flip', 'label_encoding',
                                                                     'to_tensor', 'normalize')]
        if unknown:
            raise NotImplementedError('transforms {} are outside the accelerated input pipeline'.format(unknown))
        self.items = [[np.asarray(im) for im in dataset.load(i)] for i in range(len(dataset))]      # decoded once, kept on the host
        # label kind, as LabelEncoding decides it per image (`label_level_len > 2`, my_transforms_direction.py:713-720): instance-level
        # labels (the reference's <label_dir>/train_ins layout) take the instance branch of the device transform
        kinds = {len(np.unique(it[2] if it[2].ndim == 2 else it[2][:, :, 0])) > 2 for it in self.items}
        if len(kinds) > 1:
            raise ValueError('the label directory mixes instance-level and 3-class label images')
        self.instance_labels = bool(kinds and kinds.pop())

    def __len__(self):
        n = len(self.items)
        return n // self.B if self.drop_last else -(-n // self.B)

    def _draw(self, img, weight, label):
        H, W = label.shape[:2]
        if not self.crop and not self.hflip and not self.vflip:
            # the reference's validation transform {label_encoding, to_tensor, normalize} (options.py:358): the whole image, untouched -
            # validate() then takes it whole or through split_forward_dam (train_util_dam.py:474)
            return [np.ascontiguousarray(a) for a in (img, weight, label)]
        s = self.crop or max(H, W)
        for _ in range(50):
            y0 = self.rs.randint(0, max(H - s, 0) + 1)
            x0 = self.rs.randint(0, max(W - s, 0) + 1)
            sl = (slice(y0, y0 + s), slice(x0, x0 + s))
            out = [a[sl] for a in (img, weight, label)]
            if out[2].shape[0] < s or out[2].shape[1] < s:
                out = [np.pad(a, ((0, s - a.shape[0]), (0, s - a.shape[1])) + ((0, 0),) * (a.ndim - 2)) for a in out]
            if self.hflip and self.rs.rand() < 0.5:
                out = [a[:, ::-1] for a in out]
            if self.vflip and self.rs.rand() < 0.5:
                out = [a[::-1] for a in out]
            lab0 = out[2] if out[2].ndim == 2 else out[2][:, :, 0]
            if len(np.unique(lab0)) > 1:
                break
        return [np.ascontiguousarray(a) for a in out]

    def __iter__(self):
        from .my_transforms_direction import label_encoding_batch
        order = self.rs.permutation(len(self.items)) if self.shuffle else np.arange(len(self.items))
        for b in range(len(self)):
            idx = order[b * self.B:(b + 1) * self.B]
            crops = [self._draw(*self.items[i]) for i in idx]
            img = torch.from_numpy(np.stack([c[0] for c in crops])).to(self.dev).permute(0, 3, 1, 2).float().div(255)
            if self.normalize:
                mean, std = self.normalize
                img = (img - torch.tensor(mean, device=self.dev).view(1, 3, 1, 1)) / torch.tensor(std, device=self.dev).view(1, 3, 1, 1)
            weight = torch.from_numpy(np.stack([c[1] if c[1].ndim == 2 else c[1][:, :, 0] for c in crops])).to(self.dev)
            lab0 = torch.from_numpy(np.stack([c[2] if c[2].ndim == 2 else c[2][:, :, 0] for c in crops])).to(self.dev)
            if self.instance_labels:
                from .my_transforms_direction import label_encoding_instances_batch
                l3, point, direction = label_encoding_instances_batch(lab0.to(torch.int32).contiguous())
            else:
                l3, point, direction = label_encoding_batch(lab0.to(torch.uint8).contiguous())
            yield img.contiguous(), weight.to(torch.uint8).unsqueeze(1), l3.to(torch.int64).unsqueeze(1), point, direction
