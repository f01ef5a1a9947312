"""Shims that let the *reference* (read-only, /root/reference) import in the build container.

CONTAINER-ONLY TOOLING.  Used by tests/golden/make_golden.py to generate golden vectors from the
reference itself.  Nothing here (and nothing under /root/reference) is imported by the product
(`cdnet_amd`), by the `-m gpu` tests, by `bench.py` or by `__graft_entry__.smoke()`.

What is shimmed and why:
  * numpy aliases removed in numpy>=1.24 (`np.float/np.int/np.bool`) that the reference still uses
    (getDirectionDiffMap.py:52, SegFix_offset_helper.py:144,205,329, seg_hrnet_rev1.py:376).
  * `torchvision.models.vgg16_bn` - torchvision is not installed; a plain-torch Sequential with the
    standard configuration-D layout (child names '0'..'43') stands in (model_unet_rev1.py:40-41,66-67).
  * `skimage.*`, `cv2`, `lxml`, `SimpleITK`, `albumentations`, `numba`, `tensorboardX`: absent.  The
    few skimage calls on the hot path are replaced by scipy.ndimage equivalents whose definitions
    coincide with skimage's documented behaviour ("skimage-semantics restated"):
      measure.label(x)                 -> ndimage.label(x, structure=ones(3,3))   (8-conn, raster ids)
      morphology.dilation(x, fp)       -> ndimage.grey_dilation(x, footprint=fp)
      morphology.erosion(x, fp)        -> ndimage.grey_erosion(x, footprint=fp)
      morphology.remove_small_objects  -> 4-conn label + bincount threshold (bool input) /
                                          per-label bincount threshold (int input)
  * `.cuda()` -> identity (no GPU here).
"""
import sys
import types
import numpy as np
import torch
import torch.nn as nn
from scipy import ndimage as ndi

REF = '/root/reference'


def _disk(radius, dtype=np.uint8):
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return np.array((X ** 2 + Y ** 2) <= radius ** 2, dtype=dtype)


_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.uint8)


def _dilation(image, selem=None, footprint=None, out=None):
    fp = selem if selem is not None else footprint
    if fp is None:
        fp = _CROSS
    image = np.asarray(image)
    if image.dtype == bool:
        return ndi.binary_dilation(image, structure=fp)
    return ndi.grey_dilation(image, footprint=fp, mode='constant', cval=_min_of(image.dtype))


def _erosion(image, selem=None, footprint=None, out=None):
    fp = selem if selem is not None else footprint
    if fp is None:
        fp = _CROSS
    image = np.asarray(image)
    if image.dtype == bool:
        return ndi.binary_erosion(image, structure=fp, border_value=1)
    return ndi.grey_erosion(image, footprint=fp, mode='constant', cval=_max_of(image.dtype))


def _min_of(dt):
    return np.iinfo(dt).min if np.issubdtype(dt, np.integer) else -np.inf


def _max_of(dt):
    return np.iinfo(dt).max if np.issubdtype(dt, np.integer) else np.inf


def _remove_small_objects(ar, min_size=64, connectivity=1, in_place=False):
    ar = np.asarray(ar)
    out = ar.copy()
    if min_size == 0:
        return out
    if out.dtype == bool:
        st = ndi.generate_binary_structure(ar.ndim, connectivity)
        ccs = np.zeros_like(ar, dtype=np.int32)
        ndi.label(ar, st, output=ccs)
    else:
        ccs = out
    sizes = np.bincount(ccs.ravel())
    too_small = sizes < min_size
    out[too_small[ccs]] = 0
    return out


def _label(x, connectivity=None, background=0, return_num=False):
    x = np.asarray(x)
    assert x.ndim == 2
    if set(np.unique(x)).issubset({0, 1, True, False}):
        lab, n = ndi.label(x != 0, structure=np.ones((3, 3), dtype=int))
        return (lab, n) if return_num else lab
    # integer image (utils.nuclei_accuracy_object_level, test_dam.py:613): skimage labels 8-connected regions of EQUAL value,
    # numbered in raster order of each region's first pixel
    out = np.zeros(x.shape, np.int64)
    firsts = []
    nxt = 0
    for v in np.unique(x):
        if v == background:
            continue
        lab, n = ndi.label(x == v, structure=np.ones((3, 3), dtype=int))
        for k in range(1, n + 1):
            m = lab == k
            nxt += 1
            out[m] = nxt
            firsts.append((int(np.flatnonzero(m.ravel())[0]), nxt))
    remap = np.zeros(nxt + 1, np.int64)
    for new, (_, old) in enumerate(sorted(firsts), start=1):
        remap[old] = new
    out = remap[out]
    return (out, nxt) if return_num else out


def _vgg16_bn_features():
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
    layers, c = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c = v
    return nn.Sequential(*layers)


class _FakeVGG(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = _vgg16_bn_features()


def install():
    """Install every shim into sys.modules / numpy / torch, and put the reference on sys.path."""
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name, typ in (('float', float), ('int', int), ('bool', bool)):
        if not hasattr(np, name):
            setattr(np, name, typ)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # torchvision
    tv_models = mod('torchvision.models', vgg16_bn=lambda pretrained=False, **kw: _FakeVGG())
    mod('torchvision.datasets')
    mod('torchvision.transforms')
    mod('torchvision', models=tv_models, datasets=sys.modules['torchvision.datasets'],
        transforms=sys.modules['torchvision.transforms'])

    # skimage
    selem = mod('skimage.morphology.selem', disk=_disk)
    morph = mod('skimage.morphology', dilation=_dilation, erosion=_erosion, disk=_disk, selem=selem,
                remove_small_objects=_remove_small_objects,
                watershed=_unavailable('skimage.morphology.watershed'))
    measure = mod('skimage.measure', label=_label)
    seg = mod('skimage.segmentation', watershed=_unavailable('skimage.segmentation.watershed'))
    skio = mod('skimage.io', imsave=_unavailable('skimage.io.imsave'), imread=_unavailable('skimage.io.imread'))
    feat = mod('skimage.feature')
    color = mod('skimage.color')
    filt = mod('skimage.filters')
    rank = mod('skimage.filters.rank')
    filt.rank = rank
    mod('skimage', morphology=morph, measure=measure, io=skio, segmentation=seg, feature=feat, color=color,
        filters=filt)

    # misc absent packages (import-only on the paths we run)
    mod('cv2')
    mod('lxml')
    mod('lxml.etree')
    mod('SimpleITK')
    mod('albumentations')
    mod('tensorboardX', SummaryWriter=object)
    mod('imgaug')
    mod('matplotlib')
    mod('matplotlib.pyplot')
    mod('numba', jit=lambda *a, **k: (lambda f: f))

    # no GPU here
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self


def _unavailable(name):
    def f(*a, **k):
        raise RuntimeError(name + ' is not available in this container (shim)')
    return f
