"""Deterministic synthetic inputs (SURVEY.md section 8d recipe) shared by bench.py, the tests and the golden
generator.  Pure numpy (RandomState streams are stable across numpy versions); no reference code, no oracle.

  tiles             uint8 RGB i.i.d. uniform, RandomState(2022) (2022 = reference default seed, options.py:78)
  instance labels   random non-overlapping ellipses
  network outputs   one-hot(3-class label) smoothed + noise, Gaussian centre heat-map, centripetal class map
"""
import math
import zlib
import numpy as np


# ---------------------------------------------------------------------------------------------------------
# closed-form, name-keyed parameter fill: reproducible weights without RNG state or weight blobs
# ---------------------------------------------------------------------------------------------------------
def det_fill_array(name, shape, kind):
    """value[i] = amp * (sin(0.37 i + phase) + 0.5 sin(...)), phase keyed by crc32(name); float32."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = zlib.crc32(name.encode()) & 0xffffffff
    phase = (h % 10007) * 0.001
    i = np.arange(n, dtype=np.float64)
    s = np.sin(0.37 * i + phase) + 0.5 * np.sin(0.011 * i * (1 + (h % 7)) + 2.0 * phase)
    if kind == 'weight':
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        v = s * math.sqrt(2.0 / fan_in) * 1.1
    elif kind == 'bn_weight':
        v = 1.0 + 0.2 * s
    elif kind in ('bn_bias', 'bias'):
        v = 0.1 * s
    elif kind == 'running_mean':
        v = 0.05 * s
    elif kind == 'running_var':
        v = 1.0 + 0.3 * np.abs(s)
    else:
        raise ValueError(kind)
    return v.reshape(shape).astype(np.float32)


def det_fill_state_dict(sd, bn_owners):
    """Fill a {name: torch.Tensor} state dict in place. `bn_owners`: set of module names that are BatchNorm."""
    import torch
    for k, v in sd.items():
        if k.endswith('num_batches_tracked'):
            continue
        owner, leaf = k.rsplit('.', 1)
        if owner in bn_owners:
            kind = {'weight': 'bn_weight', 'bias': 'bn_bias', 'running_mean': 'running_mean',
                    'running_var': 'running_var'}[leaf]
        else:
            kind = 'weight' if leaf == 'weight' else 'bias'
        v.copy_(torch.from_numpy(det_fill_array(k, tuple(v.shape), kind)))


def to_bf16_exact(x):
    """round float32 to the nearest bfloat16-representable value (ties to even), still stored as float32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def det_input(shape, seed, f16_exact=False, bf16_exact=False):
    x = np.random.RandomState(seed).rand(*shape).astype(np.float32)
    if f16_exact:
        x = x.astype(np.float16).astype(np.float32)
    if bf16_exact:
        x = to_bf16_exact(x)
    return x


def tiles_u8(n, h=256, w=256, seed=2022):
    return np.random.RandomState(seed).randint(0, 256, size=(n, h, w, 3), dtype=np.uint8)


# ---------------------------------------------------------------------------------------------------------
def ellipse_instances(H, W, n, rs, rmin=5, rmax=12, margin=10):
    """n attempts at random ellipses; later ones are skipped when they overlap an earlier one."""
    inst = np.zeros((H, W), np.int32)
    yy, xx = np.mgrid[:H, :W]
    k = 0
    for _ in range(n):
        cy, cx = rs.randint(margin, H - margin), rs.randint(margin, W - margin)
        a, b = rs.randint(rmin, rmax), rs.randint(rmin, rmax)
        th = rs.rand() * np.pi
        y0, y1 = max(0, cy - rmax - 1), min(H, cy + rmax + 2)
        x0, x1 = max(0, cx - rmax - 1), min(W, cx + rmax + 2)
        sy, sx = yy[y0:y1, x0:x1], xx[y0:y1, x0:x1]
        u = (sy - cy) * np.cos(th) + (sx - cx) * np.sin(th)
        v = -(sy - cy) * np.sin(th) + (sx - cx) * np.cos(th)
        m = (u / a) ** 2 + (v / b) ** 2 <= 1
        sub = inst[y0:y1, x0:x1]
        if (sub[m] != 0).any():
            continue
        k += 1
        sub[m] = k
    return inst


def _gauss1d(sigma, radius):
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    return k / k.sum()


def gaussian_blur(img, sigma=2.0, truncate=4.0):
    """separable Gaussian, reflect boundary (the scipy.ndimage.gaussian_filter definition)"""
    r = int(truncate * sigma + 0.5)
    k = _gauss1d(sigma, r)
    out = img.astype(np.float64)
    for ax in (0, 1):
        pad = [(0, 0), (0, 0)]
        pad[ax] = (r, r)
        p = np.pad(out, pad, mode='symmetric')
        acc = np.zeros_like(out)
        for i, kv in enumerate(k):
            sl = [slice(None), slice(None)]
            sl[ax] = slice(i, i + out.shape[ax])
            acc += kv * p[tuple(sl)]
        out = acc
    return out


def centroid_direction(inst):
    """direction class = quantised angle from pixel toward the instance centroid (+1), background 0;
    also the rounded centroids."""
    H, W = inst.shape
    yy, xx = np.mgrid[:H, :W]
    dcm = np.zeros((H, W), np.uint8)
    cents = []
    for k in range(1, int(inst.max()) + 1):
        m = inst == k
        if not m.any():
            continue
        cy, cx = yy[m].mean(), xx[m].mean()
        ang = np.degrees(np.arctan2(cy - yy[m], cx - xx[m]))
        cls = (np.floor((ang + 180 + 22.5) / 45).astype(int) % 8) + 1
        dcm[m] = cls
        cents.append((int(round(cy)), int(round(cx))))
    return dcm, cents


def erode8(mask):
    """binary erosion by the 3x3 square, outside = background"""
    p = np.pad(mask, 1, mode='constant')
    out = np.ones_like(mask, dtype=bool)
    for dy in range(3):
        for dx in range(3):
            out &= p[dy:dy + mask.shape[0], dx:dx + mask.shape[1]]
    return out


def postproc_case(H, W, n, seed, views=8):
    """Synthetic 8-view network outputs around an ellipse ground truth, perturbed per view.
    Returns probs f32 [V,3,H,W], points f32 [V,1,H,W], dcms u8 [V,1,H,W] (already un-flipped/un-rotated)."""
    rs = np.random.RandomState(seed)
    inst = ellipse_instances(H, W, n, rs, 5, 12, 8)
    inside = inst > 0
    ero = erode8(inside)
    lab3 = np.zeros((H, W), np.int64)
    lab3[ero] = 1
    lab3[inside & ~ero] = 2
    dcm, cents = centroid_direction(inst)
    pt = np.zeros((H, W), np.float64)
    for cy, cx in cents:
        pt[cy, cx] = 255.0
    pt = gaussian_blur(pt, 2.0)
    probs, points, dcms = [], [], []
    for v in range(views):
        eps = 0.05
        p = np.full((3, H, W), eps / 2, np.float32)
        lab_v = lab3.copy()
        flip = rs.rand(H, W) < 0.03
        lab_v[flip] = rs.randint(0, 3, size=int(flip.sum()))
        for c in range(3):
            p[c][lab_v == c] = 1 - eps
        p += rs.rand(3, H, W).astype(np.float32) * np.float32(0.2)
        p /= p.sum(0, keepdims=True)
        probs.append(p.astype(np.float32))
        points.append((pt + rs.randn(H, W) * 0.05).astype(np.float32)[None])
        d = dcm.copy()
        noise = (rs.rand(H, W) < 0.02) & (dcm > 0)
        d[noise] = rs.randint(1, 9, size=int(noise.sum()))
        dcms.append(d[None])
    return np.stack(probs), np.stack(points), np.stack(dcms)


def train_targets(B, H, W, seed, n=6):
    """labels {0,1,2}, direction 0..8, point map f16, weight map (png scale, /20 later) - fixed recipe."""
    rs = np.random.RandomState(seed)
    lab = np.zeros((B, H, W), np.uint8)
    dirn = np.zeros((B, H, W), np.uint8)
    point = np.zeros((B, H, W), np.float64)
    yy, xx = np.mgrid[:H, :W]
    for b in range(B):
        for _ in range(n):
            cy, cx = rs.randint(6, H - 6), rs.randint(6, W - 6)
            r = rs.randint(4, 9)
            d2 = (yy - cy) ** 2 + (xx - cx) ** 2
            inside = d2 <= (r - 1) ** 2
            ring = (d2 <= r ** 2) & ~inside
            lab[b][inside] = 1
            lab[b][ring] = 2
            ang = np.degrees(np.arctan2(cy - yy, cx - xx))
            cls = (np.floor((ang + 180 + 22.5) / 45).astype(int) % 8) + 1
            dirn[b][inside] = cls[inside]
            point[b, cy, cx] = 255.0
        point[b] = gaussian_blur(point[b], 2.0)
    weight = rs.randint(10, 60, size=(B, 1, H, W)).astype(np.uint8)
    return lab, dirn, point.astype(np.float16), weight


def remap_direction(dirn, classes):
    """the 0..8 direction classes of train_targets as a 4+1 / 8+1 / 16+1 class map (options.py:45): 8 -> 4 directions by pairing
    neighbours, 8 -> 16 by splitting every direction on the pixel checkerboard - fixed recipe for fixtures and tests."""
    d = np.asarray(dirn).astype(np.int64)
    if classes == 9:
        return d.astype(np.uint8)
    if classes == 5:
        return np.where(d > 0, (d - 1) // 2 + 1, 0).astype(np.uint8)
    assert classes == 17, classes
    yy, xx = np.mgrid[:d.shape[-2], :d.shape[-1]]
    return np.where(d > 0, 2 * d - ((yy + xx) & 1), 0).astype(np.uint8)


def stub_outputs(lab, dirn, point, seed, classes=9):
    """(mask, point, direction) logits of a model that almost reproduces its targets: one-hot of the label shifted by one row, every
    third instance dropped (missed objects), one spurious disk per sample (false positive), Gaussian noise - inputs for metric branches
    that need object-like predictions (validate's do_object_metric, tests/golden/validate_obj.npz).  float32 NCHW."""
    from scipy import ndimage as ndi
    rs = np.random.RandomState(seed)
    lab = np.roll(np.asarray(lab).astype(np.int64), 1, axis=1)
    B, H, W = lab.shape
    yy, xx = np.mgrid[:H, :W]
    for b in range(B):
        comp, n = ndi.label(lab[b] > 0, structure=np.ones((3, 3), dtype=int))
        for k in range(1, n + 1, 3):
            lab[b][comp == k] = 0
        free = np.argwhere(ndi.binary_dilation(lab[b] > 0, iterations=8) == 0)
        if len(free):
            cy, cx = free[rs.randint(len(free))]
            d2 = (yy - cy) ** 2 + (xx - cx) ** 2
            lab[b][d2 <= 25] = 2
            lab[b][d2 <= 16] = 1
    mask = (6.0 * (lab[:, None] == np.arange(3)[None, :, None, None]) + rs.randn(B, 3, H, W)).astype(np.float32)
    d = np.asarray(dirn).astype(np.int64)
    direction = (4.0 * (d[:, None] == np.arange(classes)[None, :, None, None]) + rs.randn(B, classes, H, W)).astype(np.float32)
    pt = (np.asarray(point).astype(np.float32) / 255.0 + 0.05 * rs.randn(B, H, W).astype(np.float32))[:, None]
    return mask, pt.astype(np.float32), direction


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c & 0xffffffff)


# ---------------------------------------------------------------------------------------------------------
# learnable synthetic histology: the image actually shows its nuclei, so a network trained on it produces
# nucleus-like outputs (used by the label-level parity gate, tests/test_gpu_label_gate.py)
# ---------------------------------------------------------------------------------------------------------
def render_nuclei(inst, rs):
    """RGB float32 [3,H,W] in [0,1]: pink background, darker violet nuclei with per-instance shade, blurred edges, noise."""
    H, W = inst.shape
    inside = (inst > 0).astype(np.float64)
    shade = np.concatenate([[0.0], 0.75 + 0.5 * rs.rand(int(inst.max()))])[inst]          # per-instance stain strength
    a = gaussian_blur(inside * shade, 1.0)
    bg = np.array([0.86, 0.70, 0.82])[:, None, None]
    fg = np.array([0.38, 0.24, 0.55])[:, None, None]
    img = bg + (fg - bg) * np.clip(a, 0, 1.2)[None]
    img += gaussian_blur(rs.randn(H, W), 3.0)[None] * 0.15                                   # slow stain variation
    img += rs.randn(3, H, W) * 0.03
    return np.clip(img, 0.0, 1.0).astype(np.float32)


def nuclei_batch(B, H, W, seed, n=60, rmin=5, rmax=12):
    """B rendered tiles with their training targets: x f32 [B,3,H,W], label u8 {0,1,2}, direction u8 0..8, point f16,
    weight u8 (constant 20), instance maps i32."""
    rs = np.random.RandomState(seed)
    x = np.zeros((B, 3, H, W), np.float32)
    lab = np.zeros((B, H, W), np.uint8)
    dirn = np.zeros((B, H, W), np.uint8)
    point = np.zeros((B, H, W), np.float16)
    insts = np.zeros((B, H, W), np.int32)
    for b in range(B):
        inst = ellipse_instances(H, W, n, rs, rmin, rmax, 10)
        insts[b] = inst
        x[b] = render_nuclei(inst, rs)
        inside = inst > 0
        ero = erode8(inside)
        lab[b][ero] = 1
        lab[b][inside & ~ero] = 2
        d, cents = centroid_direction(inst)
        d[~ero] = 0
        dirn[b] = d
        pt = np.zeros((H, W), np.float64)
        for cy, cx in cents:
            pt[cy, cx] = 255.0
        point[b] = gaussian_blur(pt, 2.0).astype(np.float16)
    weight = np.full((B, H, W), 20, np.uint8)
    return x, lab, dirn, point, weight, insts
