"""Direction class <-> vector <-> angle tables of the centripetal direction encoding.

Host-side mirror of the reference's `data_prepare/SegFix_offset_helper.py` for the symbols the hot path uses:
  label_to_vector_mapping                  (:50-89)    class id -> (d_row, d_col)
  Sobel.kernel(ksize)                      (:97-132)   2 x ksize x ksize "large Sobel" stencil
  DTOffsetHelper.label_to_vector           (:246-261)
  DTOffsetHelper.align_angle               (:311-341)  angle (deg) -> bin centre, bin index
  DTOffsetHelper.angle_to_vector           (:423-450)
  DTOffsetHelper.vector_to_label           (:486-506)
numpy only (these are tiny table / host-side helpers; the per-pixel work runs in the HIP kernels).
"""
import numpy as np

label_to_vector_mapping = {
    4: [[-1, -1], [-1, 1], [1, 1], [1, -1]],
    5: [[0, 0], [-1, -1], [-1, 1], [1, 1], [1, -1]],
    8: [[0, -1], [-1, -1], [-1, 0], [-1, 1], [0, 1], [1, 1], [1, 0], [1, -1]],
    9: [[0, 0], [0, -1], [-1, -1], [-1, 0], [-1, 1], [0, 1], [1, 1], [1, 0], [1, -1]],
    16: [[0, -2], [-1, -2], [-2, -2], [-2, -1], [-2, 0], [-2, 1], [-2, 2], [-1, 2],
         [0, 2], [1, 2], [2, 2], [2, 1], [2, 0], [2, -1], [2, -2], [1, -2]],
    17: [[0, 0], [0, -2], [-1, -2], [-2, -2], [-2, -1], [-2, 0], [-2, 1], [-2, 2], [-1, 2],
         [0, 2], [1, 2], [2, 2], [2, 1], [2, 0], [2, -1], [2, -2], [1, -2]],
}


class DTOffsetConfig:
    direction_classes = 8
    num_classes = 8


class Sobel:
    ksize = 11
    _caches = {}

    @classmethod
    def kernel(cls, ksize=None):
        """float32 [2, 1, k, k]: channel 0 = row ("y") gradient j/(i^2+j^2), channel 1 = column gradient."""
        k = cls.ksize if ksize is None else ksize
        if k not in cls._caches:
            c = (k - 1) // 2
            jj, ii = np.mgrid[-c:c + 1, -c:c + 1].astype(np.float64)
            d = ii * ii + jj * jj
            d[c, c] = 1.0
            ky = (jj / d).astype(np.float32)
            kx = (ii / d).astype(np.float32)
            ky[c, c] = 0.0
            kx[c, c] = 0.0
            cls._caches[k] = np.stack([ky, kx])[:, None]
        return cls._caches[k]


class DTOffsetHelper:

    @staticmethod
    def label_to_vector(labelmap, num_classes=DTOffsetConfig.num_classes):
        """int array [..., H, W] of class ids -> int64 [..., 2, H, W] (d_row, d_col)."""
        lab = np.asarray(labelmap)
        table = np.array(label_to_vector_mapping[num_classes], dtype=np.int64)
        valid = (lab >= 0) & (lab < len(table))
        vec = table[np.where(valid, lab, 0)]
        vec[~valid] = 0
        return np.moveaxis(vec, -1, -3)

    @staticmethod
    def align_angle(angle_map, num_classes=DTOffsetConfig.num_classes, return_tensor=False):
        a = np.asarray(angle_map, dtype=np.float64)
        step = 360.0 / num_classes
        new_angle = np.zeros(a.shape, np.float64)
        index = np.zeros(a.shape, np.int64)
        m = (a <= (-180 + step / 2)) | (a > (180 - step / 2))
        new_angle[m] = -180
        for i in range(1, num_classes):
            mid = -180 + step * i
            m = (a > (mid - step / 2)) & (a <= (mid + step / 2))
            new_angle[m] = mid
            index[m] = i
        return new_angle, index

    @staticmethod
    def angle_to_vector(angle_map, num_classes=DTOffsetConfig.num_classes, return_tensor=False):
        a = np.asarray(angle_map, dtype=np.float64)
        if num_classes is not None:
            a, _ = DTOffsetHelper.align_angle(a, num_classes=num_classes)
        r = np.deg2rad(a)
        return np.stack([np.sin(r), np.cos(r)], axis=-1)

    @staticmethod
    def vector_to_label(vector_map, num_classes=DTOffsetConfig.num_classes, return_tensor=False):
        v = np.asarray(vector_map)
        ang = np.rad2deg(np.arctan2(v[..., 0], v[..., 1]))
        return DTOffsetHelper.align_angle(ang, num_classes=num_classes)[1]
