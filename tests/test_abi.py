"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every declared symbol."""
import os
import re
import numpy as np
from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'cdnet_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(cdnet_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from cdnet_amd.csrc import build
    build.build()
    from cdnet_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n
        assert n in _lib.SIGNATURES, 'ctypes signature missing for ' + n
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.cdnet_abi_version() == 5
    assert b'gfx950' in lib.cdnet_build_info()


def test_ctypes_mirrors_have_the_library_struct_sizes():
    """every argument struct the Python host side mirrors with ctypes has the size the library was compiled with (a field added on one
    side only would shift every later field silently)"""
    from cdnet_amd import _lib, engine, runtime, trainer
    lib = _lib.load()
    mirrors = {'cdnet_conv_src': engine.ConvSrc, 'cdnet_conv_args': engine.ConvArgs, 'cdnet_pack_job': engine.PackJob,
               'cdnet_head_feat': runtime.HeadFeat, 'cdnet_wgrad_reduce_desc': _lib.WgradReduceDesc, 'cdnet_grad_in': trainer.GradIn,
               'cdnet_bn_bwd_args': trainer.BnBwdArgs, 'cdnet_fuse_term': runtime.FuseTerm, 'cdnet_grad_term': trainer.GradTerm}
    import ctypes
    for name, cls in mirrors.items():
        assert lib.cdnet_abi_sizeof(name.encode()) == ctypes.sizeof(cls), name
    assert lib.cdnet_abi_sizeof(b'no_such_struct') == 0


def test_argument_validation_without_gpu():
    """Argument checks return an error code before any HIP call (no GPU needed)."""
    from cdnet_amd import _lib
    lib = _lib.load()
    assert lib.cdnet_cc_workspace_bytes(0, 10, 10) == 0
    assert lib.cdnet_cc_workspace_bytes(2, 256, 256) >= 2 * 256 * 256 * 10
    rc = lib.cdnet_ddm_codes(None, 1, 8, 8, 9, None, 8, 0, None, None, None)
    assert rc == 1 and b'null pointer' in lib.cdnet_last_error()


def test_host_lut_matches_oracle():
    from cdnet_amd import postproc
    from oracle import postproc as orc
    for classes in (5, 9, 17):
        assert np.array_equal(postproc.ddm_lut(classes), orc.ddm_lut(classes))


def test_segfix_helper_tables(golden):
    """align_angle / vector_to_label bins (SegFix_offset_helper.py:311-341) on the bin edges."""
    from cdnet_amd.data_prepare.SegFix_offset_helper import DTOffsetHelper, Sobel
    ang = np.array([-180.0, -157.5, -157.4, -135.0, -22.5, 0.0, 22.5, 22.6, 157.5, 157.6, 180.0])
    _, idx = DTOffsetHelper.align_angle(ang, 8)
    assert idx.tolist() == [0, 0, 1, 1, 3, 4, 4, 5, 7, 0, 0]
    k = Sobel.kernel(11)
    assert k.shape == (2, 1, 11, 11) and k[0, 0, 5, 5] == 0
    assert np.isclose(k[0, 0, 6, 5], 1.0) and np.isclose(k[1, 0, 5, 7], 0.5) and np.isclose(k[0, 0, 3, 4], -2 / 5.0)
