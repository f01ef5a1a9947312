"""ctypes binding of libcdnet_hip.so (the C ABI of include/cdnet_hip.h).

There is NO fallback: if the library is missing or a symbol is absent the import of the compute path fails
loudly.  PyTorch is used by the callers only for device memory and streams; nothing torch-typed crosses the ABI.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CDNET_LIB_PATH') or os.path.join(_HERE, 'libcdnet_hip.so')      # (the override: A/B builds of tools/build_variant.sh)

_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float

class WgradReduceDesc(C.Structure):          # cdnet_wgrad_reduce_desc (include/cdnet_hip.h)
    _fields_ = [('slab', C.c_void_p), ('dw', C.c_void_p)] + [(n, C.c_int) for n in (
        'ksplit', 'npar', 'ci_blocks', 'co_blocks', 'taps', 'CI', 'CO', 'Csrc_real', 'Cin_real', 'src_coff', 'Cout', 'mode', 'block0', 'blocks')]


# name -> (restype, argtypes); must list every symbol include/cdnet_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    'cdnet_abi_version': (_i, []),
    'cdnet_abi_sizeof': (_sz, [C.c_char_p]),
    'cdnet_last_error': (C.c_char_p, []),
    'cdnet_build_info': (C.c_char_p, []),
    'cdnet_spin': (_i, [_i, _vp]),
    'cdnet_box_copy': (_i, [_vp, _vp, _sz, _vp]),
    'cdnet_box_mfma': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'cdnet_ddm_codes': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    'cdnet_ddm_normalize': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'cdnet_probmaps': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'cdnet_tta_boost_argmax': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cdnet_fuse_sum': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    'cdnet_upsample_bilinear_backward': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'cdnet_s2d_to_nhwc': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'cdnet_fuse_sum_f32': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    'cdnet_upsample_bilinear_backward_f32': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'cdnet_s2d_to_nhwc_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'cdnet_bn_backward_stats': (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'cdnet_bn_backward_apply': (_i, [_vp, _vp, _vp, _vp]),
    'cdnet_bn_backward_finalize': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'cdnet_conv_ws_eligible': (_i, [_vp]),
    'cdnet_grad_sum': (_i, [_vp, _i, _vp, C.c_longlong, _i, _vp, _vp]),
    'cdnet_grad_sum_f32': (_i, [_vp, _i, _vp, C.c_longlong, _i, _vp, _vp]),
    'cdnet_label_pair_histogram': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cdnet_remap_label': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'cdnet_watershed_workspace_bytes': (_sz, [_i, _i, _i]),
    'cdnet_watershed_process': (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    'cdnet_fill_label_process': (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    'cdnet_tile_postproc_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'cdnet_tile_postproc': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _sz] + [_vp] * 10),
    'cdnet_cc_workspace_bytes': (_sz, [_i, _i, _i]),
    'cdnet_cc_chain': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cdnet_conv_packed_weight_elems': (_sz, [_i] * 6),
    'cdnet_pack_conv_weights': (_i, [_vp, _vp] + [_i] * 7 + [_vp]),
    'cdnet_pack_conv_weights_scaled': (_i, [_vp, _vp, _vp] + [_i] * 7 + [_vp]),
    'cdnet_pack_batch_table_bytes': (_sz, [_i]),
    'cdnet_pack_conv_weights_batch': (_i, [_vp, _i, _vp, _sz, _i, _vp]),
    'cdnet_conv_forward': (_i, [_vp, _vp]),
    'cdnet_src_materialize': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'cdnet_input_pack': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'cdnet_input_pack_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'cdnet_bn_fold_eval': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    'cdnet_bn_finalize_train': (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cdnet_dam_head_forward': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'cdnet_final_conv1x1': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'cdnet_bias_grad_workspace_floats': (_sz, [_i]),
    'cdnet_bias_grad': (_i, [_vp, _sz, _i, _vp, _sz, _vp, _vp]),
    'cdnet_bias_grad_f32': (_i, [_vp, _sz, _i, _vp, _sz, _vp, _vp]),
    'cdnet_final_conv1x1_backward_workspace_floats': (_sz, []),
    'cdnet_final_conv1x1_backward': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    'cdnet_conv_wgrad_slab_floats': (_sz, [_i] * 6),
    'cdnet_conv_backward_weight': (_i, [_vp, _i, _i, _i, _vp] + [_i] * 9 + [_vp, _vp, _i, _vp]),
    'cdnet_wgrad_reduce_desc_fill': (_i, [_i] * 9 + [_vp, _vp, _i, _i, _vp]),
    'cdnet_wgrad_reduce_batch': (_i, [_vp, _i, _i, _vp]),
    'cdnet_bn_backward_workspace_floats': (_sz, [_i]),
    'cdnet_bn_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    'cdnet_dam_head_backward_workspace_floats': (_sz, [_i, _i, _i]),
    'cdnet_dam_head_backward': (_i, [_vp] * 7 + [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'cdnet_dam_loss_workspace_floats': (_sz, [_i, _i]),
    'cdnet_dam_loss': (_i, [_vp] * 7 + [_i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    'cdnet_dam_loss_classes_workspace_floats': (_sz, [_i, _i, _i]),
    'cdnet_dam_loss_classes': (_i, [_vp] * 7 + [_i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    'cdnet_dam_val_sums_workspace_floats': (_sz, [_i, _i]),
    'cdnet_dam_val_sums': (_i, [_vp] * 8 + [_i, _i, _i, _vp, _sz, _vp, _vp]),
    'cdnet_dam_val_sums_classes_workspace_floats': (_sz, [_i, _i, _i]),
    'cdnet_dam_val_sums_classes': (_i, [_vp] * 8 + [_i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    'cdnet_adam_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _f, _vp]),
    'cdnet_window_pack': (_i, [_vp] + [_i] * 9 + [_vp, _vp]),
    'cdnet_window_pack_f32': (_i, [_vp] + [_i] * 9 + [_vp, _vp]),
    'cdnet_window_stitch': (_i, [_vp] + [_i] * 9 + [_vp, _vp]),
    'cdnet_label_encoding_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'cdnet_label_encoding_instances_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'cdnet_label_encoding_instances': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cdnet_label_encoding': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


class CdnetHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise if it is not built (python -m cdnet_amd.csrc.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7, the same
    # as /opt/rocm's).  Importing torch FIRST makes the dynamic loader satisfy our NEEDED libamdhip64.so.7 with
    # the copy torch already mapped, so torch's streams / allocations and our launches share one runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CdnetHipError('libcdnet_hip.so is not built: run `python -m cdnet_amd.csrc.build` '
                            '(or __graft_entry__.build()). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.cdnet_abi_version() != 5:
        raise CdnetHipError('ABI version mismatch (libcdnet_hip.so is version %d, this binding 5): rebuild with python -m cdnet_amd.csrc.build' % lib.cdnet_abi_version())
    _lib = lib
    return lib


def check(status, what=''):
    if status != 0:
        msg = load().cdnet_last_error().decode(errors='replace')
        raise CdnetHipError('%s failed (code %d): %s' % (what, status, msg))


def call(name, *args):
    check(getattr(load(), name)(*args), name)


_RAW_STREAM = None


def stream_ptr():
    """The current PyTorch HIP stream as a void* for the ABI (the raw-handle query: torch.cuda.current_stream() builds a Stream object
    through several Python layers, ~4 us - on every one of the ~250 launches of a training step)."""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        get, dev = getattr(torch._C, '_cuda_getCurrentRawStream', None), getattr(torch._C, '_cuda_getDevice', None)
        if get is not None and dev is not None:
            _RAW_STREAM = lambda: get(dev())
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream
    return C.c_void_p(_RAW_STREAM())


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'ABI needs contiguous device tensors'
    return C.c_void_p(t.data_ptr())
