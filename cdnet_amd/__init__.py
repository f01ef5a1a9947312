"""cdnet_amd - MI355X-native (gfx950) implementation of CDNet's data-parallel hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); every compute step of
the path runs in hand-written HIP kernels behind the C ABI declared in include/cdnet_hip.h
(cdnet_amd/csrc -> cdnet_amd/libcdnet_hip.so, loaded by cdnet_amd._lib).
"""
__version__ = '0.1.0'

import os as _os

# HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), round robin in creation order.  Two streams on one queue
# run one after the other: once RCCL has made its streams (torch.distributed, any world size) the trainer's weight-gradient stream
# could land on the compute stream's queue - measured 1 780 -> 1 456 tiles/s under torch.distributed.run.  More queues make that rarer
# (read by the HIP runtime when it initialises, so this must run before the first GPU call), and the trainer also checks the stream it
# picks (cdnet_amd.streams.side_stream, shared with the inference pipeline).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def set_precision(p):
    """Arithmetic of the convolution stack: 'bf16' (16-bit NHWC activations, bf16 MFMA operands, fp32 accumulation) or
    'fp32' = "bf16x3": fp32 activations and gradients in HBM, fp32 accumulation, every product a*b evaluated as three bf16
    MFMAs over split operands (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with hi = bf16(x), lo = bf16(x - hi): operands carry ~16
    significand bits, relative error <= 2^-16 per product).  That is the storage and accumulation of the reference's fp32
    arithmetic and tighter than TF32, but not IEEE-fp32 multiplication (2^-24); kernels agree with fp64 to 4e-5 * sum|a*b|.
    Applies to models / trainers used afterwards; packed weights are rebuilt lazily."""
    from . import runtime
    runtime.set_precision(p)


def get_precision():
    from . import runtime
    return runtime.PRECISION
