"""cdnet_amd - MI355X-native (gfx950) implementation of CDNet's data-parallel hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); every compute step of
the path runs in hand-written HIP kernels behind the C ABI declared in include/cdnet_hip.h
(cdnet_amd/csrc -> cdnet_amd/libcdnet_hip.so, loaded by cdnet_amd._lib).
"""
__version__ = '0.1.0'
