"""child process of tests/test_gpu_train_step.py::test_bf16_dz_reuse_matches_the_recomputed_gradient: the parameter gradients of one bf16 training
step of the DAM-Unet (CDNET_BN_DZ_REUSE is read once per process by libcdnet_hip.so) -> an .npz file"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np


def main(out):
    import cdnet_amd
    import test_gpu_train_step as T
    cdnet_amd.set_precision('bf16')
    m, ref, x, t = T._setup(B=2, S=64)
    tr, g = T._hip_grads(m, x, t)
    np.savez(out, **{k: v.numpy() for k, v in g.items()})


if __name__ == '__main__':
    main(sys.argv[1])
