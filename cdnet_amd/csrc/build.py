"""Build libcdnet_hip.so (hipcc, gfx950 only) in-tree: cdnet_amd/libcdnet_hip.so.

  python -m cdnet_amd.csrc.build [--force]

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build container; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, 'libcdnet_hip.so')
OBJ_DIR = os.path.join(HERE, 'build')
SOURCES = ['abi.hip', 'box.hip', 'postproc.hip', 'postproc_tile.hip', 'conv.hip', 'conv32.hip', 'conv32ws.hip', 'conv16ws.hip', 'wgrad.hip', 'model.hip', 'train.hip', 'cdm.hip', 'metrics.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++20', '-fPIC', '-ffp-contract=off', '-Wall',
         '-Wno-unused-function', '-Wno-unused-result']


def _newer(a, bs):
    return os.path.exists(a) and all(os.path.getmtime(a) >= os.path.getmtime(b) for b in bs)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', 'hipcc')
    flags = FLAGS + os.environ.get('CDNET_HIPCC_FLAGS', '').split()          # debug builds (e.g. -DCDNET_WS_STAMPS)
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(PKG), 'include', 'cdnet_hip.h'))
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ_DIR, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            cmd = [hipcc] + flags + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + s)
    if force or procs or not _newer(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
