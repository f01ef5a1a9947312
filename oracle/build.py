"""Compile the plain-C oracle (gcc) into oracle/_build/liboracle.so.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'liboracle.so')
SOURCES = ['postproc_oracle.c', 'cdm_oracle.c']


def build(force=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    cmd = ['gcc', '-O2', '-fPIC', '-shared', '-std=c11', '-ffp-contract=off', '-o', LIB] + srcs + ['-lm']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
