"""ORACLE - test infrastructure only.

CPU restatements of the reference's algorithms for the hot path, used as the checker by tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg.  The product package (cdnet_amd) never imports
this package; its compute path is the HIP library behind include/cdnet_hip.h and fails loudly without it.

Pinning: every function here is checked against golden vectors produced by running the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).  Steps whose reference
implementation calls scikit-image (absent in the build container) were evaluated through scipy stand-ins and
are marked "skimage-semantics restated" (SURVEY.md 8c).  There is no compiled reference (`oracle/_ref`):
the reference is pure Python and cannot travel to the GPU box.
"""
