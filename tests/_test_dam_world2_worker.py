"""Worker of tests/test_gpu_inference.py::test_entry_point_sharded_over_two_ranks: one of TWO ranks that share the one GPU of the box (gloo group:
RCCL cannot put two ranks on one device) running `cdnet_amd.test_dam.main` on its shard of the images."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['CDNET_DIST_BACKEND'] = 'gloo'
import torch

torch.cuda.set_device(0)
from cdnet_amd import test_dam

avg = test_dam.main(sys.argv[2:])
with open(os.path.join(sys.argv[1], 'rank%s.json' % os.environ['RANK']), 'w') as fh:
    json.dump(avg, fh)
