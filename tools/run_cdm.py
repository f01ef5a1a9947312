"""label_encoding_batch on the bench's 16 synthetic label images, for profiler passes:  python tools/run_cdm.py [launches = 20]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cdnet_amd.my_transforms_direction import label_encoding_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lab0 = torch.from_numpy(bench.cdm_labels(16)).cuda()
for _ in range(n):
    label_encoding_batch(lab0)
torch.cuda.synchronize()
