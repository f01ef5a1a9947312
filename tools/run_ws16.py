"""the dominant 16-bit layer (3x3 64 -> 64 @256x256) alone, for profiler passes:  python tools/run_ws16.py <tiles> <zero 0|1> <debug bits> [launches]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine
B, zero, dbg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device('cuda:0')
x = torch.zeros((B, 256, 256, 64), device=dev, dtype=torch.bfloat16) if zero else (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
w = torch.zeros((64, 64, 3, 3), device=dev) if zero else torch.randn((64, 64, 3, 3), device=dev) * 0.06
cfg = (16, 16, 64)
wp = engine.pack_weights(w, cfg, 0)
out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
engine.CONV_DEBUG = dbg
for _ in range(n):
    engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
torch.cuda.synchronize()
