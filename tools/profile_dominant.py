"""Runs only the dominant kernel (3x3 conv 64->64 @256x256, batch 16) for rocprofv3 PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdnet_amd import engine
B = 16
dev = torch.device('cuda:0')
x = (torch.rand((B, 256, 256, 64), device=dev) - 0.3).to(torch.bfloat16)
w = torch.randn((64, 64, 3, 3), device=dev) * 0.06
cfg = engine.choose_cfg([64], 64, 256, 256)
wp = engine.pack_weights(w, cfg, 0)
out = torch.empty((B, 256, 256, 64), dtype=torch.bfloat16, device=dev)
for _ in range(20):
    engine.conv_forward([engine.Src(x)], wp, 64, cfg, out=out)
torch.cuda.synchronize()
print('done', cfg)
