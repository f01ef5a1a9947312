"""same-process A/B of the inference step (64 tiles incl. post-processing) with the round-4 eval-mode fusions switched off one at a time:
the post-processing stream beside the next batch, the max-pool beside the convolution's stores, the residual units' two-launch form, the BatchNorm fold into the weights.
usage: python tools/ab_infer.py [bf16|fp32] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cdnet_amd
from cdnet_amd import pipeline, runtime, synth
from cdnet_amd.models.dam.model_unet_rev1 import Unet

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cdnet_amd.set_precision(prec)
dev = torch.device('cuda:0')
torch.manual_seed(2022)
x = torch.from_numpy(synth.tiles_u8(B, seed=2022).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous().to(dev)
pool0 = runtime.ConvLayer.forward_eval_pool


def timed(label, steps=12, post=None):
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev).eval()
    run = lambda: pipeline.infer_tiles(m, x, post_stream=post)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('%-44s %7.3f ms  %7.0f tiles/s' % (label, dt * 1e3, B / dt), flush=True)


from cdnet_amd import streams
post_stream = streams.side_stream(dev)
for rep in range(2):
    timed('default')
    timed('post-processing on a second stream', post=post_stream)
    runtime.RU_EVAL_POINT_DOT = False
    timed('point feature stored, point_conv in the head')
    runtime.RU_EVAL_POINT_DOT = True
    runtime.ConvLayer.forward_eval_pool = lambda self, srcs: (self.forward(srcs, False), None)
    timed('no max-pool beside the stores')
    runtime.ConvLayer.forward_eval_pool = pool0
    runtime.RU_EVAL_ONE_LAUNCH = False
    timed('residual units in three launches')
    runtime.RU_EVAL_ONE_LAUNCH = True
    runtime.EVAL_FOLD_WEIGHTS = False
    timed('no BatchNorm fold (epilogue affine, older kernels)')
    runtime.EVAL_FOLD_WEIGHTS = True
