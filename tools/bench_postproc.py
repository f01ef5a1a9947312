"""The post-processing chain of a 64-tile inference batch ALONE on the device (nothing beside it): get_probmaps epilogue -> DDM codes ->
boost / arg-max -> CC chain, on logits synthesised from the centripetal-direction maps of rendered nuclei (so that the masks, the direction
classes and the union-find load look like a network's output).
    python3 tools/bench_postproc.py [tiles] [steps]             ms per batch, GB/s against the chain's algorithmic bytes
    rocprofv3 --kernel-trace --stats ... -- python3 tools/bench_postproc.py      per-kernel durations without convolutions beside them
POSTPROC_FUSED=0: the per-step chain of rounds 1-5 (~17 launches) instead of the two-launch chain.
(`bench.py`'s inference legs run this chain on a second stream beside the next batch's forward: its kernels show 2-3 x these durations there)"""
import os
import sys
import time

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cdnet_amd import postproc, synth
from cdnet_amd.my_transforms_direction import label_encoding_batch


def logits(B, dev, seed=7):
    rs = np.random.RandomState(seed)
    lab = np.stack([(synth.ellipse_instances(256, 256, 60, rs, 5, 12, 10) > 0).astype(np.uint8) * 255 for _ in range(B)])
    lab3, point, direction = label_encoding_batch(torch.from_numpy(lab).to(dev))
    g = torch.Generator(device=dev).manual_seed(seed)

    def onehot(t, C):
        return torch.nn.functional.one_hot(t.long(), C).permute(0, 3, 1, 2).float()
    mask = 6.0 * onehot((lab3.reshape(B, 256, 256).long() + 1) // 128, 3) + torch.randn((B, 3, 256, 256), device=dev, generator=g)
    dirs = 6.0 * onehot(direction.reshape(B, 256, 256), 9) + torch.randn((B, 9, 256, 256), device=dev, generator=g)
    pt = point.reshape(B, 1, 256, 256).float() + 0.05 * torch.randn((B, 1, 256, 256), device=dev, generator=g)
    return mask.contiguous(), pt.contiguous(), dirs.contiguous()


def chain(mask, point, direction, classes=9, min_area=20, radius=2):
    B, _, H, W = mask.shape
    if os.environ.get('POSTPROC_FUSED', '1') == '1' and postproc.tile_postproc_eligible(B, classes, H, W):
        return postproc.tile_postproc(mask, direction, point, min_area, radius)        # two launches (csrc/postproc_tile.hip)
    prob, dcm = postproc.probmaps(mask, direction)
    code, minmax = postproc.ddm_codes(dcm, classes)
    r = postproc.tta_boost_argmax(prob.reshape(B, 1, 3 * H * W), point.reshape(B, 1, H * W), code.reshape(B, 1, H * W), minmax.reshape(B, 1, 2),
                                  [0], H, W, want_stages=False)
    r.update(postproc.cc_chain(r['pred'], 1, min_area, radius))
    return r


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device('cuda:0')
    m, p, d = logits(B, dev)
    for _ in range(5):
        r = chain(m, p, d)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        r = chain(m, p, d)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    alg = B * 65536 * 28
    print('post-processing chain alone, %d tiles: %.3f ms per batch = %.0f tiles/s; %d nuclei per tile on average; 28 B/px algorithmic -> %.0f GB/s'
          % (B, ms, B / ms * 1e3, int(r['counts'].float().mean()), alg / ms / 1e6))
    if os.environ.get('POSTPROC_DUMP'):
        np.savez(os.environ['POSTPROC_DUMP'], final=r['final'].cpu().numpy(), pred=r['pred'].cpu().numpy(), counts=r['counts'].cpu().numpy())


if __name__ == '__main__':
    main()
