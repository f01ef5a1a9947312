"""HIP-graph replay of the training step / the inference pipeline == the eager launches, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(graphed, steps=6):
    import torch
    from cdnet_amd import trainer
    from cdnet_amd.graphs import GraphedTrainStep
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).to(dev)
    tr = trainer.Trainer(m)
    batch = trainer.synthetic_batch(4, dev, seed=5, H=128, W=128)
    losses = []
    if graphed:
        g = GraphedTrainStep(tr, batch, warmup=3)            # 3 eager steps + the first replayed one
        for _ in range(steps - 4):
            losses.append(g(*batch).clone())
    else:
        for k in range(steps):
            out = tr.train_step(*batch)
            if k >= 4:
                losses.append(out.clone())
    torch.cuda.synchronize()
    return tr.flat.P.clone(), torch.stack(losses).cpu().numpy(), m.state_dict()['backbone.1.running_mean'].clone(), tr._forwards


def test_graphed_train_step_is_bit_identical_to_eager():
    import torch
    p0, l0, rm0, f0 = _run(False)
    p1, l1, rm1, f1 = _run(True)
    assert torch.equal(p0, p1) and np.array_equal(l0, l1) and torch.equal(rm0, rm1) and f0 == f1


def test_graphed_inference_pipeline_is_bit_identical_to_eager():
    import torch
    from cdnet_amd import pipeline, synth
    from cdnet_amd.graphs import GraphedCallable
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    torch.manual_seed(3)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).cuda().eval()
    x = torch.from_numpy(synth.tiles_u8(4, 128, 128, seed=1).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous().cuda()
    want = pipeline.infer_tiles(m, x, want_prob=True)
    g = GraphedCallable(lambda t: pipeline.infer_tiles(m, t, want_prob=True), x.clone())
    got = g(x)
    torch.cuda.synchronize()
    for k in ('final', 'pred', 'prob', 'dcm', 'point'):
        assert torch.equal(got[k], want[k]), k
    x2 = torch.flip(x, dims=[3]).contiguous()                 # new input through the static buffer
    want2 = pipeline.infer_tiles(m, x2)
    got2 = g(x2)
    torch.cuda.synchronize()
    assert torch.equal(got2['final'], want2['final'])
