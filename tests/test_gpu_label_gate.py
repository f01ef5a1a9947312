"""Label-level parity gate of the inference path (north_star: "AJI/Dice within +-0.002 of the reference"; SURVEY 8c-1:
>= 99.9 % argmax agreement on mask and direction classes).

MoNuSeg images and trained weights cannot be here (no network), so the gate is built from what can: a network is trained
for a few hundred steps on rendered synthetic nuclei (cdnet_amd.synth.nuclei_batch) with the HIP trainer so that it
actually segments nuclei; the SAME weights then run (a) through the product path - HIP kernels, pipeline.infer_tiles /
infer_image incl. TTA + sliding windows + device post-processing - and (b) through the fp32 CPU oracle
(oracle.models.Unet + oracle.infer + oracle.postproc, each pinned to the reference).  The two instance-label maps are scored
against each other with the reference's own metrics (stats_utils.get_fast_aji / get_dice_1, test_dam.py:591-669): AJI and
Dice >= 0.998, i.e. a ground-truth score of either side can differ by at most 0.002.  BOTH modes hold that mutual bar on every case (fp32
mode's label maps are identical to the oracle's; bf16 mode has held the same 0.998 per tile since round 4 - the 0.997 mean / 0.995 per-tile
floors of round 3 are gone), and beside it north_star's literal bar: AJI / Dice of BOTH sides against the ground truth (the rendered instance
map) within 0.002.  Since round 5 the bf16 bar is checked on more than one draw: `test_second_draw` trains a second network (another seed,
other batches) and scores both networks on two 600x600 images (8 views x 9 windows) - four mutual AJI values, all printed, all held to the bar
(on 1000x1000 images of the same seeds round 5 measured 0.99872 / 0.99868 / 0.99893 / 0.99912: profiles/r05/label_gate.log).

Both arithmetic modes of the product path are gated in one run: every test is parametrised over 'fp32' (fp32 activations,
split-bf16 x3 MFMA products - the like-for-like mode) and 'bf16' (cdnet_amd.set_precision, restored afterwards).  The gate
network is trained ONCE, in fp32 mode with a fixed seed and fixed batches, so both gates score the same weights; the CRC of the
weights is printed with the results (the kernels are deterministic: the CRC only moves when a training kernel changes its
summation order) so that numbers of different rounds can be told apart from numbers of different networks."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AJI_MIN = 0.998
DICE_MIN = 0.998
ARGMAX_MIN = 0.999
# bf16 mode holds the SAME mutual bar as fp32 mode since round 4 (round 3 had lowered it to 0.997 mean / 0.995 per tile): with the eval-mode
# BatchNorm scale folded into the weights and the residual units' two roundings to fp16 gone (conv_ws16_kernel, one launch for conv2 + the 1x1
# branch) the gate network of profiles/r04/label_gate.log gives 0.99878-0.99890 per tile, 0.99877 on the 1000x1000 image, 0.99900 on the dense
# tile (deterministic: the network is trained in fp32 mode from a fixed seed, weights crc32 7e406dad)
TILE_FLOOR = AJI_MIN
BF16_MEAN_MIN = AJI_MIN
# the bar north_star states - ground-truth AJI / Dice of both sides within 0.002 - is asserted beside it
GT_DELTA_MAX = 0.002


def _train(precision, steps=300, B=8, seed=0, data_seed=100):
    import torch
    import cdnet_amd
    from cdnet_amd import synth, trainer
    from cdnet_amd.models.dam.model_unet_rev1 import Unet
    cdnet_amd.set_precision(precision)
    torch.manual_seed(seed)
    m = Unet(backbone_name='vgg16_bn', pretrained=False, classes=3).cuda()
    tr = trainer.Trainer(m)
    dev = torch.device('cuda:0')
    batches = []
    for k in range(4):
        x, lab, dirn, point, weight, _ = synth.nuclei_batch(B, 128, 128, data_seed + k, n=22)
        batches.append([torch.from_numpy(a).to(dev) for a in (x, lab, dirn, point, weight)])
    first = last = None
    for s in range(steps):
        loss = tr.train_step(*batches[s % len(batches)])
        if s == 0:
            first = float(loss[0])
    last = float(loss[0])
    assert last < 0.5 * first, 'training did not converge: %g -> %g' % (first, last)
    return m


def _oracle_of(m):
    from oracle import models as om
    ref = om.Unet()
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    return ref.eval()


_ORACLE = {}        # oracle results by test case: the CPU oracle does not depend on the precision of the product path


def _oracle(key, fn):
    if key not in _ORACLE:
        _ORACLE[key] = fn()
    return _ORACLE[key]


def _score(name, got, want, min_instances):
    """reference metrics between the two label maps (test_dam.py:613-617: both are re-labelled first)"""
    from cdnet_amd import stats_utils
    assert want.max() >= min_instances, '%s: the oracle found only %d instances - not a meaningful gate' % (name, want.max())
    if np.array_equal(got, want):
        return 1.0, 1.0
    # contiguous ids first, as the reference does (measure.label on the label image, test_dam.py:613-614)
    t, p = stats_utils.remap_label(want), stats_utils.remap_label(got)
    aji = float(stats_utils.get_fast_aji(t, p)[0])
    dice = float(stats_utils.get_dice_1(t, p))
    return aji, dice


@pytest.fixture(scope='module')
def gate_net():
    import zlib
    import cdnet_amd
    before = cdnet_amd.get_precision()
    m = _train('fp32')
    m.eval()
    crc = 0
    for k, v in sorted(m.state_dict().items()):
        crc = zlib.crc32(v.detach().cpu().contiguous().numpy().tobytes(), crc)
    print('label gate network: 300 fp32-mode steps, seed 0, weights crc32 %08x' % crc)
    yield m, _oracle_of(m), crc
    cdnet_amd.set_precision(before)


@pytest.fixture(scope='module')
def gate_net2():
    """a second draw of the gate network: another initialisation seed, other training batches"""
    import zlib
    import cdnet_amd
    before = cdnet_amd.get_precision()
    m = _train('fp32', seed=1, data_seed=500)
    m.eval()
    crc = 0
    for k, v in sorted(m.state_dict().items()):
        crc = zlib.crc32(v.detach().cpu().contiguous().numpy().tobytes(), crc)
    print('label gate network 2: 300 fp32-mode steps, seed 1, weights crc32 %08x' % crc)
    yield m, _oracle_of(m), crc
    cdnet_amd.set_precision(before)


@pytest.fixture(params=['fp32', 'bf16'])
def trained(request, gate_net):
    import cdnet_amd
    before = cdnet_amd.get_precision()
    cdnet_amd.set_precision(request.param)
    yield gate_net[0], gate_net[1], request.param
    cdnet_amd.set_precision(before)


def test_tiles_label_parity(trained):
    """three 256x256 tiles, single view (pipeline.infer_tiles): argmax agreement and instance-label parity"""
    import torch
    from cdnet_amd import pipeline, synth
    from oracle import infer as oinf
    m, ref, prec = trained
    x, lab, dirn, point, weight, inst = synth.nuclei_batch(3, 256, 256, 777, n=60)
    with torch.no_grad():
        r = pipeline.infer_tiles(m, torch.from_numpy(x).cuda(), want_prob=True)
    got_mask, got_dir = r['prob'].argmax(1).cpu().numpy(), r['dcm'].cpu().numpy().reshape(3, 256, 256)
    report, gt_scores = [], []
    for b in range(3):
        w = _oracle(('tile', b), lambda: oinf.infer_image(ref, x[b], tta=False, all_img_test=1))
        agree_m = (got_mask[b] == w['probs'][0].argmax(0)).mean()
        agree_d = (got_dir[b] == w['dcms'][0, 0]).mean()
        aji, dice = _score('tile %d' % b, r['final'][b].cpu().numpy(), w['final'], 20)
        report.append((b, agree_m, agree_d, aji, dice, int(w['count']), int(r['counts'][b])))
        # north_star's literal bar: the scores of both sides against the ground truth (the rendered instance map)
        gt_scores.append((_score('tile %d vs truth (HIP)' % b, r['final'][b].cpu().numpy(), inst[b], 20),
                          _score('tile %d vs truth (oracle)' % b, w['final'], inst[b], 20)))
    print('label gate [%s] tiles: ' % prec + '; '.join('tile %d mask %.5f dir %.5f AJI %.5f Dice %.5f n=%d/%d' % t for t in report))
    # arg-max agreement per tile; AJI / Dice as the reference reports them - the mean over the images of the set (test_dam.py:693-716
    # averages per-image metrics; north_star: "AJI/Dice on MoNuSeg within +-0.002") - with a per-tile floor so that no single
    # tile hides behind the others.  fp32 mode must also hold the bar on every tile.
    for b, am, ad, aji, dice, n_w, n_g in report:
        assert am >= ARGMAX_MIN and ad >= ARGMAX_MIN, (prec, b, am, ad)
        assert aji >= (AJI_MIN if prec == 'fp32' else TILE_FLOOR) and dice >= (DICE_MIN if prec == 'fp32' else TILE_FLOOR), (prec, b, aji, dice)
    m_aji, m_dice = float(np.mean([t[3] for t in report])), float(np.mean([t[4] for t in report]))
    print('label gate [%s] tiles: mean AJI %.5f mean Dice %.5f' % (prec, m_aji, m_dice))
    floor = AJI_MIN if prec == 'fp32' else BF16_MEAN_MIN
    assert m_aji >= floor and m_dice >= floor, (prec, m_aji, m_dice)
    g = np.array(gt_scores)                                   # [tile][side][AJI, Dice]
    d_aji, d_dice = float(g[:, 0, 0].mean() - g[:, 1, 0].mean()), float(g[:, 0, 1].mean() - g[:, 1, 1].mean())
    print('label gate [%s] tiles vs ground truth: AJI %.5f (oracle %.5f), Dice %.5f (oracle %.5f)'
          % (prec, g[:, 0, 0].mean(), g[:, 1, 0].mean(), g[:, 0, 1].mean(), g[:, 1, 1].mean()))
    assert abs(d_aji) <= GT_DELTA_MAX and abs(d_dice) <= GT_DELTA_MAX, (prec, d_aji, d_dice)


def test_full_image_tta_label_parity(trained):
    """BASELINE config 3 shape: one 1000x1000 image, 8 TTA views x 25 sliding windows (256/40), per-view DDM, boost, CC chain"""
    import torch
    from cdnet_amd import pipeline, synth
    from oracle import infer as oinf
    m, ref, prec = trained
    rs = np.random.RandomState(4242)
    inst = synth.ellipse_instances(1000, 1000, 700, rs, 5, 14, 10)
    img = synth.render_nuclei(inst, rs)
    with torch.no_grad():
        r = pipeline.infer_image(m, torch.from_numpy(img).cuda(), tta=True, all_img_test=0, patch_size=256, overlap=40)
    w = _oracle('image', lambda: oinf.infer_image(ref, img, tta=True, all_img_test=0, patch_size=256, overlap=40))
    got = r['final'].cpu().numpy()
    aji, dice = _score('1000x1000', got, w['final'], 200)
    agree = (r['pred'].cpu().numpy() == w['pred']).mean()
    print('label gate [%s] 1000x1000 TTA: pred agreement %.6f AJI %.5f Dice %.5f instances %d/%d' % (prec, agree, aji, dice, w['count'], r['count']))
    assert agree >= ARGMAX_MIN, (prec, agree)
    floor = AJI_MIN if prec == 'fp32' else BF16_MEAN_MIN
    assert aji >= floor and dice >= floor, (prec, aji, dice)
    # north_star's literal bar: both sides against the ground truth
    (ga, gd), (wa, wd) = _score('1000x1000 vs truth (HIP)', got, inst, 200), _score('1000x1000 vs truth (oracle)', w['final'], inst, 200)
    print('label gate [%s] 1000x1000 vs ground truth: AJI %.5f (oracle %.5f), Dice %.5f (oracle %.5f)' % (prec, ga, wa, gd, wd))
    assert abs(ga - wa) <= GT_DELTA_MAX and abs(gd - wd) <= GT_DELTA_MAX, (prec, ga, wa, gd, wd)


def test_dense_touching_nuclei_boost(trained):
    """a tile so densely packed that nuclei touch (400 placement attempts instead of 60): here the direction-difference boost of
    test_dam.py:490-539 actually decides pixels - the number of arg-max decisions it flipped is reported and must be non-zero on the
    oracle side too, otherwise the boost arithmetic would go untested at label level"""
    import torch
    from cdnet_amd import pipeline, synth
    from oracle import infer as oinf
    m, ref, prec = trained
    x, lab, dirn, point, weight, inst = synth.nuclei_batch(1, 256, 256, 4321, n=400)
    with torch.no_grad():
        r = pipeline.infer_tiles(m, torch.from_numpy(x).cuda(), want_prob=True)
    w = _oracle('dense', lambda: oinf.infer_image(ref, x[0], tta=False, all_img_test=1))
    got_pred = r['pred'][0].cpu().numpy()
    flips_got = int((r['prob'][0].argmax(0).cpu().numpy() != got_pred).sum())
    flips_want = int((w['probs'][0].argmax(0) != w['pred']).sum())
    agree = (got_pred == w['pred']).mean()
    aji, dice = _score('dense tile', r['final'][0].cpu().numpy(), w['final'], 60)
    print('label gate [%s] dense tile: %d instances (oracle %d), boost flipped %d pixels (oracle %d), pred agreement %.5f AJI %.5f Dice %.5f'
          % (prec, int(r['counts'][0]), int(w['count']), flips_got, flips_want, agree, aji, dice))
    assert flips_want > 0, 'the boost changed nothing on the dense tile: not a test of it'
    assert agree >= ARGMAX_MIN, (prec, agree)
    floor = AJI_MIN if prec == 'fp32' else BF16_MEAN_MIN
    assert aji >= floor and dice >= floor, (prec, aji, dice)
    (ga, gd), (wa, wd) = _score('dense vs truth (HIP)', r['final'][0].cpu().numpy(), inst[0], 60), _score('dense vs truth (oracle)', w['final'], inst[0], 60)
    print('label gate [%s] dense tile vs ground truth: AJI %.5f (oracle %.5f), Dice %.5f (oracle %.5f)' % (prec, ga, wa, gd, wd))
    assert abs(ga - wa) <= GT_DELTA_MAX and abs(gd - wd) <= GT_DELTA_MAX, (prec, ga, wa, gd, wd)


def test_second_draw(gate_net, gate_net2):
    """the bf16 bar on more than one draw (round 4 cleared it by 0.0007 on ONE network and ONE image): two networks (training seeds 0 / 1,
    different batches) x two 600x600 images (seeds 4242 / 9191; 8 TTA views x 9 windows of 256 / 40 each - the CPU oracle of a 1000x1000 image
    takes a minute and a half per draw on the GPU box's host) - all four mutual AJI / Dice values of the bf16 path against the fp32 CPU oracle
    are printed and held to the bar; fp32 mode is held to 0.9999 (identical label maps or a few pixels)"""
    import torch
    import cdnet_amd
    from cdnet_amd import pipeline, synth
    from oracle import infer as oinf
    before = cdnet_amd.get_precision()
    rows = []
    S = int(os.environ.get('CDNET_GATE_DRAW_SIZE', '600'))      # (1000: BASELINE config 3's size - 1.5-2 minutes of CPU oracle per draw; profiles/<round>/label_gate_1000x1000_draws.log)
    NI = 250 if S <= 600 else 700
    try:
        for ni, (m, ref, crc) in enumerate((gate_net, gate_net2)):
            for seed in (4242, 9191):
                rs = np.random.RandomState(seed)
                inst = synth.ellipse_instances(S, S, NI, rs, 5, 14, 10)
                img = synth.render_nuclei(inst, rs)
                w = _oracle(('draw', ni, seed), lambda: oinf.infer_image(ref, img, tta=True, all_img_test=0, patch_size=256, overlap=40))
                for prec in ('fp32', 'bf16'):
                    cdnet_amd.set_precision(prec)
                    with torch.no_grad():
                        r = pipeline.infer_image(m, torch.from_numpy(img).cuda(), tta=True, all_img_test=0, patch_size=256, overlap=40)
                    got = r['final'].cpu().numpy()
                    aji, dice = _score('net %d image %d' % (ni, seed), got, w['final'], 80)
                    (ga, gd), (wa, wd) = _score('vs truth (HIP)', got, inst, 80), _score('vs truth (oracle)', w['final'], inst, 80)
                    rows.append((ni, crc, seed, prec, aji, dice, ga, wa, int(r['count']), int(w['count'])))
                    print('label gate draw: network %d (crc32 %08x) image seed %d [%s]: mutual AJI %.5f Dice %.5f; vs truth AJI %.5f (oracle %.5f); '
                          'instances %d/%d' % rows[-1])
    finally:
        cdnet_amd.set_precision(before)
    for ni, crc, seed, prec, aji, dice, ga, wa, n_g, n_w in rows:
        if prec == 'fp32':
            # (identical label maps or a handful of pixels: bf16x3 products are not IEEE fp32 products - on 1000x1000 images of the same seeds
            #  round 5 measured 1.0 and 0.99999, profiles/r05/label_gate.log)
            assert aji >= 0.9999 and dice >= 0.9999, (ni, seed, aji, dice)
        else:
            assert aji >= AJI_MIN and dice >= DICE_MIN, (ni, seed, aji, dice)
        assert abs(ga - wa) <= GT_DELTA_MAX, (ni, seed, prec, ga, wa)
