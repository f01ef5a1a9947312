"""Inference entry points with the reference's names (test_dam.py).

  get_probmaps(input, model, opt, name, times)      test_dam.py:932-1035  -> [prob_maps, point_maps, pred_direction] (numpy)
  process_image(model, image, opt)                  test_dam.py:297-563   -> dict with the instance label map
  main()                                            test_dam.py:90-925    (command line; images from opt.test['img_dir'])
"""
import os
import numpy as np
import torch

from . import checkpoint, pipeline, postproc, utils
from .options import Options


def get_probmaps(input, model, opt, name=None, times=None):
    """input: float tensor [1,C,H,W].  Whole-image forward when opt.all_img_test == 1, else 256/40 sliding windows;
    returns [prob_maps f32 [3,H,W], point_maps f32 [1,H,W], pred_direction i64 [1,H,W]] as numpy arrays (the reference's
    return value for direction == 1 and mseloss == 1)."""
    x = input[0].cuda().float()
    _, H, W = x.shape
    with torch.no_grad():
        if opt.all_img_test == 1:
            (mask, point, direction), = utils.split_forward_views(model, x, max(H, W), 0, (0,), opt.direction_classes)
        else:
            (mask, point, direction), = utils.split_forward_views(model, x, opt.test['patch_size'], opt.test['overlap'], (0,),
                                                                  opt.direction_classes)
        prob, dcm = postproc.probmaps(mask[None], direction[None])
    if times is not None:
        times[0] = times[0] + 1
    return [prob[0].cpu().numpy(), point.cpu().numpy(), dcm.cpu().numpy().astype(np.int64)]


def process_image(model, image, opt):
    """image: float tensor [3,H,W] (after the test transform).  Returns dict(final=np.int32 [H,W], count=int)."""
    r = pipeline.infer_image(model, image.cuda().float(), opt)
    return dict(final=r['final'].cpu().numpy(), count=r['count'], pred=r['pred'].cpu().numpy())


def main(argv=None):
    opt = Options(isTrain=False).parse(argv)
    model = utils.chooseModel(opt).cuda()
    if os.path.exists(opt.test['model_path']):
        checkpoint.load_checkpoint(opt.test['model_path'], model, strict=False)      # DataParallel prefix (test_dam.py:158-167)
    model.eval()
    img_dir = opt.test['img_dir']
    names = sorted(f for f in os.listdir(img_dir) if f.endswith('.png')) if os.path.isdir(img_dir) else []
    os.makedirs(opt.test['save_dir'], exist_ok=True)
    from PIL import Image
    for f in names:
        img = np.asarray(Image.open(os.path.join(img_dir, f)).convert('RGB'), dtype=np.float32) / 255.0
        x = torch.from_numpy(img).permute(2, 0, 1).contiguous()
        if 'normalize' in opt.transform['test']:
            mean, std = opt.transform['test']['normalize']
            x = (x - torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)) / torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
        r = process_image(model, x, opt)
        Image.fromarray(r['final'].astype(np.uint16)).save(os.path.join(opt.test['save_dir'], f[:-4] + '_seg.tiff'))
        print('{:s}: {:d} nuclei'.format(f, r['count']))


if __name__ == '__main__':
    main()
