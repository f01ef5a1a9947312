"""`Options` - configuration of the CDNet hot path with the reference's field names and defaults.

Host-side mirror of the reference's options.py:31-516: `Options(isTrain)`, `.parse()`, `.print_options(logger)`,
`.save_options()`; nested dicts `opt.model / opt.train / opt.test / opt.post / opt.transform`, `opt.direction_classes`,
`opt.all_img_test`, `opt.dataset`.  Defaults are the ones `Options(...).parse()` yields in the reference.
Only the switches the hot path reads are live; the rest are kept so that option files / scripts keep working.
"""
import argparse
import os

import numpy as np


def get_transformString(names):
    """short tag of the transform list used in experiment names (options.py:11-28)"""
    tags = {'random_color': 'Rc', 'random_chooseAug': 'Rca', 'horizontal_flip': 'Hf', 'vertical_flip': 'Vf',
            'random_rotation': 'Rr', 'random_elastic': 'Re', 'random_crop': 'Crop', 'random_resize': 'Rs',
            'random_affine': 'Ra', 'label_encoding': 'Le', 'to_tensor': 'T', 'normalize': 'N'}
    return '_' + ''.join(tags.get(n, n[:2]) for n in names)


class Options:

    def __init__(self, isTrain):
        self.dataset = 'MoNuSeg_oridata'
        self.isTrain = isTrain
        self.all_img_test = 1
        self.momentum = 0.95
        self.direction_classes = 8 + 1
        self.model = dict(multi_class=True, in_c=3, out_c=3, direction=1, n_layers=6, growth_rate=24, drop_rate=0.1,
                          compress_ratio=0.5, is_hybrid=True, layer_type='basic', mean_std='mean_std', add_weightMap=1,
                          dice=1, boundary_loss=0, mseloss=1, modelName='UNet2RevA1_vgg16', backbone='None', pretrained=1,
                          LossName='CE1_Dice1')
        self.train = dict(branch=5, num_epochs=300, input_size=256, batch_size=8, val_overlap=40, seed=2022, early_stop=7,
                          scheduler='None', step=5, lr=0.001, lr_decay=0.995, weight_decay=1e-4, log_interval=15, workers=8,
                          gpu=[0], alpha=0.0, optimizer='adam', validation=0, checkpoint_freq=100, start_epoch=0,
                          checkpoint='',
                          trans_train=['random_color', 'random_chooseAug', 'horizontal_flip', 'random_elastic', 'random_crop',
                                       'label_encoding', 'to_tensor'])
        self.transform_str = get_transformString(self.train['trans_train'])
        self.transform = dict()
        self.post = dict(postproc=0, min_area=20, radius=2)
        self.test = dict(filename='test1', epoch='best', gpu=[0], branch=5, groundtruth=0, tta=True, save_flag=True,
                         patch_size=256, overlap=40)
        self._derive()

    # experiment-name / path derivation (options.py:116-198)
    def _derive(self):
        m, t = self.model, self.train
        first = '0_' + m['modelName'] + '[' + m['backbone'] + ']' + '[' + str(t['optimizer']) + ']' + '_sche[' + str(t['scheduler']) + ']'
        first += '_3c' if m['multi_class'] else '_2c'
        info = '_input' + str(t['input_size']) + 'over' + str(t['val_overlap']) + 'bs' + str(t['batch_size']) + '_e' + str(t['num_epochs'])
        m['exp_filename'] = first + info
        t['data_dir'] = './data/{:s}'.format(self.dataset)
        t['save_dir'] = './experiments/{:s}/{:s}'.format(self.dataset, m['exp_filename'])
        t['img_dir'] = '{:s}/images'.format(t['data_dir'])
        t['label_dir'] = '{:s}/labels'.format(t['data_dir'])
        t['weight_map_dir'] = '{:s}/weight_maps'.format(t['data_dir'])
        te = self.test
        te['img_dir'] = './data/{:s}/images/{:s}'.format(self.dataset, te['filename'])
        te['label_dir'] = './data/{:s}/labels/{:s}'.format(self.dataset, te['filename'])
        te['savefilename'] = ('br' + str(te['branch']) + '_' + te['filename'] + '_gt' + str(te['groundtruth']) + '_post' +
                              str(self.post['postproc']) + '_' + te['epoch'] + '_minarea' + str(self.post['min_area']) + '_ra' +
                              str(self.post['radius']) + ('' if te['tta'] else '_notta'))
        te['save_dir'] = './experiments/{:s}/{:s}/{:s}'.format(self.dataset, m['exp_filename'], te['savefilename'])
        te['model_path'] = './experiments/{:s}/{:s}/checkpoints/checkpoint_{:s}.pth.tar'.format(self.dataset, m['exp_filename'], te['epoch'])

    def parse(self, argv=None):
        p = argparse.ArgumentParser(description='')
        p.add_argument('--dataset', type=str, default=self.dataset)
        p.add_argument('--model-name', type=str, default=self.model['modelName'])
        p.add_argument('--gpu', type=list, default=self.train['gpu'] if self.isTrain else self.test['gpu'])
        p.add_argument('--all_img_test', type=int, default=self.all_img_test)
        p.add_argument('--direction', type=int, default=self.model['direction'])
        p.add_argument('--mseloss', type=int, default=self.model['mseloss'])
        if self.isTrain:
            p.add_argument('--branch', type=int, default=self.train['branch'])
            p.add_argument('--epochs', type=int, default=self.train['num_epochs'])
            p.add_argument('--input-size', type=int, default=self.train['input_size'])
            p.add_argument('--val-overlap', type=int, default=self.train['val_overlap'])
            p.add_argument('--batch-size', type=int, default=self.train['batch_size'])
            p.add_argument('--weight-map', type=int, default=self.model['add_weightMap'])
            p.add_argument('--backbone', type=str, default=self.model['backbone'])
            p.add_argument('--pretrained', type=int, default=self.model['pretrained'])
            p.add_argument('--LossName', type=str, default=self.model['LossName'])
            p.add_argument('--seed', type=int, default=self.train['seed'])
            p.add_argument('--early_stop', type=int, default=self.train['early_stop'])
            p.add_argument('--scheduler', type=str, default=self.train['scheduler'])
            p.add_argument('--step', type=int, default=5)
            p.add_argument('--lr', type=float, default=self.train['lr'])
            p.add_argument('--lr_decay', type=float, default=self.train['lr_decay'])
            p.add_argument('--momentum', type=float, default=0.95)
            p.add_argument('--optimizer', type=str, default=self.train['optimizer'])
            p.add_argument('--alpha', type=float, default=self.train['alpha'])
            p.add_argument('--dice', type=int, default=self.model['dice'])
            p.add_argument('--boundary-loss', type=int, default=self.model['boundary_loss'])
            p.add_argument('--log-interval', type=int, default=self.train['log_interval'])
            p.add_argument('--data-dir', type=str, default=self.train['data_dir'])
            p.add_argument('--save-dir', type=str, default=self.train['save_dir'])
            p.add_argument('--checkpoint-path', type=str, default=self.train['checkpoint'])
            p.add_argument('--transform-train', type=str, default=self.transform_str)
            p.add_argument('--exp-filename', type=str, default=self.model['exp_filename'])
            p.add_argument('--validation', type=int, default=self.train['validation'])
        else:
            p.add_argument('--epoch', type=str, default=self.test['epoch'])
            p.add_argument('--tta', type=int, default=int(self.test['tta']))
            p.add_argument('--postproc', type=int, default=self.post['postproc'])
            p.add_argument('--min-area', type=int, default=self.post['min_area'])
            p.add_argument('--radius', type=int, default=self.post['radius'])
            p.add_argument('--patch-size', type=int, default=self.test['patch_size'])
            p.add_argument('--overlap', '--test-overlap', dest='overlap', type=int, default=self.test['overlap'])     # options.py:369
            p.add_argument('--save-flag', type=lambda v: str(v).lower() not in ('0', 'false', ''), default=self.test['save_flag'])   # :373
            p.add_argument('--test-filename', type=str, default=self.test['filename'])              # :390
            p.add_argument('--groundtruth', type=int, default=self.test['groundtruth'])             # :400
            p.add_argument('--img-dir', type=str, default=self.test['img_dir'])
            p.add_argument('--label-dir', type=str, default=self.test['label_dir'])
            p.add_argument('--save-dir', type=str, default=self.test['save_dir'])
            p.add_argument('--model-path', type=str, default=self.test['model_path'])
        a = p.parse_args(argv)
        self.dataset = a.dataset
        self.model['modelName'] = a.model_name
        self.all_img_test = a.all_img_test
        self.model['direction'], self.model['mseloss'] = a.direction, a.mseloss
        if self.isTrain:
            t, m = self.train, self.model
            t['num_epochs'], t['input_size'], t['val_overlap'], t['batch_size'] = a.epochs, a.input_size, a.val_overlap, a.batch_size
            m['add_weightMap'], m['backbone'], m['pretrained'], m['LossName'] = a.weight_map, a.backbone, a.pretrained, a.LossName
            t['seed'], t['early_stop'], t['scheduler'], t['step'], t['lr'], t['lr_decay'] = a.seed, a.early_stop, a.scheduler, a.step, a.lr, a.lr_decay
            self.momentum = a.momentum
            t['optimizer'], t['alpha'], m['dice'], m['boundary_loss'] = a.optimizer, a.alpha, a.dice, a.boundary_loss
            t['log_interval'], t['gpu'], t['branch'], t['checkpoint'], t['validation'] = a.log_interval, list(a.gpu), a.branch, a.checkpoint_path, a.validation
            self._derive()
            if a.save_dir != p.get_default('save_dir'):
                t['save_dir'] = a.save_dir
            # default training transform chain as the reference builds it
            self.transform['train'] = {'random_color': 1, 'horizontal_flip': True, 'vertical_flip': True, 'random_elastic': [6, 15],
                                       'random_chooseAug': 1, 'random_crop': t['input_size'],
                                       'label_encoding': [m['out_c'], 2, m['direction']], 'to_tensor': 1}
            self.transform['val'] = {'to_tensor': 1}
        else:
            te = self.test
            te['epoch'], te['tta'], te['patch_size'], te['overlap'] = a.epoch, bool(a.tta), a.patch_size, a.overlap
            te['save_flag'], te['filename'], te['groundtruth'] = bool(a.save_flag), a.test_filename, int(a.groundtruth)
            self.post['postproc'], self.post['min_area'], self.post['radius'] = a.postproc, a.min_area, a.radius
            te['gpu'] = list(a.gpu)
            self._derive()
            for k, v in (('img_dir', a.img_dir), ('label_dir', a.label_dir), ('save_dir', a.save_dir), ('model_path', a.model_path)):
                if v != p.get_default(k):
                    te[k] = v
            # options.py:463-472: normalise with mean_std.npy unless the experiment is a "_noNorm" one
            self.transform['test'] = {'to_tensor': 1}
            ms = '{:s}/{:s}.npy'.format(self.train['data_dir'], self.model['mean_std'])
            if '_noNorm' not in te['save_dir'] and os.path.exists(ms):
                mean, std = np.load(ms)
                self.transform['test']['normalize'] = [mean, std]
        return self

    def print_options(self, logger=None):
        lines = ['# ---------- Options ---------- #', '[dataset] ' + self.dataset]
        for group in ('model', 'train', 'test', 'post', 'transform'):
            lines.append('[{}]'.format(group))
            for k, v in getattr(self, group).items():
                lines.append('\t{:s}: {:s}'.format(str(k), str(v)))
        msg = '\n'.join(lines)
        (logger.info if logger is not None else print)(msg)

    def save_options(self):
        d = self.train['save_dir'] if self.isTrain else self.test['save_dir']
        os.makedirs(d, exist_ok=True)
        name = 'train_options.txt' if self.isTrain else 'test_options.txt'
        with open(os.path.join(d, name), 'w') as f:
            for group in ('model', 'train', 'test', 'post', 'transform'):
                f.write('[{}]\n'.format(group))
                for k, v in getattr(self, group).items():
                    f.write('\t{:s}: {:s}\n'.format(str(k), str(v)))
