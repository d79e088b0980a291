"""Command-line options — Py3 counterpart of /root/reference/augmented_cyclegan/options.py (which is Python 2:
`import cPickle`, options.py:4).  Same flags, defaults, `opt.txt` format, `opt.pkl`, sub-directory creation
(options.py:7-12, 20-131).  Additions (not in the reference): --n_blocks, --precision, --synthetic, --dist."""
import argparse
import os
import pickle

import torch


def create_sub_dirs(opt, sub_dirs):
    """options.py:7-12"""
    for sub_dir in sub_dirs:
        dir_path = os.path.join(opt.expr_dir, sub_dir)
        os.makedirs(dir_path, exist_ok=True)
        setattr(opt, sub_dir, dir_path)


class TrainOptions(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.initialized = False

    def initialize(self):
        p = self.parser
        p.add_argument('--dataroot', type=str, default=None, help='path to data (trainA/B.npz, testA/B.npz)')
        p.add_argument('--name', type=str, required=True, help='name of the experiment')
        p.add_argument('--checkpoints_dir', type=str, default='./checkpoints/', help='models are saved here')
        # data
        p.add_argument('--input_nc', type=int, default=3)
        p.add_argument('--output_nc', type=int, default=3)
        p.add_argument('--grid_size', type=int, default=256)
        p.add_argument('--numpy_data', type=int, choices=[0, 1], default=1)
        # exp
        p.add_argument('--seed', type=int)
        p.add_argument('--model', type=str, choices=['cycle_gan', 'stoch_cycle_gan', 'aug_cycle_gan'], default='aug_cycle_gan')
        p.add_argument('--gpu_ids', type=str, default='0', help='gpu ids: e.g. 0  0,1,2 (only "on the GPU" matters here)')
        # supervised training
        p.add_argument('--supervised', action='store_true')
        p.add_argument('--sup_frac', type=float, default=0.1)
        p.add_argument('--lambda_sup_A', type=float, default=0.1)
        p.add_argument('--lambda_sup_B', type=float, default=0.1)
        # training
        p.add_argument('--batchSize', type=int, default=32)
        p.add_argument('--continue_train', action='store_true', help='reload <expr_dir>/<which_epoch> before training')
        p.add_argument('--which_epoch', type=str, default='latest')
        p.add_argument('--epoch_count', type=int, default=1)
        p.add_argument('--niter', type=int, default=25)
        p.add_argument('--niter_decay', type=int, default=25)
        p.add_argument('--beta1', type=float, default=0.5)
        p.add_argument('--lr', type=float, default=0.0002)
        # model
        p.add_argument('--ngf', type=int, default=32)
        p.add_argument('--nef', type=int, default=32)
        p.add_argument('--ndf', type=int, default=64)
        p.add_argument('--nlatent', type=int, default=16)
        p.add_argument('--which_model_netD', type=str, default='basic')
        p.add_argument('--which_model_netG', type=str, default='resnet')
        p.add_argument('--norm', type=str, default='instance')
        p.add_argument('--use_dropout', action='store_true')
        p.add_argument('--max_gnorm', type=float, default=500.)
        p.add_argument('--stoch_enc', action='store_true')
        p.add_argument('--z_gan', type=int, default=1, choices=[0, 1])
        p.add_argument('--enc_A_B', type=int, default=1, choices=[0, 1])
        p.add_argument('--no_lsgan', action='store_true')
        p.add_argument('--lambda_A', type=float, default=1.0)
        p.add_argument('--lambda_B', type=float, default=1.0)
        p.add_argument('--lambda_z_B', type=float, default=0.025)
        # monitoring
        p.add_argument('--monitor_gnorm', type=bool, default=True)   # type=bool as in options.py:77 (any string is True)
        p.add_argument('--display_freq', type=int, default=5000)
        p.add_argument('--print_freq', type=int, default=100)
        p.add_argument('--save_epoch_freq', type=int, default=5)
        p.add_argument('--num_multi', type=int, default=10)
        p.add_argument('--eval_A_freq', type=int, default=1)
        p.add_argument('--eval_B_freq', type=int, default=1)
        # additions
        p.add_argument('--n_blocks', type=int, default=3, help='residual blocks per generator (the reference builds 3)')
        p.add_argument('--precision', type=str, default='f32', choices=['f32', 'bf16'], help='conv arithmetic')
        p.add_argument('--synthetic', type=int, default=0, help='use N synthetic U(-1,1) samples per split instead of --dataroot')
        p.add_argument('--sync_bn', action='store_true', help='data parallel: BatchNorm (E_B, D_z_B) statistics across all ranks')
        p.add_argument('--eval_steps', type=int, default=50, help='variational-bound steps per epoch (train.py:285 uses 50)')
        self.initialized = True

    def parse(self, sub_dirs=None, argv=None):
        if not self.initialized:
            self.initialize()
        self.opt = self.parser.parse_args(argv)
        if self.opt.dataroot is None and not self.opt.synthetic:
            self.parser.error('--dataroot is required (or --synthetic N)')
        ids = [int(s) for s in self.opt.gpu_ids.split(',')]
        self.opt.gpu_ids = [i for i in ids if i >= 0]                       # options.py:92-97
        if len(self.opt.gpu_ids) > 0 and torch.cuda.is_available():
            local = int(os.environ.get('LOCAL_RANK', self.opt.gpu_ids[0]))
            torch.cuda.set_device(local)
            self.opt.gpu_ids = [local]
        expr_dir = os.path.join(self.opt.checkpoints_dir, self.opt.name)
        self.opt.expr_dir = expr_dir
        args = vars(self.opt)
        lines = ['------------ Options -------------'] + ['%s: %s' % (str(k), str(v)) for k, v in sorted(args.items())] + \
                ['-------------- End ----------------']
        print('\n'.join(lines))
        os.makedirs(expr_dir, exist_ok=True)
        with open(os.path.join(expr_dir, 'opt.txt'), 'wt') as f:          # options.py:119-124
            f.write('\n'.join(lines) + '\n')
        with open(os.path.join(expr_dir, 'opt.pkl'), 'wb') as f:          # options.py:126-128
            pickle.dump(args, f)
        if sub_dirs is not None:
            create_sub_dirs(self.opt, sub_dirs)
        return self.opt


class TestOptions(object):
    """options.py:134-143"""

    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.parser.add_argument('--chk_path', required=True, type=str)
        self.parser.add_argument('--res_dir', type=str, default='test_res')
        self.parser.add_argument('--train_logvar', type=int, default=1)
        self.parser.add_argument('--dataroot', required=True, type=str)
        self.parser.add_argument('--metric', required=True, type=str, choices=['bpp', 'mse', 'visual', 'noise_sens'])

    def parse(self, argv=None):
        return self.parser.parse_args(argv)
