#!/usr/bin/env python3
"""Training driver — Py3 counterpart of /root/reference/augmented_cyclegan/train.py:96-313 (Python 2 there).
Same flags (options.py), same `results.txt` line format (train.py:39-45), same artefacts: opt.txt / opt.pkl,
nets.txt, results.json, history_mse_A.npy / history_ubo_B.npy, best_mse_A.txt / best_bpp_B.txt, checkpoints
`latest` / `best_A` / `best_B`, PNG grids under vis_*/ (written by a small built-in PNG encoder: torchvision is not
required).  Additions: a working --continue_train, --synthetic data, one-process-per-GPU data parallelism
(`python -m torch.distributed.run --nproc-per-node N -m ...train` shards every batch by rank)."""
import itertools
import json
import os
import random
import struct
import sys
import time
import zlib
from shutil import copyfile

import numpy as np
import torch

from . import dist as D, ops
from .dataloader import AlignedIterator, DevicePrefetcher, UnalignedIterator, load_numpy_data, synthetic_data
from .evaluate import eval_mse_A, eval_ubo_B, one_to_three_channels
from .model import AugmentedCycleGAN, StochCycleGAN
from .options import TrainOptions, create_sub_dirs


def save_results(expr_dir, results_dict):
    with open(os.path.join(expr_dir, 'results.json'), 'w') as f:
        json.dump(results_dict, f, indent=4)


def print_log(out_f, message):
    if out_f is not None:
        out_f.write(message + "\n")
        out_f.flush()
        print(message)


def format_log(epoch, i, errors, t, prefix=True):
    """train.py:39-45"""
    message = '(epoch: %d, iters: %d, time: %.3f) ' % (epoch, i, t)
    if not prefix:
        message = ' ' * len(message)
    for k, v in errors.items():
        message += '%s: %.3f ' % (k, v)
    return message


def write_png(path, img_u8):
    """img_u8: (H, W, 3) uint8 -> minimal PNG (zlib + CRC chunks)"""
    h, w, _ = img_u8.shape
    raw = b''.join(b'\x00' + img_u8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack('>I', len(data)) + tag + data
        return c + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def save_image_grid(t, path, nrow, pad=2):
    """(N,3,H,W) in [-1,1] -> PNG grid with `nrow` images per row (vutils.save_image(normalize=True, range=(-1,1)))"""
    t = ((t.detach().float().cpu().clamp(-1, 1) + 1) * 127.5).round().byte().numpy()
    n, c, h, w = t.shape
    rows = (n + nrow - 1) // nrow
    grid = np.zeros((rows * (h + pad) + pad, nrow * (w + pad) + pad, 3), np.uint8)
    for i in range(n):
        r, q = divmod(i, nrow)
        grid[pad + r * (h + pad): pad + r * (h + pad) + h, pad + q * (w + pad): pad + q * (w + pad) + w] = t[i, :3].transpose(1, 2, 0)
    write_png(path, grid)


def visualize_cycle(opt, real_A, visuals, eidx, uidx, train):
    """train.py:47-59"""
    size = real_A.size()
    images = [one_to_three_channels(img.cpu()).unsqueeze(1) for img in visuals.values()]
    vis = torch.cat(images, dim=1).view(size[0] * len(images), 3, size[2], size[3])
    save_path = os.path.join(opt.train_vis_cycle if train else opt.vis_cycle, 'cycle_%02d_%04d.png' % (eidx, uidx))
    save_image_grid(vis, save_path, nrow=len(images))
    copyfile(save_path, os.path.join(opt.vis_latest, 'cycle.png'))


def visualize_multi(opt, real_A, model, eidx, uidx):
    """train.py:61-74"""
    size = real_A.size()
    z = real_A.new_empty((opt.num_multi, opt.nlatent, 1, 1)).normal_(0, 1).repeat(size[0], 1, 1, 1)
    with torch.no_grad():
        multi = model.generate_multi(real_A.detach(), z)
    multi = one_to_three_channels(multi.cpu()).view(size[0], opt.num_multi, 3, size[2], size[3])
    vis = torch.cat([one_to_three_channels(real_A.cpu()).unsqueeze(1), multi], dim=1).view(size[0] * (opt.num_multi + 1), 3, size[2], size[3])
    save_path = os.path.join(opt.vis_multi, 'multi_%02d_%04d.png' % (eidx, uidx))
    save_image_grid(vis, save_path, nrow=opt.num_multi + 1)
    copyfile(save_path, os.path.join(opt.vis_latest, 'multi.png'))


def _cuda(t, on):
    return t.cuda() if on else t


class Trainer(object):
    """One training run: data -> model -> epochs of train_instance (+ optional paired step) -> per-epoch checkpoint,
    evaluation, best-model tracking and LR decay.  Rank 0 owns all files; every rank trains on its shard of each batch."""

    BEST = ('best_dev_mse_A', 'best_test_mse_A', 'best_dev_bpp_B', 'best_test_bpp_B')

    def __init__(self, argv=None):
        self.opt = opt = TrainOptions().parse(sub_dirs=['vis_multi', 'vis_cycle', 'vis_latest', 'train_vis_cycle'], argv=argv)
        self.rank, self.ws = D.init_from_env()
        self.gpu = len(opt.gpu_ids) > 0
        self.log_f = open("%s/results.txt" % opt.expr_dir, 'a' if opt.continue_train else 'w') if self.rank == 0 else None
        ops.set_precision(opt.precision)
        if opt.seed is not None:                                                         # train.py:102-108
            print("using random seed:", opt.seed)
            random.seed(opt.seed); np.random.seed(opt.seed); torch.manual_seed(opt.seed)
            if self.gpu:
                torch.cuda.manual_seed_all(opt.seed)
        self._data()
        self._model()
        self.total_steps = 0
        self.results = {k: sys.float_info.max for k in self.BEST}
        self.history_mse_A, self.history_ubo_B = [], []
        if self.rank == 0:
            save_results(opt.expr_dir, self.results)
        create_sub_dirs(opt, ['vis_pred_B'])

    def log(self, msg):
        print_log(self.log_f, msg)

    # ---------------------------------------------------------------- setup
    def _data(self):
        o = self.opt
        if o.synthetic:
            arrays = synthetic_data(o.synthetic, o.input_nc, o.output_nc, o.grid_size, o.seed or 0)
        else:
            arrays = load_numpy_data(o.dataroot, grid_size=o.grid_size)
        trainA, trainB, devA, devB, testA, testB = arrays
        self.train_it = UnalignedIterator(trainA, trainB, batch_size=o.batchSize)
        self.test_it = AlignedIterator(testA, testB, batch_size=100)
        self.dev_it = AlignedIterator(devA, devB, batch_size=100)
        self.dev_cycle = itertools.cycle(AlignedIterator(devA, devB, batch_size=25))
        for label, it in (('training', self.train_it), ('test', self.test_it), ('dev', self.dev_it)):
            self.log('#%s images = %d' % (label, len(it)))
        self.sup_it = None
        if o.supervised:
            n_sup = int(len(trainA) * o.sup_frac)
            self.sup_it = itertools.cycle(AlignedIterator(trainA[:n_sup], trainB[:n_sup], batch_size=o.batchSize))
            self.log('#supervised images = %d' % n_sup)

    def _model(self):
        o = self.opt
        if o.model == 'aug_cycle_gan':
            self.model = AugmentedCycleGAN(o)
            create_sub_dirs(o, ['vis_inf'])
        elif o.model in ('stoch_cycle_gan', 'cycle_gan'):
            self.model = StochCycleGAN(o, ignore_noise=(o.model == 'cycle_gan'))
        else:
            raise NotImplementedError('Specified model is not implemented.')
        self.log("model [%s] was created (conv arithmetic: %s, %d rank%s)"
                 % (self.model.__class__.__name__, ops.get_precision(), self.ws, "" if self.ws == 1 else "s"))
        if getattr(o, "step_graph", False) and not getattr(o, "supervised", False):
            self.model.enable_step_graph()            # ignored while the data-parallel exchange is on (model.train_instance)
        if o.continue_train:
            chk = os.path.join(o.expr_dir, o.which_epoch)
            self.model.load(chk)
            self.log("continue_train: loaded %s" % chk)

    # ---------------------------------------------------------------- one epoch
    def _shard(self, t):
        if self.ws == 1:
            return t
        n = t.size(0) // self.ws
        return t[self.rank * n:(self.rank + 1) * n]

    def _visualize(self, real_A, visuals, epoch, it):
        o, m = self.opt, self.model
        visualize_cycle(o, real_A, visuals, epoch, it, train=True)
        batch = next(self.dev_cycle)
        dA, dB = _cuda(batch['A'], self.gpu), _cuda(batch['B'], self.gpu)
        dz = dA.new_empty((dA.size(0), o.nlatent, 1, 1)).normal_(0, 1)
        with torch.no_grad():
            visualize_cycle(o, dA, m.generate_cycle(dA, dB, dz), epoch, it, train=False)
        visualize_multi(o, dA, m, epoch, it)

    def _train_batches(self):
        """this rank's shard of every training batch; on the GPU the next batch is uploaded through pinned memory on a
        side stream while the current step computes (dataloader.DevicePrefetcher)"""
        class Sharded(object):          # host-side shard BEFORE the upload: a rank moves only its own rows
            def __init__(s, it):
                s.it = it

            def __len__(s):
                return len(s.it)

            def __iter__(s):
                iter(s.it)
                return s

            def __next__(s):
                d = next(s.it)
                full = d['A'].size(0)
                return {'A': self._shard(d['A']), 'B': self._shard(d['B']), 'n': (full, d['B'].size(0))}
        src = Sharded(self.train_it)
        return DevicePrefetcher(src) if self.gpu else src

    def train_epoch(self, epoch):
        o, m = self.opt, self.model
        seen = 0
        for data in self._train_batches():
            real_A, real_B = data['A'], data['B']
            nA, nB = data['n']
            if nA != nB:
                continue
            prior_z_B = torch.empty((nA, o.nlatent, 1, 1)).normal_(0, 1)                 # on the host, train.py:193
            self.total_steps += o.batchSize
            seen += o.batchSize
            prior_z_B = _cuda(self._shard(prior_z_B), self.gpu)
            out = m.train_instance(real_A, real_B, prior_z_B)
            sup_losses = None
            if self.sup_it is not None:
                sd = next(self.sup_it)
                sA, sB = _cuda(self._shard(sd['A']), self.gpu), _cuda(self._shard(sd['B']), self.gpu)
                sup_losses = m.supervised_train_instance(sA, sB, prior_z_B[:sA.size(0)])
            if self.total_steps % o.display_freq == 0 and self.rank == 0:
                self._visualize(real_A, out[1], epoch, seen // o.batchSize)
            if self.total_steps % o.print_freq == 0:
                t = (time.time() - self.tick) / o.batchSize                            # seconds per image, train.py:243
                self.log(format_log(epoch, seen, out[0], t))
                if sup_losses is not None:
                    self.log(format_log(epoch, seen, sup_losses, t, prefix=False))
                if o.monitor_gnorm:
                    self.log(format_log(epoch, seen, out[2], t, prefix=False) + "\n")
                self.tick = time.time()

    # ---------------------------------------------------------------- per-epoch evaluation (rank 0)
    def _track_best(self, epoch, line, dev, test, dev_key, test_key, chk, fname, banner):
        o = self.opt
        lines = [line]
        if dev < self.results[dev_key]:
            with open(os.path.join(o.expr_dir, fname), 'w') as f:
                f.write(line + '\n')
            self.results[dev_key], self.results[test_key] = dev, test
            self.model.save(chk)
            save_results(o.expr_dir, self.results)
            lines.append(banner)
        self.log("\n".join(["-" * 60] + lines + ["-" * 60]))

    def evaluate(self, epoch):
        o, m = self.opt, self.model
        if epoch % o.eval_A_freq == 0:
            t0 = time.time()
            dev, test = eval_mse_A(self.dev_it, m, self.gpu), eval_mse_A(self.test_it, m, self.gpu)
            self.history_mse_A.append((dev, test))
            np.save("%s/history_mse_A" % o.expr_dir, self.history_mse_A)
            line = "[%d] DEV_MSE_A: %.4f, TEST_MSE_A: %.4f, TIME: %.4f" % (epoch, dev, test, time.time() - t0)
            self._track_best(epoch, line, dev, test, 'best_dev_mse_A', 'best_test_mse_A', 'best_A', 'best_mse_A.txt',
                             "*** BEST DEV A ***")
        if epoch % o.eval_B_freq == 0:
            t0 = time.time()
            steps = 1 if o.model == 'cycle_gan' else o.eval_steps                      # train.py:281-285
            d_ubo, d_bpp, d_kld = eval_ubo_B(self.dev_it, m, steps, self.gpu)
            t_ubo, t_bpp, t_kld = eval_ubo_B(self.test_it, m, steps, self.gpu)
            self.history_ubo_B.append((d_ubo, d_bpp, d_kld, t_ubo, t_bpp, t_kld))
            np.save("%s/history_ubo_B" % o.expr_dir, self.history_ubo_B)
            line = "[%d] DEV_BPP_B: %.4f, TEST_BPP_B: %.4f, TIME: %.4f" % (epoch, d_bpp, t_bpp, time.time() - t0)
            self._track_best(epoch, line, d_bpp, t_bpp, 'best_dev_bpp_B', 'best_test_bpp_B', 'best_B', 'best_bpp_B.txt',
                             "*** BEST BPP B ***")

    def run(self):
        o = self.opt
        self.tick = time.time()
        last = o.niter + o.niter_decay
        for epoch in range(o.epoch_count, last + 1):
            t0 = time.time()
            self.train_epoch(epoch)
            if self.rank == 0:
                if epoch % o.save_epoch_freq == 0:
                    self.log('saving the model at the end of epoch %d, iters %d' % (epoch, self.total_steps))
                    self.model.save('latest')
                self.evaluate(epoch)
            self.log('End of epoch %d / %d \t Time Taken: %d sec' % (epoch, last, time.time() - t0))
            if epoch > o.niter:
                self.model.update_learning_rate()
        if self.log_f is not None:
            self.log_f.close()
        return self.model


def train_model(argv=None):
    """entry point with the reference's name (train.py:96)"""
    return Trainer(argv).run()


if __name__ == "__main__":
    train_model()
