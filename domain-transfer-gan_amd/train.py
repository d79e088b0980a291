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
from .dataloader import AlignedIterator, UnalignedIterator, load_numpy_data, synthetic_data
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


def _to_dev(batch, use_gpu):
    a, b = batch['A'], batch['B']
    return (a.cuda(), b.cuda()) if use_gpu else (a, b)


def _shard(t, rank, ws):
    n = t.size(0) // ws
    return t[rank * n:(rank + 1) * n] if ws > 1 else t


def train_model(argv=None):
    opt = TrainOptions().parse(sub_dirs=['vis_multi', 'vis_cycle', 'vis_latest', 'train_vis_cycle'], argv=argv)
    rank, ws = D.init_from_env()
    out_f = open("%s/results.txt" % opt.expr_dir, 'a' if opt.continue_train else 'w') if rank == 0 else None
    use_gpu = len(opt.gpu_ids) > 0
    ops.set_precision(opt.precision)
    if opt.seed is not None:
        print("using random seed:", opt.seed)
        random.seed(opt.seed); np.random.seed(opt.seed); torch.manual_seed(opt.seed)      # train.py:102-108
        if use_gpu:
            torch.cuda.manual_seed_all(opt.seed)

    if opt.synthetic:
        trainA, trainB, devA, devB, testA, testB = synthetic_data(opt.synthetic, opt.input_nc, opt.output_nc, opt.grid_size,
                                                                  opt.seed or 0)
    else:
        trainA, trainB, devA, devB, testA, testB = load_numpy_data(opt.dataroot, grid_size=opt.grid_size)
    train_dataset = UnalignedIterator(trainA, trainB, batch_size=opt.batchSize)
    print_log(out_f, '#training images = %d' % len(train_dataset))
    test_dataset = AlignedIterator(testA, testB, batch_size=100)
    print_log(out_f, '#test images = %d' % len(test_dataset))
    dev_dataset = AlignedIterator(devA, devB, batch_size=100)
    print_log(out_f, '#dev images = %d' % len(dev_dataset))
    dev_cycle = itertools.cycle(AlignedIterator(devA, devB, batch_size=25))
    if opt.supervised:
        sup_size = int(len(trainA) * opt.sup_frac)
        sup_train_dataset = itertools.cycle(AlignedIterator(trainA[:sup_size], trainB[:sup_size], batch_size=opt.batchSize))
        print_log(out_f, '#supervised images = %d' % sup_size)

    vis_inf = False
    if opt.model == 'stoch_cycle_gan':
        model = StochCycleGAN(opt)
    elif opt.model == 'cycle_gan':
        model = StochCycleGAN(opt, ignore_noise=True)
    elif opt.model == 'aug_cycle_gan':
        model = AugmentedCycleGAN(opt)
        create_sub_dirs(opt, ['vis_inf'])
        vis_inf = True
    else:
        raise NotImplementedError('Specified model is not implemented.')
    print_log(out_f, "model [%s] was created" % (model.__class__.__name__))
    if opt.continue_train:
        chk = os.path.join(opt.expr_dir, opt.which_epoch)
        model.load(chk)
        print_log(out_f, "continue_train: loaded %s" % chk)

    total_steps = 0
    print_start_time = time.time()
    results = {k: sys.float_info.max for k in ('best_dev_mse_A', 'best_test_mse_A', 'best_dev_bpp_B', 'best_test_bpp_B')}
    if rank == 0:
        save_results(opt.expr_dir, results)
    history_mse_A, history_ubo_B = [], []
    create_sub_dirs(opt, ['vis_pred_B'])

    for epoch in range(opt.epoch_count, opt.niter + opt.niter_decay + 1):
        epoch_start_time = time.time()
        epoch_iter = 0
        for i, data in enumerate(train_dataset):
            real_A, real_B = data['A'], data['B']
            if real_A.size(0) != real_B.size(0):
                continue
            prior_z_B = real_A.new_empty((real_A.size(0), opt.nlatent, 1, 1)).normal_(0, 1)      # train.py:193
            total_steps += opt.batchSize
            epoch_iter += opt.batchSize
            real_A, real_B, prior_z_B = _shard(real_A, rank, ws), _shard(real_B, rank, ws), _shard(prior_z_B, rank, ws)
            if use_gpu:
                real_A, real_B, prior_z_B = real_A.cuda(), real_B.cuda(), prior_z_B.cuda()
            out = model.train_instance(real_A, real_B, prior_z_B)
            losses, visuals = out[0], out[1]
            gnorms = out[2] if opt.monitor_gnorm else None
            if opt.supervised:
                sd = next(sup_train_dataset)
                sA, sB = _to_dev({'A': _shard(sd['A'], rank, ws), 'B': _shard(sd['B'], rank, ws)}, use_gpu)
                sup_losses = model.supervised_train_instance(sA, sB, prior_z_B[:sA.size(0)])

            if total_steps % opt.display_freq == 0 and rank == 0:
                visualize_cycle(opt, real_A, visuals, epoch, epoch_iter // opt.batchSize, train=True)
                dA, dB = _to_dev(next(dev_cycle), use_gpu)
                dz = dA.new_empty((dA.size(0), opt.nlatent, 1, 1)).normal_(0, 1)
                with torch.no_grad():
                    dev_visuals = model.generate_cycle(dA, dB, dz)
                visualize_cycle(opt, dA, dev_visuals, epoch, epoch_iter // opt.batchSize, train=False)
                visualize_multi(opt, dA, model, epoch, epoch_iter // opt.batchSize)

            if total_steps % opt.print_freq == 0:
                t = (time.time() - print_start_time) / opt.batchSize
                print_log(out_f, format_log(epoch, epoch_iter, losses, t))
                if opt.supervised:
                    print_log(out_f, format_log(epoch, epoch_iter, sup_losses, t, prefix=False))
                if opt.monitor_gnorm:
                    print_log(out_f, format_log(epoch, epoch_iter, gnorms, t, prefix=False) + "\n")
                print_start_time = time.time()

        if epoch % opt.save_epoch_freq == 0 and rank == 0:
            print_log(out_f, 'saving the model at the end of epoch %d, iters %d' % (epoch, total_steps))
            model.save('latest')

        if epoch % opt.eval_A_freq == 0 and rank == 0:
            t = time.time()
            dev_mse_A = eval_mse_A(dev_dataset, model, use_gpu)
            test_mse_A = eval_mse_A(test_dataset, model, use_gpu)
            t = time.time() - t
            history_mse_A.append((dev_mse_A, test_mse_A))
            np.save("%s/history_mse_A" % opt.expr_dir, history_mse_A)
            res = ["[%d] DEV_MSE_A: %.4f, TEST_MSE_A: %.4f, TIME: %.4f" % (epoch, dev_mse_A, test_mse_A, t)]
            if dev_mse_A < results['best_dev_mse_A']:
                with open("%s/best_mse_A.txt" % opt.expr_dir, 'w') as f:
                    f.write(res[0] + '\n')
                results['best_dev_mse_A'], results['best_test_mse_A'] = dev_mse_A, test_mse_A
                model.save('best_A')
                save_results(opt.expr_dir, results)
                res += ["*** BEST DEV A ***"]
            print_log(out_f, "\n".join(["-" * 60] + res + ["-" * 60]))

        if epoch % opt.eval_B_freq == 0 and rank == 0:
            t = time.time()
            steps = 1 if opt.model == 'cycle_gan' else opt.eval_steps
            dev_ubo_B, dev_bpp_B, dev_kld_B = eval_ubo_B(dev_dataset, model, steps, use_gpu)
            test_ubo_B, test_bpp_B, test_kld_B = eval_ubo_B(test_dataset, model, steps, use_gpu)
            t = time.time() - t
            history_ubo_B.append((dev_ubo_B, dev_bpp_B, dev_kld_B, test_ubo_B, test_bpp_B, test_kld_B))
            np.save("%s/history_ubo_B" % opt.expr_dir, history_ubo_B)
            res = ["[%d] DEV_BPP_B: %.4f, TEST_BPP_B: %.4f, TIME: %.4f" % (epoch, dev_bpp_B, test_bpp_B, t)]
            if dev_bpp_B < results['best_dev_bpp_B']:
                with open("%s/best_bpp_B.txt" % opt.expr_dir, 'w') as f:
                    f.write(res[0] + '\n')
                results['best_dev_bpp_B'], results['best_test_bpp_B'] = dev_bpp_B, test_bpp_B
                save_results(opt.expr_dir, results)
                model.save('best_B')
                res += ["*** BEST BPP B ***"]
            print_log(out_f, "\n".join(["-" * 60] + res + ["-" * 60]))

        print_log(out_f, 'End of epoch %d / %d \t Time Taken: %d sec' % (epoch, opt.niter + opt.niter_decay,
                                                                        time.time() - epoch_start_time))
        if epoch > opt.niter:
            model.update_learning_rate()
    if out_f is not None:
        out_f.close()
    return model


if __name__ == "__main__":
    train_model()
