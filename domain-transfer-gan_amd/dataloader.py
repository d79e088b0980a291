"""Data path — Py3 counterpart of /root/reference/augmented_cyclegan/dataloader.py:13-156 (Python 2 there: `print`
statements, integer `/`, `.next()`).  Same pipeline: npz['data'] -> first 3 channels -> NaN->0 -> per-sample,
per-channel min-max to [-1, 1] -> optional resize -> NHWC->NCHW float32 -> seed-123 shuffle -> 200-sample dev split;
AlignedIterator / UnalignedIterator with the reference's batch-count and last-batch-wrap semantics.

Differences: skimage.transform.resize (not installed) is replaced by an anti-aliased bilinear resize (area-averaging
prefilter when shrinking); Python 3's random.shuffle yields a different permutation than Python 2's for seed 123."""
import os
import random

import numpy as np
import torch

DEV_SIZE = 200


def minmax_scale(arr):
    """dataloader.py:24-25: per sample and channel, (x - min) / (max - min) -> [-1, 1]; constant planes -> 0.  arr: (N,H,W,C)"""
    lo = arr.min(axis=(1, 2))[:, None, None]
    hi = arr.max(axis=(1, 2))[:, None, None]
    with np.errstate(divide='ignore', invalid='ignore'):
        out = -1 + 2 * (arr - lo) / (hi - lo)
    out = np.nan_to_num(out, nan=0.0, posinf=0.0, neginf=0.0)
    return out


def resize_bilinear(arr, size):
    """(N,H,W,C) -> (N,size,size,C), anti-aliased bilinear (replaces skimage.transform.resize, dataloader.py:29-31)"""
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(0, 3, 1, 2))).float()
    t = torch.nn.functional.interpolate(t, size=(size, size), mode='bilinear', align_corners=False, antialias=True)
    return t.numpy().transpose(0, 2, 3, 1)


def prepare(arr, grid_size=None):
    arr = np.asarray(arr)
    if arr.ndim == 3:            # (N,H,W) -> one channel.  (The reference slices [..., :3] first, dataloader.py:17-21,
        arr = arr[..., None]     #  which would clip the WIDTH of a 3-D array; its data is always 4-D, so that path is dead.)
    arr = arr[..., :3]
    arr = np.nan_to_num(arr)
    arr = minmax_scale(arr.astype(np.float64))
    if grid_size is not None and (arr.shape[1] != grid_size or arr.shape[2] != grid_size):
        arr = resize_bilinear(arr, grid_size)
    return np.ascontiguousarray(np.transpose(arr, (0, 3, 1, 2))).astype('float32')


def prepare_on_device(arr, device="cuda", chunk=256):
    """`prepare` without the resize, with the arithmetic on the GPU (acg_minmax_scale_nhwc_to_nchw): the raw array is
    streamed through a pinned staging buffer in chunks of `chunk` samples, the normalised NCHW result stays on the device.
    fp32 throughout (the host path scales in fp64 and rounds once: results agree to ~1e-6)."""
    from . import _lib, ops
    arr = np.asarray(arr)
    if arr.ndim == 3:
        arr = arr[..., None]
    N, H, W, Craw = arr.shape
    C = min(3, Craw)
    out = torch.empty((N, C, H, W), device=device, dtype=torch.float32)
    stage = torch.empty((min(chunk, N), H, W, Craw), dtype=torch.float32).pin_memory()
    dev = torch.empty_like(stage, device=device)
    for i in range(0, N, chunk):
        k = min(chunk, N - i)
        stage[:k].copy_(torch.from_numpy(np.ascontiguousarray(arr[i:i + k], dtype=np.float32)))
        dev[:k].copy_(stage[:k], non_blocking=True)
        _lib.call("acg_minmax_scale_nhwc_to_nchw", ops._ptr(dev), ops._ptr(out[i:i + k]), k, H, W, Craw, C, ops._stream())
        torch.cuda.current_stream().synchronize()      # the staging buffer is reused by the next chunk
    return out


class DevicePrefetcher(object):
    """Wraps an Aligned/UnalignedIterator: batch k+1 travels host -> device through pinned buffers on a side stream while
    step k computes (the reference uploads synchronously right before the step, train.py:198-201).  Yields dicts of
    DEVICE tensors; iteration semantics (lengths, StopIteration, reset) are the wrapped iterator's."""

    def __init__(self, it, device="cuda"):
        self.it, self.device = it, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.pinned = {}
        self.nxt = None

    def __len__(self):
        return len(self.it)

    def _stage(self):
        try:
            batch = next(self.it)
        except StopIteration:
            self.nxt = None
            return
        out = {}
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if not torch.is_tensor(v):      # bookkeeping entries travel as they are
                    out[k] = v
                    continue
                buf = self.pinned.get((k, tuple(v.shape)))
                if buf is None:
                    buf = self.pinned[(k, tuple(v.shape))] = torch.empty(v.shape, dtype=v.dtype).pin_memory()
                buf.copy_(v)
                out[k] = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.nxt = (out, ev)

    def __iter__(self):
        self._stage()
        return self

    def __next__(self):
        if self.nxt is None:
            raise StopIteration
        out, ev = self.nxt
        torch.cuda.current_stream().wait_event(ev)
        for v in out.values():
            if torch.is_tensor(v):
                v.record_stream(torch.cuda.current_stream())
        self.stream.synchronize()       # the pinned buffers are about to be refilled
        self._stage()
        return out

    next = __next__


def split_train_dev(trainA, trainB, shuffle=True):
    """dataloader.py:42-57: fixed-seed shuffle (python RNG state restored), first DEV_SIZE samples become the dev set"""
    if shuffle:
        state = random.getstate()
        random.seed(123)
        idx = list(range(len(trainA)))
        random.shuffle(idx)
        trainA, trainB = trainA[idx], trainB[idx]
        random.setstate(state)
    # the reference always takes DEV_SIZE samples (dataloader.py:53-57) and is left without training data below that;
    # only there (tiny synthetic sets) the split falls back to half / half
    n = DEV_SIZE if len(trainA) > DEV_SIZE else len(trainA) // 2
    return trainA[n:], trainB[n:], trainA[:n], trainB[:n]


def load_numpy_data(root, shuffle=True, grid_size=None):
    def _load(fname):
        print("Loading %s" % os.path.join(root, fname))
        return prepare(np.load(os.path.join(root, fname))['data'], grid_size)
    trainA, trainB, testA, testB = _load("trainA.npz"), _load("trainB.npz"), _load("testA.npz"), _load("testB.npz")
    trainA, trainB, devA, devB = split_train_dev(trainA, trainB, shuffle)
    return trainA, trainB, devA, devB, testA, testB


def synthetic_data(n, nc_a, nc_b, size, seed=0):
    """U(-1,1) fields (every sample/channel of the real loader spans exactly [-1,1] too) — for runs without a dataset"""
    rs = np.random.RandomState(seed)
    mk = lambda k, c: rs.uniform(-1, 1, (k, c, size, size)).astype('float32')
    return mk(n, nc_a), mk(n, nc_b), mk(max(n // 4, 2), nc_a), mk(max(n // 4, 2), nc_b), mk(max(n // 4, 2), nc_a), mk(max(n // 4, 2), nc_b)


class AlignedIterator(object):
    """dataloader.py:61-110: A and B in the SAME order; last batch may be short"""

    def __init__(self, data_A, data_B, **kwargs):
        assert data_A.shape[0] == data_B.shape[0], 'passed data differ in number!'
        self.data_A, self.data_B = data_A, data_B
        self.num_samples = data_A.shape[0]
        self.batch_size = kwargs.get('batch_size', 100)
        self.shuffle = kwargs.get('shuffle', False)
        self.n_batches = self.num_samples // self.batch_size + (1 if self.num_samples % self.batch_size else 0)
        self.reset()

    def __iter__(self):
        return self

    def reset(self):
        self.data_indices = np.random.permutation(self.num_samples) if self.shuffle else np.arange(self.num_samples)
        self.batch_idx = 0

    def __next__(self):
        if self.batch_idx == self.n_batches:
            self.reset()
            raise StopIteration
        idx = self.batch_idx * self.batch_size
        chosen = self.data_indices[idx:idx + self.batch_size]
        self.batch_idx += 1
        return {'A': torch.from_numpy(self.data_A[chosen]), 'B': torch.from_numpy(self.data_B[chosen])}

    next = __next__

    def __len__(self):
        return self.num_samples


class UnalignedIterator(object):
    """dataloader.py:112-156: independent permutations for A and B; the last batch is shifted back to stay full
    (`idx = len - batch_size` when it would run over, dataloader.py:144-145)"""

    def __init__(self, data_A, data_B, **kwargs):
        assert data_A.shape[0] == data_B.shape[0], 'passed data differ in number!'
        self.data_A, self.data_B = data_A, data_B
        self.num_samples = data_A.shape[0]
        self.batch_size = kwargs.get('batch_size', 100)
        self.n_batches = self.num_samples // self.batch_size + (1 if self.num_samples % self.batch_size else 0)
        self.reset()

    def __iter__(self):
        return self

    def reset(self):
        self.data_indices = [np.random.permutation(self.num_samples) for _ in range(2)]
        self.batch_idx = 0

    def __next__(self):
        if self.batch_idx == self.n_batches:
            self.reset()
            raise StopIteration
        idx = self.batch_idx * self.batch_size
        if idx + self.batch_size >= len(self.data_indices[0]):
            idx = max(len(self.data_indices[0]) - self.batch_size, 0)
        ia = self.data_indices[0][idx:idx + self.batch_size]
        ib = self.data_indices[1][idx:idx + self.batch_size]
        self.batch_idx += 1
        return {'A': torch.from_numpy(self.data_A[ia]), 'B': torch.from_numpy(self.data_B[ib])}

    next = __next__

    def __len__(self):
        return self.num_samples
