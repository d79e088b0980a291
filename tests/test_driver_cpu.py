"""CPU tests of the driver side: option defaults / opt.txt, log-line format, data pipeline,
iterator semantics (incl. the last-batch wrap), PNG encoder."""
import os
import pickle
import struct
import zlib

import numpy as np
import torch

import dtgan_amd  # noqa: F401
from dtgan_amd import dataloader as DL, options as O


def test_option_defaults_and_files(tmp_path):
    opt = O.TrainOptions().parse(argv=["--name", "exp", "--checkpoints_dir", str(tmp_path), "--synthetic", "8", "--gpu_ids", "-1"],
                                 sub_dirs=["vis_multi"])
    ref = dict(input_nc=3, output_nc=3, grid_size=256, batchSize=32, niter=25, niter_decay=25, beta1=0.5, lr=2e-4, ngf=32,
               nef=32, ndf=64, nlatent=16, norm="instance", max_gnorm=500.0, stoch_enc=False, z_gan=1, enc_A_B=1, no_lsgan=False,
               lambda_A=1.0, lambda_B=1.0, lambda_z_B=0.025, lambda_sup_A=0.1, lambda_sup_B=0.1, monitor_gnorm=True,
               model="aug_cycle_gan", print_freq=100, display_freq=5000, save_epoch_freq=5, num_multi=10, sup_frac=0.1)
    for k, v in ref.items():                                               # options.py:20-85 as recorded in SURVEY §5
        assert getattr(opt, k) == v, k
    assert opt.gpu_ids == [] and opt.expr_dir == os.path.join(str(tmp_path), "exp")
    txt = open(os.path.join(opt.expr_dir, "opt.txt")).read().splitlines()
    assert txt[0] == "------------ Options -------------" and txt[-1] == "-------------- End ----------------"
    assert "batchSize: 32" in txt and txt[1:-1] == sorted(txt[1:-1])
    assert pickle.load(open(os.path.join(opt.expr_dir, "opt.pkl"), "rb"))["ngf"] == 32
    assert os.path.isdir(opt.vis_multi)


def test_format_log_matches_reference_format():
    from collections import OrderedDict
    from dtgan_amd.train import format_log
    line = format_log(3, 640, OrderedDict([("D_A", 0.25), ("G_A", 1.23456)]), 0.0123)
    assert line == "(epoch: 3, iters: 640, time: 0.012) D_A: 0.250 G_A: 1.235 "   # train.py:39-45
    cont = format_log(3, 640, OrderedDict([("gnorm_G_A_B", 2.0)]), 0.0123, prefix=False)
    assert cont == " " * len("(epoch: 3, iters: 640, time: 0.012) ") + "gnorm_G_A_B: 2.000 "


def test_data_pipeline_and_iterators():
    rs = np.random.RandomState(0)
    raw = rs.normal(5, 3, (10, 12, 12, 4))
    raw[0, 0, 0, 0] = np.nan
    raw[1, :, :, 2] = 7.0                                                  # constant plane -> 0 after min-max
    x = DL.prepare(raw)
    assert x.shape == (10, 3, 12, 12) and x.dtype == np.float32            # first 3 channels, NCHW
    assert np.allclose(x[2:].max(axis=(2, 3)), 1) and np.allclose(x[2:].min(axis=(2, 3)), -1)
    assert np.all(x[1, 2] == 0)
    assert DL.prepare(raw, grid_size=8).shape == (10, 3, 8, 8)
    assert DL.prepare(raw[..., 0]).shape == (10, 1, 12, 12)                # 3-D input gets a channel axis
    A = np.arange(10, dtype=np.float32).reshape(10, 1, 1, 1); B = -A
    al = DL.AlignedIterator(A, B, batch_size=4)
    bs = [b["A"].shape[0] for b in al]
    assert bs == [4, 4, 2] and len(al) == 10
    assert [b["A"].shape[0] for b in al] == [4, 4, 2]                      # resets after StopIteration
    for b in DL.AlignedIterator(A, B, batch_size=4):
        assert torch.equal(b["A"], -b["B"])
    un = DL.UnalignedIterator(A, B, batch_size=4)
    batches = list(un)
    assert [b["A"].shape[0] for b in batches] == [4, 4, 4]                 # last batch shifted back (dataloader.py:144-145)
    assert un.n_batches == 3
    tA, tB, dA, dB = DL.split_train_dev(np.arange(1000)[:, None], np.arange(1000)[:, None])
    assert len(dA) == 200 and len(tA) == 800 and np.array_equal(tA, tB)    # same permutation for A and B
    assert not np.array_equal(dA[:, 0], np.arange(200))


def test_data_pipeline_matches_reference_fixture(tmp_path):
    """tests/golden/data_pipeline.npz was produced by the reference's own dataloader.py (executed from its text under
    Python 3, tools/make_goldens.py data_case): load_numpy_data on raw npz files with a NaN, a constant plane and an all-zero
    sample (dataloader.py:13-59), and the two iterators (dataloader.py:61-156) under a fixed numpy seed."""
    from golden_util import load
    arr, _ = load("data_pipeline")
    root = str(tmp_path)
    for k in ("trainA", "trainB", "testA", "testB"):
        np.savez(os.path.join(root, k + ".npz"), data=arr["raw/" + k])
    out = DL.load_numpy_data(root, shuffle=False, grid_size=None)
    for k, v in zip(("trainA", "trainB", "devA", "devB", "testA", "testB"), out):
        ref = arr["out/" + k]
        assert v.shape == ref.shape and v.dtype == ref.dtype, k
        assert np.array_equal(v, ref), (k, np.abs(v - ref).max())
    A = np.arange(10, dtype=np.float32).reshape(10, 1, 1, 1)
    np.random.seed(11)
    un = DL.UnalignedIterator(A, -A, batch_size=4)
    got = [np.stack([b["A"].numpy().ravel(), b["B"].numpy().ravel()]) for _ in range(2) for b in un]
    assert np.array_equal(np.stack(got), arr["unaligned_batches"])
    assert [b["A"].shape[0] for b in DL.AlignedIterator(A, -A, batch_size=4)] == list(arr["aligned_sizes"])


def test_png_writer(tmp_path):
    from dtgan_amd.train import save_image_grid
    p = str(tmp_path / "g.png")
    save_image_grid(torch.rand(5, 3, 6, 4) * 2 - 1, p, nrow=3)
    b = open(p, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    w, h = struct.unpack(">II", b[16:24])
    assert (w, h) == (3 * 6 + 2, 2 * 8 + 2)
    idat = b[b.index(b"IDAT") + 4: b.index(b"IEND") - 8]
    assert len(zlib.decompress(idat)) == h * (1 + 3 * w)
