"""GPU end-to-end test of the Py3 training driver: two epochs on synthetic data through
options -> data iterators -> AugmentedCycleGAN.train_instance (HIP path) -> evaluation -> artefacts, then --continue_train."""
import json
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_train_driver_end_to_end(tmp_path):
    import torch
    from dtgan_amd.train import train_model
    args = ["--name", "e2e", "--checkpoints_dir", str(tmp_path), "--synthetic", "24", "--grid_size", "64", "--batchSize", "4",
            "--ngf", "8", "--nef", "8", "--ndf", "8", "--nlatent", "4", "--niter", "1", "--niter_decay", "1", "--print_freq", "8",
            "--display_freq", "16", "--save_epoch_freq", "1", "--eval_steps", "2", "--num_multi", "3", "--seed", "1",
            "--supervised", "--sup_frac", "0.5"]
    train_model(args)
    d = os.path.join(str(tmp_path), "e2e")
    for f in ("opt.txt", "opt.pkl", "nets.txt", "results.txt", "results.json", "latest", "best_A", "best_B", "history_mse_A.npy",
              "history_ubo_B.npy", "best_mse_A.txt", "best_bpp_B.txt", "vis_latest/cycle.png", "vis_latest/multi.png"):
        assert os.path.exists(os.path.join(d, f)), f
    log = open(os.path.join(d, "results.txt")).read()
    assert "#training images = 24" in log and "model [AugmentedCycleGAN] was created" in log
    # loss line exactly in the reference's format and key order (train.py:39-45, model.py:518-523)
    m = re.search(r"^\(epoch: 1, iters: 8, time: \d+\.\d{3}\) D_A: [-\d.]+ G_A: [-\d.]+ Cyc_A: [-\d.]+ Cyc_z_B: [-\d.]+ KLD_z_B: [-\d.]+ "
                  r"D_B: [-\d.]+ G_B: [-\d.]+ Cyc_B: [-\d.]+ D_z_B: [-\d.]+ P_t_A: [-\d.]+ P_f_A: [-\d.]+ P_t_B: [-\d.]+ P_f_B: [-\d.]+ $",
                  log, re.M)
    assert m, log[:2000]
    assert re.search(r"^ +S_A: [-\d.]+ S_B: [-\d.]+ KLD_z_B: ", log, re.M)            # supervised continuation line
    assert re.search(r"^ +gnorm_G_A_B: [-\d.]+ gnorm_G_B_A: [-\d.]+ gnorm_E_B: ", log, re.M)
    assert re.search(r"^\[2\] DEV_MSE_A: \d+\.\d{4}, TEST_MSE_A: \d+\.\d{4}, TIME: ", log, re.M)
    assert re.search(r"^\[1\] DEV_BPP_B: [-\d.]+, TEST_BPP_B: [-\d.]+, TIME: ", log, re.M)
    assert "End of epoch 2 / 2" in log
    res = json.load(open(os.path.join(d, "results.json")))
    assert set(res) == {"best_dev_mse_A", "best_test_mse_A", "best_dev_bpp_B", "best_test_bpp_B"} and res["best_dev_mse_A"] < 10
    assert np.load(os.path.join(d, "history_mse_A.npy")).shape == (2, 2)
    assert np.load(os.path.join(d, "history_ubo_B.npy")).shape == (2, 6)
    ck = torch.load(os.path.join(d, "latest"), map_location="cpu")
    assert "netG_A_B" in ck and "optimizer_G_B" in ck
    # resume: one more epoch starting from `latest`
    train_model(args + ["--continue_train", "--epoch_count", "2"])
    log2 = open(os.path.join(d, "results.txt")).read()
    assert "continue_train: loaded" in log2 and log2.count("End of epoch 2 / 2") == 2


def test_train_driver_with_step_graph(tmp_path):
    """train.py --step_graph: the unsupervised step replayed as one captured HIP graph between the driver's eager
    visualisation / evaluation forwards, across a learning-rate change (re-capture)"""
    from dtgan_amd.train import train_model
    args = ["--name", "graph", "--checkpoints_dir", str(tmp_path), "--synthetic", "24", "--grid_size", "64", "--batchSize", "4",
            "--ngf", "8", "--nef", "8", "--ndf", "8", "--nlatent", "4", "--niter", "1", "--niter_decay", "2", "--print_freq", "4",
            "--display_freq", "12", "--save_epoch_freq", "3", "--eval_steps", "2", "--num_multi", "2", "--seed", "1", "--step_graph"]
    train_model(args)
    log = open(os.path.join(str(tmp_path), "graph", "results.txt")).read()
    vals = [float(v) for v in re.findall(r" D_A: ([-\d.]+(?:e-?\d+)?|nan) ", log)]
    assert len(vals) >= 9 and all(np.isfinite(vals)) and "End of epoch 3 / 3" in log, log[-1500:]
    assert all(0.0 < v < 2.0 for v in vals)


def test_two_rank_training_driver_with_syncbn(tmp_path):
    """train.py under data parallelism: 2 ranks (gloo, sharing this box's GPU), --sync_bn, PNG dumps and the per-epoch
    evaluation on rank 0 only.  Those rank-0-only forwards run BatchNorm in train mode (the reference never calls eval());
    with SyncBN they must NOT post collectives the other rank never joins (they use local statistics outside a training
    step) — otherwise this run hangs or pairs a statistics all-reduce with the next step's gradient all-reduce."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--name", "dp", "--checkpoints_dir", str(tmp_path), "--synthetic", "16", "--grid_size", "64", "--batchSize", "4",
            "--ngf", "8", "--nef", "8", "--ndf", "8", "--nlatent", "4", "--niter", "1", "--niter_decay", "1", "--print_freq", "4",
            "--display_freq", "8", "--save_epoch_freq", "1", "--eval_steps", "2", "--num_multi", "2", "--seed", "1", "--sync_bn"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", WORLD_SIZE="2", ACGAN_DP_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    procs = [subprocess.Popen([sys.executable, "-c", "import dtgan_amd; from dtgan_amd.train import train_model; import sys; "
                               "train_model(sys.argv[1:])"] + args, env=dict(env, RANK=str(r), LOCAL_RANK="0"), cwd=root,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("2-rank train.py hung (a rank-0-only forward posted a collective?)")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    d = os.path.join(str(tmp_path), "dp")
    log = open(os.path.join(d, "results.txt")).read()
    assert "2 ranks" in log and "End of epoch 2 / 2" in log
    assert re.search(r"^\[2\] DEV_MSE_A: ", log, re.M) and re.search(r"^\[2\] DEV_BPP_B: ", log, re.M)
    for f in ("vis_latest/cycle.png", "vis_latest/multi.png", "latest", "best_A"):
        assert os.path.exists(os.path.join(d, f)), f


def test_device_side_data_path(tmp_path):
    """dataloader.prepare_on_device (min-max normalisation on the GPU, acg_minmax_scale_nhwc_to_nchw) equals the host
    pipeline that the reference-derived fixture pins; DevicePrefetcher yields the wrapped iterator's batches."""
    import torch
    from golden_util import load
    from dtgan_amd import dataloader as DL
    arr, _ = load("data_pipeline")
    for k in ("trainA", "trainB", "testB"):
        host = DL.prepare(arr["raw/" + k])
        dev = DL.prepare_on_device(arr["raw/" + k], chunk=64).cpu().numpy()
        assert dev.shape == host.shape and np.max(np.abs(dev - host)) < 2e-6, k
    A = np.arange(40, dtype=np.float32).reshape(10, 1, 2, 2)
    np.random.seed(3)
    ref = [(b["A"].clone(), b["B"].clone()) for b in DL.UnalignedIterator(A, -A, batch_size=4)]
    np.random.seed(3)
    pf = DL.DevicePrefetcher(DL.UnalignedIterator(A, -A, batch_size=4))
    for epoch in range(2):
        got = [(b["A"], b["B"]) for b in pf]
        assert len(got) == 3 and all(g[0].is_cuda for g in got)
        if epoch == 0:
            for (ga, gb), (ra, rb) in zip(got, ref):
                assert torch.equal(ga.cpu(), ra) and torch.equal(gb.cpu(), rb)
