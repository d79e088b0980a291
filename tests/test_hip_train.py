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
