"""GPU test of the data-parallel path: two ranks, each with half of the unpaired minibatch,
gradients averaged by all-reduce BEFORE the per-network clip  ==  one rank on the whole minibatch.
Both ranks share the box's single GPU, so the exchange runs over gloo here (RCCL needs one GPU per rank);
the code path above the backend (flat gradient buckets, ordering w.r.t. clip/Adam, scalar averaging) is the
one bench.py uses with backend "nccl".  StochCycleGAN has no BatchNorm, so the equality is exact up to fp32
summation order; the AugmentedCycleGAN variant (per-rank BatchNorm statistics, as in the reference's own
data_parallel) is checked for losses at step 0 only."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, ws, aug, port, **extra_env):
    out = str(tmp_path / ("dp_ws%d_aug%d%s.npz" % (ws, aug, "_x" if extra_env else "")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(ws), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), out, str(aug)],
                              env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(ws)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return np.load(out)


def test_two_ranks_equal_one_rank_stoch(tmp_path):
    one = _run(tmp_path, 1, 0, 29541)
    two = _run(tmp_path, 2, 0, 29542)
    for st in range(2):
        # losses are batch means -> rank average == global value; P_* monitors likewise
        assert np.allclose(two["s%d/losses" % st], one["s%d/losses" % st], rtol=2e-4, atol=1e-6), st
        # norms are taken AFTER the all-reduce -> identical on every rank and equal to the 1-rank norm
        # (step 1 follows an Adam update, which amplifies the summation-order difference between 1 and 2 ranks)
        assert np.allclose(two["s%d/gnorms" % st], one["s%d/gnorms" % st], rtol=5e-4 if st == 0 else 3e-3, atol=1e-6), st
    for k in ("probe_fake_B", "probe_fake_A"):  # weights after two steps (Adam-amplified rounding, see test_hip_step.py)
        assert np.max(np.abs(two[k] - one[k])) < 2e-2 * np.max(np.abs(one[k])), k


def test_two_ranks_aug_step0_losses(tmp_path):
    one = _run(tmp_path, 1, 1, 29543)
    two = _run(tmp_path, 2, 1, 29544)
    # generator / image-discriminator terms do not involve BatchNorm at step 0 -> equal; the latent terms
    # (Cyc_z_B, KLD_z_B, Cyc_B, D_z_B at indices 3, 4, 7, 8) see per-rank BatchNorm statistics and are only finite.
    idx = [0, 1, 2, 5, 6, 9, 10, 11, 12]  # Cyc_B (7) runs G_A_B on post_z = mu(E): BatchNorm-dependent too
    assert np.allclose(two["s0/losses"][idx], one["s0/losses"][idx], rtol=2e-4, atol=1e-6)
    assert np.all(np.isfinite(two["s1/losses"]))


def test_two_ranks_equal_one_rank_aug_with_syncbn(tmp_path):
    """SyncBN (opt.sync_bn): BatchNorm statistics of E_B / D_z_B over both ranks' shards -> the FULL Augmented CycleGAN
    step on 2 ranks equals 1 rank on the concatenated batch, all 13 losses and all 6 gradient norms."""
    one = _run(tmp_path, 1, 2, 29545)
    two = _run(tmp_path, 2, 2, 29546)
    assert np.allclose(two["s0/losses"], one["s0/losses"], rtol=2e-4, atol=1e-6), (two["s0/losses"], one["s0/losses"])
    assert np.allclose(two["s0/gnorms"][:6], one["s0/gnorms"][:6], rtol=2e-3, atol=1e-6), (two["s0/gnorms"], one["s0/gnorms"])
    assert np.allclose(two["s1/losses"], one["s1/losses"], rtol=1e-2, atol=1e-5)
    for k in ("probe_fake_B", "probe_fake_A"):
        assert np.max(np.abs(two[k] - one[k])) < 1e-2 * np.max(np.abs(one[k])), k


def test_rccl_backend_one_rank_group(tmp_path):
    """The exchange over the backend bench.py uses (torch.distributed "nccl" = RCCL).  RCCL wants one GPU per rank, so on
    this one-GPU box the group has ONE rank and ACGAN_DIST_FORCE=1 makes the step run every collective anyway (parameter
    broadcast, both gradient all-reduces on the flat buffers, SyncBN statistics, the float64 SUM/MIN/MAX of the
    reported scalars): the results must equal the step without a process group."""
    plain = _run(tmp_path, 1, 2, 29547)
    rccl = _run(tmp_path, 1, 2, 29548, ACGAN_DIST_FORCE="1", ACGAN_DP_BACKEND="nccl", RANK="0")
    for k in ("s0/losses", "s0/gnorms"):   # SyncBN forms its statistics from all-reduced sums: rounding-level differences
        assert np.allclose(rccl[k], plain[k], rtol=1e-5, atol=1e-7), k
    for k in ("s1/losses", "s1/gnorms"):   # after one Adam update (amplifies them, see test_hip_step.py)
        assert np.allclose(rccl[k], plain[k], rtol=3e-3, atol=1e-6), k
    for k in ("probe_fake_B", "probe_fake_A"):
        assert np.max(np.abs(rccl[k] - plain[k])) < 1e-2 * np.max(np.abs(plain[k])), k


def test_acg_comm_backend_one_rank_group(tmp_path):
    """The same through the library's own exchange entry points (acg_comm_unique_id / _init / _allreduce_mean / _destroy,
    ACGAN_DP_BACKEND=acg_comm): RCCL bound at run time, the flat gradient buffers averaged on a side stream, gloo as the
    control plane.  One rank (one GPU here), every collective forced."""
    plain = _run(tmp_path, 1, 2, 29549)
    comm = _run(tmp_path, 1, 2, 29550, ACGAN_DIST_FORCE="1", ACGAN_DP_BACKEND="acg_comm", RANK="0")
    for k in ("s0/losses", "s0/gnorms"):
        assert np.allclose(comm[k], plain[k], rtol=1e-5, atol=1e-7), k
    for k in ("s1/losses", "s1/gnorms"):
        assert np.allclose(comm[k], plain[k], rtol=3e-3, atol=1e-6), k


def test_acg_comm_c_abi_one_rank():
    """the four entry points called directly: id, init on the current device, in-place mean of a buffer, destroy"""
    import ctypes
    import torch
    from dtgan_amd import _lib
    blob = (ctypes.c_char * 128)()
    _lib.call("acg_comm_unique_id", ctypes.cast(blob, ctypes.c_void_p))
    assert any(blob.raw)
    comm = ctypes.c_void_p()
    _lib.call("acg_comm_init", ctypes.byref(comm), ctypes.cast(blob, ctypes.c_void_p), 1, 0)
    x = torch.randn(100_003, device="cuda")
    ref = x.clone()
    _lib.call("acg_comm_allreduce_mean", comm, ctypes.c_void_p(x.data_ptr()), x.numel(),
              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    _lib.call("acg_comm_destroy", comm)
    with pytest.raises(_lib.AcgError):
        _lib.call("acg_comm_init", ctypes.byref(comm), ctypes.cast(blob, ctypes.c_void_p), 1, 3)
