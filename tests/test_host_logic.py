"""CPU-only tests of the host side: C-ABI exports, module tree / state_dict compatibility with the
reference, flat-parameter plumbing, loud failure without a GPU, and the 2-rank gradient exchange (gloo)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import dtgan_amd  # noqa: E402
from dtgan_amd import _lib, networks as N  # noqa: E402
from golden_util import load  # noqa: E402


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "acgan_hip.h")).read()
    declared = set(re.findall(r"\b(acg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"acg_status", "acg_act", "acg_pad_mode", "acg_conv_impl", "acg_conv_desc"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()  # binds every symbol; AttributeError if one is missing
    assert lib.acg_version() == 100
    assert lib.acg_ncols_pad(16) == 32 and lib.acg_ncols_pad(64) == 64 and lib.acg_ncols_pad(256) == 256


def test_state_dict_keys_match_reference_fixture():
    """state_dict keys + shapes (incl. the aliased CINResnetBlock keys) equal the reference's"""
    arr, _ = load("statedict_keys")
    ref = json.loads(bytes(arr["keys_json"]).decode())
    nets = dict(netG_A_B=N.define_stochastic_G(16, 3, 3, 32), netG_B_A=N.define_G(3, 3, 32),
                netD_A=N.define_D_A(3, 32, "basic", "instance"), netD_B=N.define_D_B(3, 64, "basic", "instance"),
                netD_z_B=N.define_LAT_D(16, 64), netE_B=N.define_E(16, 6, 32, "batch"))
    for k, net in nets.items():
        got = [[a, list(b.shape)] for a, b in net.state_dict().items()]
        assert got == ref[k], k


def test_n_blocks_is_honoured_and_init_follows_reference_distributions():
    g3, g9 = N.define_G(3, 3, 8), N.define_G(3, 3, 8, n_blocks=9)
    assert len(g9.state_dict()) - len(g3.state_dict()) == 6 * 6
    w = N.define_G(3, 3, 32).model[4].weight
    assert abs(float(w.std()) - 0.02) < 2e-3 and abs(float(w.mean())) < 2e-3
    assert float(N.define_G(3, 3, 8).model[4].bias.abs().max()) == 0.0
    e = N.define_E(16, 6, 32, "batch")
    assert abs(float(e.conv_modules[3].weight.mean()) - 1.0) < 0.02


def test_no_cpu_fallback():
    """the product path must fail loudly without a ROCm device"""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = N.define_G(3, 3, 8)
    with pytest.raises(_lib.AcgError):
        g(torch.zeros(1, 3, 16, 16))


def test_drop_in_module_names():
    code = ("import sys; sys.path.insert(0, %r); import model, networks, modules; "
            "assert hasattr(model, 'AugmentedCycleGAN') and hasattr(model, 'StochCycleGAN') and hasattr(model, 'kld_std_guss'); "
            "assert hasattr(networks, 'define_stochastic_G') and hasattr(networks, 'define_LAT_D') and hasattr(networks, 'define_D'); "
            "assert hasattr(modules, 'CondInstanceNorm') and hasattr(modules, 'TwoInputSequential'); print('ok')"
            % os.path.join(ROOT, "domain-transfer-gan_amd", "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_flat_parameter_views_and_fused_adam_state_dict():
    from dtgan_amd.model import FlatNet, FusedAdam
    net = N.define_LAT_D(4, 8)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    f = FlatNet(net)
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    p0 = f.params[0]
    assert p0.data_ptr() == f.p.data_ptr() and p0.grad.data_ptr() == f.g.data_ptr()
    f.g.fill_(1.0)
    assert all(float(p.grad.min()) == 1.0 for p in f.params)
    f.zero_grad()
    assert all(float(p.grad.abs().max()) == 0.0 for p in f.params)
    opt = FusedAdam([f], 1e-3, (0.5, 0.999))
    sd = opt.state_dict()
    ref = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.5, 0.999)).state_dict()
    assert sd["param_groups"][0]["params"] == ref["param_groups"][0]["params"]
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    opt.load_state_dict(sd)
    f.check()


_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as td
import dtgan_amd
from dtgan_amd import dist as D, networks as N
from dtgan_amd.model import FlatNet
D.init_from_env("gloo")
r, ws = D.rank(), D.world_size()
torch.manual_seed(100 + r)                      # different init per rank on purpose
net = N.define_LAT_D(4, 8)
D.broadcast_params_([net])                      # -> identical replicas
f = FlatNet(net)
chk = f.p.clone(); td.all_reduce(chk)
assert torch.allclose(chk, f.p * ws), "broadcast failed"
f.g.fill_(float(r + 1))                         # rank-dependent gradients
D.allreduce_mean_([f.g])
assert torch.allclose(f.g, torch.full_like(f.g, (1 + ws) / 2.0)), f.g[:4]
from collections import OrderedDict
v = D.average_scalars(OrderedDict(a=float(r), lo=float(r), hi=float(r)), min_keys=["lo"], max_keys=["hi"])
assert abs(v["a"] - (ws - 1) / 2.0) < 1e-12 and v["lo"] == 0.0 and v["hi"] == ws - 1
print("rank", r, "ok")
"""


def test_two_rank_gradient_exchange_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)
