"""CPU-only tests of the host side: C-ABI exports, module tree / state_dict compatibility with the
reference, flat-parameter plumbing, loud failure without a GPU, and the 2-rank gradient exchange (gloo)."""
import json
import os
import re
import socket
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
    assert lib.acg_version() == _lib.ABI_VERSION
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


def test_seeded_initialisation_equals_the_reference_tensor_for_tensor():
    """networks.py:47 `net.apply(weights_init)` + the constructors' own draws, incl. the double visit of every CINResnetBlock
    convolution (modules.py:145-146): after the same torch.manual_seed the six networks hold the reference's values."""
    from golden_util import digest
    arr, meta = load("init_seeded")
    builders = dict(netG_A_B=lambda: N.define_stochastic_G(16, 3, 3, 32), netG_B_A=lambda: N.define_G(3, 3, 32),
                    netD_A=lambda: N.define_D_A(3, 32, "basic", "instance"), netD_B=lambda: N.define_D_B(3, 64, "basic", "instance"),
                    netD_z_B=lambda: N.define_LAT_D(16, 64), netE_B=lambda: N.define_E(16, 6, 32, "batch"))
    seen = 0
    for k, mk in builders.items():
        torch.manual_seed(meta["seed"])
        for name, v in mk().state_dict().items():
            assert np.array_equal(digest(v.detach().numpy()), arr["%s/%s" % (k, name)]), (k, name)
            seen += 1
    assert seen == len(arr)


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

# ---- PhaseExchange: all-reduce launched from the post-accumulate-grad hooks, in the armed order, overlapped with backward
class Bucket(object):
    def __init__(self, shapes):
        n = sum(int(torch.tensor(s).prod()) for s in shapes)
        self.g = torch.zeros(n + D.SCALAR_TAIL)
        self.gtail = self.g[n:]
        self.params, o = [], 0
        for s in shapes:
            p = torch.nn.Parameter(torch.ones(s)); k = p.numel()
            p.grad = self.g[o:o + k].view(s); o += k
            self.params.append(p)
        D.hook_params(self)
b1, b2 = Bucket([(3,), (2, 2)]), Bucket([(5,)])
ex = D.PhaseExchange("test")
trace = []
orig_launch = ex._launch
ex._launch = lambda i: (trace.append((i, [int(p.grad.abs().sum() > 0) for b in (b1, b2) for p in b.params])), orig_launch(i))[1]
def step(use_second_param=True):
    for b in (b1, b2):
        b.g.zero_()
    x = torch.full((1,), float(r + 1))
    # build order b1 then b2 -> backward completes b2 first; armed order says so
    y1 = (b1.params[0] * x).sum() + ((b1.params[1] * x).sum() if use_second_param else 0.0)
    y2 = (b2.params[0] * x).sum() * 2.0
    D.write_scalar_tail(b1.gtail, [torch.tensor(float(r)), torch.tensor(10.0 * r)], [torch.tensor(float(r)), torch.tensor(-float(r))])
    ex.arm([b2, b1])
    try:
        (y1 + y2).backward()
    finally:
        ex.flush()
    ex.wait(b1); ex.wait(b2)
del trace[:]; step()
assert [t[0] for t in trace] == [0, 1] and all(sum(t[1]) == 3 for t in trace), trace     # first step: nothing known -> flush launches
mean_x = (1 + ws) / 2.0
assert torch.allclose(b1.g[:7], torch.full((7,), mean_x)) and torch.allclose(b2.g[:5], torch.full((5,), 2 * mean_x))
sums, mm = D.read_scalar_tail(b1.gtail, 2, 2)
assert torch.allclose(sums, torch.tensor([(ws - 1) / 2.0, 10.0 * (ws - 1) / 2.0]))
assert float(mm[:, 0].max()) == ws - 1 and float(mm[:, 1].min()) == -(ws - 1) and mm.shape == (ws, 2)
del trace[:]; step()
# second step: bucket b2 (index 0 in the armed order) is launched as soon as its one gradient is written, i.e. BEFORE b1's
assert trace[0][0] == 0 and trace[0][1][:2] == [0, 0], trace
assert torch.allclose(b1.g[:7], torch.full((7,), mean_x)) and torch.allclose(b2.g[:5], torch.full((5,), 2 * mean_x))
# a graph that changes between steps (one gradient fewer, then one more) must not reduce partial sums silently
step(use_second_param=False)
try:
    step(use_second_param=True)
    raise SystemExit("late gradient not detected")
except RuntimeError as e:
    assert "arrived after" in str(e)
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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without RANK in the environment (how the driver runs it) starts two fresh ranks itself,
    relays rank 0's line and reports failure of any rank.  --dry-run: rendezvous + one all-reduce over gloo, no GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                                  # ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    # a rank that fails (no GPU here) makes the launcher exit non-zero instead of hanging in a collective
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    if not torch.cuda.is_available():
        assert bad.returncode != 0 and "rank exit codes" in bad.stderr


_WORKER8 = """
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as td
import dtgan_amd
from dtgan_amd import dist as D
r, ws = D.init_from_env(backend="gloo")
assert ws == D.MAX_RANKS == 8
# a flat gradient buffer with its scalar tail: gradients rank-dependent, 13 reported sums, 4 per-rank monitors
n = 1000
buf = torch.full((n + D.SCALAR_TAIL,), float(r + 1))
tail = buf[n:]
D.write_scalar_tail(tail, [torch.tensor(float(r * 10 + i)) for i in range(13)], [torch.tensor(float(100 * r + j)) for j in range(4)])
D.allreduce_mean_([buf])
assert torch.allclose(buf[:n], torch.full((n,), (ws + 1) / 2.0))
sums, mm = D.read_scalar_tail(tail, 13, 4)
want = torch.tensor([sum(q * 10 + i for q in range(ws)) / ws for i in range(13)])
assert torch.allclose(sums, want), (sums, want)
assert mm.shape == (8, 4)
for q in range(ws):                       # every rank's slot arrives intact on every rank (the last one ends the tail exactly)
    assert torch.allclose(mm[q], torch.tensor([100.0 * q + j for j in range(4)])), (q, mm[q])
assert D.SUM_SLOTS + 4 * (ws - 1) + 4 == D.SCALAR_TAIL
print("rank", r, "ok")
"""


def test_eight_rank_scalar_tail_and_bench_dry_run_gloo(tmp_path):
    """config 4's rank count on the CPU: the scalar tail behind the gradient buffers is sized for MAX_RANKS = 8 — eight gloo
    ranks fill every per-rank monitor slot and read all of them back after the averaging collective; and `bench.py --gpus 8
    --dry-run` (the driver's launch shape for the 8-GPU line) brings up eight self-launched ranks and prints ONE line."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8 % dict(root=ROOT))
    with socket.socket() as sk:   # a free port of this host (a fixed one collides when two test runs share it)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="8", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(8)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["dry_run"] is True and line["config"]["parallelism"] == "dp8"


def test_phase_exchange_async_handles_complete_out_of_order(monkeypatch):
    """The shape of the RCCL path (dist._allreduce_avg_async with backend "nccl": async handles returned from hooks that
    fire inside backward(), waited for later, per network) with a fake transport whose collectives complete OUT OF ORDER on
    another thread: launches must still be issued in the armed order from inside backward(), wait(bucket) must wait for
    that bucket's own handle only, and a gradient buffer must hold the reduced value exactly once after its wait."""
    import threading
    import time
    from dtgan_amd import dist as D

    class Work(object):
        def __init__(self, buf, idx):
            self.buf, self.idx, self.ev, self.waited = buf, idx, threading.Event(), False

        def wait(self):
            self.waited = True
            assert self.ev.wait(timeout=30), "collective %d never completed" % self.idx
            return True

    launched, completed, works = [], [], []
    lock = threading.Lock()

    def network():           # completes the LATEST launched collective first, each after a delay
        done = 0
        while done < want[0]:
            with lock:
                pend = [w for w in works if not w.ev.is_set()]
            if len(pend) < min(2, want[0] - done) and not stop.is_set():   # hold back until two are in flight
                time.sleep(0.002)
                continue
            w = pend[-1]
            time.sleep(0.01)
            w.buf.mul_(3.0)   # the "all-reduce": a marker that must be applied exactly once
            completed.append(w.idx)
            w.ev.set()
            done += 1

    def fake_async(buf):
        w = Work(buf, len(launched))
        launched.append((len(launched), in_backward[0]))
        with lock:
            works.append(w)
        return w

    monkeypatch.setattr(D, "_allreduce_avg_async", fake_async)

    class Bucket(object):
        def __init__(self, shapes):
            n = sum(int(torch.tensor(s).prod()) for s in shapes)
            self.g = torch.zeros(n + D.SCALAR_TAIL)
            self.params, o = [], 0
            for s in shapes:
                p = torch.nn.Parameter(torch.ones(s)); k = p.numel()
                p.grad = self.g[o:o + k].view(s); o += k
                self.params.append(p)
            D.hook_params(self)

    b1, b2, b3 = Bucket([(3,), (2, 2)]), Bucket([(5,)]), Bucket([(4,), (1,)])
    ex = D.PhaseExchange("async-test")
    in_backward, want, stop = [False], [3], threading.Event()
    for step in range(3):
        for b in (b1, b2, b3):
            b.g.zero_()
        del launched[:], completed[:], works[:]
        stop.clear()
        th = threading.Thread(target=network)
        th.start()
        x = torch.full((1,), 2.0)
        loss = sum((p * x).sum() for b in (b1, b2, b3) for p in b.params)
        ex.arm([b3, b2, b1])                       # completion order of the gradients: the reverse of the build order
        in_backward[0] = True
        try:
            loss.backward()
        finally:
            in_backward[0] = False
            ex.flush()
        stop.set()
        assert [i for i, _ in launched] == [0, 1, 2]
        if step > 0:                               # from the second step on every bucket is launched by its hooks, inside backward()
            assert all(inside for _, inside in launched), launched
        # wait for the FIRST-armed bucket only: it completes last in this transport, the others may or may not be done
        ex.wait(b3)
        assert works[0].waited and works[0].ev.is_set()
        assert torch.equal(b3.g[:5], torch.full((5,), 6.0))
        assert not works[1].waited and not works[2].waited
        ex.wait(b1); ex.wait(b2)
        th.join(timeout=30)
        assert completed[0] != 0, completed        # the transport really did complete out of order
        assert torch.equal(b1.g[:7], torch.full((7,), 6.0)) and torch.equal(b2.g[:5], torch.full((5,), 6.0))
        ex.wait(b1)                                # a second wait is a no-op (handle dropped): the marker is not re-applied
        assert torch.equal(b1.g[:7], torch.full((7,), 6.0))
