"""Worker for tests/test_hip_dp.py: one data-parallel rank of a StochCycleGAN / AugmentedCycleGAN step.
usage: dp_worker.py <out.npz> <aug:0|1|2>   (RANK / WORLD_SIZE / MASTER_* from the environment)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dtgan_amd  # noqa: E402
from dtgan_amd import dist as D, model as M  # noqa: E402
from hip_util import load_recipe, t, n  # noqa: E402
from oracle import recipe  # noqa: E402
from test_hip_step import make_opt  # noqa: E402

out, aug = sys.argv[1], int(sys.argv[2])
backend = os.environ.get("ACGAN_DP_BACKEND", "gloo")   # "nccl": the one-rank RCCL smoke (ACGAN_DIST_FORCE=1)
rank, ws = D.init_from_env(backend) if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or D._FORCE) else (0, 1)
kw = dict(input_nc=3, output_nc=1, ngf=8, nef=8, ndf=8, nlatent=4, n_blocks=2)
# what is tested here is the data-parallel exchange, not the conv arithmetic: exact-fp32 products keep the
# "2 ranks == 1 rank" comparison down to summation order (the bf16x3 default adds 16-bit operand rounding on top)
from dtgan_amd import ops  # noqa: E402
ops.set_precision("f32")
opt = make_opt(**kw)
opt.sync_bn = aug == 2   # aug: 0 = StochCycleGAN, 1 = AugmentedCycleGAN (per-rank BatchNorm), 2 = AugmentedCycleGAN + SyncBN
torch.manual_seed(1 + rank)
m = (M.AugmentedCycleGAN if aug else M.StochCycleGAN)(opt, testing=True)
for k, net in m._net_dict().items():
    load_recipe(net, k, 7, "rich")
GB = 4                                                       # global batch
A, B, z = recipe.inputs(9, GB, 3, 1, 64, 4)
lo, hi = rank * GB // ws, (rank + 1) * GB // ws              # this rank's shard of the unpaired minibatch
res = {}
for st in range(2):
    losses, visuals, gnorms = m.train_instance(t(A[lo:hi]), t(B[lo:hi]), t(z[lo:hi]))
    res["s%d/losses" % st] = np.array(list(losses.values()))
    res["s%d/gnorms" % st] = np.array(list(gnorms.values()))
probe = recipe.inputs(11, 2, 3, 1, 64, 4)
res["probe_fake_B"] = n(m.predict_B(t(probe[0]), t(probe[2])))
res["probe_fake_A"] = n(m.predict_A(t(probe[1])))
if rank == 0:
    np.savez(out, **res)
print("rank", rank, "done")
