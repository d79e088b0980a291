import os, sys
os.environ["ACG_DEBUG_SWITCHES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dtgan_amd
from dtgan_amd import networks as Nw, ops
from hip_util import load_recipe, rel
ops.set_precision("bf16x3")
torch.manual_seed(0)
x = (torch.rand(4, 3, 64, 64) * 2 - 1).cuda(); z = torch.randn(4, 4, 1, 1).cuda()
nets = {"G_A_B": (Nw.define_stochastic_G(4, 3, 3, 8, gpu_ids=[0], n_blocks=3), "netG_A_B"),
        "G_B_A": (Nw.define_G(3, 3, 8, gpu_ids=[0], n_blocks=3), "netG_B_A"),
        "D_A": (Nw.define_D_A(3, 8, "basic", "instance", False, [0]), "netD_A"), "D_B": (Nw.define_D_B(3, 8, "basic", "instance", False, [0]), "netD_B")}
for name, (net, key) in nets.items():
    load_recipe(net, key, 1, "rich")
    outs = {}
    for sw in (0, 1):
        if sw: os.environ["ACG_NO_GENERIC_STATS"] = "1"
        else: os.environ.pop("ACG_NO_GENERIC_STATS", None)
        calls = []
        real = ops._lib.call
        def spy(n_, *a):
            calls.append(n_); return real(n_, *a)
        ops._lib.call = spy
        xi = x.clone().requires_grad_(True)
        y = net(xi, z) if name == "G_A_B" else net(xi)
        y.square().sum().backward()
        ops._lib.call = real
        outs[sw] = (y.detach().cpu().numpy(), xi.grad.cpu().numpy(), calls.count("acg_conv2d_fwd_stats"), calls.count("acg_norm_stats"))
    print(name, "fwd rel %.2e  dgrad rel %.2e   stats-convs %d vs %d, stats passes %d vs %d" % (
        rel(outs[0][0], outs[1][0]), rel(outs[0][1], outs[1][1]), outs[0][2], outs[1][2], outs[0][3], outs[1][3]))
