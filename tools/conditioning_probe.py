import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from golden_util import load
from hip_util import t, n, rel, l2rel, load_recipe, precision
from test_hip_nets import build
from dtgan_amd import ops

def run(arr, meta, scale_noise, seed=0):
    net = load_recipe(build(meta), meta["net"], meta["seed"], meta["flavour"]); net.train()
    rs = np.random.RandomState(seed)
    ins, i = [], 0
    while "in%d" % i in arr:
        a = arr["in%d" % i]
        ins.append(t(a * (1 + scale_noise * rs.normal(size=a.shape)), grad=True)); i += 1
    out = net.forward(*ins)
    outs = list(out) if isinstance(out, tuple) else [out]
    loss = sum((o * t(arr["R%d" % j])).sum() for j, o in enumerate(outs)); loss.backward()
    return n(outs[0]), n(ins[0].grad)

def net_probe():
    for name in ("G_A_B_s32_nc1_nb3", "G_B_A_s32_nc1_nb3", "G_A_B_s16_nb9"):
        arr, meta = load(name)
        with precision("f32"):
            o0, g0 = run(arr, meta, 0.0)
            for s in (1e-6, 4e-6):
                for seed in (1, 2):
                    o1, g1 = run(arr, meta, s, seed)
                    print(name, "f32 input noise %.0e seed %d: out rel %.2e  gin l2rel %.2e maxrel %.2e" % (s, seed, rel(o1, o0), l2rel(g1, g0), rel(g1, g0)))
        with precision("bf16x3"):
            o1, g1 = run(arr, meta, 0.0)
            print(name, "bf16x3 vs f32: out rel %.2e gin l2rel %.2e maxrel %.2e | vs golden gin l2rel %.2e" % (rel(o1, o0), l2rel(g1, g0), rel(g1, g0), l2rel(g1, arr["gin0"])))


def step_probe():
    """tests/test_hip_step.py::test_step_against_oracle_with_6_blocks: how far do the step's images move in EXACT fp32
    when the inputs are perturbed by 4e-6 relative (the operand rounding of bf16x3)?"""
    from test_hip_step import make_opt
    from dtgan_amd import model as Mo
    from oracle import recipe
    kw = dict(input_nc=1, output_nc=1, ngf=8, nef=8, ndf=8, nlatent=4, n_blocks=6)
    A, B, z = recipe.inputs(5, 8, 1, 1, 64, 4)

    def run(noise, seed):
        rs = np.random.RandomState(seed)
        m = Mo.AugmentedCycleGAN(make_opt(**kw), testing=True)
        for k, net in m._net_dict().items():
            load_recipe(net, k, 3, "rich")
        pert = lambda a: a * (1 + noise * rs.normal(size=a.shape))
        _, v, _ = m.train_instance(t(pert(A)), t(pert(B)), t(pert(z)))
        return {k: n(v[k]) for k in ("fake_A", "fake_B", "rec_A", "rec_B")}

    with precision("f32"):
        v0 = run(0.0, 0)
        for s in (1, 2, 3):
            v1 = run(4e-6, s)
            print("f32, inputs perturbed 4e-6 (seed %d):" % s, {k: "%.2e" % rel(v1[k], v0[k]) for k in v0})
    with precision("bf16x3"):
        v1 = run(0.0, 0)
        print("bf16x3 vs f32:", {k: "%.2e" % rel(v1[k], v0[k]) for k in v0})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        step_probe()
    else:
        net_probe()
