#!/usr/bin/env python3
"""GPU box: conditioning of a step fixture's parameter gradients.  Runs step 0 of tests/golden/<fixture> in EXACT fp32 with
the inputs perturbed by 4e-6 relative (the operand rounding of bf16x3) and prints how far a few gradients move, then
bf16x3 against unperturbed fp32 — the measurement behind the by-name skips of tests/test_hip_step.py.
    python tools/step_grad_conditioning.py [fixture]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from golden_util import load, digest
from hip_util import precision, injected_dropout, t
import test_hip_step as T
name = sys.argv[1] if len(sys.argv) > 1 else "step_aug_small_s64_bn_dropout"
arr, meta = load(name)
keys = (("netG_B_A", "model.19.bias"), ("netG_B_A", "model.16.weight"), ("netG_B_A", "model.1.weight"), ("netG_A_B", "model.19.bias"), ("netD_A", "model.0.weight"), ("netD_A", "model.8.weight"), ("netD_B", "model.8.weight"))
def run(prec, noise, seed):
    rs = np.random.RandomState(seed)
    with precision(prec), injected_dropout(meta.get("drop_seed")):
        m = T.build_model(meta)
        A, B, z = (t(arr["s0/%s" % k] * (1 + noise * rs.normal(size=arr["s0/%s" % k].shape))) for k in ("real_A", "real_B", "prior_z_B"))
        m.train_instance(A, B, z)
        return {k: dict(m._net_dict()[k[0]].named_parameters())[k[1]].grad.detach().cpu().numpy().astype(np.float64) for k in keys}
base = run("f32", 0.0, 0)
l2 = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for s in (1, 2, 3):
    p = run("f32", 4e-6, s)
    print("f32 inputs perturbed 4e-6 seed %d:" % s, {"%s/%s" % k: "%.1e" % l2(p[k], base[k]) for k in keys})
x = run("bf16x3", 0.0, 0)
print("bf16x3 vs f32:", {"%s/%s" % k: "%.1e" % l2(x[k], base[k]) for k in keys})
