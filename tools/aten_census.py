#!/usr/bin/env python3
"""GPU box: which torch (aten) operators one training step still launches beside the C ABI, by call site.
A TorchDispatchMode logs every aten call made inside `train_instance` (autograd's backward included) with the innermost
frame of this package on the Python stack; the table says where the step's host glue (fills, adds, copies) comes from.
    python tools/aten_census.py [--batch 4] [--size 64] [--blocks 2]"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import dtgan_amd  # noqa: E402,F401
from dtgan_amd import model as M, ops  # noqa: E402
import bench  # noqa: E402

PKG = os.path.join(ROOT, "domain-transfer-gan_amd")
NO_KERNEL = ("aten.view", "aten.detach", "aten.alias", "aten._unsafe_view", "aten.reshape", "aten.slice", "aten.select",
             "aten.t.", "aten.transpose", "aten.expand", "aten.as_strided", "aten.empty", "aten.unsqueeze", "aten.squeeze",
             "aten.permute", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten._local_scalar_dense", "aten.lift_fresh",
             "aten.new_empty", "aten.empty_like", "aten.split", "aten.unbind", "aten.narrow", "aten.view_as")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        site = "(autograd engine / torch internals)"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if fr.filename.startswith(PKG) or fr.filename.endswith("bench.py"):
                site = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                break
        self.n[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=2)
    a = ap.parse_args()
    b = bench.parse_args(["--batch", str(a.batch), "--size", str(a.size), "--blocks", str(a.blocks), "--no-cpu-baseline"])
    dev = torch.device("cuda", 0)
    ops.set_precision("bf16x3")
    torch.manual_seed(0)
    model = M.AugmentedCycleGAN(bench.make_opt(b, 0), testing=True)
    g = torch.Generator(device=dev); g.manual_seed(1)
    N, S, nc = b.batch, b.size, b.nc
    rA = torch.rand((N, nc, S, S), device=dev, generator=g) * 2 - 1
    rB = torch.rand((N, nc, S, S), device=dev, generator=g) * 2 - 1
    for _ in range(2):
        model.train_instance(rA, rB, torch.randn((N, 16, 1, 1), device=dev, generator=g))
    torch.cuda.synchronize()
    c = Census()
    with c:
        model.train_instance(rA, rB, torch.randn((N, 16, 1, 1), device=dev, generator=g))
    torch.cuda.synchronize()
    launching = [(k, v) for k, v in c.n.items() if not any(k[0].startswith(p) for p in NO_KERNEL)]
    by_op = collections.Counter()
    for (op, _), v in launching:
        by_op[op] += v
    print("aten calls in one step: %d, of which launch-capable: %d" % (sum(c.n.values()), sum(v for _, v in launching)))
    print("\nby operator:")
    for op, v in by_op.most_common():
        print("%5d  %s" % (v, op))
    print("\nby operator and call site:")
    for (op, site), v in sorted(launching, key=lambda kv: -kv[1]):
        print("%5d  %-34s %s" % (v, op, site))


if __name__ == "__main__":
    main()
