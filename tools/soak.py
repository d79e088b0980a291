#!/usr/bin/env python3
"""Soak test (GPU box): N training steps of the full-width model at 128x128, checking that losses stay finite, the step is
reproducible from a fixed state, and device memory does not grow.   python tools/soak.py [--steps 200]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dtgan_amd  # noqa: E402, F401
from dtgan_amd import model as M  # noqa: E402
from bench import make_opt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--size", type=int, default=128, help="image side (256: the whole-row-tile kernels of the trunk)")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    cfg = argparse.Namespace(nc=3, blocks=6, sync_bn=False)
    torch.manual_seed(0)
    m = M.AugmentedCycleGAN(make_opt(cfg, 0), testing=True)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    mem = []
    for s in range(a.steps):
        A = torch.rand((a.batch, 3, a.size, a.size), device="cuda", generator=g) * 2 - 1
        B = torch.rand((a.batch, 3, a.size, a.size), device="cuda", generator=g) * 2 - 1
        z = torch.randn((a.batch, 16, 1, 1), device="cuda", generator=g)
        losses, _, gn = m.train_instance(A, B, z)
        assert all(v == v and abs(v) < 1e6 for v in losses.values()), (s, losses)
        assert all(v == v for v in gn.values()), (s, gn)
        if s % 20 == 0:
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
            print("step %4d  D_A %.4f G_A %.4f Cyc_A %.4f Cyc_B %.4f  gnorm_G_A_B %.3f  mem %.1f MB"
                  % (s, losses["D_A"], losses["G_A"], losses["Cyc_A"], losses["Cyc_B"], gn["gnorm_G_A_B"], mem[-1] / 1e6), flush=True)
    assert max(mem[1:]) <= mem[1] * 1.01 + 1e6, mem     # steady state after the first steps
    print("soak ok: %d steps, losses finite, memory steady (%.1f MB)" % (a.steps, mem[-1] / 1e6))


if __name__ == "__main__":
    main()
