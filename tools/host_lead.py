#!/usr/bin/env python3
"""Is the training step host-bound?  Times, per step, how long the host needs to ENQUEUE the whole step (until the one
device->host copy of the scalars) against the step's wall time.  enqueue ~= wall -> the GPU waits for Python.
    python tools/host_lead.py [--steps 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=9)
    ap.add_argument("--precision", default="bf16x3")
    a = ap.parse_args()
    from dtgan_amd import model as M, ops
    ops.set_precision(a.precision)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = M.AugmentedCycleGAN(bench.make_opt(a, 0), testing=True)
    N, S = a.batch, a.size
    A = torch.rand((N, 3, S, S), device=dev) * 2 - 1
    B = torch.rand((N, 3, S, S), device=dev) * 2 - 1
    marks = {}
    orig = M._finish_scalars

    def probe(names, tensors):
        marks["enq"] = time.perf_counter()
        return orig(names, tensors)
    M._finish_scalars = probe
    for i in range(a.steps + 2):
        z = torch.randn((N, 16, 1, 1), device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train_instance(A, B, z)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if i >= 2:
            print("step %d: host enqueue %.1f ms, return %.1f ms, drained %.1f ms" % (i, 1e3 * (marks["enq"] - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t0)))


if __name__ == "__main__":
    main()
