#!/usr/bin/env python3
"""GPU box: do two independent generator passes overlap on two HIP streams?  (the trunk convolutions are MFMA / vector-memory
bound, the norm passes between them HBM bound: side by side they could fill each other's idle pipe.)  Times G_A_B(real_A, z)
and G_B_A(real_B) — forward only, then forward + backward — back to back on one stream and side by side on two.
    python tools/two_stream_probe.py [--batch 32] [--size 256] [--iters 5]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dtgan_amd  # noqa: E402,F401
from dtgan_amd import networks, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=9)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    ops.set_precision("bf16x3")
    torch.manual_seed(0)
    N, S = a.batch, a.size
    g_ab = networks.define_stochastic_G(nlatent=16, input_nc=3, output_nc=3, ngf=32, norm='instance', which_model_netG='resnet',
                                         use_dropout=False, gpu_ids=[0], n_blocks=a.blocks)
    g_ba = networks.define_G(3, 3, 32, norm='instance', which_model_netG='resnet', use_dropout=False, gpu_ids=[0], n_blocks=a.blocks)
    xa = torch.randn(N, 3, S, S, device="cuda"); xb = torch.randn(N, 3, S, S, device="cuda")
    z = torch.randn(N, 16, 1, 1, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def f_ab():
        try:
            return g_ab(xa, z)
        except TypeError:
            return g_ab(xa)

    def f_ba():
        return g_ba(xb)

    def run(two, grad):
        main_s = torch.cuda.current_stream()
        if two:
            s1.wait_stream(main_s); s2.wait_stream(main_s)
            with torch.cuda.stream(s1):
                y1 = f_ab()
                if grad:
                    y1.square().mean().backward()
            with torch.cuda.stream(s2):
                y2 = f_ba()
                if grad:
                    y2.square().mean().backward()
            main_s.wait_stream(s1); main_s.wait_stream(s2)
        else:
            y1 = f_ab()
            if grad:
                y1.square().mean().backward()
            y2 = f_ba()
            if grad:
                y2.square().mean().backward()
        return y1, y2

    for grad in (False, True):
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            for rnd in range(2):
                line = ("fwd+bwd" if grad else "fwd    ") + " round %d:" % rnd
                for two in (False, True):
                    run(two, grad); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        run(two, grad)
                    e1.record(); torch.cuda.synchronize()
                    line += "  %s %.2f ms" % ("two streams" if two else "one stream ", e0.elapsed_time(e1) / a.iters)
                print(line, flush=True)


if __name__ == "__main__":
    main()
