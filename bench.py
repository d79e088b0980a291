#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the 256x256 Augmented CycleGAN step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL; RANK/LOCAL_RANK/WORLD_SIZE from env)

Workload at N = 1 (BASELINE.json configs[2], the configuration the metric is quoted on): 256x256x3 synthetic
unpaired batches, 9-resblock generators + latent encoder + latent discriminator (the full Augmented CycleGAN
step: AugmentedCycleGAN.train_instance), 32 (A,B) pairs per GPU, weak scaling (global batch = 32 N).
One "image" = one (A,B) pair consumed by train_instance (train.py:195).  Arithmetic (--precision): tensors, norms,
losses and Adam are fp32 throughout; the convolution products run on the matrix cores as
  bf16x3 (default) fp32 operands split hi + lo into bf16, three bf16 MFMAs per product, fp32 accumulate: 16-bit
                   operand mantissas (the config names plain bf16 = 8), parity-tested at the 1e-3 bar;
  f32              exact fp32 products on v_mfma_f32_32x32x2_f32 (strict mode, 1/16 of the bf16 MFMA rate);
  bf16             operands rounded to bf16 (what the config names; NOT inside the parity bar, reported for reference).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: the 3x3 reflect-pad 128->128 resblock
convolution forward, timed live with HIP events on its launch stream) and `cpu_baseline` (the oracle "port"
timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def make_opt(a, local_rank):
    return argparse.Namespace(input_nc=3, output_nc=3, ngf=32, nef=32, ndf=64, nlatent=16, lr=2e-4, beta1=0.5,
                              max_gnorm=500.0, lambda_A=1.0, lambda_B=1.0, lambda_z_B=0.025, lambda_sup_A=0.1,
                              lambda_sup_B=0.1, stoch_enc=False, z_gan=1, enc_A_B=1, no_lsgan=False, norm="instance",
                              use_dropout=False, which_model_netG="resnet", which_model_netD="basic",
                              gpu_ids=[local_rank], monitor_gnorm=True, niter_decay=25, expr_dir="/tmp",
                              n_blocks=a.blocks)


def cpu_baseline(a):
    """Oracle ("port": from-scratch C/NumPy restatement of the reference's networks.py/model.py path, pinned to the
    reference by tests/golden) timed on the host cores.  Bounded sample: ONE (A,B) pair of the same step."""
    # the 1-GPU box's CPU share is 16 cores; pin the OpenMP pool BEFORE the C library loads
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
    cores = int(os.environ["OMP_NUM_THREADS"])
    import numpy as np
    from oracle import recipe, step
    opt = step.Opt(input_nc=3, output_nc=3, n_blocks=a.blocks)
    m = step.AugStep(opt, dtype=np.float32)
    m.load({n: recipe.values_for(net.shapes, n, 0, "init") for n, net in m.nets().items()})
    A, B, z = recipe.inputs(0, 1, 3, 3, a.size, 16)
    t0 = time.time()
    m.train_instance(A, B, z)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "1 step of batch 1 (one A,B pair) of the same %dx%dx3 %d-resblock full Augmented CycleGAN step, "
                      "fp32, %.1f s" % (a.size, a.size, a.blocks, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="(A,B) pairs per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"],
                    help="conv arithmetic, see the module docstring")
    a = ap.parse_args()

    import dtgan_amd
    from dtgan_amd import dist as D, model as M, ops
    rank, ws = D.init_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert ws == a.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, ws)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ops.set_precision(a.precision)
    torch.manual_seed(0)  # identical replicas by construction (and broadcast from rank 0 anyway)
    model = M.AugmentedCycleGAN(make_opt(a, local_rank), testing=True)

    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    N, S = a.batch, a.size
    real_A = torch.rand((N, 3, S, S), device=dev, generator=g) * 2 - 1
    real_B = torch.rand((N, 3, S, S), device=dev, generator=g) * 2 - 1

    def step():
        z = torch.randn((N, 16, 1, 1), device=dev, generator=g)       # train.py:193: fresh prior every step
        return model.train_instance(real_A, real_B, z)

    def barrier():
        if ws > 1:
            torch.distributed.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    # dominant kernel: resblock 3x3 reflect conv 128->128 at S/2 (forward launches only)
    timer = ops.ConvTimer(lambda d: d.K == 3 and d.Ci == 128 and d.Co == 128 and d.stride == 1 and d.pad_mode == 1)
    ops.CONV_TIMER = timer
    barrier()
    t0 = time.time()
    for _ in range(a.steps):
        losses, _, _ = step()
    barrier()
    dt = time.time() - t0
    ops.CONV_TIMER = None
    if ws > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    if rank != 0:
        return
    ms = timer.ms()
    kern_ms = sum(ms) / max(len(ms), 1)
    flops = 2.0 * N * (S // 2) * (S // 2) * 128 * 128 * 9
    achieved = flops / (kern_ms * 1e-3) / 1e12 if ms else None
    # dense MFMA peaks (MI355X_MICROARCH.md): fp32 157.3, bf16 2500 TFLOP/s; bf16x3 issues 3 bf16 MFMAs per
    # algorithmic product, so its ceiling for ALGORITHMIC flops is 2500 / 3
    peak = {"f32": 157.3, "bf16x3": round(2500.0 / 3, 1), "bf16": 2500.0}[a.precision]
    kname = {"f32": "igemm_conv_f32<128,128,2,2,32,REFLECT,!THIN>", "bf16x3": "igemm_conv_x3_ws<REFLECT,STATS> (wave-specialised, 16x16x32 bf16 MFMA x3)",
             "bf16": "igemm_conv_bf16<128,128,2,2,64,REFLECT,!SPLIT>"}[a.precision]
    dtype = {"f32": "f32", "bf16x3": "f32 tensors; conv products as 3 bf16 MFMAs on hi/lo-split fp32 operands (~2^-17 operand rounding), fp32 accumulate",
             "bf16": "bf16 (MFMA operands; fp32 accumulate and fp32 tensors)"}[a.precision]
    traffic = None
    # measured by tools/profile_traffic.sh + tools/summarize_traffic.py (separate --pmc passes), committed per kernel
    tj = os.path.join(ROOT, "profiles", {"f32": "r01_c_resblock_conv_traffic_f32.json",
                                         "bf16x3": "r01_f_resblock_conv_traffic_bf16x3.json"}.get(a.precision, "-"))
    if os.path.exists(tj) and (N, S) == (32, 256):
        traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
    out = {
        "metric": "training images/sec, 256x256 Augmented CycleGAN step, 1/2/4/8 MI355X",
        "value": round(ws * N * a.steps / dt, 3), "unit": "images/s", "n_gpus": ws, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": "%dx%dx3 synthetic unpaired, %d-resblock G + latent encoder (full Augmented CycleGAN "
                               "train_instance), batch=%d per GPU (global %d)" % (S, S, a.blocks, N, N * ws),
                   "parallelism": "dp%d" % ws, "loss_G_A": round(losses["G_A"], 5)},
        "roofline": {"bound": "mfma", "kernel": kname + " (resblock 3x3 reflect 128->128 fwd)",
                     "achieved": None if achieved is None else round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": None if achieved is None else round(achieved / peak, 4), "traffic": traffic,
                     "launches_timed": len(ms), "avg_launch_ms": round(kern_ms, 4), "flops_per_launch": flops},
    }
    if ws == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
