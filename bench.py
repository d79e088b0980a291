#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the Augmented CycleGAN step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|5]

N > 1: when RANK is not in the environment this process is only a LAUNCHER — before touching the GPU it starts N fresh
child processes of this same file (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), one rank per
GPU over RCCL, relays rank 0's JSON line and exits non-zero if any child fails.  Under an external
`python -m torch.distributed.run` the ranks find RANK in the environment and run directly.

Workloads (BASELINE.json `configs`; `--config`):
  3 (default) configs[2], the configuration the metric is quoted on: 256x256x3 synthetic unpaired batches, 9-resblock
              generators + latent encoder + latent discriminator (the full AugmentedCycleGAN.train_instance), 32 (A,B) pairs
              per GPU, conv arithmetic bf16x3.  With --gpus 8 this is configs[3] (global batch 256, weak scaling).
  2           configs[1]: 128x128x3, 6 resblocks, batch 16, exact-fp32 conv arithmetic.
  5           configs[4]: 512x512x1 "Livneh-shaped" fields, 9 resblocks, 16 pairs per GPU (global 128 on 8 GPUs), bf16x3.
One "image" = one (A,B) pair consumed by train_instance (train.py:195).  Arithmetic (--precision): tensors, norms,
losses and Adam are fp32 throughout; the convolution products run on the matrix cores as
  bf16x3  fp32 operands split hi + lo into bf16, three bf16 MFMAs per product, fp32 accumulate: 16-bit operand
          mantissas (config 3 names plain bf16 = 8), parity-tested at the 1e-3 bar;
  f32     exact fp32 products on v_mfma_f32_32x32x2_f32 (strict mode, 1/16 of the bf16 MFMA rate);
  (a plain-bf16 mode is not offered: DESIGN_LOG.md A.4 — an MFMA-only ablation bounds it at 1.24x, a storage mode short of
  330 images/s; `ops.set_precision("bf16")` remains as the operand-rounding probe of tests/test_hip_bf16.py.)

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: the 3x3 reflect-pad 128->128 resblock convolution forward),
`roofline_hbm` (the stride-2 64->128 downsample convolution forward, HBM-bound) — both timed live with HIP events on the
launch stream, kernel names read back from the dispatcher — and `cpu_baseline` (the oracle "port" on the host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    2: dict(size=128, nc=3, blocks=6, batch=16, precision="f32", name="configs[1]"),
    3: dict(size=256, nc=3, blocks=9, batch=32, precision="bf16x3", name="configs[2] (configs[3] at 8 GPUs)"),
    5: dict(size=512, nc=1, blocks=9, batch=16, precision="bf16x3", name="configs[4]"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5], help="BASELINE.json workload (4 = 3 on 8 GPUs)")
    ap.add_argument("--batch", type=int, default=None, help="(A,B) pairs per GPU (default: the config's)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--nc", type=int, default=None, help="image channels of both domains")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-step-graph", action="store_true",
                    help="enqueue every step kernel by kernel.  Default on ONE GPU: the step is replayed as one captured HIP graph "
                         "(model.enable_step_graph, train.py --step_graph: the same launches without ~1 400 launch boundaries, "
                         "138.8-139.1 against 140.9-141.7 ms), except on the sampled steps that carry the HIP-event brackets of "
                         "the roofline block, which run eagerly.  With several ranks the data-parallel exchange keeps the eager path.")
    ap.add_argument("--sync-scalars", action="store_true",
                    help="wait for every replayed step's losses on the host (the reference's loop does, train.py:198-243) instead of "
                         "reading them one step late")
    ap.add_argument("--timer-every", type=int, default=10, metavar="N",
                    help="bracket the roofline kernels with HIP events on every N-th timed step only (0: never; the events are "
                         "barrier packets between back-to-back launches and cost the step time)")
    ap.add_argument("--sync-bn", action="store_true", help="BatchNorm statistics over all ranks (default: per rank, as the "
                                                           "reference's data_parallel)")
    ap.add_argument("--precision", default=None, choices=["f32", "bf16x3"],
                    help="conv arithmetic, see the module docstring (default: the config's)")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, metavar="THREADS",
                    help="(internal) time the oracle step on THREADS host threads and print its JSON object")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher test: ranks rendezvous over gloo on the CPU, exchange one all-reduce and print the line "
                         "without touching a GPU")
    a = ap.parse_args(argv)
    c = CONFIGS[3 if a.config == 4 else a.config]
    for k in ("batch", "size", "blocks", "nc", "precision"):
        if getattr(a, k) is None:
            setattr(a, k, c[k])
    a.config_name = c["name"]
    return a


# --------------------------------------------------------------------------------------------------------------------
# launcher: N fresh ranks (never re-exec a process that touched the GPU; the parent makes no GPU call at all)
# --------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv):
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    env = dict(os.environ, WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import tempfile
    procs = []
    cap = tempfile.TemporaryFile(mode="w+")      # rank 0's stdout (the JSON line); every other rank's stdout goes to stderr
    for r in range(a.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=cap if r == 0 else sys.stderr, text=True))
    codes = [None] * a.gpus
    while any(c is None for c in codes):         # a rank that dies would leave the others waiting in a collective
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()                     # exactly the children started above
                    codes[i] = p.wait()
        time.sleep(0.05)
    cap.seek(0)
    out0 = cap.read()
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    for ln in (out0 or "").splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if any(codes) or not lines:
        print("bench.py: rank exit codes %s" % codes, file=sys.stderr)
        sys.exit(1)
    print(lines[-1], flush=True)


# --------------------------------------------------------------------------------------------------------------------
def make_opt(a, local_rank):
    return argparse.Namespace(input_nc=a.nc, output_nc=a.nc, ngf=32, nef=32, ndf=64, nlatent=16, lr=2e-4, beta1=0.5,
                              max_gnorm=500.0, lambda_A=1.0, lambda_B=1.0, lambda_z_B=0.025, lambda_sup_A=0.1,
                              lambda_sup_B=0.1, stoch_enc=False, z_gan=1, enc_A_B=1, no_lsgan=False, norm="instance",
                              use_dropout=False, which_model_netG="resnet", which_model_netD="basic",
                              gpu_ids=[local_rank], monitor_gnorm=True, niter_decay=25, expr_dir="/tmp",
                              n_blocks=a.blocks, sync_bn=a.sync_bn)


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_run(a, threads):
    """Oracle ("port": from-scratch C/NumPy restatement of the reference's networks.py/model.py path, pinned to the
    reference by tests/golden) timed on `threads` host cores.  Bounded sample: ONE pair of the same step, one warm-up step
    (page faults, thread pool), then THREE timed steps (CPU time is linear in the batch)."""
    os.environ["OMP_NUM_THREADS"] = str(threads)          # the OpenMP pool is sized when the C library loads
    import numpy as np
    from oracle import recipe, step
    opt = step.Opt(input_nc=a.nc, output_nc=a.nc, n_blocks=a.blocks)
    m = step.AugStep(opt, dtype=np.float32)
    m.load({n: recipe.values_for(net.shapes, n, 0, "init") for n, net in m.nets().items()})
    nb, timed = 1, 3
    batches = [recipe.inputs(s, nb, a.nc, a.nc, a.size, 16) for s in range(1 + timed)]
    m.train_instance(*batches[0])
    t0 = time.time()
    for b in batches[1:]:
        m.train_instance(*b)
    dt = (time.time() - t0) / timed
    return {"value": round(nb / dt, 4), "unit": "images/s", "cores": threads, "seconds_per_step": round(dt, 2), "timed_steps": timed}


def cpu_baseline(a, argv):
    """the host baseline at 8 threads (comparable with the survey container's 8-core numbers) and at the
    1-GPU box's CPU share of one socket (16 cores), each in a fresh process so that the OpenMP pool has exactly that size;
    the headline object is the faster of the two."""
    # (the affinity mask of a 1-GPU box shows every core of the host, its CPU share is 16 of them: 256 threads on that share
    # ran the step 11x slower than 8)
    avail = min(len(os.sched_getaffinity(0)), 16)
    runs = []
    for th in sorted(set([min(8, avail), avail])):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--cpu-baseline-only", str(th)],
                           stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(th)))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            runs.append(json.loads(lines[-1]))
    if not runs:
        return None
    best = dict(max(runs, key=lambda r: r["value"]))
    best.update({"cpu": _cpu_model(), "kind": "port",
                 "sample": "3 timed steps (after 1 warm-up step) of ONE (A,B) pair of the same %dx%dx%d %d-resblock full Augmented "
                           "CycleGAN step, fp32, %.1f s per step on %d threads" % (a.size, a.size, a.nc, a.blocks,
                                                                                   best["seconds_per_step"], best["cores"]),
                 "by_threads": runs,
                 "note": "plain-C loops (cache-blocked rows, register-blocked forward strips, no FMA contraction), not a tuned library: SURVEY.md §6 measured the reference's own "
                         "torch-CPU (MKL-DNN) path at 0.45 images/s on 8 threads for the 256x256x3 3-resblock StochCycleGAN step "
                         "(593 GFLOP/pair = 267 GFLOP/s); the 9-resblock full step is 1289 GFLOP/pair, i.e. about 0.2 images/s "
                         "for the real reference on 8 cores.  This port is a baseline, slower than that library path; the GPU/CPU "
                         "ratio is not a quality measure, roofline.frac is."})
    return best


def dry_run(a):
    """ranks rendezvous over gloo without touching a GPU (CPU test of the self-launch path)"""
    import torch
    import torch.distributed as td
    ws, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if ws > 1:
        td.init_process_group(backend="gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        td.barrier()
        assert float(t) == ws * (ws + 1) / 2.0
    if rank == 0:   # every key of the contract line the driver's SCALE parser reads, with the measured fields empty
        print(json.dumps({"metric": "training images/sec, 256x256 Augmented CycleGAN step, 1/2/4/8 MI355X", "dry_run": True,
                          "value": None, "unit": "images/s", "n_gpus": ws, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": a.precision, "data": "synthetic",
                          "config": {"workload": "%s: %dx%dx%d, %d resblocks, batch=%d per GPU (global %d)"
                                                 % (a.config_name, a.size, a.size, a.nc, a.blocks, a.batch, a.batch * ws),
                                     "parallelism": "dp%d" % ws,
                                     "backend": td.get_backend() if ws > 1 else "none (single process)",
                                     "world_size_seen": td.get_world_size() if ws > 1 else 1}}), flush=True)


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    if a.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(a, argv)          # parent: no torch.cuda / HIP call has happened in this process
    if a.dry_run:
        return dry_run(a)
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline_run(a, a.cpu_baseline_only)), flush=True)
        return

    import torch
    import dtgan_amd  # noqa: F401
    from dtgan_amd import dist as D, model as M, ops
    rank, ws = D.init_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert ws == a.gpus, "WORLD_SIZE=%d but --gpus %d" % (ws, a.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ops.set_precision(a.precision)
    torch.manual_seed(0)  # identical replicas by construction (and broadcast from rank 0 anyway)
    model = M.AugmentedCycleGAN(make_opt(a, local_rank), testing=True)
    use_graph = ws == 1 and not a.no_step_graph
    if use_graph:
        # (defer_scalars: a replayed step hands back a DeferredStep instead of waiting for its 23 scalars; the host enqueues the
        # next replay meanwhile and the line's loss is read from the last step behind the timed region)
        model.enable_step_graph(defer_scalars=not a.sync_scalars)
    graph_obj = model._step_graph

    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    N, S, nc = a.batch, a.size, a.nc
    real_A = torch.rand((N, nc, S, S), device=dev, generator=g) * 2 - 1
    real_B = torch.rand((N, nc, S, S), device=dev, generator=g) * 2 - 1

    def step(eager=False):
        nonlocal_graph = graph_obj
        z = torch.randn((N, 16, 1, 1), device=dev, generator=g)       # train.py:193: fresh prior every step
        if eager and nonlocal_graph is not None:     # a sampled step: the same train_instance, enqueued kernel by kernel
            model._step_graph = None
            try:
                return model.train_instance(real_A, real_B, z)
            finally:
                model._step_graph = nonlocal_graph
        return model.train_instance(real_A, real_B, z)

    def barrier():
        if ws > 1:
            torch.distributed.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    # the graph is captured on its third call (two eager calls settle the lazily built state): never inside the timed region
    try:
        while graph_obj is not None and graph_obj.graph is None:
            step()
        if graph_obj is not None:
            # torch.cuda.graph() empties the caching allocator before it captures: the first eagerly enqueued step behind the
            # capture allocates its ~50 GB of activations afresh (measured: 1.1 s instead of 0.14 s, in 2 of 18 runs inside the
            # timed region) — it happens here, untimed, and one replay behind it
            step(eager=True)
            step()
    except Exception as e:   # a capture that fails leaves no graph behind (model.StepGraph): the run goes on eagerly, and says so
        print("bench.py: step-graph capture failed (%s: %s); continuing with eager launches" % (type(e).__name__, e), file=sys.stderr)
        model.enable_step_graph(False)
        graph_obj = None
    # dominant kernel: resblock 3x3 reflect conv 128->128 at S/2; HBM-bound companion: the stride-2 64->128 downsample
    # (networks.py:168, 220) at full resolution — forward launches only, HIP events on the launch stream
    is_res = lambda d: d.K == 3 and d.Ci == 128 and d.Co == 128 and d.stride == 1 and d.pad_mode == 1
    t_res = ops.ConvTimer(is_res)
    t_res_d, t_res_w, t_res_ds = ops.ConvTimer(is_res, "dgrad"), ops.ConvTimer(is_res, "wgrad"), ops.ConvTimer(is_res, "dgrad_sums")
    t_s2 = ops.ConvTimer(lambda d: d.K == 3 and d.Ci == 64 and d.Co == 128 and d.stride == 2 and d.Hi == S)
    timers = [t_res, t_res_d, t_res_w, t_res_ds, t_s2]
    ops.FUSED.clear()
    barrier()
    t0 = time.time()
    eager_steps = 0
    step_marks = [] if os.environ.get("ACG_BENCH_STEP_TIMES") else None   # diagnostic: one event behind every timed step
    for i in range(a.steps):
        sampled = a.timer_every > 0 and i % a.timer_every == 0
        ops.CONV_TIMERS[:] = timers if sampled else []
        eager_steps += 1 if (sampled or graph_obj is None) else 0
        last = step(eager=sampled)
        if step_marks is not None:
            ev = torch.cuda.Event(enable_timing=True); ev.record(); step_marks.append((ev, time.time()))
    barrier()
    dt = time.time() - t0
    losses = (last.result() if hasattr(last, "result") else last)[0]
    ops.CONV_TIMERS[:] = []
    if step_marks:
        print("per-step: GPU ms between step ends %s | host s at step ends %s" % (
            " ".join("%.1f" % step_marks[k][0].elapsed_time(step_marks[k + 1][0]) for k in range(len(step_marks) - 1)),
            " ".join("%.3f" % (tm - t0) for _, tm in step_marks)), file=sys.stderr)
    # what the matrix pipe holds on THIS device in its present clock / power state: the library's register-only MFMA loop
    # (acg_probe_mfma_rate), timed here, right BEHIND the timed region — printed beside the spec peak, never instead of it.
    # (In front of the timed region its 150 ms at full matrix power cost the step 3 %: 145.8 against 141.6 ms, every trunk
    # kernel 3-4 % slower — the chip starts the timed steps at a lower clock.)
    sustained = None
    if a.precision == "bf16x3" and rank == 0:
        import ctypes
        from dtgan_amd import _lib
        scratch = torch.empty(512 * 1024, device=dev)
        fl = ctypes.c_double(0.0)
        best = 0.0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.call("acg_probe_mfma_rate", ops._ptr(scratch), scratch.numel(), 120000, ctypes.byref(fl), ops._stream())
            e1.record()
            torch.cuda.synchronize()
            best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        sustained = best
    # (ops.FUSED counts in Python: a graph replay does not pass there — the counts are those of the eagerly enqueued steps)
    fused_paths = {k: round(v / float(max(eager_steps, 1)), 2) for k, v in sorted(ops.FUSED.items())}
    if ws > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    if rank != 0:
        return
    sampled_steps = max(len(range(0, a.steps, a.timer_every)) if a.timer_every > 0 else 0, 1)   # timed steps that carried the event brackets
    ms = t_res.ms()
    kern_ms = sum(ms) / max(len(ms), 1)
    flops = 2.0 * N * (S // 2) * (S // 2) * 128 * 128 * 9
    achieved = flops / (kern_ms * 1e-3) / 1e12 if ms else None
    # dense MFMA peaks (MI355X_MICROARCH.md): fp32 157.3, bf16 2500 TFLOP/s; bf16x3 issues 3 bf16 MFMAs per
    # algorithmic product, so its ceiling for ALGORITHMIC flops is 2500 / 3
    peak = {"f32": 157.3, "bf16x3": round(2500.0 / 3, 1)}[a.precision]
    dtype = {"f32": "f32",
             "bf16x3": "f32 tensors; conv products as 3 bf16 MFMAs on hi/lo-split fp32 operands (~2^-17 operand rounding), fp32 "
                       "accumulate; parity vs the reference goldens: losses / single-pass images / cycle reconstructions <= 1e-3 on every "
                       "fixture at step 0 (measured rec_A / rec_B <= 7.9e-4, profiles/r06_rec_errors.txt)"}[a.precision]
    # HBM traffic of the dominant kernel: measured by tools/profile_traffic.sh + tools/summarize_traffic.py (separate
    # rocprofv3 --pmc passes; counters cannot be read inside the timed run), committed per kernel under profiles/
    traffic, traffic_src = None, None
    if (N, S, nc) != (32, 256, 3):
        traffic_src = "not measured for this shape (the committed counter runs are batch 32, 256x256x3)"
    for cand in ({"f32": ["r01_c_resblock_conv_traffic_f32.json"],
                  "bf16x3": ["r06_resblock_conv_traffic_bf16x3.json", "r05_resblock_conv_traffic_bf16x3.json", "r04_resblock_conv_traffic_bf16x3.json", "r03_resblock_conv_traffic_bf16x3.json", "r02_z_resblock_conv_traffic_bf16x3.json", "r02_p_resblock_conv_traffic_bf16x3.json", "r02_resblock_conv_traffic_bf16x3.json", "r01_f_resblock_conv_traffic_bf16x3.json"]}.get(a.precision, [])):
        tj = os.path.join(ROOT, "profiles", cand)
        if os.path.exists(tj) and (N, S, nc) == (32, 256, 3):
            traffic, traffic_src = json.load(open(tj)).get("hbm_bytes_per_launch"), "profiles/" + cand
            break
    # the other two passes of the same layer (the data gradient is the largest line of the profile): per-pass launch time
    # and the three-pass aggregate = 3 x the forward's FLOPs over the sum of the three mean launch times.  The weight
    # gradient's main kernel and its split-K reduction launch (~10 us) are timed apart; the aggregate charges both.
    passes = {}
    for nm, tm in (("fwd", t_res), ("dgrad", t_res_d), ("wgrad", t_res_w)):
        v = tm.ms_main()   # the kernel alone: the weight gradient's split-K reduction launch is timed apart (acg_debug_mid_event)
        if v:
            m_ = sum(v) / len(v)
            passes[nm] = {"kernel": tm.kernel, "launches_timed": len(v), "avg_launch_ms": round(m_, 4),
                          "achieved": round(flops / (m_ * 1e-3) / 1e12, 2), "frac": round(flops / (m_ * 1e-3) / 1e12 / peak, 4)}
            if nm == "wgrad":
                full = tm.ms()
                passes[nm]["with_split_k_reduce_ms"] = round(sum(full) / len(full), 4)
                passes[nm]["split_k_reduce_ms"] = round(sum(full) / len(full) - m_, 4)
    # data-gradient launches that also carry the first pass of the backward of the norm in front of the layer (its input read
    # as one more side stream, per-tile sums written: acg_conv2d_bwd_data_s16_sums) are timed apart; the aggregate below takes
    # the launch-weighted mean over BOTH kinds, i.e. it charges that norm work to the convolution
    vds = t_res_ds.ms()
    dgrad_all = t_res_d.ms() + vds
    if vds:
        m_ = sum(vds) / len(vds)
        passes["dgrad_sums"] = {"kernel": t_res_ds.kernel, "launches_timed": len(vds), "avg_launch_ms": round(m_, 4),
                                "achieved": round(flops / (m_ * 1e-3) / 1e12, 2), "frac": round(flops / (m_ * 1e-3) / 1e12 / peak, 4),
                                "also": "sum gy, sum gy*xhat of the norm whose output gradient it writes (replaces a norm_bwd_partial pass)"}
    agg = None
    if "fwd" in passes and "wgrad" in passes and dgrad_all:
        dmean = sum(dgrad_all) / len(dgrad_all)
        tot = passes["fwd"]["avg_launch_ms"] + dmean + passes["wgrad"]["with_split_k_reduce_ms"]
        agg = {"ms_fwd_dgrad_wgrad": round(tot, 4), "dgrad_mean_all_launches_ms": round(dmean, 4),
               "achieved": round(3 * flops / (tot * 1e-3) / 1e12, 2), "frac": round(3 * flops / (tot * 1e-3) / 1e12 / peak, 4)}
    # the kernel instance of the resblock layer that costs the step most (launches per step x mean launch time) — what a
    # rocprofv3 kernel table of the same run shows as its top line: the data gradient that also emits the norm sums, not the
    # forward instance the headline `roofline` block is quoted on
    dominant = None
    inst = {}
    for nm, tm in (("fwd", t_res), ("dgrad", t_res_d), ("dgrad_sums", t_res_ds), ("wgrad", t_res_w)):
        for k, v in tm.by_kernel(main=True).items():   # kernels alone, as a rocprofv3 kernel table lists them
            inst.setdefault((nm, k), []).extend(v)
    if inst:
        (nm, k), v = max(inst.items(), key=lambda kv: sum(kv[1]))
        m_ = sum(v) / len(v)
        dominant = {"pass": nm, "kernel": k, "launches_per_step": round(len(v) / float(sampled_steps), 1),
                    "ms_per_step": round(sum(v) / sampled_steps, 2), "avg_launch_ms": round(m_, 4),
                    "achieved": round(flops / (m_ * 1e-3) / 1e12, 2), "frac": round(flops / (m_ * 1e-3) / 1e12 / peak, 4)}
    ms2 = t_s2.ms()
    k2 = sum(ms2) / max(len(ms2), 1)
    bytes2 = 4.0 * (N * S * S * 64 + N * (S // 2) * (S // 2) * 128 + 9 * 64 * 128)   # in + out + weights, each once
    out = {
        "metric": "training images/sec, 256x256 Augmented CycleGAN step, 1/2/4/8 MI355X",
        "value": round(ws * N * a.steps / dt, 3), "unit": "images/s", "n_gpus": ws, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d synthetic unpaired, %d-resblock G + latent encoder + latent discriminator (full "
                               "Augmented CycleGAN train_instance), batch=%d per GPU (global %d)"
                               % (a.config_name, S, S, nc, a.blocks, N, N * ws),
                   "parallelism": "dp%d" % ws, "batchnorm": "sync" if a.sync_bn else "per-rank",
                   "launch": ("one captured HIP graph per step (model.enable_step_graph); %d of the %d timed steps — the ones carrying the "
                              "HIP-event brackets — enqueued kernel by kernel; %s" % (eager_steps, a.steps,
                              "every step's scalars awaited on the host" if a.sync_scalars else
                              "a replayed step's 23 scalars travel to pinned memory asynchronously (read behind the timed region)"))
                             if graph_obj is not None else "eager",
                   # what torch.distributed itself reports (a SCALE record shows RCCL saw N ranks)
                   "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else "none (single process)"),
                   "world_size_seen": (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1),
                   "loss_G_A": round(losses["G_A"], 5)},
        "roofline": {"bound": "mfma", "kernel": "%s (resblock 3x3 reflect 128->128 fwd)" % t_res.kernel,
                     "achieved": None if achieved is None else round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": None if achieved is None else round(achieved / peak, 4),
                     # beside — never instead of — the spec-peak fraction: what the matrix pipe HOLDS under load on this device,
                     # measured in this process right behind the timed region (best of three ~50 ms runs of the library's
                     # register-only v_mfma_f32_16x16x32_bf16 loop on random data); / 3 = algorithmic bf16x3 work
                     "peak_sustained": (None if sustained is None else round(sustained / 3, 1)),
                     "frac_of_sustained": (None if achieved is None or sustained is None else round(achieved / (sustained / 3), 4)),
                     "peak_sustained_source": (None if sustained is None else "measured live: acg_probe_mfma_rate (register-only bf16 MFMA loop, random "
                                               "operands, 8 waves per CU) ran at %.0f TFLOP/s executed on this device right behind the timed region" % sustained),
                     "traffic": traffic,
                     "traffic_source": traffic_src, "launches_timed": len(ms), "avg_launch_ms": round(kern_ms, 4),
                     # the HIP-event brackets are barrier packets between back-to-back launches: on every step they cost it
                     # 1.0-1.5 ms (143.4 / 143.7 against 141.9 / 142.4 ms without any, one box); they ride on a sample of the timed steps
                     "timer_sampling": "HIP events on every %d-th timed step (%d of %d)" % (a.timer_every, sampled_steps if a.timer_every > 0 else 0, a.steps),
                     "flops_per_launch": flops, "passes": passes, "three_pass_aggregate": agg, "dominant_by_time": dominant},
        "roofline_hbm": {"bound": "hbm", "kernel": "%s (3x3 stride-2 64->128 downsample fwd)" % t_s2.kernel,
                         "achieved": round(bytes2 / (k2 * 1e-3) / 1e9, 1) if ms2 else None, "peak": 8000.0, "unit": "GB/s",
                         "frac": round(bytes2 / (k2 * 1e-3) / 1e9 / 8000.0, 4) if ms2 else None,
                         "launches_timed": len(ms2), "avg_launch_ms": round(k2, 4), "bytes_per_launch": bytes2},
    }
    out["fused_paths"] = {"per_step": fused_paths, "note": "launches per training step that took each fused path (ops.FUSED); "
                          "config 3 expects 70 norm_bwd_sums_from_dgrad (54 trunk + 16 full-resolution layers), 72 wgrad_s16, 18 relu bitmask links"}
    if ws == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a, [x for x in argv if x != "--no-cpu-baseline"])
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
