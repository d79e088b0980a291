"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).

The unpaired A/B minibatch shards by rank with no data-path collective; the only exchange is the
gradient average of each optimiser phase: one all-reduce over the flat gradient
buffers of D_A+D_B+D_z_B after loss_D.backward() and one over G_B_A+G_A_B+E_B after loss_G.backward(),
both BEFORE the per-network clip (the clip coefficient depends on the global-batch gradient norm,
model.py:447-449, 510-512).  The reference's nn.parallel.data_parallel (networks.py:194-195 etc.) is
replaced, not translated.
"""
import os

import torch
import torch.distributed as td


# test hook: ACGAN_DIST_FORCE=1 runs every collective in a ONE-rank group too, so that the RCCL path (init, all-reduce of
# the flat gradient buffers, the float64 scalar reductions) can be exercised on a single-GPU box
_FORCE = os.environ.get("ACGAN_DIST_FORCE") == "1"


def is_on():
    return td.is_available() and td.is_initialized()


def exchange_on():
    """True when the step has to exchange gradients/statistics (more than one rank, or the one-rank test hook)."""
    return is_on() and (td.get_world_size() > 1 or _FORCE)


def world_size():
    return td.get_world_size() if is_on() else 1


def rank():
    return td.get_rank() if is_on() else 0


def init_from_env(backend=None):
    """torchrun-style init (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if (ws <= 1 and not _FORCE) or is_on():
        return rank(), world_size()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    td.init_process_group(backend=backend)
    return rank(), world_size()


def allreduce_mean_(bufs):
    """In-place average of each flat buffer across ranks (async launches, one wait at the end)."""
    ws = world_size()
    if not exchange_on():
        return
    if td.get_backend() == "gloo" and bufs[0].is_cuda:
        # test-only path (2 ranks sharing one GPU under gloo): stage through the host
        for b in bufs:
            h = b.cpu()
            td.all_reduce(h, op=td.ReduceOp.SUM)
            b.copy_(h.mul_(1.0 / ws))
        return
    works = [td.all_reduce(b, op=td.ReduceOp.SUM, async_op=True) for b in bufs]
    for w in works:
        w.wait()
    for b in bufs:
        b.mul_(1.0 / ws)


def broadcast_params_(nets):
    """Rank 0's initial parameters/buffers to every rank (replicas must start identical)."""
    if not exchange_on():
        return
    stage = td.get_backend() == "gloo"
    for n in nets:
        for t in list(n.parameters()) + list(n.buffers()):
            if stage and t.is_cuda:
                h = t.data.cpu()
                td.broadcast(h, src=0)
                t.data.copy_(h)
            else:
                td.broadcast(t.data, src=0)


def average_scalars(vals, sq_keys=(), min_keys=(), max_keys=()):
    """Reported scalars: rank-mean of the losses; gradient norms are already global (the gradients
    were averaged before the norm); min/max monitors reduce accordingly.  `vals`: OrderedDict of floats."""
    ws = world_size()
    if not exchange_on():
        return vals
    keys = list(vals.keys())
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    t = torch.tensor([vals[k] for k in keys], dtype=torch.float64, device=dev)
    mean = t.clone(); td.all_reduce(mean, op=td.ReduceOp.SUM); mean /= ws
    out = type(vals)(zip(keys, mean.tolist()))
    if min_keys:
        mn = t.clone(); td.all_reduce(mn, op=td.ReduceOp.MIN)
        for k in min_keys:
            out[k] = float(mn[keys.index(k)])
    if max_keys:
        mx = t.clone(); td.all_reduce(mx, op=td.ReduceOp.MAX)
        for k in max_keys:
            out[k] = float(mx[keys.index(k)])
    return out
