"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).

The unpaired A/B minibatch shards by rank with no data-path collective; the only exchange is the
gradient average of each optimiser phase: the flat gradient buffers of D_A, D_B, D_z_B
after loss_D.backward() and of G_B_A, G_A_B, E_B after loss_G.backward(), each BEFORE that network's
clip (the clip coefficient depends on the global-batch gradient norm, model.py:447-449, 510-512).
The reference's nn.parallel.data_parallel (networks.py:194-195 etc.) is replaced, not translated.

Overlap (PhaseExchange): every parameter carries a post-accumulate-grad hook; when the last gradient of a
network has been written — in the middle of the backward pass — that network's all-reduce is launched on
RCCL's stream and runs beside the rest of the backward (and, for the D phase, beside the G-phase forwards
that do not depend on the discriminators, model.py:467-494).  Nothing waits until the network's own
clip + Adam.  The step's reported scalars travel in a small tail behind one gradient buffer, so a step
issues exactly six collectives (seven with SyncBN's statistics excluded) and ONE device->host copy.
"""
import os

import torch
import torch.distributed as td


# test hook: ACGAN_DIST_FORCE=1 runs every collective in a ONE-rank group too, so that the RCCL path (init, all-reduce of
# the flat gradient buffers, the scalar tail) can be exercised on a single-GPU box
_FORCE = os.environ.get("ACGAN_DIST_FORCE") == "1"

# floats reserved behind every flat gradient buffer: [0, SUM_SLOTS) rank-averaged scalars, then 4 per rank for min/max monitors
SUM_SLOTS = 32
MAX_RANKS = 8
SCALAR_TAIL = SUM_SLOTS + 4 * MAX_RANKS


def is_on():
    return td.is_available() and td.is_initialized()


def exchange_on():
    """True when the step has to exchange gradients/statistics (more than one rank, or the one-rank test hook)."""
    return is_on() and (td.get_world_size() > 1 or _FORCE)


def world_size():
    return td.get_world_size() if is_on() else 1


def rank():
    return td.get_rank() if is_on() else 0


_ACG_COMM = None   # ACGAN_DP_BACKEND=acg_comm: (communicator handle, side stream) of the library's own RCCL wrappers


def init_from_env(backend=None):
    """torchrun-style init (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK).
    ACGAN_DP_BACKEND: "nccl" (default on a GPU: torch.distributed over RCCL), "gloo" (tests that put several ranks on one GPU;
    RCCL wants one GPU per rank) or "acg_comm": the gradient buffers travel through the library's own C entry points
    (acg_comm_*, include/acgan_hip.h: RCCL ncclAvg on a side stream) and torch.distributed/gloo is only the control plane
    that ships the communicator id, the initial parameters and SyncBN's statistics."""
    global _ACG_COMM
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if (ws <= 1 and not _FORCE) or is_on():
        return rank(), world_size()
    if backend is None:
        backend = os.environ.get("ACGAN_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend in ("nccl", "acg_comm"):
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    td.init_process_group(backend="gloo" if backend == "acg_comm" else backend)
    if td.get_world_size() > MAX_RANKS:
        raise RuntimeError("the scalar tail of the gradient buffers holds %d ranks (one node); got %d" % (MAX_RANKS, td.get_world_size()))
    if backend == "acg_comm":
        import ctypes
        from . import _lib
        ident = torch.zeros(128, dtype=torch.uint8)
        if td.get_rank() == 0:
            blob = (ctypes.c_char * 128)()
            _lib.call("acg_comm_unique_id", ctypes.cast(blob, ctypes.c_void_p))
            ident = torch.frombuffer(bytearray(blob.raw), dtype=torch.uint8).clone()
        td.broadcast(ident, src=0)
        raw = (ctypes.c_char * 128).from_buffer_copy(bytes(ident.numpy().tobytes()))
        comm = ctypes.c_void_p()
        _lib.call("acg_comm_init", ctypes.byref(comm), ctypes.cast(raw, ctypes.c_void_p), td.get_world_size(), td.get_rank())
        _ACG_COMM = (comm, torch.cuda.Stream())
    return rank(), world_size()


def shutdown():
    """release the communicator of the acg_comm backend (tests; a training process simply exits)"""
    global _ACG_COMM
    if _ACG_COMM is not None:
        from . import _lib
        torch.cuda.synchronize()
        _lib.call("acg_comm_destroy", _ACG_COMM[0])
        _ACG_COMM = None


def _staged():
    """test-only situation: gloo with CUDA tensors (2 ranks sharing the box's one GPU) -> stage through the host"""
    return td.get_backend() == "gloo"


class _Done(object):
    def wait(self):
        return True


class _OnEvent(object):
    """handle of a collective enqueued on the side stream: wait() orders the CURRENT stream behind it"""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)
        return True


def _allreduce_avg_async(buf):
    """Average `buf` over the ranks in place; returns a handle whose wait() orders the current stream behind it."""
    ws = world_size()
    if _ACG_COMM is not None and buf.is_cuda:   # the library's RCCL wrapper, on its own stream behind everything enqueued so far
        import ctypes
        from . import _lib
        comm, side = _ACG_COMM
        side.wait_stream(torch.cuda.current_stream())
        _lib.call("acg_comm_allreduce_mean", comm, ctypes.c_void_p(buf.data_ptr()), buf.numel(), ctypes.c_void_p(side.cuda_stream))
        ev = torch.cuda.Event()
        ev.record(side)
        return _OnEvent(ev)
    if td.get_backend() == "nccl":   # RCCL averages in the collective itself (ncclAvg): no extra kernel
        return td.all_reduce(buf, op=td.ReduceOp.AVG, async_op=True)
    if buf.is_cuda:                  # gloo + device tensors (tests): blocking, through the host
        h = buf.cpu()
        td.all_reduce(h, op=td.ReduceOp.SUM)
        buf.copy_(h.mul_(1.0 / ws))
        return _Done()
    td.all_reduce(buf, op=td.ReduceOp.SUM)   # gloo on host tensors (CPU tests)
    buf.mul_(1.0 / ws)
    return _Done()


def allreduce_mean_(bufs):
    """In-place average of each flat buffer across ranks (async launches, one wait at the end)."""
    if not exchange_on():
        return
    for w in [_allreduce_avg_async(b) for b in bufs]:
        w.wait()


class PhaseExchange(object):
    """The gradient exchange of ONE optimiser phase, overlapped with the backward pass that produces the gradients.

    arm(buckets) before backward(): `buckets` are objects with `.g` (flat gradient buffer incl. its scalar tail) and
    `.params`, in the order their gradients are expected to complete.  The per-parameter hooks (hook_params) call
    param_done(); when a bucket has seen as many gradients as it did in this phase's previous step, its all-reduce is
    launched — strictly in the armed order, so every rank issues the same sequence of collectives whatever order its
    autograd engine happened to use.  flush() after backward() launches whatever is left (always everything on a phase's
    first step, when nothing is known yet).  A gradient that arrives AFTER its bucket was launched means the graph
    changed between steps: that raises instead of silently reducing a partial sum.
    """

    def __init__(self, name):
        self.name = name
        self.expected = {}     # id(bucket) -> gradients seen in the previous step of this phase
        self.buckets, self.works, self.seen, self.launched = [], [], [], []
        self.armed = False

    def arm(self, buckets):
        self.buckets = list(buckets)
        n = len(self.buckets)
        self.works, self.seen, self.launched, self.next = [None] * n, [0] * n, [False] * n, 0
        for i, b in enumerate(self.buckets):
            b._exchange = (self, i)
        self.armed = True

    def param_done(self, i):
        if self.launched[i]:
            raise RuntimeError("%s: a gradient of bucket %d arrived after its all-reduce was launched (the autograd graph "
                               "changed between steps)" % (self.name, i))
        self.seen[i] += 1
        self._pump()

    def _pump(self):
        while self.next < len(self.buckets):
            i = self.next
            exp = self.expected.get(id(self.buckets[i]))
            if exp is None or self.seen[i] < exp:
                return
            self._launch(i)

    def _launch(self, i):
        self.works[i] = _allreduce_avg_async(self.buckets[i].g)
        self.launched[i] = True
        self.next = i + 1

    def flush(self):
        """after backward(): launch the rest, remember how many gradients each bucket received"""
        for i in range(self.next, len(self.buckets)):
            self._launch(i)
        for i, b in enumerate(self.buckets):
            self.expected[id(b)] = self.seen[i]
            b._exchange = None
        self.armed = False

    def wait(self, bucket):
        for i, b in enumerate(self.buckets):
            if b is bucket and self.works[i] is not None:
                self.works[i].wait()
                self.works[i] = None


def hook_params(bucket):
    """post-accumulate-grad hooks on every parameter of `bucket` (a model.FlatNet): route to the armed exchange, if any"""
    def hook(_p):
        ex = bucket._exchange
        if ex is not None:
            ex[0].param_done(ex[1])
    bucket._exchange = None
    for p in bucket.params:
        p.register_post_accumulate_grad_hook(hook)
        p._acg_grad_hook = hook   # ops fires it by hand where a kernel adds into .grad itself (no AccumulateGrad runs)


def broadcast_params_(nets):
    """Rank 0's initial parameters/buffers to every rank (replicas must start identical)."""
    if not exchange_on():
        return
    stage = _staged()
    for n in nets:
        for t in list(n.parameters()) + list(n.buffers()):
            if stage and t.is_cuda:
                h = t.data.cpu()
                td.broadcast(h, src=0)
                t.data.copy_(h)
            else:
                td.broadcast(t.data, src=0)


def write_scalar_tail(tail, sums, minmax=None):
    """Fill a gradient buffer's tail before its all-reduce: `sums` (device scalars, rank-AVERAGED by the collective) and
    `minmax` = up to 4 device scalars kept per rank (slot of this rank; the other ranks contribute zeros, and the factor
    world_size undoes the average), all without a host sync."""
    if len(sums) > SUM_SLOTS:
        raise RuntimeError("scalar tail: %d > %d slots" % (len(sums), SUM_SLOTS))
    tail.zero_()
    tail[:len(sums)].copy_(torch.stack([t.detach().reshape(()).float() for t in sums]))
    if minmax:
        r, ws = rank(), world_size()
        v = torch.stack([t.detach().reshape(()).float() for t in minmax]) * float(ws)
        tail[SUM_SLOTS + 4 * r: SUM_SLOTS + 4 * r + len(minmax)].copy_(v)


def read_scalar_tail(tail, nsums, nminmax=0):
    """-> (averaged sums [nsums], per-rank monitor matrix [world_size, nminmax]) as device tensors"""
    ws = world_size()
    mm = tail[SUM_SLOTS: SUM_SLOTS + 4 * ws].view(ws, 4)[:, :nminmax] if nminmax else None
    return tail[:nsums], mm


def average_scalars(vals, sq_keys=(), min_keys=(), max_keys=()):
    """Blocking host-side variant (kept for tools and tests; the training step uses the scalar tail instead):
    rank-mean of `vals`, MIN / MAX for the named keys.  `vals`: OrderedDict of floats."""
    ws = world_size()
    if not exchange_on():
        return vals
    keys = list(vals.keys())
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    sign = [(-1.0 if k in min_keys else 1.0) for k in keys]          # min(x) = -max(-x): one MAX serves both
    t = torch.tensor([vals[k] for k in keys], dtype=torch.float64, device=dev)
    mean = t.clone(); td.all_reduce(mean, op=td.ReduceOp.SUM); mean /= ws
    out = type(vals)(zip(keys, mean.tolist()))
    if min_keys or max_keys:
        mx = t * torch.tensor(sign, dtype=torch.float64, device=dev)
        td.all_reduce(mx, op=td.ReduceOp.MAX)
        for k in list(min_keys) + list(max_keys):
            i = keys.index(k)
            out[k] = float(mx[i]) * sign[i]
    return out
