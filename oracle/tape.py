"""ORACLE (test infrastructure): a minimal reverse-mode tape over NumPy arrays.

Stands in for torch.autograd, which the reference relies on for every
`loss.backward()` (/root/reference/augmented_cyclegan/model.py:158,186,445,509).
Each op in oracle/ops.py records an explicit adjoint closure, so the whole
backward pass of the step is a from-scratch restatement, not a torch call.
"""
import numpy as np


class T(object):
    """A value on the tape. `g` accumulates dL/dvalue during backward()."""
    __slots__ = ("v", "g", "parents", "bw", "req", "name")

    def __init__(self, v, parents=(), bw=None, req=False, name=None):
        self.v = v
        self.g = None
        self.parents = parents
        self.bw = bw
        self.req = req or any(p.req for p in parents)
        self.name = name

    @property
    def shape(self):
        return self.v.shape

    def detach(self):
        # model.py:424-430 `.detach()`: same value, no gradient path
        return T(self.v)


def leaf(v, req=True, name=None):
    return T(v, req=req, name=name)


def backward(loss, seed=None):
    """Reverse sweep from a scalar `loss` (autograd `loss.backward()`).
    Gradients accumulate into `.g` of every reachable node with req=True;
    leaves keep accumulating across calls until zero_grad() (torch semantics)."""
    order, seen = [], set()
    stack = [(loss, False)]
    while stack:
        node, done = stack.pop()
        if done:
            order.append(node)
            continue
        if id(node) in seen:
            continue
        seen.add(id(node))
        stack.append((node, True))
        for p in node.parents:
            if p.req and id(p) not in seen:
                stack.append((p, False))
    g0 = np.ones_like(loss.v) if seed is None else seed
    loss.g = g0 if loss.g is None else loss.g + g0
    for node in reversed(order):
        if node.bw is None or node.g is None:
            continue
        grads = node.bw(node.g)
        for p, gp in zip(node.parents, grads):
            if gp is None or not p.req:
                continue
            p.g = gp if p.g is None else p.g + gp
        if node.parents:
            node.g = None  # interior node: free


def zero_grad(params):
    for p in params:
        p.g = None
