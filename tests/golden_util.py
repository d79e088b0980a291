"""Helpers to read tests/golden/*.npz fixtures (written by tools/make_goldens.py)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    arr = {k: z[k] for k in z.files if k != "__meta__"}
    meta = json.loads(bytes(z["__meta__"]).decode())
    return arr, meta


def names(kind):
    out = []
    for f in sorted(os.listdir(GOLDEN_DIR)):
        if f.endswith(".npz"):
            _, meta = load(f[:-4])
            if meta["kind"] == kind:
                out.append(f[:-4])
    return out


def rel_err(a, b):
    """max|a-b| / max|b| — the relative measure used for the 1e-3 bar (BASELINE.json north_star)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def digest(a):
    a = np.asarray(a, np.float64).ravel()
    idx = (np.arange(8) * 2654435761 % max(a.size, 1)).astype(np.int64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], a[idx]])
