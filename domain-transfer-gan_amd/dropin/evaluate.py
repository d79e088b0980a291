"""Drop-in for the reference's bare `import evaluate`: put this directory on sys.path instead of
/root/reference/augmented_cyclegan (the reference's own evaluate.py is Python 2)."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
import dtgan_amd as _pkg  # noqa: E402
from dtgan_amd.evaluate import *  # noqa: E402,F401,F403
from dtgan_amd import evaluate as _m  # noqa: E402

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
if __name__ == "__main__" and hasattr(_m, "train_model"):
    _m.train_model()
