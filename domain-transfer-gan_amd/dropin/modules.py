"""Drop-in for the reference's bare `import modules` (train.py:14, model.py:10, networks.py:7):
put this directory on sys.path instead of /root/reference/augmented_cyclegan."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
import dtgan_amd as _pkg  # noqa: E402
from dtgan_amd.modules import *  # noqa: E402,F401,F403
from dtgan_amd import modules as _m  # noqa: E402

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
