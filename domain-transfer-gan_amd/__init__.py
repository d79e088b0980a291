"""MI355X-native Augmented CycleGAN training step (hot path of adrianalbert/domain-transfer-GAN).

Import as `dtgan_amd` (repo-root shim `dtgan_amd.py`; the directory name carries a hyphen), or put
`domain-transfer-gan_amd/dropin` on sys.path to get the reference's bare module names
(`import model, networks, modules`) so the reference's train.py-style callers drop in.
"""
from . import _lib, dist, model, modules, networks, ops  # noqa: F401
from . import dataloader, evaluate, options  # noqa: F401  (driver side; `train` is imported on demand)
from .model import AugmentedCycleGAN, AugmentedCycleGAN_Model, StochCycleGAN  # noqa: F401

__version__ = "0.1.0"
