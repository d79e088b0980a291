"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (NumPy + plain C for the conv loops) of the Augmented CycleGAN
training step of adrianalbert/domain-transfer-GAN:
  /root/reference/augmented_cyclegan/modules.py   (all)
  /root/reference/augmented_cyclegan/networks.py  (13-21, 149-252, 308-482)
  /root/reference/augmented_cyclegan/model.py     (15-72, 126-208, 327-539)
plus the torch semantics those files delegate to (Conv2d, ConvTranspose2d,
ReflectionPad2d, BatchNorm train mode, LeakyReLU, mse/l1 loss, clip_grad_norm,
Adam) — torch is an unpinned third-party dependency of the reference.

Parity pin: the reference ships NO tests / golden vectors, so the
oracle is pinned against outputs of the reference itself, imported in the build
container from /root/reference (tools/make_goldens.py, torch 2.10.0 CPU) and
committed as small fixtures under tests/golden/.  tests/test_oracle_golden.py
checks the oracle against every one of them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.
"""
