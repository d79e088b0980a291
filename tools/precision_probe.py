"""Conv forward / data-gradient error of the three arithmetic modes against an fp64 torch CPU conv (GPU box tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from dtgan_amd import ops, modules as M

def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())

def rms(a, b):
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))

torch.manual_seed(0)
for (Ci, Co, K, H) in [(128, 128, 3, 64), (64, 128, 3, 64), (256, 256, 4, 32)]:
    x = torch.randn(2, Ci, H, H, dtype=torch.float64); w = torch.randn(Co, Ci, K, K, dtype=torch.float64) * 0.05
    r = torch.randn(2, Co, H - K + 1 + 2, H - K + 1 + 2, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    y64 = F.conv2d(xr, w, None, 1, 1); y64.backward(r)
    for prec in ("f32", "bf16x3", "bf16"):
        ops.set_precision(prec)
        conv = M.Conv2d(Ci, Co, K, 1, 1, bias=False).cuda()
        with torch.no_grad():
            conv.weight.copy_(w.float())
        xg = x.float().cuda().requires_grad_(True)
        y = conv(xg)
        y.backward(r.float().cuda())
        print("%dx%d k%d %-7s fwd max %.2e rms %.2e | dgrad max %.2e rms %.2e" % (
            Ci, Co, K, prec, rel(y.detach().cpu().double().numpy(), y64.detach().numpy()), rms(y.detach().cpu().double().numpy(), y64.detach().numpy()),
            rel(xg.grad.cpu().double().numpy(), xr.grad.numpy()), rms(xg.grad.cpu().double().numpy(), xr.grad.numpy())))
