"""trainconv.BNAct (fused train-mode BatchNorm + skip + ReLU) vs torch fp32 vs fp64: relative L2 errors of output, statistics and gradients; time per launch pair."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigma_zero_amd.trainconv import BNAct
g = torch.Generator(device="cuda").manual_seed(3)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for B in (8, 128, 300):
    for with_res in (False, True):
        C = 256
        x = (torch.randn(B, C, 8, 8, device="cuda", generator=g) * 1.7 + 0.3)
        res = torch.randn(B, C, 8, 8, device="cuda", generator=g) if with_res else None
        gy = torch.randn(B, C, 8, 8, device="cuda", generator=g)
        out = {}
        for kind in ("fp64", "torch", "fused"):
            dt = torch.float64 if kind == "fp64" else torch.float32
            bn = torch.nn.BatchNorm2d(C).cuda().to(dt).train()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C)); bn.running_mean.fill_(0.1); bn.running_var.fill_(0.8)
            xx = x.detach().to(dt).requires_grad_(True)
            rr = res.detach().to(dt).requires_grad_(True) if with_res else None
            if kind == "fused":
                y = BNAct.apply(xx, bn.weight, bn.bias, rr, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
            else:
                y = torch.relu(bn(xx) if rr is None else bn(xx) + rr)
            y.backward(gy.to(dt))
            out[kind] = [y.detach(), bn.running_mean.clone(), bn.running_var.clone(), xx.grad, bn.weight.grad, bn.bias.grad] + ([rr.grad] if with_res else [])
        names = ["y", "running_mean", "running_var", "dx", "dgamma", "dbeta"] + (["dres"] if with_res else [])
        print("B=%d residual=%s: " % (B, with_res) + "  ".join("%s %.1e/%.1e" % (n, rel(out["fused"][i], out["fp64"][i]), rel(out["torch"][i], out["fp64"][i])) for i, n in enumerate(names)) + "   (fused / torch fp32, rel L2 vs fp64)", flush=True)
