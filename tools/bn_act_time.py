"""Launch times of the fused BatchNorm + skip + ReLU kernels (k_bn_act_fwd / k_bn_act_bwd) at B boards, beside torch's own launches for the same site."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigma_zero_amd.trainconv import BNAct
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(B, 256, 8, 8, device="cuda", requires_grad=True); res = torch.randn(B, 256, 8, 8, device="cuda", requires_grad=True); gy = torch.randn(B, 256, 8, 8, device="cuda")
bn = torch.nn.BatchNorm2d(256).cuda().train()
def timeit(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
def fused():
    y = BNAct.apply(x, bn.weight, bn.bias, res, bn.running_mean, bn.running_var, bn.momentum, bn.eps); y.backward(gy)
def plain():
    y = torch.relu(bn(x) + res); y.backward(gy)
print("B = %d: BatchNorm + skip + ReLU forward + backward: fused %.1f us, torch %.1f us (wall per pair of passes, eager)" % (B, timeit(fused), timeit(plain)))
