"""times FastPolicyNet's head stage (fused sz_nn_heads_bf16 vs conv_p1 + policy head + value head launches) at B boards"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet, planes_nchw_to_nhwc128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
planes = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
real_tower = fast.tower
x, scratch = real_tower(planes)
fast.tower = lambda p: (x, scratch)          # time the heads only
def timeit(n=50):
    for _ in range(5): fast(planes)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fast(planes)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for rep in range(3):
    for fused in (False, True):
        fast.fused_heads = fused
        print("B=%d fused=%s heads %.1f us" % (B, fused, timeit()), flush=True)
