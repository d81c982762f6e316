"""FastPolicyNet forward at B boards: per-block launches vs the persistent whole-tower kernel (interleaved samples)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet, planes_nchw_to_nhwc128
from sigma_zero_amd.network import FLOPS_PER_BOARD
torch.manual_seed(0)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
for B in [int(b) for b in sys.argv[1:]] or [4096]:
    planes = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
    def timeit(n=10):
        for _ in range(3): fast(planes)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fast(planes)
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    for rep in range(4):
        fast.persistent_max_boards = 0
        a = timeit()
        fast.persistent_max_boards = 1 << 30
        b = timeit()
        print("B=%d forward: per-block %.3f ms (%.0f TFLOP/s)   persistent tower %.3f ms (%.0f TFLOP/s)" % (B, a, B * FLOPS_PER_BOARD / a / 1e9, b, B * FLOPS_PER_BOARD / b / 1e9), flush=True)
