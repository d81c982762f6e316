"""wall time of the persistent tower at B boards (bit-packed planes), mean over N back-to-back launches after a 2 s warm-up; used with SIGMAZERO_LIB for A/B builds"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet, planes_nchw_to_nhwc128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
planes = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
img = planes.float().to(torch.uint8).view(B, 16, 4, 16, 8)
planes = (img << torch.arange(8, device="cuda", dtype=torch.uint8)).sum(-1).to(torch.uint8).permute(0, 2, 3, 1).reshape(B, 1024).contiguous()
t_end = time.time() + 2.0
while time.time() < t_end:
    fast.tower(planes)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    fast.tower(planes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("%s: tower B=%d %.4f ms per launch (%.1f TFLOP/s algorithmic)" % (os.environ.get("SIGMAZERO_LIB", "in-tree").split("/")[-1], B, dt * 1e3, 2.0 * B * 64 * 256 * 9 * (119 + 38 * 256) / dt / 1e12))
