"""Forward latency at small batches (a search of one position: eval.py / play.py of the reference; the tail of a self-play run): the persistent towers with
one board per workgroup (default up to #CUs boards) against the two-board form, for the three MFMA networks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import FastPolicyNet, SplitPolicyNet, planes_nchw_to_nhwc128
torch.manual_seed(0)
net = sz.policyNN({}).cuda().eval()
nets = {"bf16": FastPolicyNet(net), "fp16": FastPolicyNet(net, operands="fp16"), "split": SplitPolicyNet(net)}
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("forward ms (whole network), one board per workgroup / two boards per workgroup")
for B in (1, 16, 64, 128, 256):
    x = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
    row = []
    for name, f in nets.items():
        w1, w2 = (N.SZ_NN_SPLIT_WGB1, N.SZ_NN_SPLIT_WGB2) if name == "split" else (N.SZ_NN_TOWER_WGB1, N.SZ_NN_TOWER_WGB2)
        f.force_wgb = w1; t1 = timeit(lambda: f(x))
        f.force_wgb = w2; t2 = timeit(lambda: f(x))
        f.force_wgb = 0
        row.append("%s %.3f / %.3f" % (name, t1, t2))
    print("B = %3d: %s" % (B, "   ".join(row)))
