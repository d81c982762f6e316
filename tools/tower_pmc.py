# minimal launcher for rocprofv3 passes: the persistent whole-tower kernel (k_tower16_bf16; with a third argument `split`: k_tower_split) at B boards, 12 launches
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet, planes_nchw_to_nhwc128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
fast.persistent_max_boards = 1 << 30
planes = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
if len(sys.argv) > 2 and sys.argv[2] == "bits":          # the engine's bit-packed image (what bench.py feeds by default)
    img = planes.float().to(torch.uint8).view(B, 16, 4, 16, 8)
    planes = (img << torch.arange(8, device="cuda", dtype=torch.uint8)).sum(-1).to(torch.uint8).permute(0, 2, 3, 1).reshape(B, 1024).contiguous()
if len(sys.argv) > 3 and sys.argv[3] == "split":          # the split-precision tower (k_tower_split) instead
    from sigma_zero_amd.fastnet import SplitPolicyNet
    fast = SplitPolicyNet(sz.policyNN({}).cuda().eval())
if len(sys.argv) > 4:                                     # diagnostic build of the bf16 tower under the counters: 3 = no LDS fragment reads in the K loop, ...
    import ctypes as C
    from sigma_zero_amd import _native as N
    stamps = torch.zeros(256 * 4 * 16, dtype=torch.int64, device="cuda")
    N.check(N.lib().sz_nn_debug_tower_stamps(C.c_void_p(stamps.data_ptr()), int(sys.argv[4])), "stamps")
for _ in range(12):
    fast.tower(planes)
torch.cuda.synchronize()
