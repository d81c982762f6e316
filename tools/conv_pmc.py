# minimal launcher for rocprofv3 --pmc passes: N launches of the 3x3 C=256 conv at B=4096 (with residual), nothing else
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import _pack
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
use_res = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
x = torch.randn(B, 64, 256, device="cuda").to(torch.bfloat16)
res = torch.randn(B, 64, 256, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
w = _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda")
bias = torch.zeros(256, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(12):
    N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()),
                            C.c_void_p(res.data_ptr()) if use_res else None, C.c_void_p(out.data_ptr()), B, 256, 3, 1, st)
torch.cuda.synchronize()
