# minimal launcher for rocprofv3 --pmc passes: N launches of the fused BasicBlock kernel (16x16x32 path) at B=4096
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import _pack
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
x = torch.randn(B, 64, 256, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
w1 = _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda", w16=True)
w2 = _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda", w16=True)
bias = torch.zeros(256, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(24):
    N.lib().sz_nn_block_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w1.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(w2.data_ptr()),
                             C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), B, N.SZ_NN_W16, st)
torch.cuda.synchronize()
