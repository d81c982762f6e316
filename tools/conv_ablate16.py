import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import _pack
def timeit(fn, n=30, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
B = 4096
x = torch.randn(B, 64, 256, device="cuda").to(torch.bfloat16); res = torch.randn(B, 64, 256, device="cuda").to(torch.bfloat16); out = torch.empty_like(x)
w = _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda", w16=True); bias = torch.zeros(256, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(mode, r=None):
    def f():
        N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(res.data_ptr()) if r else None, C.c_void_p(out.data_ptr()), B, 256, 3, mode | N.SZ_NN_W16, st)
    return f
for rep in range(2):
    for mode, name in ((1, "full"), (1 | 0x100000, "full, L1-hot weights"), (1 | 0x10000, "full, no stagger"), (1 | 32 | (1 << 8), "stagger n=1"), (1 | 32 | (6 << 8), "stagger n=6"), (3, "no tile load"), (5, "no store"), (7, "K loop only (zeros)"), (5 | 0, "load+K (real data)"), (9, "no K loop")):
        print("%-22s %.3f ms" % (name, timeit(mk(mode)) * 1e3))
    print("full+res               %.3f ms" % (timeit(mk(1, True)) * 1e3))
