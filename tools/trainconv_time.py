"""Launch time of the training convolution kernel alone (k_conv3x3_split_f32, hi + lo f16 operands) at B boards; with SIGMAZERO_LIB=ab/lib_conv_abl<N>.so
(SIGMAZERO_EXTRA_FLAGS=-DSZ_CONV_ABL=N at build time: 1 = no K loop, 2 = no LDS stage writes, 4 = a quarter of the stores) it prices the kernel's phases."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from sigma_zero_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
x = torch.randn(B, 256, 8, 8, device="cuda")
w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
buf = torch.empty(72 * 2048 * 16, dtype=torch.uint8, device="cuda")
zero = torch.zeros(256, device="cuda")
y = torch.empty_like(x)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
N.check(N.lib().sz_nn_pack_conv_split_dev(C.c_void_p(w.data_ptr()), 0, 1, C.c_void_p(buf.data_ptr()), None, st), "pack")
def run():
    N.check(N.lib().sz_nn_conv3x3_split_f32(C.c_void_p(x.data_ptr()), C.c_void_p(buf.data_ptr()), C.c_void_p(zero.data_ptr()), C.c_void_p(y.data_ptr()), B, 1, None, st), "conv")
for _ in range(20): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): run()
e1.record(); torch.cuda.synchronize()
print("%s: B = %d: %.2f us per launch" % (os.environ.get("SIGMAZERO_LIB", "shipped library"), B, e0.elapsed_time(e1) / 300 * 1e3))
