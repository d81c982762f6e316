"""Launch time of the weight-gradient kernels alone (k_wgrad3x3_split + k_wgrad_reduce, sz_nn_wgrad3x3_split_f32) at B boards; with SIGMAZERO_LIB=ab/lib_wg_abl<N>.so
(-DWG_ABL=N at build time: 1 = no partial-sum stores, 2 = no MFMA loop, 4 = no stage / fragment preparation) the cost of the kernel's phases."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigma_zero_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
x = torch.randn(B, 256, 8, 8, device="cuda"); gy = torch.randn(B, 256, 8, 8, device="cuda") * 1e-3
amax = torch.stack((x.abs().amax(), gy.abs().amax())).view(torch.int32)
part = torch.empty(16 * 9 * 256 * 256, device="cuda"); dw = torch.empty(256, 256, 3, 3, device="cuda")
st = torch.cuda.current_stream().cuda_stream
run = lambda: N.check(N.lib().sz_nn_wgrad3x3_split_f32(gy.data_ptr(), x.data_ptr(), amax[1:].data_ptr(), amax[:1].data_ptr(), part.data_ptr(), dw.data_ptr(), B, st), "wgrad")
for _ in range(20): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): run()
e1.record(); torch.cuda.synchronize()
print("%s: B = %d: weight gradient + reduction %.2f us per call" % (os.environ.get("SIGMAZERO_LIB", "shipped library"), B, e0.elapsed_time(e1) / 300 * 1e3))
