"""FETCH_SIZE calibration for the two 16-B-per-lane read paths the kernels use (MI355X_MICROARCH.md, HBM section: 'calibrate on a known byte count in your
own access pattern'): k_stream_read reads a 1 GiB buffer exactly once with flat global_load_dwordx4 (mode 0) and with buffer_load_dwordx4 through a
descriptor (mode 1, the way k_tower16_bf16 streams its weights since round 2).  Run under  rocprofv3 --pmc FETCH_SIZE  and divide."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from sigma_zero_amd import _native as N
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 1 << 30
buf = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
flush = torch.zeros(1 << 29, dtype=torch.uint8, device="cuda")
sink = torch.zeros(16, dtype=torch.uint8, device="cuda")
for _ in range(3):
    flush.add_(1)                                      # 512 MiB of other traffic: the buffer's lines leave L2 and the Infinity Cache
    N.check(N.lib().sz_debug_stream_read(C.c_void_p(buf.data_ptr()), n, mode, C.c_void_p(sink.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stream_read")
torch.cuda.synchronize()
print("mode", mode, "bytes read per launch", n)
