import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import sigma_zero_amd as sz
from sigma_zero_amd.network import FLOPS_PER_BOARD

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

torch.manual_seed(0)
net = sz.policyNN({}).cuda().eval()
print(torch.__version__, torch.cuda.get_device_name(0))
for B in (512, 4096):
    for dtype, cl in ((torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True), (torch.float16, True)):
        m = sz.policyNN({}).cuda().eval().to(dtype)
        if cl: m = m.to(memory_format=torch.channels_last)
        x = (torch.rand(B, 119, 8, 8, device="cuda") < 0.15).to(dtype)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            try:
                dt = timeit(lambda: m(x, inference=True))
                print("B=%d %s cl=%s: %.2f ms  %.1f TFLOP/s  %.0f evals/s" % (B, dtype, cl, dt*1e3, B*FLOPS_PER_BOARD/dt/1e12, B/dt))
            except Exception as e:
                print("B=%d %s cl=%s failed: %s" % (B, dtype, cl, e))
# single conv layer microbench
for B in (4096,):
    for dtype in (torch.bfloat16,):
        for cl in (False, True):
            conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).cuda().to(dtype)
            x = torch.randn(B, 256, 8, 8, device="cuda", dtype=dtype)
            if cl:
                conv = conv.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                dt = timeit(lambda: conv(x), n=20)
            fl = 2*B*64*256*256*9
            print("conv3x3 B=%d %s cl=%s: %.3f ms %.1f TFLOP/s" % (B, dtype, cl, dt*1e3, fl/dt/1e12))
# same GEMM as matmul for reference: [B*64, 2304] x [2304, 256]
a = torch.randn(4096*64, 2304, device="cuda", dtype=torch.bfloat16); w = torch.randn(2304, 256, device="cuda", dtype=torch.bfloat16)
dt = timeit(lambda: a @ w, n=20)
print("gemm 262144x2304x256 bf16: %.3f ms %.1f TFLOP/s" % (dt*1e3, 2*4096*64*2304*256/dt/1e12))
