import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import FastPolicyNet, _pack, planes_nchw_to_nhwc128
from sigma_zero_amd.network import FLOPS_PER_BOARD

def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

for B in (512, 4096):
    x = torch.randn(B, 64, 256, device="cuda").to(torch.bfloat16)
    res = torch.randn(B, 64, 256, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    w = _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda")
    bias = torch.zeros(256, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(r=None):
        N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(res.data_ptr()) if r else None, C.c_void_p(out.data_ptr()), B, 256, 3, 1, st)
    dt = timeit(lambda: run()); fl = 2 * B * 64 * 256 * 2304
    print("conv3x3 B=%d: %.3f ms  %.1f TFLOP/s" % (B, dt * 1e3, fl / dt / 1e12))
    dt = timeit(lambda: run(True))
    print("conv3x3+res B=%d: %.3f ms  %.1f TFLOP/s" % (B, dt * 1e3, fl / dt / 1e12))
    for rep in range(3):
        for mode, name in ((1, "32x32x16 (real)"), (1 | 0x20000, "16x16x32 probe")):
            def runp():
                N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, C.c_void_p(out.data_ptr()), B, 256, 3, mode, st)
            dt = timeit(runp, n=30)
            print("   shape %-18s: %.3f ms" % (name, dt * 1e3))
    for sel, sname in ((0x10000, "off"),):
        for n in ((0,) if sel == 0x10000 else (1, 3)):
            mode = 1 | sel | (n << 8)
            def runs(r=None):
                N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(res.data_ptr()) if r else None, C.c_void_p(out.data_ptr()), B, 256, 3, mode, st)
            dt = timeit(lambda: runs()); dt2 = timeit(lambda: runs(True))
            print("   stagger %-10s n=%d: %.3f ms  (+res %.3f ms)" % (sname, n, dt * 1e3, dt2 * 1e3))
    for mode, name in ((17, "4-board WG (1/CU)"), (3, "no prologue loads"), (5, "no epilogue stores"), (7, "K loop only"), (9, "no K loop"), (15, "empty")):
        def runm():
            N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, C.c_void_p(out.data_ptr()), B, 256, 3, mode, st)
        dt = timeit(runm)
        print("   ablation %-20s: %.3f ms" % (name, dt * 1e3))
    w16a = _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda", w16=True)
    for rep in range(2):
        for r in (None, True):
            def run16():
                N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w16a.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(res.data_ptr()) if r else None, C.c_void_p(out.data_ptr()), B, 256, 3, 1 | N.SZ_NN_W16, st)
            dt = timeit(run16, n=30)
            print("conv3x3 mfma16%s B=%d: %.3f ms  %.1f TFLOP/s" % ("+res" if r else "", B, dt * 1e3, fl / dt / 1e12))
    def runb16():
        N.lib().sz_nn_block_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w16a.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(w16a.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), B, N.SZ_NN_W16, st)
    dt = timeit(runb16)
    print("fused block mfma16 B=%d: %.3f ms  %.1f TFLOP/s" % (B, dt * 1e3, 2 * fl / dt / 1e12))
    def runb16_1():
        N.lib().sz_nn_block_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w16a.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(w16a.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), B, N.SZ_NN_W16 | 0x80000, st)
    for rep in range(2):
        dt = timeit(runb16_1); dt2 = timeit(runb16)
        print("fused block mfma16 1-board WGs B=%d: %.3f ms  (2-board: %.3f ms)" % (B, dt * 1e3, dt2 * 1e3))
    w2 = _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda")
    def runb():
        N.lib().sz_nn_block_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), B, 0, st)
    dt = timeit(runb)
    print("fused block B=%d: %.3f ms  %.1f TFLOP/s" % (B, dt * 1e3, 2 * fl / dt / 1e12))
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    fast32 = FastPolicyNet(net, mfma16=False)
    fast = FastPolicyNet(net)
    planes = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
    dt = timeit(lambda: fast(planes, inference=True), n=10)
    print("FastPolicyNet B=%d: %.2f ms  %.1f TFLOP/s  %.0f evals/s" % (B, dt * 1e3, B * FLOPS_PER_BOARD / dt / 1e12, B / dt))
    dt = timeit(lambda: fast32(planes, inference=True), n=10)
    print("FastPolicyNet(mfma32) B=%d: %.2f ms  %.1f TFLOP/s  %.0f evals/s" % (B, dt * 1e3, B * FLOPS_PER_BOARD / dt / 1e12, B / dt))
    for rep in range(2):
        fast.persistent_tower = False
        dt = timeit(lambda: fast(planes, inference=True), n=10)
        print("FastPolicyNet(per-block launches) B=%d: %.2f ms  %.0f evals/s" % (B, dt * 1e3, B / dt))
        fast.persistent_tower = True; fast.persistent_max_boards = 1 << 30
        dt = timeit(lambda: fast(planes, inference=True), n=10)
        print("FastPolicyNet(persistent tower) B=%d: %.2f ms  %.0f evals/s" % (B, dt * 1e3, B / dt))
        fast.persistent_max_boards = 1024
    dt = timeit(lambda: fast.tower(planes), n=10)
    print("  tower only: %.2f ms" % (dt * 1e3))
