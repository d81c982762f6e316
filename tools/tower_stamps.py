"""In-kernel phase timing of the persistent tower (diagnostic build, MI355X_MICROARCH.md 'DVFS give-back' item 6):
cycles of K loop / epilogue / barrier for block 3 of each workgroup's second tile, and the shader clock held under load."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import FastPolicyNet, planes_nchw_to_nhwc128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
OPERANDS = os.environ.get("SZ_OPERANDS", "bf16")          # SZ_OPERANDS=fp16: the f16-operand tower (mode 1 only)
torch.manual_seed(0)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval(), operands=OPERANDS)
fast.persistent_max_boards = 1 << 30
planes = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
t_end = time.time() + 2.0
while time.time() < t_end:                       # >= 2 s of back-to-back launches so the clock has settled
    fast.tower(planes)
torch.cuda.synchronize()
modes = [int(m) for m in sys.argv[2:]] or [1]
names = ["conv1 K loop", "epilogue1 (acc->LDS)", "barrier1", "conv2 K loop", "epilogue2 (residual)", "barrier2"]
WGB = 1 if B <= torch.cuda.get_device_properties(0).multi_processor_count else 2
print("ideal K loop = 72 k-steps x %d = %d MFMA cycles (%d board%s per workgroup)" % (256 * WGB, 72 * 256 * WGB, WGB, "s" if WGB == 2 else ""))
for mode in modes:
    buf = torch.zeros(256 * 4 * 16, dtype=torch.int64, device="cuda")
    N.check(N.lib().sz_nn_debug_tower_stamps(C.c_void_p(buf.data_ptr()), mode), "stamps")
    for _ in range(3):
        fast.tower(planes)
    torch.cuda.synchronize()
    N.lib().sz_nn_debug_tower_stamps(None, 1)
    raw = buf.cpu().numpy().astype(np.float64)
    tl = raw[8192:].reshape(256 * 4, 8)
    tl = tl[tl[:, 0] > 0]
    if len(tl):
        td = np.diff(tl[:, :7], axis=1)
        for i, n in enumerate(["tile: pacing wait + barrier", "tile: stage planes + barrier", "tile: stem K loop", "tile: stem epilogue + barrier", "tile: %d blocks" % len(fast.blocks), "tile: output store"]):
            print("  %-30s median %8.0f  p10 %8.0f  p90 %8.0f cycles" % (n, np.median(td[:, i]), np.percentile(td[:, i], 10), np.percentile(td[:, i], 90)))
        print("  tile total median %.0f cycles" % np.median(tl[:, 6] - tl[:, 0]))
    s = raw[:8192].reshape(256, 4, 8)
    s = s[s[:, :, 0] > 0].reshape(-1, 8)
    d = np.diff(s[:, :7], axis=1)
    print("mode %d (%s): waves with stamps: %d" % (mode, {1: "full", 2: "no weight loads", 3: "no LDS reads", 4: "MFMA only", 5: "border-row skipping OFF (all 72 tap-tiles multiply)"}[mode], len(s)))
    for i, n in enumerate(names):
        print("  %-22s median %8.0f  p10 %8.0f  p90 %8.0f cycles" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    tot = s[:, 6] - s[:, 0]
    mf = 2 * 72 * 256 * WGB * (1.0 if mode == 5 else 66.0 / 72.0)
    print("  block total median %.0f cycles; MFMA-busy fraction %.3f (%.0f MFMA cycles per block)" % (np.median(tot), mf / np.median(tot), mf))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fast.tower(planes)
    e1.record(); torch.cuda.synchronize()
    print("  operands %s: %.3f ms per launch of the shipped kernel (same process, right after)" % (OPERANDS, e0.elapsed_time(e1) / 10))
