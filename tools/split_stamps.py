"""In-kernel phase timing of the split-precision tower (diagnostic build; MI355X_MICROARCH.md 'DVFS give-back' item 6): cycles of K loop / barrier /
epilogue / barrier for convolutions 7 (a conv1) and 8 (a conv2) of each workgroup's second tile, and the shader clock held under load."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import SplitPolicyNet, planes_nchw_to_nhwc128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
split = SplitPolicyNet(sz.policyNN({}).cuda().eval())
planes = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, device="cuda") < 0.12).float())
t_end = time.time() + 2.0
while time.time() < t_end:                       # >= 2 s of back-to-back launches so the clock has settled
    split.tower(planes)
torch.cuda.synchronize()
modes = [int(m) for m in sys.argv[2:]] or [1]
names = ["conv1 K loop", "barrier", "epilogue (t -> hi/lo images)", "barrier", "conv2 K loop", "barrier", "epilogue (residual, x -> images)", "barrier"]
ideal = 72 * 96 * 16 * 66.0 / 72.0
print("ideal K loop = 72 k-steps x 96 MFMAs x 16 cycles x 66/72 (border rows skipped) = %.0f MFMA cycles" % ideal)
for mode in modes:
    buf = torch.zeros(256 * 4 * 16, dtype=torch.int64, device="cuda")
    N.check(N.lib().sz_nn_debug_split_stamps(C.c_void_p(buf.data_ptr()), mode), "stamps")
    t0 = time.perf_counter()
    for _ in range(3):
        split.tower(planes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    N.lib().sz_nn_debug_split_stamps(None, 1)
    s = buf.cpu().numpy().reshape(256, 4, 16).astype(np.float64)
    s = s[s[:, :, 0] > 0].reshape(-1, 16)
    d = np.concatenate([np.diff(s[:, 0:5], axis=1), np.diff(s[:, 5:10], axis=1)], axis=1)
    print("mode %d (%s): waves with stamps: %d; launch %.2f ms" % (mode, {1: "full", 2: "no weight loads", 3: "no LDS reads", 4: "MFMA only"}[mode], len(s), dt * 1e3))
    for i, n in enumerate(names):
        print("  %-34s median %8.0f  p10 %8.0f  p90 %8.0f cycles" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    tot = s[:, 9] - s[:, 0]
    print("  two convolutions: median %.0f cycles; MFMA-busy fraction %.3f" % (np.median(tot), 2 * ideal / np.median(tot)))
