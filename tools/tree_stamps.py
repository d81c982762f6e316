"""In-kernel phase timing of k_search_step (diagnostic stamps): cycles per phase, median over boards, for a mid-search simulation"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from sigma_zero_amd import _native as N
from sigma_zero_amd.selfplay import SelfPlayEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = 200
eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, chess960=False, learning=True, planes_dtype="bits128")
eng.new_games([-1] * B)
g = torch.Generator(device="cuda").manual_seed(0)
NPOL = 8 if B >= 2048 else 32            # rotate through more policy tensors than the Infinity Cache holds: every step reads cold data,
policies = [torch.softmax(torch.randn(B, N.SZ_ACTIONS, generator=g, device="cuda"), 1).contiguous() for _ in range(NPOL)]   # as after a real forward
value = (torch.rand(B, generator=g, device="cuda") * 2 - 1).contiguous()
eng.begin()
for i in range(150):
    eng.step(policies[i % NPOL], value)
buf = torch.zeros(B * 8, dtype=torch.int64, device="cuda")
N.check(N.lib().sz_debug_step_stamps(eng._e, C.c_void_p(buf.data_ptr())), "stamps")
acc = []
import time
for i in range(20):
    buf.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.step(policies[i % NPOL], value)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    s = buf.cpu().numpy().reshape(B, 8).astype(np.float64)
    ok = (s[:, :5] > 0).all(1)
    acc.append(np.diff(s[ok, :5], axis=1))
N.lib().sz_debug_step_stamps(eng._e, None)
d = np.concatenate(acc)
names = ["expand + backprop", "select (descent)", "move + movegen + repetition + terminal", "history + encode"]
print("B=%d: %d samples" % (B, len(d)))
for i, n in enumerate(names):
    print("  %-40s median %7.0f  p10 %7.0f  p90 %7.0f cycles" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("  total median %.0f cycles; last launch wall (incl. launch + sync) %.1f us" % (np.median(d.sum(1)), wall * 1e6))
eng.close()
