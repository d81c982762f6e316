# minimal launcher for rocprofv3 --pmc passes on the tree kernels: 4096 boards, 128 lock-step simulations with a fixed
# synthetic (policy, value) — no network, so the counter files only hold k_search_begin / k_search_step / k_play
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sigma_zero_amd import _native as N
from sigma_zero_amd.selfplay import SelfPlayEngine
B, S = 4096, 128
eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, chess960=False, learning=True, planes_dtype=(sys.argv[1] if len(sys.argv) > 1 else "bits128"))
eng.new_games([-1] * B)
g = torch.Generator(device="cuda").manual_seed(0)
policy = torch.softmax(torch.randn(B, N.SZ_ACTIONS, generator=g, device="cuda"), 1).contiguous()
value = (torch.rand(B, generator=g, device="cuda") * 2 - 1).contiguous()
for ply in range(2):
    eng.begin()
    for _ in range(S):
        eng.step(policy, value)
    eng.play(np.random.RandomState(ply).random_sample(B))
    eng.fetch_ply()
st = eng.check_errors()
print("sims", st["simulations"], "mean depth", st["sum_depth"] / st["simulations"], "mean K", st["sum_children"] / st["expansions"])
