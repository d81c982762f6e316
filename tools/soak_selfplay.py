"""Soak run of the product API at full width: N games on 4096 board slots to their natural end (refill, compaction, every kind of game end),
FastPolicyNet, S searches.  Prints a progress line every 20 plies; ends with per-result counts and the throughput."""
import sys, os, time, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet
import sigma_zero_amd.sim as sim
N_GAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
SLOTS = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
torch.manual_seed(0); random.seed(2); np.random.seed(2)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
st = {}
t0 = time.time()
_print = print
ply_counter = [0]
def quiet_print(*a, **k):
    ply_counter[0] += 1
    if ply_counter[0] % 20 == 0:
        _print("[%6.0f s]" % (time.time() - t0), *a, flush=True)
import builtins
builtins.print = quiet_print
games = sim.play_games(fast, {"C": 2, "num_searches": S}, N_GAMES, c960=True, n_boards=SLOTS, max_plies=1200, stats=st, verbose=True)
builtins.print = _print
dt = time.time() - t0
lens = np.array([len(g["actions"]) for g in games])
res = collections.Counter(g["result"] for g in games)
print("soak: %d games on %d slots x %d searches: %d plies of self-play, %d simulations in %.1f s = %.0f simulations/s; network rows / simulations %.4f"
      % (N_GAMES, SLOTS, S, st["plies"], st["sims"], dt, st["sims"] / dt, st["nn_rows"] / st["sims"]))
print("game length min / median / max: %d / %d / %d; results %s" % (lens.min(), np.median(lens), lens.max(), dict(res)))
assert all(g["result"] is not None or len(g["actions"]) == 1200 for g in games)
