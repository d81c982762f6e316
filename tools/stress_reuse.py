import sys, os, time, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet
torch.manual_seed(0); random.seed(1); np.random.seed(1)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
for reuse in (False, True):
    st = {}
    t0 = time.time()
    args = {"C": 2, "num_searches": 200}
    if reuse: args["reuse_subtree"] = True
    games = sz.sim.play_games(fast, args, 1536, c960=True, n_boards=1024, max_plies=60, stats=st)
    dt = time.time() - t0
    n = sum(len(g["actions"]) for g in games)
    print("reuse=%s: %d games, %d samples, %d sims in %.1f s = %.0f sims/s; nn_rows/sims %.3f; finished %d" % (reuse, len(games), n, st["sims"], dt, st["sims"] / dt, st["nn_rows"] / st["sims"], sum(g["result"] is not None for g in games)), flush=True)
