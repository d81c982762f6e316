"""One simulation step (tree kernel + network forward) at the batch sizes of a self-play run's tail: ms per step of SelfPlayEngine.search at B boards with
the default self-play network (f16 operands).  A full-length cycle is bounded below by its longest game x searches x this latency.
Under `rocprofv3 --kernel-trace --stats -- python3 tools/tail_step_latency.py 64` the kernel statistics split the step into its launches."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet
from sigma_zero_amd.selfplay import SelfPlayEngine
sizes = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 128, 256, 512, 1024]
S = int(os.environ.get("SZ_SEARCHES", "800"))
torch.manual_seed(0)
net = sz.policyNN({}).cuda().eval()
fast = FastPolicyNet(net, operands=os.environ.get("SZ_OPERANDS", "fp16"))
for B in sizes:
    eng = SelfPlayEngine(fast, {"C": 2, "num_searches": S}, B, chess960=True, planes_dtype="bits128")
    eng.new_games([random.Random(B).randrange(960) for _ in range(B)])
    eng.search(); eng.play(np.random.RandomState(0).random_sample(B)); eng.fetch_ply()        # warm-up ply
    torch.cuda.synchronize(); t = time.perf_counter()
    plies = 2
    for p in range(plies):
        eng.search(); eng.play(np.random.RandomState(p + 1).random_sample(B)); eng.fetch_ply()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    # the forward alone at this batch, and the Python loop alone (launches with the GPU idle would hide behind the kernels)
    planes = eng.planes
    for _ in range(5): fast(planes, inference=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): fast(planes, inference=True)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 100
    print("B = %4d: %.3f ms per simulation step (%d searches x %d plies; %.0f simulations/s)   forward alone %.3f ms" % (B, dt / (plies * S) * 1e3, S, plies, B * plies * S / dt, tf * 1e3), flush=True)
    eng.close()
