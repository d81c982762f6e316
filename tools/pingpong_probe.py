"""A/B probe: one 4096-board engine on one stream vs two 2048-board engines ping-ponged on two streams
(SURVEY §8(f)#3).  Prints ms per simulation step for both arrangements."""
import copy, sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet
from sigma_zero_amd.selfplay import SelfPlayEngine

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = FastPolicyNet(sz.policyNN({}).eval(), device=dev)
args = {"C": 2, "num_searches": S}

def lane(m):
    o = copy.copy(m); o._bufs = {}; return o

def run(n_lanes):
    engs = [SelfPlayEngine(None, args, B // n_lanes, device=dev, planes_dtype="nhwc128") for _ in range(n_lanes)]
    nets = [lane(model) for _ in range(n_lanes)]
    streams = [torch.cuda.Stream(dev) for _ in range(n_lanes)]
    for e in engs: e.new_games(None)
    torch.cuda.synchronize()
    out = []
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n_lanes):
            with torch.cuda.stream(streams[k]): engs[k].begin()
        for it in range(S):
            for k in range(n_lanes):
                with torch.cuda.stream(streams[k]):
                    p, v = nets[k](engs[k].planes, inference=True)
                    engs[k].step(p, v.reshape(-1))
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / S * 1e3)
    for e in engs: e.check_errors(); e.close()
    return out

for n in (1, 2, 1, 2, 4):
    print("lanes", n, "ms/step", ["%.3f" % x for x in run(n)], flush=True)
