"""does capturing the simulation step (tower + heads + value MLP + tree step) in a HIP graph pay? eager vs graph replay, per iteration"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet
from sigma_zero_amd.selfplay import SelfPlayEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
UNROLL = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
eng = SelfPlayEngine(fast, {"C": 2, "num_searches": S}, B, planes_dtype="bits128")
eng.new_games(None)
def it():
    p, v = fast(eng.planes, inference=True)
    eng.step(p, v.reshape(-1))
def eager():
    eng.begin()
    for _ in range(S): it()
for _ in range(2): eager()
torch.cuda.synchronize()
t = time.perf_counter(); eager(); torch.cuda.synchronize(); te = (time.perf_counter() - t) / S * 1e3
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
eng.begin()
with torch.cuda.stream(side):
    it(); 
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(UNROLL): it()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
def graphed():
    eng.begin()
    for _ in range(S // UNROLL): g.replay()
for _ in range(2): graphed()
torch.cuda.synchronize()
t = time.perf_counter(); graphed(); torch.cuda.synchronize(); tg = (time.perf_counter() - t) / (S // UNROLL * UNROLL) * 1e3
eng.check_errors()
print("B=%d S=%d: eager %.4f ms/iteration, graph(%d per replay) %.4f ms/iteration, ratio %.3f" % (B, S, te, UNROLL, tg, tg / te))
