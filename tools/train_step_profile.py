"""The RL optimiser step (train_RL.py:103-122: fp32 forward + backward + Adam, batch 128) on the device: eager against one HIP-graph replay per step.
Under `rocprofv3 --kernel-trace --stats -- python3 tools/train_step_profile.py eager` the kernel statistics say where the GPU time of a step goes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd import train_rl as T
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda:0")
torch.manual_seed(0)
Bn = 128
g = torch.Generator(device=dev).manual_seed(1)
batch = {"states": (torch.rand(Bn, 119, 8, 8, device=dev, generator=g) < 0.15).float(),
         "actions": torch.softmax(torch.randn(Bn, 4672, device=dev, generator=g) * 3, 1), "rewards": torch.randint(-1, 2, (Bn,), device=dev, generator=g).float()}


def timeit(step, n=30):
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


if mode in ("eager", "both"):
    model = sz.policyNN({}).to(dev).train()
    if os.environ.get("SZ_TRAINCONVS", "0") == "1":        # the default train step of train_rl: the tower's convolutions on the matrix cores (trainconv.py)
        from sigma_zero_amd.trainconv import enable_split_convs
        enable_split_convs(model)
    opt, sched = T.make_optimiser(model)
    def step():
        opt.zero_grad()
        loss, mse, ce = T.loss_fn(model, batch, dev)
        loss.backward(); opt.step(); sched.step()
    print("eager fp32 step, batch %d: %.2f ms" % (Bn, timeit(step)), flush=True)
if mode in ("graph", "both"):
    model = sz.policyNN({}).to(dev).train()
    if os.environ.get("SZ_TRAINCONVS", "0") == "1":
        from sigma_zero_amd.trainconv import enable_split_convs
        enable_split_convs(model)
    opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-4, device=dev), weight_decay=1e-4, fused=True, capturable=True)
    static = {k: v.clone() for k, v in batch.items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss, mse, ce = T.loss_fn(model, static, dev)
            loss.backward(); opt.step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss, mse, ce = T.loss_fn(model, static, dev)
        loss.backward(); opt.step()
    def gstep():
        graph.replay()
    print("HIP-graph fp32 step (forward + backward + fused Adam in one replay), batch %d: %.2f ms (loss %.4f)" % (Bn, timeit(gstep), float(loss)), flush=True)
