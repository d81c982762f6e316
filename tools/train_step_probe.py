"""where does a train_RL optimiser step spend its time? data path vs forward/backward variants (batch 128)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd import train_rl as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
Bn = 128
rng = np.random.RandomState(0)
N = 4096
packed = [rng.randint(0, 256, size=(119, 8)).astype(np.uint8) for _ in range(N)]
aidx = [np.sort(rng.choice(4672, size=30, replace=False)) for _ in range(N)]
aprob = [np.full(30, 1 / 30.0) for _ in range(N)]
rew = [float(rng.choice([-1, 0, 1])) for _ in range(N)]
ds = T.SelfPlayDataset(packed, aidx, aprob, rew)
dl = torch.utils.data.DataLoader(ds, batch_size=Bn, shuffle=True, drop_last=True, collate_fn=T.SelfPlayDataset.collate)
t = time.perf_counter(); n = 0
for b in dl:
    n += 1
print("DataLoader + collate: %.2f ms per batch" % ((time.perf_counter() - t) / n * 1e3), flush=True)
batch = {k: v.to(dev) for k, v in b.items()}
def run(model, opt, amp, cl, n=10):
    x = batch["states"].contiguous(memory_format=torch.channels_last) if cl else batch["states"]
    bb = dict(batch, states=x)
    def step():
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss, mse, ce = T.loss_fn(model, bb, dev)
        loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for amp, cl in ((False, False), (False, True), (True, False), (True, True)):
    model = sz.policyNN({}).to(dev).train()
    if cl: model = model.to(memory_format=torch.channels_last)
    opt, _ = T.make_optimiser(model)
    print("fwd+bwd+Adam batch %d: amp(bf16)=%s channels_last=%s: %.2f ms" % (Bn, amp, cl, run(model, opt, amp, cl)), flush=True)
# fp32 variants that keep the reference's arithmetic class: MIOpen find mode (benchmark), fused Adam
for bench, fused in ((True, False), (False, True), (True, True)):
    torch.backends.cudnn.benchmark = bench
    model = sz.policyNN({}).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=fused)
    print("fp32 fwd+bwd+Adam batch %d: cudnn.benchmark=%s fused_adam=%s: %.2f ms" % (Bn, bench, fused, run(model, opt, False, False)), flush=True)
torch.backends.cudnn.benchmark = False
