"""Bare optimiser step (batch 128, matrix-core convolutions, eager) timed in the tree this script is started from: ms per step with the GPU and the host's issue time.
For a same-box A/B of two trees:  python tools/train_step_ab.py; (cd ab/prev && python tools/train_step_ab.py); ... alternating."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sigma_zero_amd as sz
from sigma_zero_amd import train_rl as T
from sigma_zero_amd.trainconv import split_convs
dev = torch.device("cuda:0")
torch.manual_seed(0)
g = torch.Generator(device=dev).manual_seed(1)
batch = {"states": (torch.rand(128, 119, 8, 8, device=dev, generator=g) < 0.15).float(), "actions": torch.softmax(torch.randn(128, 4672, device=dev, generator=g) * 3, 1),
         "rewards": torch.randint(-1, 2, (128,), device=dev, generator=g).float()}
model = sz.policyNN({}).to(dev).train()
opt, sched = T.make_optimiser(model)
def step():
    opt.zero_grad(); loss, mse, ce = T.loss_fn(model, batch, dev); loss.backward(); opt.step(); sched.step()
with split_convs(model):
    for _ in range(15): step()
    res = []
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(60): step()
        th = time.perf_counter() - t
        torch.cuda.synchronize(); res.append(((time.perf_counter() - t) / 60 * 1e3, th / 60 * 1e3))
print("%s: %s ms per step (host issue %s ms)" % (ROOT, " / ".join("%.2f" % a for a, _ in res), " / ".join("%.2f" % b for _, b in res)))

# train() itself (BatchNorm step counters set aside, history kept on the device) over 120 prepared batches
model2 = sz.policyNN({}).to(dev).train()
opt2, sched2 = T.make_optimiser(model2)
batches = [batch] * 120
T.train(model2, batches[:10], opt2, total_steps=0, lr_scheduler=sched2, device=dev)
torch.cuda.synchronize(); t = time.perf_counter()
T.train(model2, batches, opt2, total_steps=0, lr_scheduler=sched2, device=dev)
torch.cuda.synchronize(); print("%s: train() over 120 steps: %.2f ms per step" % (ROOT, (time.perf_counter() - t) / 120 * 1e3))
