"""train_rl.train() as the cycle runs it (DeviceBatches on the GPU, batch 128): ms per optimiser step of the whole loop, of the data path alone, and the
host's issue time per step (the Python + launch cost with nothing waited for: when it equals the step time the loop is host-bound, not GPU-bound)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd import train_rl as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
rng = np.random.RandomState(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 150
packed = [rng.randint(0, 256, size=(119, 8)).astype(np.uint8) for _ in range(n)]
aidx = [np.sort(rng.choice(4672, size=30, replace=False)) for _ in range(n)]
aprob = [np.full(30, 1 / 30.0) for _ in range(n)]
rew = [float(rng.choice([-1, 0, 1])) for _ in range(n)]
dl = T.DeviceBatches(packed, aidx, aprob, rew, batch_size=128, device=dev, generator=torch.Generator().manual_seed(1))
for b in dl: pass
torch.cuda.synchronize(); t = time.perf_counter()
for b in dl: pass
t_issue = time.perf_counter() - t
torch.cuda.synchronize(); print("data path alone: %.3f ms per batch (host issue %.3f ms)" % ((time.perf_counter() - t) / len(dl) * 1e3, t_issue / len(dl) * 1e3), flush=True)
for convs in (True, False, True):
    model = sz.policyNN({}).to(dev).train()
    opt, sched = T.make_optimiser(model)
    T.train(model, [b for b in list(dl)[:5]], opt, total_steps=0, lr_scheduler=sched, device=dev, split_convs=convs, graph=False)       # warm-up
    torch.cuda.synchronize(); t = time.perf_counter()
    T.train(model, dl, opt, total_steps=0, lr_scheduler=sched, device=dev, split_convs=convs, graph=True)
    torch.cuda.synchronize(); dtg = time.perf_counter() - t
    print("train(graph=True) split_convs=%s: %.2f ms per step over %d steps (2 eager steps + capture included)" % (convs, dtg / len(dl) * 1e3, len(dl)), flush=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    T.train(model, dl, opt, total_steps=0, lr_scheduler=sched, device=dev, split_convs=convs, graph=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    # the same steps issued without the final wait: host time per step
    batches = list(dl)[:40]
    from sigma_zero_amd.trainconv import split_convs as ctx
    import contextlib
    with (ctx(model) if convs else contextlib.nullcontext()):
        torch.cuda.synchronize(); t = time.perf_counter()
        for b in batches:
            opt.zero_grad(); loss, mse, ce = T.loss_fn(model, b, dev); loss.backward(); opt.step(); sched.step()
        t_host = (time.perf_counter() - t) / len(batches)
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / len(batches)
    print("train(graph=False) split_convs=%s: %.2f ms per step over %d steps;  bare step on prepared batches: host issue %.2f ms, with the GPU %.2f ms" % (convs, dt / len(dl) * 1e3, len(dl), t_host * 1e3, t_all * 1e3), flush=True)
