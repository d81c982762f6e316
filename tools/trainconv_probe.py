"""Training-step convolutions on the split-precision kernel (sigma_zero_amd/trainconv.py) against torch/MIOpen fp32 and an fp64 reference: forward, input gradient,
whole train step time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import sigma_zero_amd as sz
from sigma_zero_amd import train_rl as T
import sigma_zero_amd.trainconv as TC
from sigma_zero_amd.trainconv import SplitConv3x3, split_convs, enable_split_convs
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(B, 256, 8, 8, device=dev) * (torch.rand(B, 256, 8, 8, device=dev) < 0.5)
w = (torch.randn(256, 256, 3, 3, device=dev) * 0.03).requires_grad_()
gy = torch.randn(B, 256, 8, 8, device=dev)
xr = x.clone().requires_grad_()
y64 = F.conv2d(xr.double(), w.double(), padding=1); y64.backward(gy.double())
gx64 = xr.grad.clone(); gw64 = w.grad.clone(); w.grad = None
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
xr.grad = None
y32 = F.conv2d(xr, w, padding=1); y32.backward(gy); gx32 = xr.grad.clone(); gw32 = w.grad.clone()
for f16 in (True, False):
    TC.OPERANDS_F16 = f16
    xr.grad = None; w.grad = None
    ys = SplitConv3x3.apply(xr, w); ys.backward(gy); gxs = xr.grad.clone(); gws = w.grad.clone()
    print("operands %s:" % ("hi+lo f16, scaled" if f16 else "hi+lo bf16"))
    print("  forward   rel L2 vs fp64: torch fp32 %.2e   split kernel %.2e" % (rel(y32, y64), rel(ys, y64)))
    print("  grad(x)   rel L2 vs fp64: torch fp32 %.2e   split kernel %.2e" % (rel(gx32, gx64), rel(gxs, gx64)))
    print("  grad(w)   rel L2 vs fp64: torch fp32 %.2e   %s %.2e" % (rel(gw32, gw64), "wgrad kernel" if f16 else "torch wgrad", rel(gws, gw64)))
    # the same with tiny gradients and large activations: the per-board scaling must not care
    with torch.no_grad():
        print("  forward on x*1e3 %.2e   on x*1e-6 %.2e   (rel L2 vs fp64)" % (rel(SplitConv3x3.apply(x * 1e3, w), y64 * 1e3), rel(SplitConv3x3.apply(x * 1e-6, w), y64 * 1e-6)))
TC.OPERANDS_F16 = True
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
def bwd_time(fn_conv):
    xx = x.clone().requires_grad_(); ww = w.detach().clone().requires_grad_()
    yy = fn_conv(xx, ww)
    def f():
        xx.grad = None; ww.grad = None
        yy.backward(gy, retain_graph=True)
    return timeit(f)
print("B=%d conv backward (grad x + grad w): torch %.1f us   split kernels %.1f us" % (B, bwd_time(lambda a, b: F.conv2d(a, b, padding=1)), bwd_time(SplitConv3x3.apply)))
with torch.no_grad():
    print("B=%d conv forward: torch %.1f us   split kernel (pack + conv) %.1f us" % (B, timeit(lambda: F.conv2d(x, w, padding=1)), timeit(lambda: SplitConv3x3.apply(x, w))))
g = torch.Generator(device=dev).manual_seed(1)
batch = {"states": (torch.rand(128, 119, 8, 8, device=dev, generator=g) < 0.15).float(),
         "actions": torch.softmax(torch.randn(128, 4672, device=dev, generator=g) * 3, 1), "rewards": torch.randint(-1, 2, (128,), device=dev, generator=g).float()}
for split in (False, True, "bf16", True):
    TC.OPERANDS_F16 = split is True
    torch.manual_seed(0)
    model = sz.policyNN({}).to(dev).train()
    if split: enable_split_convs(model)
    opt, sched = T.make_optimiser(model)
    losses = []
    def step():
        opt.zero_grad()
        loss, mse, ce = T.loss_fn(model, batch, dev)
        loss.backward(); opt.step(); sched.step()
        losses.append(loss.detach())
    ms = timeit(step, 30) / 1e3
    print("train step batch 128, split convs %s: %.2f ms   loss after 35 steps %.6f" % ({False: "off", True: "f16x2", "bf16": "bf16x2"}[split], ms, float(losses[-1])))
# whole-network gradient at batch 128 against an fp64 run of the same step: what each fp32 path loses through 39 train-mode BatchNorms
def grads(split, dtype):
    torch.manual_seed(0)
    model = sz.policyNN({}).to(dev).to(dtype).train()
    if split: enable_split_convs(model)
    bb = {k: v.to(dtype) for k, v in batch.items()}
    loss, mse, ce = T.loss_fn(model, bb, dev)
    loss.backward()
    return torch.cat([p.grad.flatten() for p in model.parameters()]).double(), float(loss.detach())
g64, l64 = grads(False, torch.float64)
for name, split, f16 in (("torch/MIOpen fp32", False, True), ("split convolutions, hi+lo f16 scaled", True, True), ("split convolutions, hi+lo bf16", True, False)):
    TC.OPERANDS_F16 = f16
    g, l = grads(split, torch.float32)
    print("batch 128 gradient rel L2 vs the fp64 step: %-38s %.2e   (loss %.7f vs %.7f)" % (name, float((g - g64).norm() / g64.norm()), l, l64))
