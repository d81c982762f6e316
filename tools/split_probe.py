"""SplitPolicyNet (hi/lo bf16 operands, 3 MFMAs per product) vs the fp32 torch network and vs the bf16 tower: accuracy of the tower
activation and of the logits, forward time at B boards."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd.fastnet import FastPolicyNet, SplitPolicyNet
from sigma_zero_amd.selfplay import SelfPlayEngine, unpack_bits128
from sigma_zero_amd.network import FLOPS_PER_BOARD
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = sz.policyNN({}).cuda().eval()
# give BatchNorm non-trivial statistics and the weights some spread, like a trained network would have
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.05)
fast, split, fast16, split16 = FastPolicyNet(net), SplitPolicyNet(net), FastPolicyNet(net, operands="fp16"), SplitPolicyNet(net, operands="fp16")
eng = SelfPlayEngine(fast, {"C": 2, "num_searches": 4}, B, chess960=True, planes_dtype="bits128")
import random
eng.new_games([random.Random(1).randrange(960) for _ in range(B)])
for _ in range(3):
    eng.search(); eng.play(np.random.RandomState(0).random_sample(B)); eng.fetch_ply()
eng.begin()
planes = eng.planes.clone()
x = unpack_bits128(planes)[:, :, :119].transpose(1, 2).reshape(B, 119, 8, 8).float()
with torch.no_grad():
    nb = min(B, 512)
    t_ref = net.resnet_blocks(F.relu(net.norm_layer(net.conv1(x[:nb].double().cuda()))).double()) if False else None
    net64 = sz.policyNN({}).cuda().eval().double(); net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    y64 = net64.resnet_blocks(F.relu(net64.norm_layer(net64.conv1(x[:nb].double()))))
    y32 = net.resnet_blocks(F.relu(net.norm_layer(net.conv1(x[:nb]))))
    ys = split.tower(planes[:nb]).view(nb, 8, 8, 256).permute(0, 3, 1, 2)
    yb = fast.tower(planes[:nb])[0].float().view(nb, 8, 8, 256).permute(0, 3, 1, 2)
    yh = fast16.tower(planes[:nb])[0].view(torch.float16).float().view(nb, 8, 8, 256).permute(0, 3, 1, 2)
    ys16 = split16.tower(planes[:nb]).clone().view(nb, 8, 8, 256).permute(0, 3, 1, 2)
    def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
    print("tower activation, relative L2 error vs fp64: fp32 torch %.2e   split %.2e   bf16 %.2e   fp16 %.2e" % (rel(y32, y64), rel(ys, y64), rel(yb, y64), rel(yh, y64)))
    p64, v64 = net64(x[:nb].double(), inference=False)
    p32, v32 = net(x[:nb], inference=False)
    ps, vs = (t.clone() for t in split(planes[:nb], inference=False))
    pb, vb = (t.clone() for t in fast(planes[:nb], inference=False))
    ph, vh = (t.clone() for t in fast16(planes[:nb], inference=False))
    ps16, vs16 = (t.clone() for t in split16(planes[:nb], inference=False))
    print("split with hi+lo f16 operands: tower activation %.2e, centred logits %.2e relative L2 vs fp64; value max abs %.2e" % (rel(ys16, y64), rel(ps16 - ps16.mean(1, keepdim=True), p64 - p64.mean(1, keepdim=True)), float((vs16.double().view(-1) - v64.view(-1)).abs().max())))
    c = lambda p: p - p.mean(1, keepdim=True)
    print("centred logits, relative L2 error vs fp64:   fp32 torch %.2e   split %.2e   bf16 %.2e   fp16 %.2e" % (rel(c(p32), c(p64)), rel(c(ps), c(p64)), rel(c(pb), c(p64)), rel(c(ph), c(p64))))
    print("value, max abs error vs fp64:                fp32 torch %.2e   split %.2e   bf16 %.2e" % (float((v32.double() - v64).abs().max()), float((vs.double() - v64).abs().max()), float((vb.double().view(-1) - v64.view(-1)).abs().max())))
    def timeit(fn, n=5):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    ts, tst, tb, t32 = timeit(lambda: split(planes)), timeit(lambda: split.tower(planes)), timeit(lambda: fast(planes)), timeit(lambda: net(x, inference=True), n=2)
    th, tb2, th2 = timeit(lambda: fast16(planes)), timeit(lambda: fast(planes)), timeit(lambda: fast16(planes))
    print("split forward: bf16x2 %.2f ms   f16x2 %.2f ms   bf16x2 %.2f ms   f16x2 %.2f ms" % (timeit(lambda: split(planes)), timeit(lambda: split16(planes)), timeit(lambda: split(planes)), timeit(lambda: split16(planes))))
    print("B=%d forward: split %.2f ms (tower %.2f ms, %.0f TFLOP/s algorithmic)   bf16 %.2f / %.2f ms   fp16 %.2f / %.2f ms   fp32 torch %.1f ms" % (B, ts, tst, B * FLOPS_PER_BOARD / tst / 1e9, tb, tb2, th, th2, t32))
    split.module_heads = True
    pm, vm = (t.clone() for t in split(planes[:nb], inference=False))
    tm = timeit(lambda: split(planes))
    split.module_heads = False
    print("heads as fp32 GEMMs vs the module's own heads: max |dlogit| %.2e, max |dv| %.2e; forward with module heads %.2f ms" % (float((pm - ps).abs().max()), float((vm - vs).abs().max()), tm))
