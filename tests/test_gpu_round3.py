"""Round-3 GPU tests: the RL train step on the device against the reference's golden loss and the CPU step,
BASELINE configs[1] (512 boards x 100 searches, bf16 MFMA network), and bench.py's own multi-rank launch."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

import sigma_zero_amd as sz
from sigma_zero_amd import train_rl
from sigma_zero_amd.selfplay import SelfPlayEngine

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_step_on_the_device_matches_reference_loss_and_the_cpu_step(golden_dir):
    """/root/reference/train_RL.py:103-122,187: loss = mse(v.squeeze(-1), z) + cross_entropy(logits, pi) in train mode, Adam(1e-4, wd 1e-4).
    On cuda: (i) mse / ce within 1e-4 of the values the reference's own network.py produced for this mini-batch (train_loss_golden.npz);
    (ii) the gradient within 5e-3 relative L2 of the CPU gradient (measured 2.2e-3 on MI355X: MIOpen's fp32 convolution algorithms against
    the CPU's direct ones, through 39 train-mode BatchNorms over a small batch); (iii) one fused-Adam step moves the parameters like one CPU Adam step."""
    z = np.load(os.path.join(golden_dir, "train_loss_golden.npz"))
    batch = {"states": torch.from_numpy(z["x"].astype(np.float32)), "actions": torch.from_numpy(z["p_target"]), "rewards": torch.from_numpy(z["v_target"])}
    res = {}
    for dev in ("cpu", "cuda"):
        torch.manual_seed(0)
        net = sz.policyNN({}).to(dev)
        net.train()
        before = torch.cat([p.detach().flatten().cpu() for p in net.parameters()]).double()
        opt, _ = train_rl.make_optimiser(net)
        opt.zero_grad()
        loss, mse, ce = train_rl.loss_fn(net, batch, dev)
        loss.backward()
        grad = torch.cat([p.grad.flatten().cpu() for p in net.parameters()]).double()
        opt.step()
        after = torch.cat([p.detach().flatten().cpu() for p in net.parameters()]).double()
        res[dev] = (float(mse.detach()), float(ce.detach()), grad, after - before)
    mse, ce, g_gpu, d_gpu = res["cuda"]
    assert abs(mse - float(z["mse"])) < 1e-4 and abs(ce - float(z["ce"])) < 1e-4, (mse, ce, float(z["mse"]), float(z["ce"]))
    _, _, g_cpu, d_cpu = res["cpu"]
    rel_g = float((g_gpu - g_cpu).norm() / g_cpu.norm())
    assert rel_g < 5e-3, rel_g
    # first Adam step: delta = -lr * g / (|g| + eps), i.e. +-1e-4 per element; elements whose gradient is below the fp32 noise of the two
    # backward passes may flip, everything else must agree
    assert abs(float(d_cpu.abs().max()) - 1e-4) < 2e-6 and abs(float(d_gpu.abs().max()) - 1e-4) < 2e-6
    flipped = float(((d_gpu - d_cpu).abs() > 5e-5).double().mean())
    rel_d = float((d_gpu - d_cpu).norm() / d_cpu.norm())
    print("device train step: mse %.6f ce %.6f (reference %.6f %.6f); gradient rel L2 vs CPU %.2e; Adam update rel L2 %.2e, %.4f %% of the elements differ by more than lr/2"
          % (mse, ce, float(z["mse"]), float(z["ce"]), rel_g, rel_d, 100 * flipped))
    assert flipped < 0.01 and rel_d < 0.1, (flipped, rel_d)


def test_split_precision_training_convolution_forward_and_backward():
    """trainconv.SplitConv3x3 (k_conv3x3_split_f32 / k_wgrad3x3_split: hi + lo f16 operands with power-of-two scaling, f32 accumulation) against an fp64 convolution:
    forward, input gradient (the same kernel with transposed + flipped weights) and weight gradient (its own kernel: positions as the MFMA's reduction dimension, partial
    sums over board groups); fewer boards than CUs (two workgroups per board), odd counts, more boards than CUs (the persistent board loop); tiny and huge magnitudes
    (per-board / per-tensor scaling); the weight gradient without a backward-data pass; the hi + lo bf16 form."""
    import torch.nn.functional as F
    from sigma_zero_amd.trainconv import SplitConv3x3
    g = torch.Generator(device="cuda").manual_seed(5)
    w = (torch.randn(256, 256, 3, 3, device="cuda", generator=g) * 0.03).requires_grad_()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for B in (1, 37, 300):
        x = (torch.randn(B, 256, 8, 8, device="cuda", generator=g) * (torch.rand(B, 256, 8, 8, device="cuda", generator=g) < 0.5)).requires_grad_()
        gy = torch.randn(B, 256, 8, 8, device="cuda", generator=g)
        y64 = F.conv2d(x.double(), w.double(), padding=1)
        y64.backward(gy.double())
        gx64, gw64 = x.grad.clone(), w.grad.clone()
        x.grad = None; w.grad = None
        y = SplitConv3x3.apply(x, w)
        y.backward(gy)
        assert rel(y, y64) < 2e-6 and rel(x.grad, gx64) < 2e-6 and rel(w.grad, gw64) < 2e-6, (B, rel(y, y64), rel(x.grad, gx64), rel(w.grad, gw64))    # measured 5.0e-7 / 5.1e-7 / 2.7e-7 (torch fp32: 4.9e-7 / 5.1e-7 / 2.5e-7)
        x.grad = None; w.grad = None
        with torch.no_grad():
            for mag in (1e3, 1e-6, 1e-20):                 # activations of 1e+3 and gradients of 1e-6 alike; far below f16's range too
                assert rel(SplitConv3x3.apply(x.detach() * mag, w), y64 * mag) < 2e-6, mag
            xz = x.detach().clone(); xz[B // 2] = 0       # an all-zero board beside others
            yz = SplitConv3x3.apply(xz, w)
            assert float(yz[B // 2].abs().max()) == 0.0 and rel(yz, F.conv2d(xz.double(), w.double(), padding=1)) < 2e-6 or B == 1
            import sigma_zero_amd.trainconv as TC
        # weight gradient alone (input without requires_grad: no backward-data kernel leaves max|gy| behind) and on tiny gradients
        xn = x.detach().clone()
        yn = SplitConv3x3.apply(xn, w)
        yn.backward(gy * 1e-7)
        assert rel(w.grad, gw64 * 1e-7) < 2e-6
        w.grad = None
        with torch.no_grad():
            TC.OPERANDS_F16 = False
            try:
                assert rel(SplitConv3x3.apply(x.detach(), w), y64) < 2e-5                     # hi + lo bf16: 16 bits (measured 4.5e-6)
            finally:
                TC.OPERANDS_F16 = True


def test_train_step_with_split_convolutions_matches_reference_loss_and_the_cpu_gradient(golden_dir):
    """the default train step on the device (train_rl.train enables trainconv's convolutions: hi + lo f16 operands, fp32's accuracy class): loss within 1e-4 of the
    reference's golden values; the gradient of this 8-sample batch as close to the CPU's as MIOpen's fp32 path is (both ~2.7e-3: 39 train-mode BatchNorms over 8
    samples amplify every rounding)"""
    from sigma_zero_amd.trainconv import split_convs
    from sigma_zero_amd import train_rl as T
    z = np.load(os.path.join(golden_dir, "train_loss_golden.npz"))
    batch = {"states": torch.from_numpy(z["x"].astype(np.float32)), "actions": torch.from_numpy(z["p_target"]), "rewards": torch.from_numpy(z["v_target"])}
    grads = {}
    for kind in ("cpu", "miopen", "split"):
        dev = "cpu" if kind == "cpu" else "cuda"
        torch.manual_seed(0)
        net = sz.policyNN({}).to(dev)
        net.train()
        if kind == "split":
            with split_convs(net):
                loss, mse, ce = train_rl.loss_fn(net, batch, dev)
                loss.backward()
            assert abs(float(mse.detach()) - float(z["mse"])) < 1e-4 and abs(float(ce.detach()) - float(z["ce"])) < 1e-4
        else:
            loss, mse, ce = train_rl.loss_fn(net, batch, dev)
            loss.backward()
        grads[kind] = torch.cat([p.grad.flatten().cpu() for p in net.parameters()]).double()
    r_mi = float((grads["miopen"] - grads["cpu"]).norm() / grads["cpu"].norm())
    r_sp = float((grads["split"] - grads["cpu"]).norm() / grads["cpu"].norm())
    print("gradient rel L2 vs the CPU's: MIOpen fp32 %.2e, split-precision convolutions %.2e" % (r_mi, r_sp))
    # train() plumbs the option through and restores the modules' own forward afterwards
    hist = {}
    for flag in (False, None):                            # None = the default: on
        torch.manual_seed(0)
        net = sz.policyNN({}).cuda()
        opt, sched = T.make_optimiser(net)
        dl = [{k: v.cuda() for k, v in batch.items()}] * 3
        hist[flag] = T.train(net, dl, opt, total_steps=0, lr_scheduler=sched, device="cuda", split_convs=flag)
        assert not any(hasattr(m, "_sz_orig_forward") for m in net.modules())
    assert len(hist[None]) == 3 and np.allclose(np.array(hist[None]), np.array(hist[False]), rtol=2e-3, atol=2e-3), (hist[None], hist[False])
    # two fp32 evaluations of this 8-sample step differ by a few 1e-3 whatever produces them (MIOpen vs the CPU: 2.6e-3; the matrix-core convolutions with torch's
    # BatchNorm: 2.7e-3; with the fused BatchNorm + ReLU launches, each as close to fp64 as torch's own: 5.9e-3): against an fp64 step at batch 128 the three are at
    # 3.4e-3 / 3.5e-3 / 3.8e-3 (tools/trainconv_probe.py)
    assert r_sp < 1.2e-2 and r_mi < 5e-3


def test_graphed_train_steps_equal_eager_steps():
    """train_rl.GraphedStep — gradient zeroing + forward + backward of a step as one HIP-graph replay — against eager forward + backward on a twin model with the same
    weights: the same losses and the same gradient for every batch (two eager steps on the capture stream, the capture, three replays with new batch contents, a batch
    of another shape that drops the graph and runs eagerly, a second capture, three replays), with per-parameter gradient tensors and with GradSync's flat buffer; BatchNorm's running statistics advance
    exactly once per step (nothing is computed twice around the capture).  Then train(graph=True) against train(graph=False) over 10 optimiser steps: same losses
    within the divergence two eager runs show among themselves (train-mode BatchNorm over 32 samples amplifies last-bit differences of the atomics in torch's kernels)."""
    from sigma_zero_amd import train_rl as T
    from sigma_zero_amd.trainconv import split_convs
    g = torch.Generator().manual_seed(5)
    def batch(n):
        return {"states": (torch.rand(n, 119, 8, 8, generator=g) < 0.15).float().cuda(), "actions": torch.softmax(torch.randn(n, 4672, generator=g) * 3, 1).cuda(),
                "rewards": torch.randint(-1, 2, (n,), generator=g).float().cuda()}
    batches = [batch(32) for _ in range(5)] + [batch(16)] + [batch(32) for _ in range(4)]
    for use_sync in (False, True):
        torch.manual_seed(0)
        ref = sz.policyNN({}).cuda().train()
        net = sz.policyNN({}).cuda().train()
        net.load_state_dict(ref.state_dict())
        sync = T.GradSync(net) if use_sync else None
        with split_convs(ref), split_convs(net):
            gs = T.GraphedStep(net, "cuda", sync)
            for i, b in enumerate(batches):
                ref.zero_grad()
                loss, mse, ce = T.loss_fn(ref, b, "cuda")
                loss.backward()
                m2, c2 = gs.step(b)
                assert (gs.graph is not None) == (i in (1, 2, 3, 4, 6, 7, 8, 9))          # captured at the end of the second step; dropped by the odd batch (5), captured again after 6
                assert abs(float(m2) - float(mse)) < 1e-6 and abs(float(c2) - float(ce)) < 1e-5, (use_sync, i, float(m2), float(mse), float(c2), float(ce))
                ga = torch.cat([p.grad.flatten() for p in ref.parameters()]).double()
                gb = torch.cat([p.grad.flatten() for p in net.parameters()]).double()
                r = float((ga - gb).norm() / ga.norm())
                assert r < 1e-5, (use_sync, i, r)
        for (n1, b1), (n2, b2) in zip(ref.named_buffers(), net.named_buffers()):
            assert torch.allclose(b1.double(), b2.double(), rtol=1e-5, atol=1e-7), n1
        assert int(net.norm_layer.num_batches_tracked) == len(batches)
    hist = {}
    for kind in ("eager", "eager2", "graph"):
        torch.manual_seed(0)
        net = sz.policyNN({}).cuda()
        opt, sched = T.make_optimiser(net)
        hist[kind] = np.array(T.train(net, batches, opt, total_steps=0, lr_scheduler=sched, device="cuda", graph=(kind == "graph")))
        # BatchNorm's step counters are set aside during the loop (41 launches per step) and advanced once at its end: the state_dict is what eager torch leaves
        assert all(int(m.num_batches_tracked) == len(batches) for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
        assert "norm_layer.num_batches_tracked" in net.state_dict()
    noise = np.abs(hist["eager2"] - hist["eager"]).max()
    diff = np.abs(hist["graph"] - hist["eager"]).max()
    print("10 optimiser steps: max |loss difference| graph vs eager %.2e, eager vs eager %.2e" % (diff, noise))
    assert len(hist["graph"]) == 10 and np.allclose(hist["graph"][0], hist["eager"][0], rtol=1e-6) and diff < max(5e-3, 10 * noise), (hist["graph"], hist["eager"])


def test_fused_batchnorm_skip_relu_matches_torch():
    """trainconv.BNAct (k_bn_act_fwd / k_bn_act_bwd: train-mode BatchNorm2d + skip connection + ReLU in one launch per direction, network.py:62-83) against the same
    layers in torch computed in fp64: output, running statistics and the gradients of input, gamma, beta and the skip connection — as close to fp64 as torch's own fp32
    kernels are; 3 boards, 128 and 300 (beyond the kernels' register cache of 256 boards); with and without the skip connection.  Then a whole ResidualBlock through
    split_convs(model) (fused sites + matrix-core convolutions) against the plain torch block."""
    from sigma_zero_amd.trainconv import BNAct, split_convs
    from sigma_zero_amd.network import ResidualBlock
    g = torch.Generator(device="cuda").manual_seed(3)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for B in (3, 128, 300):
        for with_res in (False, True):
            C = 256
            x = (torch.randn(B, C, 8, 8, device="cuda", generator=g) * 1.7 + 0.3).requires_grad_(True)
            res = torch.randn(B, C, 8, 8, device="cuda", generator=g).requires_grad_(True) if with_res else None
            gy = torch.randn(B, C, 8, 8, device="cuda", generator=g)
            out = {}
            for kind in ("fp64", "torch", "fused"):
                dt = torch.float64 if kind == "fp64" else torch.float32
                bn = torch.nn.BatchNorm2d(C).cuda().to(dt).train()
                with torch.no_grad():
                    bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C)); bn.running_mean.fill_(0.1); bn.running_var.fill_(0.8)
                xx = x.detach().to(dt).requires_grad_(True)
                rr = res.detach().to(dt).requires_grad_(True) if with_res else None
                if kind == "fused":
                    y = BNAct.apply(xx, bn.weight, bn.bias, rr, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
                else:
                    y = torch.relu(bn(xx) if rr is None else bn(xx) + rr)
                y.backward(gy.to(dt))
                out[kind] = [y.detach(), bn.running_mean.clone(), bn.running_var.clone(), xx.grad, bn.weight.grad, bn.bias.grad] + ([rr.grad] if with_res else [])
            for i, name in enumerate(["y", "running_mean", "running_var", "dx", "dgamma", "dbeta"] + (["dres"] if with_res else [])):
                ef, et = rel(out["fused"][i], out["fp64"][i]), rel(out["torch"][i], out["fp64"][i])
                assert ef < max(3 * et, 3e-7), (B, with_res, name, ef, et)
    torch.manual_seed(1)
    blk = ResidualBlock(256).cuda().train()
    ref = ResidualBlock(256).cuda().train()
    ref.load_state_dict(blk.state_dict())
    x = torch.randn(64, 256, 8, 8, device="cuda", generator=g)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    with split_convs(blk):
        ya = blk(xa)
        ya.square().mean().backward()
    yb = ref(xb)
    yb.square().mean().backward()
    # (gradients: a ReLU input within rounding of zero takes the other branch in one of the two evaluations — measured 1.6e-4)
    assert rel(ya, yb) < 1e-5 and rel(xa.grad, xb.grad) < 1e-3
    for (n1, p1), (_, p2) in zip(blk.named_parameters(), ref.named_parameters()):
        assert rel(p1.grad, p2.grad) < 2e-3, (n1, rel(p1.grad, p2.grad))
    for (n1, b1), (_, b2) in zip(blk.named_buffers(), ref.named_buffers()):
        assert torch.allclose(b1.double(), b2.double(), rtol=1e-5, atol=1e-6), n1
    assert not hasattr(blk, "_sz_orig_block_forward")


def test_device_batches_on_the_device_equal_dataloader_with_collate():
    """train_RL.py:14-49 (chessDataset + collatefn) vs DeviceBatches on cuda: the same batches bit for bit"""
    rng = np.random.RandomState(1)
    n = 300
    planes = rng.rand(n, 119, 8, 8) < 0.2
    packed = list((planes.astype(np.uint8) * (1 << np.arange(8)).astype(np.uint8)).sum(-1).astype(np.uint8))
    aidx = [np.sort(rng.choice(4672, size=rng.randint(1, 40), replace=False)) for _ in range(n)]
    aprob = [(lambda v: v / v.sum())(rng.rand(len(a))) for a in aidx]
    rew = [float(rng.choice([-1, 0, 1])) for _ in range(n)]
    ds = train_rl.SelfPlayDataset(packed, aidx, aprob, rew)
    dl = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=False, drop_last=True, collate_fn=train_rl.SelfPlayDataset.collate)
    db = train_rl.DeviceBatches(packed, aidx, aprob, rew, batch_size=128, device="cuda", shuffle=False)
    assert len(db) == len(dl) == 2
    for a, b in zip(dl, db):
        for k in ("states", "actions", "rewards"):
            assert b[k].is_cuda and a[k].dtype == b[k].dtype and torch.equal(a[k], b[k].cpu()), k
    g = torch.Generator(device="cuda").manual_seed(3)
    seen = torch.cat([b["rewards"] for b in train_rl.DeviceBatches(packed, aidx, aprob, rew, batch_size=128, device="cuda", shuffle=True, generator=g)])
    assert seen.numel() == 256


def test_config1_512_boards_100_searches_bf16_network():
    """BASELINE.json configs[1]: 512 concurrent vanilla-chess boards, num_searches = 100, bf16 policy/value net (FastPolicyNet, bit-packed planes).
    Size-independent invariants over two plies, bitwise reproducibility, and (below) a lock-step oracle comparison with the same network as evaluator."""
    from sigma_zero_amd.fastnet import FastPolicyNet
    B, S = 512, 100
    torch.manual_seed(0)
    fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
    runs = []
    for rep in range(2):
        eng = SelfPlayEngine(fast, {"C": 2, "num_searches": S}, B, chess960=False, learning=True, planes_dtype="bits128")
        eng.new_games([-1] * B)
        prng = np.random.RandomState(5)
        out = []
        for ply in range(2):
            st0 = eng.stats()
            eng.search()
            st = eng.check_errors()
            assert st["simulations"] - st0["simulations"] == B * S
            assert st["expansions"] + st["terminal_hits"] - st0["expansions"] - st0["terminal_hits"] == B * S
            action, visits, n_child, prior, wsum = eng.root_children()
            k = np.arange(visits.shape[1])[None, :] < n_child[:, None]
            assert (n_child >= 1).all() and (n_child <= 218).all()
            if ply == 0:
                assert (n_child == 20).all()                                                            # the start position has 20 legal moves
            assert ((visits * k).sum(1) == S - 1).all()                                                 # mcts.py:46,118
            assert (np.diff(np.where(k, action, 1 << 30), axis=1)[:, :-1][k[:, 1:-1]] > 0).all()        # ascending action order
            assert (np.abs(np.where(k, wsum, 0)) <= np.where(k, visits, 0) + 1e-9).all()                 # |W| <= N
            psum = np.where(k, prior, 0).sum(1)
            assert np.allclose(psum, 0.75 + 0.25 * n_child * float(np.float32(1) - np.float32(2.0 ** -24)), atol=1e-3)
            eng.play(prng.random_sample(B))
            rec = eng.fetch_ply()
            assert rec["active"].all() and (rec["n_child"] == n_child).all()
            for b in range(0, B, 37):
                kk = int(n_child[b])
                assert rec["chosen"][b] in action[b, :kk] and visits[b, :kk][list(action[b, :kk]).index(rec["chosen"][b])] > 0
            out.append((action.copy(), visits.copy(), prior.copy(), wsum.copy(), rec["chosen"].copy()))
        eng.close()
        runs.append(out)
    for p0, p1 in zip(*runs):
        for a, b in zip(p0, p1):
            assert np.array_equal(a, b)


def test_config1_lockstep_with_the_bf16_network_as_evaluator():
    """8 classical boards x 100 searches in lock-step with the oracle, both fed by the shipped bf16 MFMA network (FastPolicyNet): planes, legal masks,
    leaf depths at every step; root children, visits, priors and value sums bit for bit."""
    from sigma_zero_amd.fastnet import FastPolicyNet, planes_nchw_to_nhwc128
    from test_gpu_parity import Mirror, lockstep_search
    torch.manual_seed(0)
    fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
    rng = random.Random(21)
    boards = [Mirror(c960=False, scharnagl=518, pre_moves=rng.randrange(0, 30), rng=rng) for _ in range(8)]

    def ev(planes, step):
        with torch.no_grad():
            p, v = fast(planes_nchw_to_nhwc128(planes.float()), inference=True)
        return p.clone(), v.reshape(-1).clone()

    st = lockstep_search(boards, 100, True, ev)
    assert st["simulations"] == 8 * 100


def _bench_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` outside a launcher starts the two ranks itself (gloo here: both share the one GPU of the box) and rank 0 reports the
    whole job; the reference spawns its own workers the same way (train_RL.py:215-227)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--boards", "256", "--searches", "20",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _bench_line(r.stdout)
    assert line["n_gpus"] == 2 and line["sim_count_ok"] is True
    assert line["config"]["boards_per_gpu"] == 256 and line["value"] > 0
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 2 * 256 * 20) < 1.0               # whole-job aggregate: both ranks' simulations over the slowest rank's time
