#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations/sec of batched self-play (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W          (driver contract)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one complete self-play ply of the hot path over one batch of boards: for every board a full
MCTS0.search (num_searches simulations: select -> move -> terminal test -> encode -> policy/value net ->
expand -> backprop), then the visit-count readout, move sampling and move application (sim.py:46-76).
Workload at N=1 (BASELINE.json configs[2], the configuration the metric is quoted on): 4096 concurrent
classical-chess boards per GPU, num_searches=800, bf16 policy/value network with seeded random-init weights.
Boards are independent, so with N ranks each rank owns 4096 boards (weak scaling, no data-path collective).

One JSON line is printed by rank 0.  Extra objects:
  roofline      the dominant kernel of the job (97.7 % of GPU time): k_tower16_bf16, the whole policy/value tower in one persistent
                MFMA launch — algorithmic FLOP per launch / mean launch duration measured with HIP events on the launch stream over
                the timed region, against the 2.5 PFLOP/s dense bf16 peak; `traffic` = HBM bytes per launch from the committed
                rocprofv3 --pmc passes (profiles/pmc_tower_latest.json)
  roofline_tree the hand-written tree kernel (k_search_step), HBM roofline: algorithmic bytes per launch / mean launch duration
  roofline_nn   the whole network forward (tower + heads): 2.915 GFLOP x boards / mean forward duration
  roofline_split the same workload with the network at the REFERENCE's precision class on the matrix cores (SplitPolicyNet / k_tower_split: hi + lo
                bf16 operands, 3 MFMAs per product — the configuration that meets north_star's 1e-4 on visit policies): 1 warm-up + 3 timed plies,
                simulations/s, launch duration of k_tower_split by HIP events, algorithmic AND issued MFMA fraction of the 2.5 PFLOP/s peak (N=1 only)
  fp16_config   the same workload on f16 operands (the product API's default self-play network; one timed ply, N=1 only)
  parity_config simulations/s of the same boards-per-GPU with the fp32 network the reference uses (bounded sample, N=1 only)
  train_step    the optimiser step of the same loop (train_RL.py:103-122, batch 128, synthetic batches): ms per step with the matrix-core convolutions + fused
                BatchNorm (train_rl.train's default) and with torch / MIOpen fp32 beside it (a few seconds, N=1 only)
`--gpus N` with N > 1 outside a launcher (no RANK in the environment) starts N ranks itself (torch.distributed.run on 127.0.0.1) before any GPU call,
as the reference spawns its own self-play workers (train_RL.py:215-227); under a launcher WORLD_SIZE must equal --gpus.
  cpu_baseline  the oracle (reference algorithm restated on the CPU: one leaf per step, per-game pointer tree,
                batch-1 fp32 forward on the host cores) on BASELINE.json configs[0], timed on rank 0 over a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def tree_bytes_per_launch(boards, sims, expansions, sum_depth, sum_children, plane_bytes):
    """Algorithmic HBM bytes of one k_search_step launch, from the measured mean depth / fan-out.
    per simulation:   select   d*(16*K + 16)      child stat spans (W f64, N i32, P f32) + one 16-B meta per level
                      backprop (d+1)*24           N,W read + write per level
    per expansion:    position 80 + 80            parent record read, child record write
                      history  7*64 + 150*16/8    ancestors for planes; key/meta window for repetition (mean clock ~ small)
                      planes   119*64*plane_bytes network input written
                      mask     584*2              legal-move mask written, re-read after the network
                      policy   4*K                legal probabilities gathered (+4 value)
                      children 32*K               EdgeStat + EdgeMeta written
                      path     8*(d+1)            descent path spilled and re-read
    """
    if sims == 0:
        return 0.0
    d = sum_depth / sims
    k = sum_children / max(expansions, 1)
    per_sim = d * (16.0 * k + 16.0) + (d + 1.0) * 24.0
    per_exp = 160.0 + 7 * 64.0 + 64.0 + 119 * 64 * plane_bytes + 584 * 2 + 4.0 * k + 4.0 + 32.0 * k + 8.0 * (d + 1.0)
    return per_sim * sims + per_exp * expansions


def cpu_baseline(budget_s=15.0):
    """BASELINE.json configs[0] — the reference's own CPU-runnable case (sim.py:125-137): 1 self-play game, num_searches=10, Chess960
    start, random-init network — played by the oracle (the reference algorithm restated on the CPU: per-game pointer tree, one leaf per
    step, state copy + replay per leaf) with the batch-1 fp32 policyNN forward on the host cores; bounded to ~budget_s seconds of it."""
    import random
    import sigma_zero_amd as sz
    from oracle import oracle as O
    # the GPU box exposes every host core but grants one GPU job a 16-core share; oversubscribing 256 threads
    # on a batch-1 forward is ~1000x slower, so use the share (and say so in `cores`)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = sz.policyNN({}).eval()
    random.seed(0)
    sch = random.randint(0, 959)                       # chess_tensor.py:69
    game = O.ChessTensor(chess960=True, scharnagl=sch)
    S = 10
    sims = 0
    rng = np.random.RandomState(0)
    t0 = time.perf_counter()
    plies = 0
    with torch.no_grad():
        while time.perf_counter() - t0 < budget_s:
            s = O.Search.on_chess(game, c=2.0, num_searches=S, learning=True)
            while s.advance():
                x = torch.from_numpy(s.leaf_planes().astype(np.float32)).unsqueeze(0)
                p, v = net(x, inference=True)
                s.feed(p[0].numpy(), float(v[0, 0]))
            e, t = s.counters()
            sims += e + t
            idx, vis, moves = s.root_children()
            if sum(vis) == 0 or game.get_value_and_terminated()[1]:
                break
            game.move_piece(moves[O.sample_move(vis, rng.random_sample())])
            plies += 1
            if game.get_value_and_terminated()[1]:
                break                                      # the game of configs[0] ended inside the budget
    dt = time.perf_counter() - t0
    return {"value": sims / dt, "unit": "simulations/s", "cores": cores, "kind": "port",
            "config": "BASELINE.json configs[0]: 1 self-play game, num_searches=10, Chess960 start %d (random.seed(0)), random-init fp32 network, CPU" % sch,
            "sample": "the first %d plies / %d simulations of that game in %.1f s (batch-1 fp32 forward on %d host threads)" % (plies, sims, dt, cores)}


def parity_config_sample(dev, B, chess960, n_searches=40):
    """Throughput at the REFERENCE's own arithmetic (network.py is fp32 end to end; no mixed precision anywhere), same boards-per-GPU: fp32 policyNN
    through torch/MIOpen on fp32 NCHW planes, one whole ply at a small search budget (the cost of a simulation does not depend on the budget: trees
    stay shallow).  A bounded sample next to the bf16 headline, not the headline; the matrix-core path of the same precision class is `roofline_split`."""
    import sigma_zero_amd as sz
    from sigma_zero_amd.selfplay import SelfPlayEngine
    torch.manual_seed(0)
    net = sz.policyNN({}).eval().to(dev)
    eng = SelfPlayEngine(net, {"C": 2, "num_searches": n_searches}, B, chess960=bool(chess960), learning=True, device=dev, planes_dtype=torch.float32)
    eng.new_games([-1] * B)
    with torch.no_grad():
        eng.begin()
        eng.evaluate(eng.planes)                           # warm-up forward (MIOpen algorithm search) outside the timed region
        torch.cuda.synchronize(dev)
        st0 = eng.stats()
        t0 = time.perf_counter()
        eng.search()
        eng.play(np.random.RandomState(7).random_sample(B))
        eng.fetch_ply()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    st1 = eng.check_errors()
    eng.close()
    sims = st1["simulations"] - st0["simulations"]
    return {"value": sims / dt, "unit": "simulations/s", "dtype": "f32", "network": "policyNN fp32 (torch / MIOpen), NCHW fp32 planes",
            "sample": "%d boards x num_searches=%d, one ply, %d simulations in %.2f s" % (B, n_searches, sims, dt),
            "search_level_gap_of_bf16": "profiles/r02a_bf16_vs_fp32_search_S{100,800}.json (tests/test_gpu_train_and_precision.py)"}


def split_tower_flops(B):
    """(algorithmic, issued) FLOP of one k_tower_split launch over B boards.  Algorithmic: the standard convolution FLOP of the tower (9 taps x 64
    positions, padding included: 2.915 GFLOP per board minus the heads).  Issued: what its MFMAs execute — three products per 256-channel convolution,
    two for the stem (0/1 planes have no lo part; 128 padded input channels), 66 of 72 (tap, board-row) tiles (the all-zero border rows are skipped)."""
    alg = 2.0 * B * 64 * 256 * 9 * (119 + 38 * 256)
    mfma_per_wave_tile = 66 * 4 * 2 * 4 + 38 * (66 * 8 * 3 * 4)          # (tap-tiles) x k32-steps x products x channel tiles
    issued = ((B + 1) // 2) * 4 * mfma_per_wave_tile * (2.0 * 16 * 16 * 32)
    return alg, issued


def split_sample(dev, B, S, chess960, plies=3, warmup=1):
    """The headline workload with the network at the reference's precision class on the matrix cores: `warmup` + `plies` full plies of B boards x S
    searches (finished games restart), timed like the headline; the tower launch is sampled with HIP events on the launch stream."""
    import random
    import sigma_zero_amd as sz
    from sigma_zero_amd.selfplay import SelfPlayEngine
    from sigma_zero_amd.fastnet import SplitPolicyNet
    torch.manual_seed(0)
    model = SplitPolicyNet(sz.policyNN({}).eval().to(dev), device=dev)
    eng = SelfPlayEngine(model, {"C": 2, "num_searches": S}, B, chess960=bool(chess960), learning=True, device=dev, planes_dtype="bits128")
    prng, rng = random.Random(0), np.random.RandomState(99)
    eng.new_games([prng.randrange(960) if chess960 else -1 for _ in range(B)])
    events = []

    @torch.no_grad()
    def ply(timed):
        eng.begin()
        for it in range(S):
            model.timing = events if (timed and it % 50 == 25) else None
            policy, value = model(eng.planes, inference=True)
            eng.step(policy, value.reshape(-1))
        model.timing = None
        eng.play(rng.random_sample(B))
        rec = eng.fetch_ply()
        over = rec["game_over"].astype(bool) & rec["active"].astype(bool)
        if over.any():
            eng.new_games([prng.randrange(960) if chess960 else -1 for _ in range(B)], active=over.astype(np.uint8))

    for _ in range(warmup):
        ply(False)
    st0 = eng.check_errors()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(plies):
        ply(True)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    st1 = eng.check_errors()
    eng.close()
    sims = st1["simulations"] - st0["simulations"]
    ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    alg, issued = split_tower_flops(B)
    atf, itf = alg / (ms * 1e-3) / 1e12, issued / (ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    try:
        import sigma_zero_amd.build as _b
        with open(os.path.join(ROOT, "profiles", "pmc_split_latest.json")) as f:
            traffic_src = json.load(f)
        if B == 4096 and traffic_src.get("source_hash") == _b.source_hash("split"):
            traffic = traffic_src["hbm_bytes_per_launch"]
        else:
            traffic_src = {"source": traffic_src.get("source"), "stale": "other batch size, or sz_nn_split.hip changed since these counters were collected"}
    except Exception:
        pass
    return {"kernel": "k_tower_split<%d> (persistent: stem + 19 BasicBlocks per launch on hi + lo bf16 operands, 3 MFMAs per product, f32 accumulate; f32 heads)" % (2 if B > 256 else 1),
            "bound": "mfma", "achieved": atf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": atf / MFMA_BF16_PEAK_TFLOPS,
            "issued": itf, "issued_frac": itf / MFMA_BF16_PEAK_TFLOPS, "launch_ms": ms, "sampled_launches": len(events),
            "algorithmic_flop_per_launch": alg, "issued_mfma_flop_per_launch": issued, "traffic": traffic, "traffic_source": traffic_src,
            "value": sims / dt, "value_unit": "simulations/s", "dtype": "bf16x2", "plies": plies, "warmup": warmup, "ms_per_step": 1e3 * dt / plies,
            "sim_count_ok": bool(sims == B * S * plies),
            "config": {"workload": "selfplay_%dboards_%dsearches" % (B, S), "network": "SplitPolicyNet (reference precision class: identical visit counts to the fp32 network in tests)"}}


def fp16_sample(dev, B, S, chess960):
    """The headline workload on f16 operands (FastPolicyNet(operands="fp16"): the product API's default self-play network): 1 warm-up + 1 timed ply."""
    import random
    import sigma_zero_amd as sz
    from sigma_zero_amd.selfplay import SelfPlayEngine
    from sigma_zero_amd.fastnet import FastPolicyNet
    torch.manual_seed(0)
    model = FastPolicyNet(sz.policyNN({}).eval(), device=dev, operands="fp16")
    eng = SelfPlayEngine(model, {"C": 2, "num_searches": S}, B, chess960=bool(chess960), learning=True, device=dev, planes_dtype="bits128")
    prng, rng = random.Random(0), np.random.RandomState(98)
    eng.new_games([prng.randrange(960) if chess960 else -1 for _ in range(B)])
    events = []

    @torch.no_grad()
    def ply(timed):
        eng.begin()
        for it in range(S):
            model.timing = events if (timed and it % 50 == 25) else None
            policy, value = model(eng.planes, inference=True)
            eng.step(policy, value.reshape(-1))
        model.timing = None
        eng.play(rng.random_sample(B))
        eng.fetch_ply()

    ply(False)
    st0 = eng.check_errors()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ply(True)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    st1 = eng.check_errors()
    eng.close()
    sims = st1["simulations"] - st0["simulations"]
    ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    flop = 2.0 * B * 64 * 256 * 9 * (119 + 38 * 256)
    return {"value": sims / dt, "unit": "simulations/s", "dtype": "f16", "network": "FastPolicyNet(operands='fp16'): k_tower16_bf16<ElemF16> + fused heads",
            "sample": "%d boards x num_searches=%d, one ply after one warm-up ply, %d simulations in %.2f s" % (B, S, sims, dt),
            "tower_launch_ms": ms, "tower_frac_of_mfma_peak": flop / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "fidelity": "64/64 test positions with the fp32 network's exact visit counts at 100 and 800 searches (bf16: 61-62/64); tests/test_gpu_train_and_precision.py"}


def train_step_sample(dev, batch=128, steps=40):
    """The optimiser step of the same loop (train_RL.py:103-122: fp32 forward + backward + Adam at the reference's batch size; configs[3]'s other half) on synthetic
    batches: ms per step with the matrix-core convolutions + fused BatchNorm (train_rl.train's default) and with torch / MIOpen fp32.  Side measurement, a few seconds."""
    import sigma_zero_amd as sz
    from sigma_zero_amd import train_rl as T
    from sigma_zero_amd.trainconv import split_convs
    import contextlib
    g = torch.Generator(device=dev).manual_seed(1)
    b = {"states": (torch.rand(batch, 119, 8, 8, device=dev, generator=g) < 0.15).float(), "actions": torch.softmax(torch.randn(batch, 4672, device=dev, generator=g) * 3, 1),
         "rewards": torch.randint(-1, 2, (batch,), device=dev, generator=g).float()}
    out = {"unit": "ms per optimiser step", "batch": batch, "steps": steps, "dtype": "f32 (convolutions: hi + lo f16 operands on the matrix cores, f32 accumulate)",
           "data": "synthetic"}
    for name, ctx in (("torch_miopen_fp32", contextlib.nullcontext), ("value", split_convs)):
        torch.manual_seed(0)
        model = sz.policyNN({}).to(dev).train()
        opt, sched = T.make_optimiser(model)
        with (ctx(model) if ctx is split_convs else ctx()):
            def step():
                opt.zero_grad(); loss, _, _ = T.loss_fn(model, b, dev); loss.backward(); opt.step(); sched.step()
            for _ in range(10):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            out[name] = (time.perf_counter() - t0) / steps * 1e3
    out["samples_per_s"] = batch / out["value"] * 1e3
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) before anything in this process has touched the GPU, relay their output
    (rank 0 prints the JSON line) and return their exit code.  The reference spawns its own workers the same way (train_RL.py:215-227)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--boards", type=int, default=4096, help="boards per GPU")
    ap.add_argument("--searches", type=int, default=800)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chess960", type=int, default=0)
    ap.add_argument("--net", default="fast", choices=["fast", "torch", "split"],
                    help="fast: hand-written bf16 MFMA tower (csrc/sz_nn.hip); split: the same on hi+lo bf16 operands (reference precision class); torch: MIOpen/ATen kernels")
    ap.add_argument("--operands", default="bf16", choices=["bf16", "fp16"],
                    help="MFMA operand element of --net fast: bf16 (BASELINE.json's configuration) or fp16 (11 bits of mantissa, same cycles, ~5 %% lower clock)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) or gloo (rehearsal of the N>1 path on one GPU)")
    ap.add_argument("--planes", default="bits128", choices=["bits128", "nhwc128"],
                    help="network-input image written by the tree kernel on the fast path: bit-packed (1 KiB/board) or bf16 NHWC (16 KiB/board)")
    ap.add_argument("--edges-per-board", type=int, default=0, help="child slots per board (0 = engine default: worst case when it fits in half of the free HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-config", action="store_true", help="skip the fp32-network sample (`parity_config`)")
    ap.add_argument("--no-split", action="store_true", help="skip the reference-precision run on the matrix cores (`roofline_split`)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the optimiser-step side sample (`train_step`)")
    ap.add_argument("--split-plies", type=int, default=3, help="timed plies of the `roofline_split` run (after one warm-up ply)")
    ap.add_argument("--no-kernel-events", action="store_true")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))        # nothing above this line has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; the line would report the wrong job size" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the search path has no CPU fallback")
    if a.dist_backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)      # rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.dist_backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import sigma_zero_amd as sz
    from sigma_zero_amd.selfplay import SelfPlayEngine

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    split = a.net == "split"
    fast = (a.net == "fast" and dtype == torch.bfloat16) or split
    B, S = a.boards, a.searches
    if fast:
        from sigma_zero_amd.fastnet import FastPolicyNet, SplitPolicyNet
        model = SplitPolicyNet(sz.policyNN({}).eval().to(dev), device=dev) if split else FastPolicyNet(sz.policyNN({}).eval(), device=dev, operands=a.operands)
        eng = SelfPlayEngine(model, {"C": 2, "num_searches": S}, B, chess960=bool(a.chess960), learning=True, device=dev, planes_dtype=a.planes, edges_per_board=a.edges_per_board)
    else:
        model = sz.policyNN({}).eval().to(dev).to(dtype).to(memory_format=torch.channels_last)
        eng = SelfPlayEngine(model, {"C": 2, "num_searches": S}, B, chess960=bool(a.chess960), learning=True, device=dev, planes_dtype=dtype, edges_per_board=a.edges_per_board)
    rng = np.random.RandomState(1234 + rank)
    import random
    prng = random.Random(rank)
    eng.new_games([prng.randrange(960) if a.chess960 else -1 for _ in range(B)])

    ev_nn, ev_tree, conv_events = [], [], []
    use_events = not a.no_kernel_events

    def evaluate(planes):
        if fast:
            policy, value = model(planes, inference=True)
            return policy, value.reshape(-1)
        x = planes.contiguous(memory_format=torch.channels_last)
        policy, value = model(x, inference=True)
        return policy.float().contiguous(), value.float().reshape(-1).contiguous()

    @torch.no_grad()
    def one_ply(timed):
        eng.begin()
        for it in range(S):
            if fast:
                # sample the conv kernel's own launch duration on every 50th iteration (HIP events on the launch stream)
                model.timing = conv_events if (timed and use_events and it % 50 == 25) else None
            if timed and use_events:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
                e0.record()
                policy, value = evaluate(eng.planes)
                e1.record()
                eng.step(policy, value)
                e2.record()
                ev_nn.append((e0, e1)); ev_tree.append((e1, e2))
            else:
                policy, value = evaluate(eng.planes)
                eng.step(policy, value)
        eng.play(rng.random_sample(B))
        rec = eng.fetch_ply()               # the training record of this ply goes to the host like in sim.py:71-73
        # finished games restart so that every board keeps working (steady-state self-play)
        over = rec["game_over"].astype(bool) & rec["active"].astype(bool)
        if over.any():
            eng.new_games([prng.randrange(960) if a.chess960 else -1 for _ in range(B)], active=over.astype(np.uint8))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        one_ply(False)
    st0 = eng.check_errors()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_ply(True)
    barrier()
    dt = time.perf_counter() - t0
    st1 = eng.check_errors()

    sims = st1["simulations"] - st0["simulations"]
    exps = st1["expansions"] - st0["expansions"]
    sum_depth = st1["sum_depth"] - st0["sum_depth"]
    sum_k = st1["sum_children"] - st0["sum_children"]
    from sigma_zero_amd.train_rl import aggregate_throughput
    # every board ran a full search in every timed ply (finished games are refilled, the counters survive the refill); reported, not asserted:
    # a rank-local assert in front of the collectives would leave the other ranks hanging in them
    bad = 0.0 if sims == B * S * a.steps else 1.0
    (total_sims, total_exps, bad_ranks), dt_max = aggregate_throughput([sims, exps, bad], dt, device="cpu" if a.dist_backend == "gloo" else dev)

    if rank == 0:
        out = {
            "metric": "MCTS simulations/sec (self-play)", "value": total_sims / dt_max, "unit": "simulations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_max / max(a.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x2" if split else (("f16" if (fast and a.operands == "fp16") else "bf16") if dtype == torch.bfloat16 else "f32"),
            "data": "synthetic (seeded random-init policyNN weights, self-play from %s start positions)" % ("Chess960" if a.chess960 else "classical"),
            "config": {"workload": "selfplay_%dboards_%dsearches" % (B, S), "boards_per_gpu": B, "num_searches": S, "C": 2,
                       "learning": True, "chess960": bool(a.chess960), "step": "one ply = full search of every board + sample + play",
                       "parallelism": "games sharded, %d rank(s), no data-path collective" % world},
            "expansions_per_s": total_exps / dt_max,
            "mean_leaf_depth": sum_depth / max(sims, 1), "mean_children": sum_k / max(exps, 1),
            "max_edges_used": st1["max_edges_used"],
            "sim_count_ok": bad_ranks == 0,     # simulations == boards x searches x steps on every rank
        }
        if use_events and ev_tree:
            tree_ms = float(np.mean([s.elapsed_time(e) for s, e in ev_tree]))
            nn_ms = float(np.mean([s.elapsed_time(e) for s, e in ev_nn]))
            launches = len(ev_tree)
            # network-input bytes per board: NCHW 119x64 elements, or the NHWC image of 64 x 128 bf16 on the fast path
            # network-input bytes per plane cell: bit-packed image 1024 B / (119*64), bf16 NHWC image 16 KiB / (119*64), NCHW 2 or 4
            plane_bytes = ((1024.0 / (119 * 64)) if a.planes == "bits128" else (128.0 * 2 / 119)) if fast else (2 if dtype == torch.bfloat16 else 4)
            bytes_per_launch = tree_bytes_per_launch(B, sims, exps, sum_depth, sum_k, plane_bytes) / launches
            ach = bytes_per_launch / (tree_ms * 1e-3) / 1e9
            tree_traffic, tree_src = None, None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_tree_latest.json")) as f:
                    tree_src = json.load(f)
                tree_traffic = tree_src["hbm_bytes_per_launch"] if (fast and B == 4096 and tree_src.get("planes") == a.planes) else None
                import sigma_zero_amd.build as _b
                if tree_traffic is not None and tree_src.get("source_hash") != _b.source_hash("tree"):
                    tree_traffic = None
                    tree_src = dict(tree_src, stale="sz_engine.hip / sz_chess.h changed since these counters were collected: run tools/profile_round.sh")
            except Exception:
                pass
            tree_roof = {"kernel": "k_search_step", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": tree_traffic, "traffic_source": tree_src, "launch_ms": tree_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "share_of_step_time": tree_ms / (tree_ms + nn_ms)}
            fl = exps / launches * sz.network.FLOPS_PER_BOARD
            tf = fl / (nn_ms * 1e-3) / 1e12
            nn_roof = {"kernel": "policyNN forward, all kernels (%s)" % ("sz_nn.hip MFMA tower + native heads" if fast else "MIOpen/ATen, bf16 channels_last"),
                       "bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
                       "forward_ms": nn_ms, "useful_boards_per_forward": exps / launches}
            if fast and conv_events:
                # dominant kernel of the whole job (>90 % of GPU time): k_block_bf16 = one fused BasicBlock (two 3x3 convs)
                # per launch, or k_conv_bf16<256,9> when block fusion is off
                conv_ms = float(np.mean([s.elapsed_time(e) for s, e in conv_events]))
                fused = getattr(model, "fuse_blocks", False)
                whole = split or (getattr(model, "persistent_tower", False) and B <= getattr(model, "persistent_max_boards", 0))
                conv_flop = 2.0 * B * 64 * 256 * 2304 * (2 if fused else 1)
                if whole:                                   # stem (119 real input planes) + 38 tower convolutions in one launch
                    conv_flop = 2.0 * B * 64 * 256 * 9 * (119 + 38 * 256)          # ALGORITHMIC flop (the split tower spends 3 MFMAs per product: not counted)
                ctf = conv_flop / (conv_ms * 1e-3) / 1e12
                traffic, traffic_src = None, None          # HBM bytes per launch from committed rocprofv3 --pmc passes
                try:
                    with open(os.path.join(ROOT, "profiles", "pmc_tower_latest.json" if whole else "pmc_conv_latest.json")) as f:
                        traffic_src = json.load(f)
                    traffic = traffic_src["hbm_bytes_per_launch"] if ((fused or whole) and not split and B == 4096 and (not whole or traffic_src.get("planes") == a.planes)) else None
                    # the counters belong to the kernel sources they were collected from: a record from other sources is not this binary's traffic
                    import sigma_zero_amd.build as _b
                    if traffic is not None and whole and traffic_src.get("source_hash") != _b.source_hash("tower"):
                        traffic = None
                        traffic_src = dict(traffic_src, stale="the kernel sources changed since these counters were collected (source_hash %s, now %s): run tools/profile_round.sh"
                                           % (traffic_src.get("source_hash"), _b.source_hash("tower")))
                except Exception:
                    pass
                out["roofline"] = {"kernel": "k_tower_split (persistent, hi+lo bf16 operands: 3 MFMAs per algorithmic product)" if split
                                   else "k_tower16_bf16 (persistent: stem + 19 BasicBlocks per launch, activations resident in LDS; %s operands)" % a.operands if whole
                                   else "k_block16_bf16<2> (fused BasicBlock: conv3x3+BN+ReLU -> LDS -> conv3x3+BN+residual+ReLU, 16x16x32 MFMA)" if fused
                                   else "k_conv_bf16<256,9,2> (fused 3x3 conv + folded BN + bias + residual + ReLU)", "bound": "mfma",
                                   "achieved": ctf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ctf / MFMA_BF16_PEAK_TFLOPS,
                                   "launch_ms": conv_ms, "algorithmic_flop_per_launch": conv_flop, "sampled_launches": len(conv_events),
                                   "traffic": traffic, "traffic_source": traffic_src}
                if split:
                    _, issued = split_tower_flops(B)
                    out["roofline"]["issued"] = issued / (conv_ms * 1e-3) / 1e12
                    out["roofline"]["issued_frac"] = out["roofline"]["issued"] / MFMA_BF16_PEAK_TFLOPS
                out["roofline_tree"] = tree_roof
                out["roofline_nn"] = nn_roof
            else:
                out["roofline"] = tree_roof
                out["roofline_nn"] = nn_roof
        # the two side measurements must never cost the headline line: a failure is reported in their place
        if world == 1 and fast:
            eng.close()
        if not a.no_split and world == 1 and fast and not split:
            try:
                out["roofline_split"] = split_sample(dev, B, S, a.chess960, plies=a.split_plies)
            except Exception as ex:                     # noqa: BLE001
                out["roofline_split"] = {"value": None, "error": "%s: %s" % (type(ex).__name__, ex)}
        if not a.no_split and world == 1 and fast and not split and a.operands == "bf16":
            try:
                out["fp16_config"] = fp16_sample(dev, B, S, a.chess960)
            except Exception as ex:                     # noqa: BLE001
                out["fp16_config"] = {"value": None, "error": "%s: %s" % (type(ex).__name__, ex)}
        if not a.no_parity_config and world == 1 and fast:
            try:
                out["parity_config"] = parity_config_sample(dev, B, a.chess960)
            except Exception as ex:                     # noqa: BLE001
                out["parity_config"] = {"value": None, "error": "%s: %s" % (type(ex).__name__, ex)}
        if not a.no_train_step and world == 1 and fast:
            try:
                out["train_step"] = train_step_sample(dev)
            except Exception as ex:                     # noqa: BLE001
                out["train_step"] = {"value": None, "error": "%s: %s" % (type(ex).__name__, ex)}
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (bounded ~15 s sample)
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as ex:                     # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "error": "%s: %s" % (type(ex).__name__, ex)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
