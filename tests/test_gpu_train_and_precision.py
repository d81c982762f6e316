"""GPU tests of (1) the RCCL code path of the optimiser step (backend "nccl" = RCCL, one rank: the collectives execute on the device),
(2) how far the shipped bf16 MFMA network moves a SEARCH away from the fp32 network the reference uses (network.py has no mixed
precision): same positions, same seeds, root visit distributions compared at S = 100 and S = 800, (3) ragged self-play through the
product API: slot refill + batch compaction give bit-identical per-game records."""
import json
import os
import random

import numpy as np
import pytest
import torch

import sigma_zero_amd as sz
from sigma_zero_amd import train_rl
from sigma_zero_amd.fastnet import FastPolicyNet
from sigma_zero_amd.selfplay import SelfPlayEngine

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ 1. RCCL
def test_rccl_gradsync_and_aggregation_execute_on_the_device():
    """north star: 'RCCL all-reduce over xGMI only for gradient sync in the optimiser step' (the loop of train_RL.py:219-264).
    One rank, backend nccl: GradSync's bucketed all-reduces and bench.py's aggregation run through RCCL on device tensors."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        torch.manual_seed(0)
        net = sz.policyNN({}).to(dev)
        net.eval()
        g = torch.Generator().manual_seed(5)
        batch = {"states": (torch.rand(4, 119, 8, 8, generator=g) < 0.15).float(), "actions": torch.softmax(torch.randn(4, 4672, generator=g) * 3, 1),
                 "rewards": torch.tensor([1., -1., 0., 1.])}
        loss, _, _ = train_rl.loss_fn(net, batch, dev)
        loss.backward()
        ref = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        net.zero_grad(set_to_none=True)
        sync = train_rl.GradSync(net, n_buckets=4, always_sync=True)
        assert sync.collective and sync.flat.is_cuda and sync.flat.numel() == 22_809_420
        sync.zero(); sync.begin_step()
        loss, _, _ = train_rl.loss_fn(net, batch, dev)
        loss.backward()
        assert len(sync.handles) == 4                                 # one asynchronous RCCL all-reduce per bucket, launched from backward hooks
        sync.finish_step()
        # SUM over one rank / 1 = the local gradient (torch's own conv backward is not run-to-run bitwise on the GPU, hence allclose here;
        # the collective itself is checked exactly below)
        assert torch.allclose(sync.flat, ref, rtol=1e-3, atol=1e-6)
        t = torch.randn(1 << 20, device=dev)
        t2 = t.clone()
        dist.all_reduce(t2, op=dist.ReduceOp.SUM)
        assert torch.equal(t, t2)
        assert sync.common_batches(7) == 7                            # all_reduce(MIN) on the device
        (a, b), t = train_rl.aggregate_throughput([123.0, 45.0], 2.5, device=dev, force=True)
        assert (a, b, t) == (123.0, 45.0, 2.5)
        opt, sched = train_rl.make_optimiser(net)
        dl = train_rl.DeviceBatches([np.zeros((119, 8), np.uint8)] * 8, [np.array([1, 2])] * 8, [np.array([0.5, 0.5])] * 8, [1] * 8, batch_size=4, device=dev)
        hist = train_rl.train(net, dl, opt, total_steps=0, lr_scheduler=sched, sync=sync, device=dev)
        assert len(hist) == 2 and np.isfinite(hist).all()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ 2. bf16 network vs fp32 network, at search level
def _positions(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        ct = sz.ChessTensor(chess960=True, scharnagl=rng.randrange(960))
        for _ in range(rng.randrange(0, 60)):
            if ct.board.is_game_over():
                break
            idx = ct.legal_action_indices()
            ct.push_action(idx[rng.randrange(len(idx))])
        if not ct.board.is_game_over():
            out.append(ct)
    return out


def _root_distributions(model, positions, S, planes_dtype):
    eng = SelfPlayEngine(model, {"C": 2, "num_searches": S}, len(positions), chess960=True, learning=True, planes_dtype=planes_dtype)
    for b, ct in enumerate(positions):
        eng.upload_game(b, ct)
    eng.search()
    eng.check_errors()
    action, visits, n_child, _, _ = eng.root_children()
    eng.close()
    return action, visits, n_child


@pytest.mark.parametrize("operands,S", [("bf16", 100), ("bf16", 800), ("fp16", 100), ("fp16", 800)])
def test_bf16_network_search_divergence_from_fp32(operands, S):
    """The fast inference path (FastPolicyNet: MFMA tower on bf16 or fp16 operands) against the reference-precision network (fp32 policyNN) on the
    SAME 64 positions: legal-move sets and child order are identical by construction; what moves is the visit distribution.  The figures are
    printed (and quoted in DESIGN.md §4); the bounds below are what the random-init network gives with margin."""
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    fast = FastPolicyNet(net, operands=operands)
    pos = _positions(64, seed=S)
    a32, v32, n32 = _root_distributions(net, pos, S, torch.float32)
    a16, v16, n16 = _root_distributions(fast, pos, S, "bits128")
    assert np.array_equal(n32, n16)
    top_same, l1, linf, kl, top_share = [], [], [], [], []
    for b in range(len(pos)):
        k = int(n32[b])
        assert np.array_equal(a32[b, :k], a16[b, :k])                          # same children, same order: indices are exact
        p, q = v32[b, :k] / v32[b, :k].sum(), v16[b, :k] / v16[b, :k].sum()
        top_same.append(int(np.argmax(p) == np.argmax(q)))
        l1.append(float(np.abs(p - q).sum())); linf.append(float(np.abs(p - q).max()))
        eps = 1e-9
        kl.append(float(np.sum(p * np.log((p + eps) / (q + eps)))))
        top_share.append(float(q[np.argmax(p)] / max(p.max(), eps)))           # how much of fp32's favourite's visits bf16 gives that move
    rep = dict(operands=operands, S=S, boards=len(pos), identical_boards=int(sum(int(np.array_equal(v32[b, :int(n32[b])], v16[b, :int(n32[b])])) for b in range(len(pos)))),
               top_move_agreement=float(np.mean(top_same)), mean_L1=float(np.mean(l1)), max_Linf=float(np.max(linf)),
               mean_Linf=float(np.mean(linf)), mean_KL=float(np.mean(kl)), max_KL=float(np.max(kl)), favourite_share=float(np.mean(top_share)))
    print("%s-vs-fp32 search divergence:" % operands, json.dumps(rep))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "%s_vs_fp32_search_S%d.json" % (operands, S)), "w") as f:
        json.dump(rep, f)
    # not within the north star's 1e-4 (that tolerance is met by the fp32 path: tests/test_gpu_parity.py, test_gpu_reference_golden.py);
    # bounded so that a regression of the bf16 kernels shows up here
    # measured on MI355X (round 2): S=100 mean L1 6.3e-4, max Linf 1.0e-2 (= one visit of 99), KL 1.1e-4; S=800 mean L1 1.2e-4, max Linf
    # 1.3e-3 (= one visit of 799), KL 2.1e-6; the most-visited move agreed on 64/64 boards at both budgets
    assert rep["mean_L1"] < 0.01 and rep["mean_KL"] < 1e-3 and rep["top_move_agreement"] >= 0.95 and rep["max_Linf"] <= 3.0 / (S - 1) + 1e-9, rep


def test_split_precision_network_searches_like_fp32():
    """SplitPolicyNet (reference-precision class on the matrix cores) against fp32 policyNN at search level: same 64 positions, S = 200"""
    from sigma_zero_amd.fastnet import SplitPolicyNet
    S = 200
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    split = SplitPolicyNet(net)
    pos = _positions(64, seed=77)
    a32, v32, n32 = _root_distributions(net, pos, S, torch.float32)
    a16, v16, n16 = _root_distributions(split, pos, S, "bits128")
    assert np.array_equal(n32, n16) and np.array_equal(a32, a16)
    same = sum(int(np.array_equal(v32[b], v16[b])) for b in range(len(pos)))
    worst = max(float(np.abs(v32[b] / v32[b].sum() - v16[b] / v16[b].sum()).max()) for b in range(len(pos)))
    print("split-precision vs fp32 search: %d/64 boards with identical visit counts, worst |delta fraction| %.2e" % (same, worst))
    with open(os.path.join("gpurun_out", "split_vs_fp32_search_S%d.json" % S), "w") as f:
        json.dump(dict(S=S, boards=64, identical_boards=same, worst_delta_fraction=worst), f)
    assert same >= 60 and worst <= 2.0 / (S - 1) + 1e-9


# ------------------------------------------------------------------------------------------------ 3. ragged self-play: refill + compaction
@pytest.mark.parametrize("kind,n_games,n_slots", [("fast", 12, 5), ("split", 12, 5), ("split", 300, 130), ("fast", 300, 130), ("fp16", 300, 130),
                                                   ("fp16", 700, 300), ("split", 700, 300)])
def test_refill_and_compaction_give_identical_per_game_records(kind, n_games, n_slots):
    """sim.py:102-123 plays exactly num_games games.  The product runs them on fewer board slots than games (slot refill) and evaluates
    only the boards that still play (compaction); per-game records must be bit-identical to the plain run (one slot per game, no
    compaction), because a game's results do not depend on what runs beside it.  All MFMA networks (bf16 / f16 operands, split precision); the
    300-game cases also cross from two boards per workgroup (300 boards) to one (130 and fewer); the 700-game cases run the plain side above 2 x #CUs boards, where
    the towers' last round goes out as a launch of its own in the one-board form."""
    from sigma_zero_amd.fastnet import SplitPolicyNet
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    fast = FastPolicyNet(net) if kind == "fast" else FastPolicyNet(net, operands="fp16") if kind == "fp16" else SplitPolicyNet(net)
    args = {"C": 2, "num_searches": 10 if n_games == 12 else 4}
    sch = [(3 * g + 5) % 960 for g in range(n_games)]
    caps = [6, 9, 14, 40, 11, 40, 7, 40, 25, 40, 40, 13] if n_games == 12 else [3 + (g * 7) % 6 for g in range(n_games)]    # ragged: games are cut at different plies (stand-in for different game lengths)

    def uni(g, ply):
        return ((g * 7919 + ply * 104729) % 1000003) / 1000003.0

    def run(**kw):
        st = {}
        random.seed(1); np.random.seed(1)
        games = sz.sim.play_games(fast, args, n_games, c960=True, scharnagl=sch, uniforms=uni, max_plies=caps, stats=st, **kw)
        return games, st

    plain, st_plain = run(compact=False)
    packed, st_packed = run(n_boards=n_slots, compact=True)
    assert len(plain) == len(packed) == n_games
    for g in range(n_games):
        a, b = plain[g], packed[g]
        assert len(a["actions"]) == len(b["actions"]) == caps[g], "game %d length" % g
        assert a["rewards"] == b["rewards"] and a["colours"] == b["colours"] and a["result"] == b["result"]
        for x, y in zip(a["states"], b["states"]):
            assert torch.equal(x, y)
        for x, y in zip(a["actions"], b["actions"]):
            assert list(x.keys()) == list(y.keys()) and list(x.values()) == list(y.values())
    assert st_plain["sims"] == st_packed["sims"]                       # the same work was done ...
    S = args["num_searches"]
    assert st_packed["nn_rows"] <= st_packed["sims"] + S * 2 * st_packed["plies"]       # ... on network batches that held (almost) only live boards
    assert st_plain["nn_rows"] == n_games * S * st_plain["plies"]


# ------------------------------------------------------------------------------------------------ 4. subtree reuse (non-reference option)
def _subtree_rows(tree, root_child_action):
    """rows of the DFS dump below the root child with that action index, depths shifted so that its children are depth 0"""
    d, a, v, w, p = tree
    k0 = next(k for k in range(1, len(d)) if d[k] == 0 and a[k] == root_child_action)
    k1 = next((k for k in range(k0 + 1, len(d)) if d[k] == 0), len(d))
    return (int(v[k0]), float(w[k0])), (d[k0 + 1:k1] - 1, a[k0 + 1:k1], v[k0 + 1:k1], w[k0 + 1:k1], p[k0 + 1:k1])


def test_subtree_reuse_option_carries_the_chosen_subtree_exactly_and_default_is_off():
    """optional reuse of the search tree across plies (the reference builds a fresh tree per ply, sim.py:53 — default).
    With args['reuse_subtree'] the tree at the start of ply p+1 must be EXACTLY the subtree below the move played at ply p (same nodes,
    visits, value sums, priors, child order), and the next search adds num_searches simulations on top of it."""
    from test_gpu_parity import random_evaluator
    S, B = 160, 48
    sch = [17 * b + 3 for b in range(B)]

    def run(reuse, plies=5):
        args = {"C": 2, "num_searches": S}
        if reuse:
            args["reuse_subtree"] = True
        eng = SelfPlayEngine(None, args, B, chess960=True, learning=True)
        eng.new_games(sch)
        ev = random_evaluator(123)
        urng = np.random.RandomState(9)
        out = []
        for ply in range(plies):
            eng.begin()
            torch.cuda.synchronize()
            start = [eng.debug_tree(b) for b in range(B)]
            for step in range(S):
                policy, value = ev(eng.planes, step)
                eng.step(policy, value)
            st = eng.check_errors()
            end = [eng.debug_tree(b) for b in range(B)]
            eng.play(urng.random_sample(B))
            rec = eng.fetch_ply()
            out.append((start, end, rec, st))
        eng.close()
        return out

    fresh = run(False)
    for start, end, rec, st in fresh:                                  # default: every ply starts from a bare root (visit_count 1, mcts.py:46)
        for b in range(B):
            assert len(start[b][0]) == 1 and start[b][2][0] == 1
            assert end[b][2][0] == 1 + S
    kept = run(True)
    n_reused = 0
    for ply in range(1, len(kept)):
        prev_end, rec_prev = kept[ply - 1][1], kept[ply - 1][2]
        start, end = kept[ply][0], kept[ply][1]
        for b in range(B):
            (n_c, w_c), sub = _subtree_rows(prev_end[b], int(rec_prev["chosen"][b]))
            d, a, v, w, p = start[b]
            if len(sub[0]) == 0:                                       # the played move was never expanded: nothing to keep, fresh root
                assert len(d) == 1 and v[0] == 1
                continue
            n_reused += 1
            assert (int(v[0]), float(w[0])) == (n_c, w_c), "ply %d board %d root stats" % (ply, b)
            assert np.array_equal(d[1:], sub[0]) and np.array_equal(a[1:], sub[1]) and np.array_equal(v[1:], sub[2]), "ply %d board %d subtree" % (ply, b)
            assert np.array_equal(w[1:], sub[3]) and np.array_equal(p[1:].view(np.uint32), sub[4].view(np.uint32))
            # the search continued on it: exactly S more simulations through the root, children account for all but the root's own first visit
            de, ae, ve, we, pe = end[b]
            assert ve[0] == n_c + S
            assert ve[1:][de[1:] == 0].sum() == n_c - 1 + S
            # ply 1's records are those of the fresh run (the first search is identical with or without the option)
    assert n_reused >= B // 2
    for b in range(B):
        assert np.array_equal(kept[0][2]["visits"][b], fresh[0][2]["visits"][b]) and kept[0][2]["chosen"][b] == fresh[0][2]["chosen"][b]
    assert kept[-1][3]["simulations"] == fresh[-1][3]["simulations"] == B * S * len(kept)


def test_play_games_to_the_end_on_fewer_slots_than_games():
    """64 Chess960 games on 16 slots, played to their natural end (mate, stalemate, material, 75 moves, fivefold): every game is complete —
    a result unless it hit the ply cap, one sample per ply, rewards derived from the result (sim.py:86-97), replayable move by move"""
    torch.manual_seed(0)
    fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
    random.seed(5); np.random.seed(5)
    st = {}
    games = sz.sim.play_games(fast, {"C": 2, "num_searches": 6}, 64, c960=True, n_boards=16, max_plies=600, stats=st)
    assert len(games) == 64
    kinds = set()
    total = 0
    for g in games:
        n = len(g["actions"])
        total += n
        assert n == len(g["states"]) == len(g["colours"]) == len(g["rewards"]) > 0
        assert (g["result"] is not None) or n == 600, "a game that stopped before the cap must be over"
        reward = {"1-0": 1, "0-1": -1}.get(g["result"], 0)
        assert g["rewards"] == [reward if i % 2 == 0 else -reward for i in range(n)]
        assert g["colours"][0] is True and all(a != b for a, b in zip(g["colours"], g["colours"][1:]))
        kinds.add(g["result"])
    assert st["sims"] == 6 * total and st["nn_rows"] == st["sims"]           # every network row carried a live board
    assert len(kinds) >= 2, kinds


def test_engine_follows_the_models_device():
    """a rank with local_rank > 0 must search on ITS GPU.  Needs two visible devices; the one-GPU box checks the default."""
    fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
    eng = SelfPlayEngine(fast, {"C": 2, "num_searches": 2}, 2, planes_dtype="bits128")
    assert eng.planes.device == fast.device == torch.device("cuda", torch.cuda.current_device())
    eng.close()
    if torch.cuda.device_count() > 1:
        before = torch.cuda.current_device()
        net1 = sz.policyNN({}).to("cuda:1").eval()
        fast1 = FastPolicyNet(net1)
        assert fast1.device == torch.device("cuda:1")
        eng1 = SelfPlayEngine(fast1, {"C": 2, "num_searches": 2}, 2, planes_dtype="bits128")
        assert eng1.planes.device == torch.device("cuda:1") and torch.cuda.current_device() == before
        eng1.new_games([-1, -1]); eng1.search(); eng1.check_errors()
        eng1.close()


def test_run_cycle_with_split_precision_self_play_and_reuse_option():
    """train_rl.run_cycle with the self-play network at the reference's precision class on the matrix cores (fast_inference="split"), and the
    subtree-reuse option passed through the args dict of the product API"""
    torch.manual_seed(0)
    random.seed(0); np.random.seed(0)
    net = sz.policyNN({}).cuda()
    opt, sched = train_rl.make_optimiser(net)
    before = net.conv1.weight.detach().clone()
    hist, games = train_rl.run_cycle(net, opt, sched, {"C": 2, "num_searches": 4, "max_plies": 10}, n_games=8, chess960=True, batch_size=8, total_steps=0,
                                     fast_inference="split")
    assert len(games) == 8 and all(len(g["actions"]) == 10 for g in games) and len(hist) == 10 and np.isfinite(hist).all()
    assert not torch.equal(before, net.conv1.weight.detach())
    fast = FastPolicyNet(net.eval())
    plain = sz.sim.play_games(fast, {"C": 2, "num_searches": 16}, 6, c960=True, scharnagl=[5, 6, 7, 8, 9, 10], uniforms=lambda g, p: 0.37, max_plies=6)
    reuse = sz.sim.play_games(fast, {"C": 2, "num_searches": 16, "reuse_subtree": True}, 6, c960=True, scharnagl=[5, 6, 7, 8, 9, 10], uniforms=lambda g, p: 0.37, max_plies=6)
    for a, b in zip(plain, reuse):
        assert list(a["actions"][0].items()) == list(b["actions"][0].items())          # the first ply's search is the same with or without the option
        assert all(abs(sum(d.values()) - 1.0) < 1e-12 for d in b["actions"]) and len(b["actions"]) == 6


def test_board_revived_without_recompaction_is_an_error_not_a_silent_skip():
    from sigma_zero_amd import _native as N
    from test_gpu_parity import random_evaluator
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": 3}, 4)
    eng.new_games([-1] * 4)
    eng.set_active([1, 1, 0, 1])
    assert eng.compact() == 3
    ev = random_evaluator(3)
    eng.search(lambda planes: ev(planes, 0))
    eng.check_errors()
    eng.set_active([1, 1, 1, 1])                       # board 2 comes back, but the batch rows were not renumbered
    eng.search(lambda planes: ev(planes, 0))
    with pytest.raises(N.NativeError) as ei:
        eng.check_errors()
    assert ei.value.code == N.SZ_ERR_STATE
    eng.close()


def test_root_noise_and_subtree_reuse_exclude_each_other():
    """the two non-reference options cannot be combined (a reused root is never expanded again, so its children would never see the noise): refused at both levels"""
    with pytest.raises(ValueError):
        SelfPlayEngine(None, {"C": 2, "num_searches": 8, "reuse_subtree": True, "root_dirichlet_alpha": 0.3}, 2, chess960=True, learning=True)
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": 8, "reuse_subtree": True}, 2, chess960=True, learning=True)
    from sigma_zero_amd import _native as N
    with pytest.raises(N.NativeError):
        eng.set_root_noise(torch.ones(2, 218, device="cuda"))
    eng.set_root_noise(None)
    eng.close()
