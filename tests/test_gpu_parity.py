"""GPU parity: the HIP engine (through the C ABI) against the oracle, on the same seeded inputs.
Integer/index results must be bit-exact; priors and value sums are compared bitwise too because both
sides are fed the identical (policy, value) numbers."""
import random

import numpy as np
import pytest
import torch

import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
from sigma_zero_amd.selfplay import SelfPlayEngine, unpack_planes, NOISE_REFERENCE
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FENS = [
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
    "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1",
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8",
    "7k/5Q2/5K2/8/8/8/8/8 w - - 0 1",                 # mates and stalemates one ply away: terminal leaves everywhere
    "8/8/8/8/8/5k2/4p3/4K3 b - - 0 1",
    "6k1/5ppp/8/8/8/8/5PPP/R5K1 w - - 0 1",           # back-rank mate in one
    "8/P7/8/8/8/8/7k/K7 w - - 0 1",                   # promotions (queen + under-promotions)
    "4k3/8/8/8/8/8/8/4K2R w K - 0 1",                 # castling with few pieces
    "8/8/5k2/8/8/3KR3/8/8 w - - 140 100",             # 75-move rule inside the tree
    "R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1",   # 218 legal moves: the maximum child span
    "rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f6 0 3",   # en passant available at the root
]


class Mirror:
    """One board: product host game (for upload) + oracle game + oracle search."""

    def __init__(self, c960=False, scharnagl=518, fen=None, pre_moves=0, rng=None):
        self.ct = sz.ChessTensor(chess960=c960, scharnagl=scharnagl, fen=fen)
        self.oct = O.ChessTensor.from_fen(fen, chess960=c960) if fen else O.ChessTensor(chess960=c960, scharnagl=scharnagl)
        for _ in range(pre_moves):
            if self.ct.board.is_game_over():
                break
            idx, moves = self.oct.legal_action_indices()
            k = rng.randrange(len(idx))
            self.ct.push_action(idx[k])
            self.oct.move_piece(moves[k])
        self.search = None


def make_boards(n, seed, c960=False):
    rng = random.Random(seed)
    boards = []
    for i in range(n):
        if i < len(FENS) and not c960:
            boards.append(Mirror(fen=FENS[i]))
        else:
            boards.append(Mirror(c960=c960, scharnagl=rng.randrange(960), pre_moves=rng.randrange(0, 40), rng=rng))
    return boards


def lockstep_search(boards, S, learning, evaluator, c960=False, dtype=torch.float32):
    B = len(boards)
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, chess960=c960, learning=learning, planes_dtype=dtype)
    for b, m in enumerate(boards):
        eng.upload_game(b, m.ct)
        m.search = O.Search.on_chess(m.oct, c=2.0, num_searches=S, learning=learning, noise_value=NOISE_REFERENCE)
    eng.begin()
    n_evals = 0
    for step in range(S + 1):
        torch.cuda.synchronize()
        mask, depth, n_nodes, n_edges, status = eng.debug_pending()
        planes = eng.planes.float().cpu().numpy()
        pending = [(status[b] & 2) != 0 for b in range(B)]
        o_pending = [m.search.advance() for m in boards]
        assert pending == o_pending, "step %d: pending sets differ" % step
        if not any(pending):
            break
        for b, m in enumerate(boards):
            if not pending[b]:
                continue
            want = m.search.leaf_planes()
            got = planes[b].astype(np.uint8)
            if not np.array_equal(got, want):
                bad = [c for c in range(119) if not np.array_equal(got[c], want[c])]
                raise AssertionError("step %d board %d planes differ at %s (depth %d, oracle trace %s)"
                                     % (step, b, bad, depth[b], m.search.trace()))
            legal = m.search.leaf_actions()
            mine = [p * 64 + v for p in range(73) for v in range(64) if (int(mask[b, p]) >> v) & 1]
            assert mine == legal, "step %d board %d legal mask" % (step, b)
            assert depth[b] == len(m.search.trace()), "step %d board %d depth" % (step, b)
        policy, value = evaluator(eng.planes, step)
        pol_h, val_h = policy.cpu().numpy(), value.cpu().numpy()
        for b, m in enumerate(boards):
            if pending[b]:
                m.search.feed(pol_h[b], val_h[b])
                n_evals += 1
        eng.step(policy, value)
    st = eng.check_errors()
    action, visits, n_child, prior, wsum = eng.root_children()
    for b, m in enumerate(boards):
        idx, vis, _ = m.search.root_children()
        k = int(n_child[b])
        assert k == len(idx), "board %d child count" % b
        assert action[b, :k].tolist() == idx, "board %d child order" % b
        assert visits[b, :k].tolist() == vis, "board %d visits" % b
        pr, ws = m.search.root_stats()
        assert np.array_equal(prior[b, :k].view(np.uint32), pr.view(np.uint32)), "board %d priors" % b
        assert np.array_equal(wsum[b, :k], ws), "board %d value sums" % b
    exp = sum(m.search.counters()[0] for m in boards)
    term = sum(m.search.counters()[1] for m in boards)
    assert st["expansions"] == exp == n_evals and st["terminal_hits"] == term
    eng.close()
    return st


def random_evaluator(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)

    def ev(planes, step):
        B = planes.shape[0]
        logits = torch.randn(B, N.SZ_ACTIONS, generator=g, device="cuda") * 2.0
        return torch.softmax(logits, 1).contiguous(), (torch.rand(B, generator=g, device="cuda") * 2 - 1).contiguous()
    return ev


@pytest.mark.parametrize("learning", [False, True])
def test_lockstep_search_matches_oracle(learning):
    boards = make_boards(24, seed=7)
    st = lockstep_search(boards, S=64, learning=learning, evaluator=random_evaluator(11 + learning))
    assert st["terminal_hits"] > 0 and st["expansions"] > 0


def test_lockstep_search_chess960_deeper():
    boards = make_boards(12, seed=21, c960=True)
    lockstep_search(boards, S=200, learning=True, evaluator=random_evaluator(5), c960=True)


def test_peaked_policies_drop_zero_priors_and_go_deep():
    # very peaked policies: some legal moves get probability exactly 0 (dropped children), trees go deep
    g = torch.Generator(device="cuda").manual_seed(3)

    def ev(planes, step):
        B = planes.shape[0]
        logits = torch.randn(B, N.SZ_ACTIONS, generator=g, device="cuda") * 60.0
        return torch.softmax(logits, 1).contiguous(), (torch.rand(B, generator=g, device="cuda") * 2 - 1).contiguous()
    boards = make_boards(16, seed=9)
    lockstep_search(boards, S=120, learning=False, evaluator=ev)


def test_real_network_fp32_end_to_end():
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()

    def ev(planes, step):
        with torch.no_grad():
            p, v = net(planes, inference=True)
        return p.float().contiguous(), v.float().reshape(-1).contiguous()
    boards = make_boards(8, seed=33)
    lockstep_search(boards, S=24, learning=True, evaluator=ev)


def test_bf16_planes_are_the_same_bits():
    boards = make_boards(12, seed=4)
    lockstep_search(boards, S=16, learning=False, evaluator=random_evaluator(2), dtype=torch.bfloat16)


def test_selfplay_games_match_oracle():
    """Several plies of self-play: sampling (np.random.choice semantics), move application, game-over test,
    training records — against oracle search + oracle sampler on the same uniforms."""
    S, B, PLIES = 12, 12, 40
    rng = random.Random(5)
    sch = [rng.randrange(960) for _ in range(B)]
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, chess960=True, learning=True)
    eng.new_games(sch)
    octs = [O.ChessTensor(chess960=True, scharnagl=n) for n in sch]
    alive = [True] * B
    ev = random_evaluator(77)
    urng = np.random.RandomState(1)
    for ply in range(PLIES):
        searches = [O.Search.on_chess(o, c=2.0, num_searches=S, learning=True, noise_value=NOISE_REFERENCE) if alive[b] else None
                    for b, o in enumerate(octs)]
        eng.begin()
        for step in range(S):
            policy, value = ev(eng.planes, step)
            pol_h, val_h = policy.cpu().numpy(), value.cpu().numpy()
            for b, s in enumerate(searches):
                if s is not None and s.advance():
                    s.feed(pol_h[b], val_h[b])
            eng.step(policy, value)
        u = urng.random_sample(B)
        eng.play(u)
        rec = eng.fetch_ply()
        eng.check_errors()
        for b, s in enumerate(searches):
            if s is None:
                assert not rec["active"][b]
                continue
            assert not s.advance()
            idx, vis, moves = s.root_children()
            k = int(rec["n_child"][b])
            assert rec["active"][b] and k == len(idx)
            assert rec["action"][b, :k].tolist() == idx and rec["visits"][b, :k].tolist() == vis
            assert bool(rec["colour"][b]) == bool(octs[b].turn)
            assert np.array_equal(unpack_planes(rec["packed"][b]).astype(np.uint8), octs[b].get_representation())
            choice = O.sample_move(vis, u[b])
            assert int(rec["chosen"][b]) == idx[choice], "ply %d board %d sampled move" % (ply, b)
            octs[b].move_piece(moves[choice])
            v, t = octs[b].get_value_and_terminated()
            assert bool(rec["game_over"][b]) == t
            if t:
                o, w = octs[b].board.outcome()
                assert int(rec["result"][b]) == (0 if w < 0 else (1 if w == 1 else -1))
                alive[b] = False
            pos, gply = eng.debug_position(b)
            assert [int(x) for x in pos[:7]] == octs[b].board.bitboards()[:7] and gply == octs[b].board.ply
        if not any(alive):
            break
    eng.close()


def test_mcts0_api_single_position():
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    game = sz.ChessTensor()
    for u in ("e2e4", "e7e5", "g1f3"):
        game.move_piece(sz.Move.from_uci(u))
    args = {"C": 2, "num_searches": 40}
    probs = sz.MCTS0(game=game, args=args, model=net).search(game.board, verbose=False, learning=False)
    assert abs(sum(probs.values()) - 1.0) < 1e-12 and len(probs) == len(game.get_moves())
    keys = [sz.chess_tensor.action_index(m, game.board.turn) for m in probs]
    assert keys == sorted(keys)
    # same search through the oracle with the same fp32 network on the same device
    oct_ = O.ChessTensor()
    for u in ("e2e4", "e7e5", "g1f3"):
        oct_.move_piece(O.Move.from_uci(u))

    def ev(planes):
        with torch.no_grad():
            p, v = net(torch.from_numpy(planes.astype(np.float32)).cuda().unsqueeze(0), inference=True)
        return p[0].float().cpu().numpy(), float(v.reshape(-1)[0])
    s = O.search_with_evaluator(oct_, ev, c=2.0, num_searches=40, learning=False)
    ref = s.action_probs()
    assert keys == list(ref.keys())
    # batch-1 evaluations on both sides: visit fractions agree to 1e-4 (north_star tolerance), indices exactly
    assert np.allclose(list(probs.values()), list(ref.values()), atol=1e-4)
    with pytest.raises(ZeroDivisionError):
        sz.MCTS0(game=game, args={"C": 2, "num_searches": 1}, model=net).search(game.board, verbose=False)


def test_sqrt_visits_rounding_matches_host():
    # f32(sqrt_f64(N)) on the device equals the host's for every parent visit count a search can reach
    n = torch.arange(1, 1 << 20, device="cuda", dtype=torch.float64)
    dev = torch.sqrt(n).float().cpu().numpy()
    host = np.sqrt(np.arange(1, 1 << 20, dtype=np.float64)).astype(np.float32)
    assert np.array_equal(dev, host)


def test_generate_training_data_api():
    """sim.generate_training_data drop-in: dict layout of sim.py:38-43, rewards of :86-97, replayable games."""
    import random
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    random.seed(3)
    np.random.seed(3)
    args = {"C": 2, "num_searches": 6}
    games = sz.sim.play_games(net, args, 6, c960=True, max_plies=30)
    for g in games:
        n = len(g["actions"])
        assert n == len(g["states"]) == len(g["colours"]) == len(g["rewards"]) and n > 0
        for st, act, col in zip(g["states"], g["actions"], g["colours"]):
            assert st.shape == (119, 8, 8) and st.dtype == torch.bool
            assert abs(sum(act.values()) - 1.0) < 1e-12
            assert bool(st[112, 0, 0]) == col                       # colour plane of the mover's view
        assert g["colours"][0] is True and all(a != b for a, b in zip(g["colours"], g["colours"][1:]))
        assert set(g["rewards"]) <= {0, 1, -1}
    import os
    rd = {}
    data = sz.generate_training_data(net, num_games=3, args={"C": 2, "num_searches": 6, }, return_dict=rd, c960=False, n_boards=2)   # 3 games on 2 slots: refill
    assert list(rd.keys()) == [os.getpid()] and rd[os.getpid()] is data                  # sim.py:120-121
    assert set(data.keys()) == {"states", "actions", "rewards", "colours"}
    n = len(data["actions"])
    assert n > 0 and len(data["states"]) == len(data["rewards"]) == len(data["colours"]) == n
    assert sum(1 for st in data["states"] if not st[113].any()) == 3                      # exactly num_games games: 3 start positions (plane 113 = any move played)
    # FastPolicyNet path through the same API
    from sigma_zero_amd.fastnet import FastPolicyNet
    games2 = sz.sim.play_games(FastPolicyNet(net), args, 4, c960=False, max_plies=4)
    assert all(len(g["actions"]) == 4 for g in games2)


def test_train_cycle_smoke_on_gpu():
    from sigma_zero_amd import train_rl
    import random
    torch.manual_seed(0)
    random.seed(0); np.random.seed(0)
    net = sz.policyNN({}).cuda()
    opt, sched = train_rl.make_optimiser(net)
    sd_before = net.conv1.weight.detach().clone()
    hist, games = train_rl.run_cycle(net, opt, sched, {"C": 2, "num_searches": 4, "max_plies": 12}, n_games=16, chess960=True, batch_size=8, total_steps=0)
    assert len(hist) >= 1 and all(np.isfinite(h).all() for h in hist)
    assert not torch.equal(sd_before, net.conv1.weight.detach())


def test_capacity_overflow_is_reported_not_silent():
    """a board that runs out of child slots raises SZ_ERR_CAPACITY through the stats path (sticky per-board flag)"""
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": 64}, 4, edges_per_board=230)
    eng.new_games([-1] * 4)
    ev = random_evaluator(1)
    eng.begin()
    for step in range(64):
        p, v = ev(eng.planes, step)
        eng.step(p, v)
    with pytest.raises(N.NativeError) as ei:
        eng.check_errors()
    assert ei.value.code == N.SZ_ERR_CAPACITY
    eng.close()


def test_engine_refuses_without_model_device():
    with pytest.raises(TypeError):
        sz.MCTS0(game=object(), args={"C": 2, "num_searches": 4}, model=sz.policyNN({})).search(None)


def test_inactive_and_ragged_boards():
    """boards can be switched off (empty slots of a ragged batch): they never ask for an evaluation and keep their state"""
    boards = make_boards(6, seed=2)
    S = 10
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, 6)
    for b, m in enumerate(boards):
        eng.upload_game(b, m.ct)
    active = [1, 0, 1, 1, 0, 1]
    eng.set_active(active)
    ev = random_evaluator(9)
    eng.begin()
    for step in range(S):
        torch.cuda.synchronize()
        _, _, _, _, status = eng.debug_pending()
        for b in range(6):
            assert bool(status[b] & 2) == bool(active[b]) or (status[b] & 4), (step, b)
        p, v = ev(eng.planes, step)
        eng.step(p, v)
    eng.check_errors()
    action, visits, n_child, _, _ = eng.root_children()
    for b in range(6):
        assert (n_child[b] > 0) == bool(active[b])
        if active[b]:
            assert visits[b, :n_child[b]].sum() == S - 1          # root.visit_count = 1 + S, first simulation expands the root
    eng.play(np.full(6, 0.5))
    rec = eng.fetch_ply()
    assert rec["active"].tolist() == active
    eng.close()


def test_arena_greedy_match():
    """model-vs-model gating (test_update.py semantics): greedy play, learning off; a model against itself with mirrored
    colours must produce mirrored results (the search is deterministic)."""
    from sigma_zero_amd.arena import play_match
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    out = play_match(net, net, {"C": 2, "num_searches": 6}, n_games=4, max_plies=10)
    assert out["a_wins"] + out["b_wins"] + out["draws"] + out["unfinished"] == 4
    # boards 0 and 2 (same colours, same deterministic play) agree, and so do boards 1 and 3
    assert out["results"][0] == out["results"][2] and out["results"][1] == out["results"][3]
    # greedy = first maximum: check against the oracle on one position
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": 12}, 2, learning=False)
    eng.new_games([-1, -1])
    ev = random_evaluator(4)
    oct_ = O.ChessTensor()
    s = O.Search.on_chess(oct_, c=2.0, num_searches=12, learning=False)
    eng.begin()
    for step in range(12):
        p, v = ev(eng.planes, step)
        if s.advance():
            s.feed(p[0].cpu().numpy(), float(v[0]))
        eng.step(p, v)
    eng.play(np.array([-1.0, 0.3]))
    rec = eng.fetch_ply()
    idx, vis, _ = s.root_children()
    assert int(rec["chosen"][0]) == idx[int(np.argmax(vis))]
    eng.close()


def test_full_games_to_the_end_match_oracle():
    """whole self-play games (tiny search budget) until every game is over: move sampling, repetition planes, 75-move /
    fivefold / insufficient-material / mate detection and results on the device, ply by ply against the oracle"""
    S, B, MAX_PLIES = 3, 10, 700
    rng = random.Random(11)
    sch = [rng.randrange(960) for _ in range(B)]
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, chess960=True, learning=True)
    eng.new_games(sch)
    octs = [O.ChessTensor(chess960=True, scharnagl=n) for n in sch]
    alive = [True] * B
    ev = random_evaluator(123)
    urng = np.random.RandomState(7)
    ended = {}
    for ply in range(MAX_PLIES):
        searches = [O.Search.on_chess(o, c=2.0, num_searches=S, learning=True, noise_value=NOISE_REFERENCE) if alive[b] else None
                    for b, o in enumerate(octs)]
        eng.begin()
        for step in range(S):
            policy, value = ev(eng.planes, step)
            pol_h, val_h = policy.cpu().numpy(), value.cpu().numpy()
            for b, s in enumerate(searches):
                if s is not None and s.advance():
                    s.feed(pol_h[b], val_h[b])
            eng.step(policy, value)
        u = urng.random_sample(B)
        eng.play(u)
        rec = eng.fetch_ply()
        eng.check_errors()
        for b, s in enumerate(searches):
            if s is None:
                assert not rec["active"][b]
                continue
            assert not s.advance()
            idx, vis, moves = s.root_children()
            k = int(rec["n_child"][b])
            assert rec["action"][b, :k].tolist() == idx and rec["visits"][b, :k].tolist() == vis, (ply, b)
            assert np.array_equal(unpack_planes(rec["packed"][b]).astype(np.uint8), octs[b].get_representation()), (ply, b)
            choice = O.sample_move(vis, u[b])
            assert int(rec["chosen"][b]) == idx[choice], (ply, b)
            octs[b].move_piece(moves[choice])
            v, t = octs[b].get_value_and_terminated()
            assert bool(rec["game_over"][b]) == t, (ply, b)
            if t:
                kind, w = octs[b].board.outcome()
                assert int(rec["result"][b]) == (0 if w < 0 else (1 if w == 1 else -1))
                ended[b] = kind
                alive[b] = False
        if not any(alive):
            break
    eng.close()
    assert len(ended) >= B - 2, "games should finish within %d plies: %s" % (MAX_PLIES, ended)
    assert len(set(ended.values())) >= 2, ended          # more than one kind of game end was exercised


def test_train_rl_main_two_ranks_on_one_gpu(tmp_path):
    """the train_RL.main loop as two gloo ranks sharing the GPU: per-rank self-play, bucketed gradient all-reduce, rank-0 saves
    with the reference's file names and state_dict keys"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "tools", "run_train_rl.py"), "--epochs", "1", "--games-per-rank", "6,9", "--searches", "4", "--batch-size", "8", "--total-steps", "0",
           "--max-plies", "10", "--backend", "gloo", "--save-dir", str(tmp_path / "saves"), "--games-dir", str(tmp_path / "games"), "--log-dir", str(tmp_path / "logs")]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    # per-step loss log (train_RL.py:124-125): one JSON line per optimiser step
    import json
    steps = [json.loads(l) for l in open(tmp_path / "logs" / "RL_train.jsonl")]
    assert len(steps) == 7 and [r["step"] for r in steps] == list(range(7)) and all(np.isfinite(r["loss"]) and abs(r["loss"] - r["mse"] - r["cross_entropy"]) < 1e-6 and r["lr"] == 1e-4 for r in steps)
    # rank 0 holds 60 samples = 7 batches of 8, rank 1 holds 90 = 11: both run MIN = 7 all-reduced steps (no deadlock)
    assert "epoch 1: 2 ranks x 6 games" in out.stdout and "7 optimiser steps" in out.stdout
    other = torch.load(tmp_path / "games" / "RL_960_1.rank1.pt", weights_only=True)
    assert len(other["states"]) == 90
    sd = torch.load(tmp_path / "saves" / "RL_1.pt", weights_only=True)
    assert list(sd.keys()) == list(sz.policyNN({}).state_dict().keys())
    games = torch.load(tmp_path / "games" / "RL_960_1.pt", weights_only=True)
    assert len(games["states"]) == len(games["actions"]) == len(games["rewards"]) == len(games["colours"]) > 0
    assert tuple(games["states"][0].shape) == (119, 8) and games["states"][0].dtype == torch.uint8
    # --merge-games: the reference's single games file (train_RL.py:229-241) = rank 0's samples followed by rank 1's
    # ... and, in the same second run, --train-graph on: forward + backward of a step as one HIP-graph replay, all gradient buckets reduced over the two ranks after it
    out = subprocess.run([("29534" if c == "29533" else c) for c in cmd] + ["--merge-games", "--games-dir", str(tmp_path / "games2"), "--save-dir", str(tmp_path / "saves2"),
                                                                            "--log-dir", str(tmp_path / "logs2"), "--train-graph", "on"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    steps2 = [json.loads(l) for l in open(tmp_path / "logs2" / "RL_train.jsonl")]
    assert len(steps2) == 7 and abs(steps2[0]["loss"] - steps[0]["loss"]) < 1e-4 * abs(steps[0]["loss"])          # the same games, the same first batch
    assert all(abs(a["loss"] - b["loss"]) < 5e-2 * abs(b["loss"]) for a, b in zip(steps2, steps)), (steps2, steps)   # later steps: within what two eager runs differ by
    merged = torch.load(tmp_path / "games2" / "RL_960_1.pt", weights_only=True)
    assert len(merged["states"]) == len(merged["actions"]) == len(merged["rewards"]) == len(merged["colours"]) == 150
    assert not (tmp_path / "games2" / "RL_960_1.rank1.pt").exists()
    assert all(torch.equal(a, b) for a, b in zip(merged["states"][:60], games["states"])) and all(torch.equal(a, b) for a, b in zip(merged["states"][60:], other["states"]))


def test_slot_refill_keeps_other_boards_intact():
    """finished games are replaced by new ones in place (steady-state self-play): only the flagged boards restart"""
    B, S = 8, 3
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, chess960=True, learning=True)
    sch = [11, 22, 33, 44, 55, 66, 77, 88]
    eng.new_games(sch)
    ev = random_evaluator(5)
    urng = np.random.RandomState(3)
    for ply in range(6):
        eng.search(lambda planes: ev(planes, 0))
        eng.play(urng.random_sample(B))
        eng.fetch_ply()
    before = [eng.debug_position(b) for b in range(B)]
    assert all(p[1] == 6 for p in before)
    mask = np.array([0, 1, 0, 0, 1, 0, 0, 0], dtype=np.uint8)
    eng.new_games([900 + b for b in range(B)], active=mask)
    torch.cuda.synchronize()
    for b in range(B):
        pos, gply = eng.debug_position(b)
        if mask[b]:
            assert gply == 0 and [int(x) for x in pos[:7]] == O.Board.from_chess960_pos(900 + b).bitboards()[:7]
        else:
            assert gply == 6 and np.array_equal(pos, before[b][0])
    eng.search(lambda planes: ev(planes, 0))
    eng.check_errors()
    _, visits, n_child, _, _ = eng.root_children()
    assert all(visits[b, :n_child[b]].sum() == S - 1 for b in range(B))
    eng.close()


@pytest.mark.parametrize("chess960,operands", [(True, "bf16"), (False, "bf16"), (True, "fp16")])
def test_full_size_invariants_4096_boards_800_searches(chess960, operands):
    """BASELINE configs[2] (classical starts) and configs[4] (Chess960 starts) at full size through size-independent properties (the oracle cannot follow 3.3 M simulations): every
    simulation is accounted for, every root's child visits sum to S-1 (mcts.py:46,118), the sampled move is a visited root child,
    records are complete, no board reports an error, edge capacity is not approached."""
    from sigma_zero_amd.fastnet import FastPolicyNet
    B, S = 4096, 800
    torch.manual_seed(0)
    fast = FastPolicyNet(sz.policyNN({}).cuda().eval(), operands=operands)      # bf16: the bench's headline network; fp16: the product's default self-play network
    eng = SelfPlayEngine(fast, {"C": 2, "num_searches": S}, B, chess960=chess960, learning=True, planes_dtype="bits128")
    prng = np.random.RandomState(5)
    eng.new_games(prng.randint(0, 960, size=B).tolist() if chess960 else [-1] * B)
    if not chess960:                                      # 4096 identical start positions are 4096 identical searches: check the ply after the first sampled moves
        eng.search()
        eng.play(prng.random_sample(B))
        eng.fetch_ply()
    st0 = eng.stats()
    eng.search()
    st = eng.check_errors()
    assert st["simulations"] - st0["simulations"] == B * S
    assert st["expansions"] + st["terminal_hits"] - st0["expansions"] - st0["terminal_hits"] == B * S
    assert st["max_edges_used"] < S * 64 + 256
    action, visits, n_child, prior, wsum = eng.root_children()
    k = np.arange(visits.shape[1])[None, :] < n_child[:, None]
    assert (n_child >= 1).all() and (n_child <= 218).all()
    assert ((visits * k).sum(1) == S - 1).all()
    assert (np.diff(np.where(k, action, 1 << 30), axis=1)[:, :-1][k[:, 1:-1]] > 0).all()          # ascending action order
    assert (np.abs(np.where(k, wsum, 0)) <= np.where(k, visits, 0) + 1e-9).all()                   # |W| <= N
    psum = np.where(k, prior, 0).sum(1)                                                             # 0.75*1 + 0.25*K*noise
    assert np.allclose(psum, 0.75 + 0.25 * n_child * float(np.float32(1) - np.float32(2.0 ** -24)), atol=1e-3)
    u = prng.random_sample(B)
    eng.play(u)
    rec = eng.fetch_ply()
    assert rec["active"].all() and (rec["n_child"] == n_child).all()
    for b in range(0, B, 97):
        kk = int(n_child[b])
        assert rec["chosen"][b] in action[b, :kk] and visits[b, :kk][list(action[b, :kk]).index(rec["chosen"][b])] > 0
    eng.close()


def test_search_is_bitwise_reproducible_with_the_mfma_network():
    """same seeds, same weights -> identical visit counts and priors on two fresh engines (no atomics, fixed summation orders)"""
    from sigma_zero_amd.fastnet import FastPolicyNet
    torch.manual_seed(3)
    fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
    outs = []
    for rep in range(2):
        eng = SelfPlayEngine(fast, {"C": 2, "num_searches": 60}, 130, chess960=True, learning=True, planes_dtype="bits128")
        eng.new_games(list(range(200, 330)))
        urng = np.random.RandomState(9)
        res = []
        for ply in range(3):
            eng.search()
            res.append([a.copy() for a in eng.root_children()])
            eng.play(urng.random_sample(130))
            eng.fetch_ply()
        eng.check_errors()
        eng.close()
        outs.append(res)
    for r0, r1 in zip(*outs):
        for a, b in zip(r0, r1):
            assert np.array_equal(a, b)


def test_root_dirichlet_option_is_root_only_and_exact():
    """NON-REFERENCE option (off by default): with Gamma draws g supplied, root prior k == 0.75*p_k + 0.25*g_k/sum(g[:K]) and the
    search below the root runs on un-noised priors; without it the reference constant is back."""
    B, S = 6, 40
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, chess960=True, learning=True)
    eng.new_games([100 + b for b in range(B)])
    g = torch.Generator(device="cuda").manual_seed(1)
    gamma = torch._standard_gamma(torch.full((B, N.SZ_MAX_MOVES), 0.3, device="cuda"), generator=g)
    ev = random_evaluator(3)
    eng.set_root_noise(gamma)
    eng.begin()
    pol0 = None
    for it in range(S):
        policy, value = ev(eng.planes, it)
        if it == 0:
            pol0 = policy.cpu().numpy()
        eng.step(policy, value)
    eng.check_errors()
    action, visits, n_child, prior, wsum = eng.root_children()
    gam = gamma.cpu().numpy()
    for b in range(B):
        k = int(n_child[b])
        p = pol0[b, action[b, :k]].astype(np.float32)
        p = p / p.sum(dtype=np.float32)
        n = gam[b, :k] / gam[b, :k].sum(dtype=np.float32)
        want = 0.75 * p + 0.25 * n
        assert np.allclose(prior[b, :k], want, rtol=2e-6, atol=1e-8), b
        assert abs(float(prior[b, :k].sum()) - 1.0) < 1e-5 and visits[b, :k].sum() == S - 1
    # back to the reference noise
    eng.set_root_noise(None)
    eng.begin()
    for it in range(2):
        policy, value = ev(eng.planes, it)
        eng.step(policy, value)
    _, _, n_child, prior, _ = eng.root_children()
    k = int(n_child[0])
    assert abs(float(prior[0, :k].sum()) - (0.75 + 0.25 * k * NOISE_REFERENCE)) < 1e-4
    eng.close()


@pytest.mark.parametrize("c960", [False, True])
def test_many_random_positions_movegen_planes_and_first_expansions(c960):
    """breadth instead of depth: 640 positions reached by random playouts of up to 150 plies (captures, promotions, castling, en passant,
    repetition windows, long games), each searched for 3 simulations in lock-step with the oracle: root and first leaves' planes, legal
    masks, child order, priors and value sums bit-exact"""
    rng = random.Random(77 + int(c960))
    boards = [Mirror(c960=c960, scharnagl=rng.randrange(960) if c960 else 518, pre_moves=rng.randrange(0, 150), rng=rng) for _ in range(640)]
    st = lockstep_search(boards, 3, True, random_evaluator(5 + int(c960)), c960=c960)
    assert st["expansions"] + st["terminal_hits"] >= 640


def test_long_games_at_scale_finish_cleanly():
    """stress: 768 Chess960 games played to the end (or 700 plies) with the MFMA network and 12 searches per move — history ring wrap-around
    (> 256 plies), 75-move / repetition / material draws and mates at scale; every record stays well-formed and no board reports an error"""
    from sigma_zero_amd.fastnet import FastPolicyNet
    from sigma_zero_amd.sim import play_games
    torch.manual_seed(1)
    np.random.seed(7)
    random.seed(7)
    fast = FastPolicyNet(sz.policyNN({}).cuda().eval())
    games = play_games(fast, {"C": 2, "num_searches": 12}, 768, c960=True, max_plies=700)
    finished = [g for g in games if g["result"] is not None]
    assert len(finished) >= 0.5 * len(games)
    longest = max(len(g["actions"]) for g in games)
    assert longest > 256, longest                                   # the ring wrapped for some game
    for g in games:
        n = len(g["actions"])
        assert n == len(g["states"]) == len(g["colours"]) == len(g["rewards"]) and n >= 1
        assert all(abs(sum(a.values()) - 1.0) < 1e-6 and len(a) >= 1 for a in g["actions"][::37])
        assert all(c0 != c1 for c0, c1 in zip(g["colours"], g["colours"][1:]))       # colours alternate
        if g["result"] == "1/2-1/2":
            assert set(g["rewards"]) == {0}
        elif g["result"] is not None:
            assert set(g["rewards"]) == {1, -1} or n == 1
    results = {r: sum(1 for g in finished if g["result"] == r) for r in ("1-0", "0-1", "1/2-1/2")}
    assert sum(results.values()) == len(finished)
