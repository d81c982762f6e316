"""GPU parity against fixtures produced by the REFERENCE's own chess_tensor.py / mcts.py / mctsnode.py / sim.py (run unmodified in the
build container, tests/golden/gen_reference_chess_fixtures.py and gen_reference_fixtures.py).  Everything goes through the C ABI
(ctypes) of libsigmazero_hip.so; the oracle is not involved here: HIP engine vs reference output directly."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch

import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
from sigma_zero_amd.selfplay import SelfPlayEngine
from hashmodel import HashModel, evaluate_packed, pack_planes

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def _cases(z):
    for i in range(int(z["n_cases"])):
        yield i, {k[len("c%d_" % i):]: z[k] for k in z if k.startswith("c%d_" % i)}


@pytest.fixture(scope="module")
def games(golden_dir):
    return _load(golden_dir, "chess_tensor_games.npz")


def _host_game(z, g, upto=None):
    ct = sz.ChessTensor(chess960=bool(z["c960"][g]), scharnagl=int(z["scharnagl"][g]) if z["c960"][g] else None)
    lo, hi = z["move_off"][g], z["move_off"][g + 1]
    hi = hi if upto is None else lo + upto
    for f, t, p in z["moves"][lo:hi]:
        ct.move_piece(sz.Move(int(f), int(t), int(p) or None))
    return ct


# ------------------------------------------------------------------------------------------------ A. encode + movegen kernels on every position
@pytest.mark.parametrize("c960", [0, 1])
def test_engine_root_planes_and_legal_masks_match_reference_every_ply(games, c960):
    """every position of the reference-played games is uploaded as a live game (history ring included); k_search_begin's root planes
    (chess_tensor.py:131-142) and legal mask (actionsToTensor of the legal moves) must equal what the reference produced"""
    z = games
    snaps = []
    for g in range(int(z["n_games"])):
        if int(z["c960"][g]) != c960:
            continue
        ct = sz.ChessTensor(chess960=bool(c960), scharnagl=int(z["scharnagl"][g]) if c960 else None)
        mlo = z["move_off"][g]
        for k, s in enumerate(range(z["snap_off"][g], z["snap_off"][g + 1])):
            if k:
                f, t, p = z["moves"][mlo + k - 1]
                ct.move_piece(sz.Move(int(f), int(t), int(p) or None))
            if not z["term"][s]:
                snaps.append((s, ct.export_ring()))
    B = len(snaps)
    assert B > 1000
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": 2}, B, chess960=bool(c960), learning=False, edges_per_board=512)
    for b, (s, (ring, ply, _)) in enumerate(snaps):
        N.check(N.lib().sz_upload_game(eng._e, b, ring, int(ply), eng._stream()), "sz_upload_game")
    eng.begin()
    torch.cuda.synchronize()
    mask, depth, n_nodes, n_edges, status = eng.debug_pending()
    planes = pack_planes(eng.planes.cpu().numpy())
    for b, (s, _) in enumerate(snaps):
        assert status[b] & 2, "position %d should wait for the network" % s
        assert np.array_equal(planes[b], z["rep"][s]), "planes of fixture position %d" % s
        mine = [p * 64 + v for p in range(73) for v in range(64) if (int(mask[b, p]) >> v) & 1]
        assert mine == z["idx"][z["idx_off"][s]:z["idx_off"][s + 1]].tolist(), "legal mask of fixture position %d" % s
    eng.close()


# ------------------------------------------------------------------------------------------------ B. whole searches vs the reference's trees
def _engine_search(z, case, dtype=torch.float32):
    ct = _host_game(z, int(case["game"]), int(case["ply"]))
    S = int(case["S"])
    eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, 1, chess960=bool(z["c960"][int(case["game"])]), learning=bool(case["learning"]),
                         planes_dtype=dtype)
    eng.upload_game(0, ct)
    eng.begin()
    k = 0
    for step in range(S + 1):
        torch.cuda.synchronize()
        _, _, _, _, status = eng.debug_pending()
        if not (status[0] & 2):
            break
        planes = pack_planes(eng.planes.float().cpu().numpy())[0]
        assert k < len(case["leaf_planes"]) and np.array_equal(planes, case["leaf_planes"][k]), "network input %d differs from the reference's" % k
        pol, val = evaluate_packed(planes, str(case["mode"]), int(case["salt"]))
        eng.step(torch.from_numpy(pol).cuda().unsqueeze(0).contiguous(), torch.tensor([val], dtype=torch.float32, device="cuda"))
        k += 1
    assert k == len(case["leaf_planes"])
    st = eng.stats()
    assert st["boards_error"] == 0 or int(case["S"]) == 1
    assert st["expansions"] == k and st["terminal_hits"] == S - k
    tree = eng.debug_tree(0)
    rc = eng.root_children()
    eng.close()
    return tree, rc, ct


def test_engine_search_trees_match_reference_on_real_positions(games, golden_dir):
    tr = _load(golden_dir, "chess_search_traces.npz")
    n_exact = n_tol = 0
    for i, case in _cases(tr):
        tag = "case %d game %d ply %d S=%d learning=%d %s" % (i, case["game"], case["ply"], case["S"], case["learning"], case["mode"])
        (d, a, v, w, p), (action, visits, n_child, prior, wsum), ct = _engine_search(games, case)
        # row 0 is the root; the reference dump starts at its children
        assert int(v[0]) == int(case["root_visits"]), tag
        assert np.array_equal(d[1:], case["tree_depth"]) and np.array_equal(a[1:], case["tree_action"]), tag
        assert np.array_equal(v[1:], case["tree_visits"]), tag
        if str(case["mode"]) == "dyadic":
            assert np.array_equal(p[1:].view(np.uint32), case["tree_prior"].view(np.uint32)), tag
            assert np.array_equal(w[1:], case["tree_value_sum"]) and w[0] == float(case["root_value_sum"]), tag
            n_exact += 1
        else:
            assert np.allclose(p[1:], case["tree_prior"], rtol=3e-7, atol=0), tag          # torch.sum order: <= 1 ulp on the normaliser
            assert np.allclose(w[1:], case["tree_value_sum"], rtol=0, atol=1e-12), tag
            n_tol += 1
        k = int(n_child[0])
        if str(case["error"]):
            assert int(visits[0, :k].sum()) == 0, tag
            continue
        assert action[0, :k].tolist() == case["root_actions"].tolist(), tag
        probs = visits[0, :k].astype(np.float64) / int(visits[0, :k].sum())
        assert np.abs(probs - case["root_probs"]).max() <= 1e-4 and np.array_equal(probs, case["root_probs"]), tag
        moves = [ct.move_from_index(int(x)) for x in action[0, :k]]
        assert [(m.from_square, m.to_square, m.promotion or 0) for m in moves] == [tuple(int(x) for x in r) for r in case["root_moves"]], tag
    assert n_exact >= 30 and n_tol >= 8


def test_mcts0_api_and_node_view_match_reference(games, golden_dir):
    """MCTS0(game, args, model).search(board, verbose, learning) -> {Move: fraction}, and the Node view of the finished tree
    (mctsnode.py:7-18 fields) against the reference's tree"""
    tr = _load(golden_dir, "chess_search_traces.npz")
    n = 0
    for i, case in _cases(tr):
        if int(case["S"]) > 120 or i % 3:
            continue
        ct = _host_game(games, int(case["game"]), int(case["ply"]))
        model = HashModel(mode=str(case["mode"]), salt=int(case["salt"]))
        m = sz.MCTS0(game=ct, args={"C": 2, "num_searches": int(case["S"])}, model=model)
        if str(case["error"]):
            with pytest.raises(ZeroDivisionError):
                m.search(ct.board, verbose=False, learning=bool(case["learning"]))
            continue
        probs = m.search(ct.board, verbose=False, learning=bool(case["learning"]))
        assert [(k.from_square, k.to_square, k.promotion or 0) for k in probs] == [tuple(int(x) for x in r) for r in case["root_moves"]]
        assert np.abs(np.array(list(probs.values())) - case["root_probs"]).max() <= 1e-4
        root = m.root
        assert isinstance(root, sz.Node) and root.parent is None and root.game is not None and root.color == ct.board.turn
        assert root.visit_count == int(case["root_visits"])
        rows = []

        def walk(node, depth):
            for ch in node.children:
                assert ch.parent is node and ch.color == (not node.color)
                assert (ch.game is not None) == (ch.visit_count > 0)          # visited nodes own a position (mcts.py:57-59)
                rows.append((depth, ch.action_index, ch.visit_count, ch.value_sum, ch.prior, ch.action_taken.from_square, ch.action_taken.to_square,
                             ch.action_taken.promotion or 0))
                walk(ch, depth + 1)
        walk(root, 0)
        assert [r[0] for r in rows] == case["tree_depth"].tolist() and [r[1] for r in rows] == case["tree_action"].tolist()
        assert [r[2] for r in rows] == case["tree_visits"].tolist()
        assert [tuple(r[5:8]) for r in rows] == [tuple(int(x) for x in r) for r in case["tree_moves"]]
        if str(case["mode"]) == "dyadic":
            assert np.array_equal(np.array([r[3] for r in rows]), case["tree_value_sum"])
            assert np.array_equal(np.array([r[4] for r in rows], np.float32).view(np.uint32), case["tree_prior"].view(np.uint32))
        # the view's own select() (reference arithmetic on the Node objects) picks the child the engine would descend into next
        if root.children:
            sel = root.select()
            assert sel in root.children
        with pytest.raises(ValueError):
            class _Wrong:
                turn = not ct.board.turn
            m.search(_Wrong(), verbose=False)
        n += 1
    assert n >= 8


# ------------------------------------------------------------------------------------------------ C. play_game / generate_training_data
def _check_history(h, case, lo=0):
    n = len(h["actions"])
    for k in range(n):
        tag = "sample %d" % (lo + k)
        st = h["states"][k]
        assert st.dtype == torch.bool and tuple(st.shape) == (119, 8, 8)
        assert np.array_equal(pack_planes(st.numpy()), case["states"][lo + k]), tag
        a, b = case["act_off"][lo + k], case["act_off"][lo + k + 1]
        assert [(m.from_square, m.to_square, m.promotion or 0) for m in h["actions"][k]] == [tuple(int(x) for x in r) for r in case["act_moves"][a:b]], tag
        assert list(h["actions"][k].values()) == case["act_probs"][a:b].tolist(), tag
        assert int(bool(h["colours"][k])) == int(case["colours"][lo + k]) and h["rewards"][k] == int(case["rewards"][lo + k]), tag
    return n


def test_play_game_matches_reference_records(golden_dir):
    """sz.play_game(model, args, c960) under the same random / np.random seeds as the reference run: same start position, same search
    results, same sampled moves, same records and rewards, ply for ply to the end of the game"""
    z = _load(golden_dir, "chess_play_records.npz")
    n = 0
    for i, case in _cases(z):
        if str(case["kind"]) != "play_game":
            continue
        seed = int(case["seed"])
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        model = HashModel(mode=str(case["mode"]), salt=int(case["salt"]))
        h = sz.play_game(model, {"C": 2, "num_searches": int(case["S"])}, c960=bool(case["c960"]))
        assert set(h.keys()) == {"states", "actions", "rewards", "colours"}
        assert len(h["actions"]) == len(case["rewards"]), "case %d: game length" % i
        _check_history(h, case)
        n += 1
    assert n >= 6


def test_generate_training_data_matches_reference_records(golden_dir):
    """sim.generate_training_data (sim.py:102-123): key-wise concatenation over the games and return_dict[os.getpid()]"""
    z = _load(golden_dir, "chess_play_records.npz")
    case = [c for _, c in _cases(z) if str(c["kind"]) == "generate_training_data"][0]
    seed = int(case["seed"])
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    model = HashModel(mode=str(case["mode"]), salt=int(case["salt"]))
    rd = {}
    h = sz.generate_training_data(model, int(case["num_games"]), {"C": 2, "num_searches": int(case["S"])}, rd, bool(case["c960"]), rng_order="reference")
    assert list(rd.keys()) == [os.getpid()] and rd[os.getpid()] is h
    assert len(h["actions"]) == len(case["rewards"])
    _check_history(h, case)
    # the default (all games concurrently) has the same layout and plays the same number of games
    random.seed(seed); np.random.seed(seed)
    rd2 = {}
    h2 = sz.generate_training_data(model, 3, {"C": 2, "num_searches": 4}, rd2, True)
    assert rd2[os.getpid()] is h2 and len(h2["states"]) == len(h2["actions"]) == len(h2["rewards"]) == len(h2["colours"]) > 0
    assert sum(1 for k in range(1, len(h2["states"])) if not h2["states"][k][113].any()) == 2          # plane 113 (any move played) is 0 only at a game's first sample


# ------------------------------------------------------------------------------------------------ D. the device's Node.select on the reference UCB vectors
def test_device_select_on_reference_ucb_vectors_bitwise(golden_dir):
    """3,000 cases / 90,009 children produced by the reference's Node.get_ucb / Node.select: the DEVICE function the search runs
    (wave_select_child -> ucb_value) must give the same float32 bits and the same argmax"""
    z = _load(golden_dir, "ucb_vectors.npz")
    dev = "cuda"
    off = torch.from_numpy(z["offsets"].astype(np.int32)).to(dev)
    vc = torch.from_numpy(z["vc"].astype(np.int32)).to(dev)
    ws = torch.from_numpy(z["vsum"].astype(np.float64)).to(dev)
    pr = torch.from_numpy(z["prior"].astype(np.float32)).to(dev)
    pv = torch.from_numpy(z["parent_visits"].astype(np.int32)).to(dev)
    cc = torch.from_numpy(z["C"].astype(np.float32)).to(dev)
    n_cases = len(z["argmax"])
    ucb = torch.zeros(len(z["vc"]), dtype=torch.float32, device=dev)
    arg = torch.zeros(n_cases, dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    N.check(N.lib().sz_debug_select(P(off), P(vc), P(ws), P(pr), P(pv), P(cc), P(ucb), P(arg), n_cases,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sz_debug_select")
    torch.cuda.synchronize()
    got = ucb.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), z["ucb"].view(np.uint32))
    assert np.array_equal(arg.cpu().numpy().astype(np.int64), z["argmax"])


# ------------------------------------------------------------------------------------------------ E. the north-star statement: reference CPU path vs the GPU engine
def test_gpu_engine_with_fp32_network_matches_reference_cpu_path(games, golden_dir):
    """Fixture: the reference's mcts.py + the reference's policyNN (seed 0, fp32, CPU, batch 1).  Here: MCTS0 of the product = HIP engine +
    the product's policyNN on the GPU (fp32 through MIOpen).  Move indices must be exact; visit fractions within 1e-4 (BASELINE.json north_star)."""
    z = _load(golden_dir, "chess_real_network_searches.npz")
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    worst, n_exact, n = 0.0, 0, 0
    for i, case in _cases(z):
        ct = _host_game(games, int(case["game"]), int(case["ply"]))
        with torch.no_grad():
            p0, v0 = net(ct.get_representation().float().unsqueeze(0).cuda(), inference=True)
        assert np.allclose(p0[0].cpu().numpy(), case["root_policy"], rtol=1e-3, atol=1e-8) and abs(float(v0) - float(case["root_value"])) < 1e-4
        m = sz.MCTS0(game=ct, args={"C": 2, "num_searches": int(case["S"])}, model=net)
        probs = m.search(ct.board, verbose=False, learning=bool(case["learning"]))
        assert [(k.from_square, k.to_square, k.promotion or 0) for k in probs] == [tuple(int(x) for x in r) for r in case["root_moves"]], "case %d: moves" % i
        d = float(np.abs(np.array(list(probs.values())) - case["root_probs"]).max())
        worst = max(worst, d)
        n_exact += int(d == 0.0)
        n += 1
        assert d <= 1e-4, "case %d: visit fractions differ by %.3g (one visit = %.3g)" % (i, d, 1.0 / (int(case["S"]) - 1))
    print("reference CPU path vs GPU engine + fp32 network: %d/%d searches with identical visit counts, worst |delta fraction| %.2e" % (n_exact, n, worst))
    assert n >= 14


def test_config0_whole_game_matches_reference(golden_dir):
    """BASELINE.json configs[0] — the reference's own runnable case (`python sim.py`, sim.py:125-137: 1 self-play game, num_searches=10,
    Chess960, random-init policyNN; the reference ran it on the CPU in fp32 with seed 0): the product plays the SAME game on the GPU (HIP engine +
    fp32 policyNN), all 361 plies: start position, every state, every visit distribution, every sampled move, rewards."""
    case = _load(golden_dir, "chess_config0_game.npz")
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    h = sz.generate_training_data(net, 1, {"C": 2, "num_searches": 10}, None, True)
    n_ref = len(case["rewards"])
    # report how far the two games agree before asserting (a single flipped visit would change a sampled move sooner or later)
    agree = 0
    for k in range(min(len(h["actions"]), n_ref)):
        a, b = case["act_off"][k], case["act_off"][k + 1]
        if not np.array_equal(pack_planes(h["states"][k].numpy()), case["states"][k]) or list(h["actions"][k].values()) != case["act_probs"][a:b].tolist():
            break
        agree += 1
    print("configs[0] game: %d of %d plies identical to the reference's CPU run" % (agree, n_ref))
    assert len(h["actions"]) == n_ref and agree == n_ref
    _check_history(h, case)


def test_config0_whole_game_with_split_precision_network(golden_dir):
    """the same configs[0] game with the network on the matrix cores at the reference's precision class (SplitPolicyNet): how many plies of the
    reference's fp32 CPU game it reproduces (a single visit that falls the other way changes a sampled move sooner or later)"""
    from sigma_zero_amd.fastnet import SplitPolicyNet
    case = _load(golden_dir, "chess_config0_game.npz")
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    net = SplitPolicyNet(sz.policyNN({}).cuda().eval())
    h = sz.generate_training_data(net, 1, {"C": 2, "num_searches": 10}, None, True)
    n_ref = len(case["rewards"])
    agree = 0
    for k in range(min(len(h["actions"]), n_ref)):
        a, b = case["act_off"][k], case["act_off"][k + 1]
        if not np.array_equal(pack_planes(h["states"][k].numpy()), case["states"][k]) or list(h["actions"][k].values()) != case["act_probs"][a:b].tolist():
            break
        agree += 1
    print("configs[0] game with the split-precision network: %d of %d plies identical to the reference's CPU run" % (agree, n_ref))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "config0_split_agreement.txt"), "w") as f:
        f.write("%d of %d plies identical\n" % (agree, n_ref))
    assert len(h["actions"]) == n_ref and agree == n_ref            # measured: 361 of 361 (deterministic: same kernel, same order on every MI355X)


@pytest.mark.parametrize("operands", ["fp16", "bf16"])
def test_config0_whole_game_with_the_fast_networks(golden_dir, operands):
    """the same configs[0] game with the fast inference network (FastPolicyNet on f16 / bf16 operands): how many plies of the reference's fp32 CPU game it
    reproduces before a visit falls the other way (a record, written to gpurun_out/; the game is one chain of 361 searches: the first differing visit
    distribution changes everything after it, so this is a much harsher measure than the per-position agreement of test_gpu_train_and_precision.py)"""
    from sigma_zero_amd.fastnet import FastPolicyNet
    case = _load(golden_dir, "chess_config0_game.npz")
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    net = FastPolicyNet(sz.policyNN({}).cuda().eval(), operands=operands)
    h = sz.generate_training_data(net, 1, {"C": 2, "num_searches": 10}, None, True)
    n_ref = len(case["rewards"])
    agree = 0
    for k in range(min(len(h["actions"]), n_ref)):
        a, b = case["act_off"][k], case["act_off"][k + 1]
        if not np.array_equal(pack_planes(h["states"][k].numpy()), case["states"][k]) or list(h["actions"][k].values()) != case["act_probs"][a:b].tolist():
            break
        agree += 1
    print("configs[0] game with %s operands: %d of %d plies identical to the reference's CPU run" % (operands, agree, n_ref))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "config0_%s_agreement.txt" % operands), "w") as f:
        f.write("%d of %d plies identical\n" % (agree, n_ref))
    assert agree >= 1 and np.array_equal(pack_planes(h["states"][0].numpy()), case["states"][0])        # same start position (Chess960 index 864), same first search at least
