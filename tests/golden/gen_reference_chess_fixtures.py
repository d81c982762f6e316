"""Golden vectors for ChessTensor, real-chess MCTS0.search and sim.play_game, produced by running the REFERENCE's own
source UNMODIFIED in the build container:

    /root/reference/chess_tensor.py   ChessTensor (:30-188) + the codecs
    /root/reference/mcts.py           MCTS0.search (:39-122)        /root/reference/mctsnode.py  Node
    /root/reference/sim.py            play_game (:31-99), generate_training_data (:102-123)

    python tests/golden/gen_reference_chess_fixtures.py        # writes tests/golden/chess_*.npz

python-chess is absent from this image, so `import chess` resolves to tests/golden/_chess_stub/chess.py: a Board adapter whose
RULES are oracle/oc_chess.c (perft-pinned; see that file's docstring).  Everything the reference does on top of the rules is
the reference's code executing: plane stacks, history shift, views and flips, saturating counters, mask/codec use in the
search, deepcopy + move_piece per leaf, Dirichlet mix, back-up, root readout, np.random.choice sampling, records, rewards.
The network is tests/hashmodel.HashModel, an integer-hash function of the planes that the tests recompute.

The fixtures are data only (inputs + expected outputs); /root/reference does not exist on the GPU box.
"""
import io
import contextlib
import os
import sys
import random
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "_chess_stub"))
sys.path.insert(1, "/root/reference")
sys.path.append(ROOT)
sys.path.append(os.path.join(ROOT, "tests"))

import numpy as np
import torch

import chess                      # the adapter (rules = oracle/oc_chess.c)
import chess_tensor as ref_ct     # reference, unmodified
import mctsnode as ref_node       # reference, unmodified
import mcts as ref_mcts           # reference, unmodified
import sim as ref_sim             # reference, unmodified
from hashmodel import HashModel, pack_planes

assert ref_ct.__file__.startswith("/root/reference/") and ref_sim.__file__.startswith("/root/reference/")
assert ref_mcts.device == "cpu"
torch.set_num_threads(1)
OUT = HERE


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def new_game(c960, scharnagl):
    """ChessTensor(chess960) exactly as the reference constructs it; the Scharnagl index is what its random.randint(0, 959) draws"""
    if c960:
        state = random.getstate()
        # make the reference's own `random.randint(0, 959)` (chess_tensor.py:69) return `scharnagl`
        orig = ref_ct.random.randint
        ref_ct.random.randint = lambda a, b: scharnagl
        try:
            with quiet():
                g = ref_ct.ChessTensor(chess960=True)
        finally:
            ref_ct.random.randint = orig
            random.setstate(state)
        return g
    return ref_ct.ChessTensor()


# --------------------------------------------------------------------------- A. ChessTensor along whole games
SCRIPTED = [
    # (chess960 start or -1, uci moves)
    (-1, "e2e4 e7e5 g1f3 b8c6 f1c4 f8c5 e1g1 g8f6 d2d3 e8g8 c1g5 h7h6 g5h4 d7d6 b1c3 c8g4".split()),            # both sides castle short
    (-1, "d2d4 d7d5 b1c3 b8c6 c1f4 c8f5 d1d2 d8d7 e1c1 e8c8 g1f3 g8f6".split()),                                # both castle long
    (-1, "e2e4 a7a6 e4e5 d7d5 e5d6 c7d6 d2d4 b7b5 d4d5 e7e5 d5e6 f7e6 g1f3 b5b4 c2c4 b4c3".split()),           # three en-passant captures
    (-1, "g1f3 g8f6 f3g1 f6g8 g1f3 g8f6 f3g1 f6g8 g1f3 g8f6 f3g1 f6g8 g1f3 g8f6 f3g1 f6g8".split()),           # repetition planes up to fivefold (game over)
    (-1, "a2a4 b7b5 a4b5 a7a6 b5a6 c8b7 a6b7 b8c6 b7a8n g7g5 h2h4 g5h4 g2g3 h4g3 f1h3 g3g2 e2e3 g2h1q".split()),  # capture-promotions N and q
    (-1, "h2h4 g7g5 h4g5 h7h6 g5h6 f8g7 h6g7 g8f6 g7h8r f6g8 b2b4 a7a5 g1f3 a5b4 a2a3 b4a3 f3g1 a3a2 g1f3 a2b1b".split()),  # promotions r and b (both colours)
    (-1, "e2e4 e7e5 e1e2 e8e7 e2e1 e7e8 g1f3 g8f6 h1g1 h8g8 g1h1 g8h8 a2a3 a7a6".split()),                      # castling rights lost by king / rook moves
    (-1, "f2f3 e7e5 g2g4 d8h4".split()),                                                                         # fool's mate: terminal, white mated
    (-1, "e2e4 e7e5 d1h5 b8c6 f1c4 g8f6 h5f7".split()),                                                          # scholar's mate: black mated
    (518, "e2e4 e7e5 g1f3 b8c6 f1c4 f8c5 e1h1 g8f6 d2d3 e8h8".split()),                                          # the classical array as a Chess960 board: castling spelled king-takes-rook
]


def pick_move(rng, game, prev_own, style):
    """biased random choice that makes the rare move types common; legality is entirely the board's"""
    board = game.board
    legal = list(board.legal_moves)
    castles, eps, promos, undo, pawns = [], [], [], [], []
    for m in legal:
        p = board.piece_at(m.from_square)
        tgt = board.piece_at(m.to_square)
        if p.piece_type == chess.KING and ((tgt is not None and tgt.color == p.color) or abs(m.to_square % 8 - m.from_square % 8) > 1):
            castles.append(m)
        if p.piece_type == chess.PAWN:
            pawns.append(m)
            if m.promotion:
                promos.append(m)
            elif (m.to_square % 8) != (m.from_square % 8) and tgt is None:
                eps.append(m)
        if prev_own is not None and m.from_square == prev_own.to_square and m.to_square == prev_own.from_square and not m.promotion:
            undo.append(m)
    r = rng.random()
    if castles and r < 0.6:
        return rng.choice(castles)
    if eps and r < 0.8:
        return rng.choice(eps)
    if promos and r < 0.85:
        return rng.choice(promos)
    if undo and rng.random() < style.get("undo", 0.1):
        return undo[0]
    if pawns and rng.random() < style.get("pawn", 0.0):
        return rng.choice(pawns)
    return rng.choice(legal)


def snapshot(game):
    """everything the reference exposes about the current position"""
    board = game.board
    rep = game.get_representation()
    assert rep.dtype == torch.bool and tuple(rep.shape) == (119, 8, 8)
    value, term = game.get_value_and_terminated()
    turn = bool(board.turn)
    legal = game.get_valid_moves(board)
    mask, qp = ref_ct.actionsToTensor(legal, turn) if legal else (torch.zeros(4672), {})
    idx = mask.nonzero().flatten().tolist()
    assert len(idx) == len(legal), "two legal moves share an action index"
    back = ref_ct.tensorToAction(mask, turn, queen_promotion=qp) if legal else []
    return dict(rep=pack_planes(rep.numpy()), white_stack=pack_planes(game.representation.numpy()),
                black_stack=pack_planes(game.black_representation.numpy()), value=int(value), term=int(term), turn=int(turn),
                idx=np.array(idx, np.int32), back=np.array([(m.from_square, m.to_square, m.promotion or 0) for m in back], np.int32).reshape(-1, 3))


def gen_games():
    games = []
    stats = dict(castle=0, ep=0, promo={2: 0, 3: 0, 4: 0, 5: 0}, rep2=0, rep3=0, over=0, black_views=0)

    def run(c960, sch, mover, max_plies):
        game = new_game(c960, sch)
        snaps, moves = [snapshot(game)], []
        prev = {True: None, False: None}
        for ply in range(max_plies):
            if game.board.is_game_over():
                break
            mv = mover(game, ply, prev[game.board.turn])
            if mv is None:
                break
            board = game.board
            p, tgt = board.piece_at(mv.from_square), board.piece_at(mv.to_square)
            if p.piece_type == chess.KING and ((tgt is not None and tgt.color == p.color) or abs(mv.to_square % 8 - mv.from_square % 8) > 1):
                stats["castle"] += 1
            if p.piece_type == chess.PAWN and (mv.to_square % 8) != (mv.from_square % 8) and tgt is None:
                stats["ep"] += 1
            if mv.promotion:
                stats["promo"][mv.promotion] += 1
            prev[board.turn] = mv
            game.move_piece(mv)                                    # the reference's move_piece (raises ValueError if illegal)
            moves.append((mv.from_square, mv.to_square, mv.promotion or 0))
            snaps.append(snapshot(game))
            stats["rep2"] += int(snaps[-1]["white_stack"][12].any())
            stats["rep3"] += int(snaps[-1]["white_stack"][13].any())
            stats["black_views"] += int(not snaps[-1]["turn"])
        stats["over"] += int(game.board.is_game_over())
        res = game.board.result()
        games.append(dict(c960=int(c960), scharnagl=int(sch), moves=np.array(moves, np.int32).reshape(-1, 3), snaps=snaps,
                          result={"1-0": 1, "0-1": -1, "1/2-1/2": 0, "*": 2}[res]))

    for sch, ucis in SCRIPTED:
        it = iter(ucis)
        run(sch >= 0, sch if sch >= 0 else -1, lambda g, ply, prev: (lambda u: chess.Move.from_uci(u) if u else None)(next(it, None)), len(ucis) + 1)
    plan = [(False, -1, dict(undo=0.10, pawn=0.0), 400), (False, -1, dict(undo=0.45, pawn=0.0), 260), (False, -1, dict(undo=0.02, pawn=0.5), 400),
            (False, -1, dict(undo=0.05, pawn=0.25), 400), (False, -1, dict(undo=0.3, pawn=0.3), 120), (False, -1, dict(undo=0.0, pawn=0.6), 500)]
    rs = random.Random(960)
    for k in range(8):
        plan.append((True, rs.randrange(960), dict(undo=[0.05, 0.3, 0.1, 0.0][k % 4], pawn=[0.0, 0.2, 0.5, 0.3][k % 4]), [300, 150, 400, 500][k % 4]))
    plan += [(True, 0, dict(undo=0.1, pawn=0.2), 80), (True, 959, dict(undo=0.1, pawn=0.2), 80)]
    for gi, (c960, sch, style, cap) in enumerate(plan):
        rng = random.Random(1000 + gi)
        run(c960, sch, lambda g, ply, prev, rng=rng, style=style: pick_move(rng, g, prev, style), cap)
    # pack (ragged -> offsets)
    flat = dict(n_games=np.int64(len(games)))
    flat["c960"] = np.array([g["c960"] for g in games], np.int32)
    flat["scharnagl"] = np.array([g["scharnagl"] for g in games], np.int32)
    flat["result"] = np.array([g["result"] for g in games], np.int32)
    flat["move_off"] = np.cumsum([0] + [len(g["moves"]) for g in games]).astype(np.int64)
    flat["moves"] = np.concatenate([g["moves"] for g in games]).astype(np.int8)
    snaps = [s for g in games for s in g["snaps"]]
    flat["snap_off"] = np.cumsum([0] + [len(g["snaps"]) for g in games]).astype(np.int64)
    for key in ("rep", "white_stack", "black_stack"):
        flat[key] = np.stack([s[key] for s in snaps])
    for key in ("value", "term", "turn"):
        flat[key] = np.array([s[key] for s in snaps], np.int8)
    flat["idx_off"] = np.cumsum([0] + [len(s["idx"]) for s in snaps]).astype(np.int64)
    flat["idx"] = np.concatenate([s["idx"] for s in snaps]).astype(np.int16)
    flat["back"] = np.concatenate([s["back"] for s in snaps]).astype(np.int8)
    np.savez_compressed(os.path.join(OUT, "chess_tensor_games.npz"), **flat)
    print("chess_tensor_games: %d games, %d positions; stats %s" % (len(games), len(snaps), stats))
    assert stats["castle"] >= 12 and stats["ep"] >= 5 and all(v >= 2 for v in stats["promo"].values()) and stats["rep3"] >= 5 and stats["over"] >= 6
    return games


# --------------------------------------------------------------------------- B. MCTS0.search on real positions
class RecordingNode(ref_node.Node):
    created = []

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        RecordingNode.created.append(self)


def dump_tree(root, color_root):
    rows = []

    def rec(node, depth, parent_color):
        for ch in node.children:
            idx = ref_ct.actionToTensor(ch.action_taken, parent_color).nonzero().item()
            rows.append((depth, idx, ch.visit_count, ch.value_sum, np.float32(ch.prior), ch.action_taken.from_square, ch.action_taken.to_square,
                         ch.action_taken.promotion or 0))
            rec(ch, depth + 1, not parent_color)
    rec(root, 0, color_root)
    return rows


def replay(games, gi, ply):
    g = games[gi]
    game = new_game(bool(g["c960"]), g["scharnagl"])
    for (f, t, p) in g["moves"][:ply]:
        game.move_piece(chess.Move(int(f), int(t), int(p) or None))
    return game


def gen_search(games):
    ref_mcts.Node = RecordingNode                # same class, only remembers the objects it creates (to dump the tree afterwards)
    n_scripted = len(SCRIPTED)
    spec = []
    rs = random.Random(77)
    # (game, ply, S, learning, mode)
    spec += [(0, 0, 2, False, "dyadic"), (0, 0, 16, True, "dyadic"), (0, 0, 60, False, "dyadic"), (0, 0, 200, True, "dyadic"),
             (0, 1, 40, True, "dyadic"), (0, 9, 60, True, "dyadic"), (1, 8, 60, False, "dyadic"), (2, 3, 50, True, "dyadic"),
             (3, 11, 80, True, "dyadic"), (3, 14, 80, False, "dyadic"),       # repetition in the game history AND reachable in the tree (fivefold terminal)
             (4, 8, 60, True, "dyadic"), (4, 17, 60, True, "dyadic"), (5, 17, 60, False, "dyadic"),   # promotions among the children (queen_promotion dict path)
             (7, 3, 30, True, "dyadic"), (8, 6, 30, True, "dyadic"),          # mate in one in the tree: terminal leaves, value -1
             (9, 6, 60, True, "dyadic"), (9, 9, 60, True, "dyadic"),          # Chess960 castling among the children
             (0, 0, 1, False, "dyadic")]                                       # num_searches == 1 -> ZeroDivisionError
    for k in range(14):                                                        # random positions from the random games, both colours, all phases
        gi = n_scripted + rs.randrange(len(games) - n_scripted)
        n = len(games[gi]["moves"])
        ply = rs.randrange(0, max(1, n - 1))
        spec.append((gi, ply, [24, 60, 120][k % 3], bool(k % 2), "dyadic"))
    for k in range(8):                                                         # tolerance cases: sums that round
        gi = rs.randrange(len(games))
        n = len(games[gi]["moves"])
        spec.append((gi, rs.randrange(0, max(1, n - 1)), [60, 200][k % 2], bool(k % 2), "rational"))
    spec.append((0, 0, 800, True, "dyadic"))                                   # the AlphaZero default budget once
    cases = []
    for ci, (gi, ply, S, learning, mode) in enumerate(spec):
        game = replay(games, gi, ply)
        if game.board.is_game_over():
            ply = max(0, ply - 1)
            game = replay(games, gi, ply)
        model = HashModel(mode=mode, salt=ci, record=True)
        RecordingNode.created = []
        torch.manual_seed(ci)
        engine = ref_mcts.MCTS0(game=game, args={"C": 2, "num_searches": S}, model=model)
        err = ""
        t0 = time.time()
        try:
            probs = engine.search(game.board, verbose=False, learning=learning)
        except ZeroDivisionError:
            probs, err = {}, "ZeroDivisionError"
        root = RecordingNode.created[0]
        tree = dump_tree(root, game.board.turn)
        priors = np.array([r[4] for r in tree], np.float32)
        assert not np.isnan(priors).any(), "NaN prior (all legal probabilities were 0): pick another salt"
        root_actions = [ref_ct.actionToTensor(m, game.board.turn).nonzero().item() for m in probs.keys()]
        n_term = S - len(model.calls)
        cases.append(dict(game=gi, ply=ply, S=S, learning=int(learning), mode=mode, salt=ci, error=err,
                          leaf_planes=np.stack(model.calls) if model.calls else np.zeros((0, 119, 8), np.uint8),
                          root_actions=np.array(root_actions, np.int32), root_probs=np.array(list(probs.values()), np.float64),
                          root_moves=np.array([(m.from_square, m.to_square, m.promotion or 0) for m in probs.keys()], np.int32).reshape(-1, 3),
                          root_visits=np.int64(root.visit_count), root_value_sum=np.float64(root.value_sum),
                          tree_depth=np.array([r[0] for r in tree], np.int32), tree_action=np.array([r[1] for r in tree], np.int32),
                          tree_visits=np.array([r[2] for r in tree], np.int64), tree_value_sum=np.array([r[3] for r in tree], np.float64),
                          tree_prior=priors, tree_moves=np.array([r[5:8] for r in tree], np.int32).reshape(-1, 3)))
        print("search case %2d: game %2d ply %3d S=%3d learning=%d %-8s -> %3d expansions, %3d terminal visits, %5d nodes, %.1fs %s"
              % (ci, gi, ply, S, learning, mode, len(model.calls), n_term, len(tree), time.time() - t0, err))
    flat = {"n_cases": np.int64(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat["c%d_%s" % (i, k)] = np.array(v) if not isinstance(v, np.ndarray) else v
    np.savez_compressed(os.path.join(OUT, "chess_search_traces.npz"), **flat)
    ref_mcts.Node = ref_node.Node


# --------------------------------------------------------------------------- C. sim.play_game / generate_training_data
def pack_history(h):
    n = len(h["actions"])
    assert len(h["states"]) == n and len(h["rewards"]) == n and len(h["colours"]) == n
    states = np.stack([pack_planes(s.numpy()) for s in h["states"]]) if n else np.zeros((0, 119, 8), np.uint8)
    off, mv, pr = [0], [], []
    for d in h["actions"]:
        for m, p in d.items():
            mv.append((m.from_square, m.to_square, m.promotion or 0))
            pr.append(float(p))
        off.append(len(mv))
    return dict(states=states, act_off=np.array(off, np.int64), act_moves=np.array(mv, np.int8).reshape(-1, 3), act_probs=np.array(pr, np.float64),
                rewards=np.array(h["rewards"], np.int8), colours=np.array([int(bool(c)) for c in h["colours"]], np.int8))


def gen_play():
    cases = []
    spec = [(0, False, 8, "dyadic"), (1, True, 8, "dyadic"), (2, True, 12, "dyadic"), (3, False, 20, "dyadic"), (4, True, 5, "dyadic"), (5, False, 2, "dyadic")]
    for seed, c960, S, mode in spec:
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        state = random.getstate()
        sch = random.randint(0, 959) if c960 else -1            # what ChessTensor(chess960=True) will draw (chess_tensor.py:69)
        random.setstate(state)
        model = HashModel(mode=mode, salt=1000 + seed)
        t0 = time.time()
        with quiet():
            h = ref_sim.play_game(model, {"C": 2, "num_searches": S}, c960=c960)
        c = pack_history(h)
        c.update(seed=seed, c960=int(c960), S=S, mode=mode, salt=1000 + seed, scharnagl=sch, kind="play_game")
        cases.append(c)
        print("play_game seed %d c960=%d S=%d: %d plies, rewards[0]=%s, %.1fs" % (seed, c960, S, len(h["actions"]), h["rewards"][:1], time.time() - t0))
    # generate_training_data: 3 games concatenated + return_dict[os.getpid()]
    seed, c960, S = 11, True, 6
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    state = random.getstate()
    sch = [random.randint(0, 959) for _ in range(3)]
    random.setstate(state)
    rd = {}
    model = HashModel(mode="dyadic", salt=2000)
    with quiet():
        h = ref_sim.generate_training_data(model, 3, {"C": 2, "num_searches": S}, rd, c960)
    assert list(rd.keys()) == [os.getpid()] and rd[os.getpid()] is h
    c = pack_history(h)
    c.update(seed=seed, c960=int(c960), S=S, mode="dyadic", salt=2000, scharnagl=np.array(sch, np.int32), kind="generate_training_data", num_games=3)
    cases.append(c)
    print("generate_training_data: 3 games, %d samples" % len(h["actions"]))
    flat = {"n_cases": np.int64(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat["c%d_%s" % (i, k)] = np.array(v) if not isinstance(v, np.ndarray) else v
    np.savez_compressed(os.path.join(OUT, "chess_play_records.npz"), **flat)


if __name__ == "__main__":
    which = sys.argv[1:] or ["games", "search", "play"]
    games = gen_games() if ("games" in which or "search" in which or "realnet" in which) else None
    if "search" in which:
        gen_search(games)
    if "play" in which:
        gen_play()


# --------------------------------------------------------------------------- D. the north-star statement itself: reference mcts.py + reference network.py (CPU, fp32)
def gen_real_network(games):
    """MCTS0.search of the reference with the reference's own policyNN (torch.manual_seed(0) random init, eval mode, fp32, CPU, batch 1 —
    exactly what `python mcts.py` / `python sim.py` run) on real positions.  The product must reproduce the visit distributions with ITS
    network module (same seed -> same weights, golden-checked) on the GPU: 'within 1e-4 fp32, move indices bit-exact' (BASELINE.json north_star)."""
    import network as ref_net
    torch.set_num_threads(8)
    torch.manual_seed(0)
    net = ref_net.policyNN({})
    net.eval()
    rs = random.Random(2024)
    spec = [(0, 0, 30, False), (0, 0, 100, True), (0, 5, 60, True), (1, 9, 60, False), (9, 4, 60, True), (3, 6, 40, True), (4, 12, 50, False)]
    n_scripted = len(SCRIPTED)
    for k in range(7):
        gi = n_scripted + rs.randrange(len(games) - n_scripted)
        spec.append((gi, rs.randrange(0, max(1, len(games[gi]["moves"]) - 1)), [40, 80][k % 2], bool(k % 2)))
    cases = []
    for ci, (gi, ply, S, learning) in enumerate(spec):
        game = replay(games, gi, ply)
        if game.board.is_game_over():
            ply -= 1
            game = replay(games, gi, ply)
        torch.manual_seed(100 + ci)
        t0 = time.time()
        engine = ref_mcts.MCTS0(game=game, args={"C": 2, "num_searches": S}, model=net)
        probs = engine.search(game.board, verbose=False, learning=learning)
        with torch.no_grad():
            p0, v0 = net(game.get_representation().float().unsqueeze(0), inference=True)
        root_actions = [ref_ct.actionToTensor(m, game.board.turn).nonzero().item() for m in probs.keys()]
        cases.append(dict(game=gi, ply=ply, S=S, learning=int(learning), root_actions=np.array(root_actions, np.int32),
                          root_probs=np.array(list(probs.values()), np.float64),
                          root_moves=np.array([(m.from_square, m.to_square, m.promotion or 0) for m in probs.keys()], np.int32).reshape(-1, 3),
                          root_policy=p0[0].numpy().copy(), root_value=np.float32(v0.item())))
        print("real-network case %2d: game %2d ply %3d S=%3d learning=%d -> %d children, %.1fs" % (ci, gi, ply, S, learning, len(probs), time.time() - t0))
    flat = {"n_cases": np.int64(len(cases)), "torch_version": np.array(torch.__version__)}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat["c%d_%s" % (i, k)] = np.array(v) if not isinstance(v, np.ndarray) else v
    np.savez_compressed(os.path.join(OUT, "chess_real_network_searches.npz"), **flat)
    torch.set_num_threads(1)


if __name__ == "__main__" and "realnet" in sys.argv[1:]:
    gen_real_network(games if games is not None else gen_games())


# --------------------------------------------------------------------------- E. BASELINE.json configs[0] itself: `python sim.py` (sim.py:125-137)
def gen_config0():
    """1 self-play game, num_searches=10, Chess960, random-init policyNN on the CPU — the reference's own runnable case, with the seeds fixed."""
    import network as ref_net
    torch.set_num_threads(8)
    seed = 0
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    net = ref_net.policyNN({})
    net.eval()
    state = random.getstate()
    sch = random.randint(0, 959)
    random.setstate(state)
    t0 = time.time()
    with quiet():
        h = ref_sim.generate_training_data(net, 1, {"C": 2, "num_searches": 10}, None, True)
    c = pack_history(h)
    c.update(seed=seed, c960=1, S=10, scharnagl=sch, kind="config0")
    print("configs[0]: Chess960 start %d, %d plies, %.1fs" % (sch, len(h["actions"]), time.time() - t0))
    np.savez_compressed(os.path.join(OUT, "chess_config0_game.npz"), **{k: (np.array(v) if not isinstance(v, np.ndarray) else v) for k, v in c.items()})
    torch.set_num_threads(1)


if __name__ == "__main__" and "config0" in sys.argv[1:]:
    gen_config0()
