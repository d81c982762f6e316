"""Self-play of /root/reference/sim.py:31-123 — same functions, same return dict — with all games of a
call advanced concurrently on the GPU (one board per game, lock-step searches, one batched network call
per simulation)."""
import os

import numpy as np
import torch

from .chess_tensor import Move, index_to_move, QUEEN
from .selfplay import SelfPlayEngine, unpack_planes, model_device

device = "cuda" if torch.cuda.is_available() else "cpu"      # module global of the reference (sim.py:12); the engine itself follows the model's device


_DECODE = {}            # colour -> (from[4672], to[4672], promo[4672], maybe_queen[4672]) int arrays; from = -1 where the index leaves the board
_MOVES = {}             # (from, to, promo) -> Move: one shared object per distinct move (treat the Moves of a record as read-only)


def _decode_tables(colour_white):
    """index -> (from, to, under-promotion piece) for one colour, computed once with index_to_move (tensorToAction, chess_tensor.py:309-410)"""
    if colour_white not in _DECODE:
        frm = np.full(4672, -1, np.int64); to = np.zeros(4672, np.int64); pr = np.zeros(4672, np.int64); mq = np.zeros(4672, bool)
        for a in range(4672):
            try:
                mv = index_to_move(a, colour_white)
            except IndexError:
                continue
            frm[a], to[a], pr[a] = mv.from_square, mv.to_square, mv.promotion or 0
            plane, cell = divmod(a, 64)
            mq[a] = plane < 56 and cell // 8 == 1 and plane // 7 in (0, 1, 7) and plane % 7 == 0       # one step up / up-diagonal from view row 1
        _DECODE[colour_white] = (frm, to, pr, mq)
    return _DECODE[colour_white]


def _moves_for_record(action_idx, colour_white, packed_root):
    """Decode child action indices of one root into Move objects.  A sliding-plane move of a pawn from the
    7th to the 8th rank (mover's view rows 1 -> 0) is a queen promotion (tensorToAction + queen_promotion dict)."""
    frm, to, pr, mq = _decode_tables(bool(colour_white))
    a = np.asarray(action_idx, dtype=np.int64)
    f, t, p = frm[a], to[a], pr[a].copy()
    cand = mq[a]
    if cand.any():
        own_pawn_row1 = int(packed_root[0][1])                # plane 0 = mover's pawns, view row 1: bit j = column j
        cols = a[cand] % 8
        p[np.nonzero(cand)[0][((own_pawn_row1 >> cols) & 1).astype(bool)]] = QUEEN
    moves = []
    for key in zip(f.tolist(), t.tolist(), p.tolist()):
        mv = _MOVES.get(key)
        if mv is None:
            mv = _MOVES[key] = Move(key[0], key[1], key[2] or None)
        moves.append(mv)
    return moves


def play_games(model, args, n_games, c960=False, scharnagl=None, uniforms=None, learning=True, planes_dtype=None, max_plies=100000,
               verbose=False, n_boards=None, compact=True, stats=None):
    """Plays n_games games to the end and returns a list of per-game history dicts (sim.py:38-43 layout), game g at index g.

    The games run concurrently on `n_boards` board slots of one engine (default: one slot per game).  A slot whose game ends is
    REFILLED with the next game that has not started yet, so the GPU stays full until fewer than n_boards games remain; from then
    on the batch is COMPACTED (sz_compact): the network only evaluates the boards that still play.  With the MFMA networks
    (FastPolicyNet, SplitPolicyNet) a game's results do not depend on the slot it runs in, on the batch size or on what runs beside it —
    bit for bit (their kernels are per board and every reduction has a fixed order; tests/test_gpu_train_and_precision.py).  A plain
    torch module goes through MIOpen / hipBLASLt, whose algorithm choice and hence rounding can change with the batch size: there the
    records are reproducible for a fixed (n_games, n_boards) schedule only; pass compact=False to keep the batch size constant.

    scharnagl: start index per game (default: python `random.randint(0,959)` per game like chess_tensor.py:69, drawn in game order).
    uniforms(game, ply) -> float: the np.random.random_sample() draw of sim.py:68.  Default: the global numpy RNG, drawn once per ply
      for every running game in game order (n_games == 1 reproduces the reference's stream; for several concurrent games the order of
      the draws necessarily differs from the reference's one-game-after-another order — generate_training_data(rng_order="reference")).
    max_plies: a game still running after that many plies is cut (result None, rewards 0); one number or one per game.
    stats: optional dict, receives 'sims', 'nn_rows', 'plies' (work done; nn_rows = network rows evaluated) and 'host_seconds' (the host's wall time per phase
      of the ply loop: enqueue_search returns before the GPU is done, wait_search is the wait for it)."""
    import random
    if not torch.cuda.is_available():
        raise RuntimeError("self-play needs an MI355X (HIP) device: the search has no CPU fallback")
    dev = model_device(model)                               # the model's own GPU (a rank with local_rank > 0 plays on ITS device)
    model = model.to(dev)
    if planes_dtype is None:
        if hasattr(model, "tower"):                       # FastPolicyNet: hand-written MFMA tower, NHWC planes
            planes_dtype = "bits128" if getattr(model, "w16", False) else "nhwc128"     # bit-packed: 1 KiB per board instead of 16
        else:
            planes_dtype = next(model.parameters()).dtype
            if planes_dtype not in (torch.float32, torch.bfloat16):
                planes_dtype = torch.float32
    if c960 and scharnagl is None:
        scharnagl = [random.randint(0, 959) for _ in range(n_games)]
    if not c960:
        scharnagl = [-1] * n_games
    cap = np.broadcast_to(np.asarray(max_plies, dtype=np.int64), (n_games,))
    B = max(1, min(int(n_games), int(n_boards) if n_boards else int(n_games)))
    S = int(args["num_searches"])
    eng = SelfPlayEngine(model, args, B, chess960=c960, learning=learning, planes_dtype=planes_dtype, device=dev)
    games = [dict(states=[], actions=[], rewards=[], colours=[], result=None, packed_states=[]) for _ in range(n_games)]   # packed_states: the (119,8) uint8 form (train_RL.py:42)
    slot_game = np.full(B, -1, dtype=np.int64)            # game running on each board slot, -1 = none
    plies = np.zeros(n_games, dtype=np.int64)             # plies played so far per game
    next_game = 0
    work = dict(sims=0, nn_rows=0, plies=0)

    def refill(slots):
        """start the next waiting games on these slots (ChessTensor.__init__/start_board for each); the rest go dark"""
        nonlocal next_game
        sch, act = np.full(B, -1, dtype=np.int32), np.zeros(B, dtype=np.uint8)
        for s_ in slots:
            if next_game < n_games:
                slot_game[s_] = next_game
                sch[s_], act[s_] = scharnagl[next_game], 1
                next_game += 1
            else:
                slot_game[s_] = -1
        if act.any():
            eng.new_games(sch, act)
        eng.set_active((slot_game >= 0).astype(np.uint8))

    def absorb(rec, game_of_slot):
        """host-side bookkeeping of one ply's records (sim.py:71-73); runs while the GPU searches the next ply"""
        slots = np.nonzero((game_of_slot >= 0) & rec["active"].astype(bool))[0]
        if not len(slots):
            return
        states = torch.from_numpy(unpack_planes(rec["packed"][slots]))          # one unpack for the whole ply; a game's state is a view of it
        for i, s_ in enumerate(slots.tolist()):
            g = int(game_of_slot[s_])
            k = int(rec["n_child"][s_])
            white = bool(rec["colour"][s_])
            vis = rec["visits"][s_, :k].astype(np.int64)
            moves = _moves_for_record(rec["action"][s_, :k], white, rec["packed"][s_])
            games[g]["states"].append(states[i])
            games[g]["packed_states"].append(rec["packed"][s_])
            games[g]["actions"].append(dict(zip(moves, (vis / int(vis.sum())).tolist())))      # int / int in float64: the reference's v / sum_values
            games[g]["colours"].append(white)
            if rec["game_over"][s_]:
                games[g]["result"] = {1: "1-0", -1: "0-1", 0: "1/2-1/2"}[int(rec["result"][s_])]

    import time
    host = dict(compact=0.0, enqueue_search=0.0, absorb=0.0, wait_search=0.0, play_fetch=0.0, refill=0.0)     # host wall time per phase (stats["host_seconds"])
    def lap(key, t0):
        t1 = time.perf_counter()
        host[key] += t1 - t0
        return t1
    refill(range(B))
    pending = None
    while (slot_game >= 0).any():
        t0 = time.perf_counter()
        n_rows = eng.compact() if compact else B
        t0 = lap("compact", t0)
        eng.search()                                       # enqueues num_searches x (network + tree step); returns before the GPU is done
        t0 = lap("enqueue_search", t0)
        if pending is not None:
            absorb(*pending)                               # previous ply's records, overlapped with this ply's search
            pending = None
        t0 = lap("absorb", t0)
        eng.check_errors()
        t0 = lap("wait_search", t0)
        u = np.zeros(B, dtype=np.float64)
        running = np.nonzero(slot_game >= 0)[0]
        for s_ in running[np.argsort(slot_game[running], kind="stable")]:      # draws in game order
            g = int(slot_game[s_])
            u[s_] = uniforms(g, int(plies[g])) if uniforms is not None else np.random.random_sample()
        eng.play(u)
        rec = eng.fetch_ply()
        if eng.stats()["boards_error"]:
            eng.check_errors()
        t0 = lap("play_fetch", t0)
        pending = (rec, slot_game.copy())
        work["sims"] += S * len(running); work["nn_rows"] += S * n_rows; work["plies"] += 1
        plies[slot_game[running]] += 1
        done = [int(s_) for s_ in running if (rec["game_over"][s_] and rec["active"][s_]) or plies[slot_game[s_]] >= cap[slot_game[s_]]]
        if done:
            refill(done)
        lap("refill", t0)
        if verbose:
            print("ply %d: %d games running, %d waiting, %d network rows" % (work["plies"], int((slot_game >= 0).sum()), n_games - next_game, n_rows))
    if pending is not None:
        absorb(*pending)
    for g in range(n_games):
        reward = {"1-0": 1, "0-1": -1}.get(games[g]["result"], 0)
        games[g]["rewards"] = [reward if i % 2 == 0 else -reward for i in range(len(games[g]["actions"]))]   # sim.py:94-97
    eng.close()
    if stats is not None:
        stats.update(work)
        stats["host_seconds"] = host
    return games


def play_game(model, args, c960=False):
    """One self-play game (sim.py:31-99): {'states','actions','rewards','colours'}."""
    g = play_games(model, args, 1, c960=c960)[0]
    return {k: g[k] for k in ("states", "actions", "rewards", "colours")}


def generate_training_data(model, num_games=1, args=None, return_dict=None, c960=False, rng_order="batched", n_boards=None):
    """sim.py:102-123: concatenated histories of num_games games; also stored under return_dict[os.getpid()].
    rng_order="batched" (default): all games run concurrently; python's `random` (Chess960 starts) and numpy's global RNG (move
      sampling) are consumed per ply across the running games, so the games differ from the ones the reference would draw from the
      same seeds (same distribution).
    rng_order="reference": one game after another, exactly the reference's order of RNG draws — same seeds, same games as
      sim.py:114-118 (bit for bit with a deterministic network); one board on the GPU, for checks rather than for throughput."""
    games_history = {"states": [], "actions": [], "rewards": [], "colours": []}
    if rng_order == "reference":
        games = [play_games(model, args, 1, c960=c960)[0] for _ in range(num_games)]
    elif rng_order == "batched":
        games = play_games(model, args, num_games, c960=c960, n_boards=n_boards)
    else:
        raise ValueError("rng_order must be 'batched' or 'reference'")
    for g in games:
        for key in games_history:
            games_history[key] += g[key]
    if return_dict is not None:
        return_dict[os.getpid()] = games_history
    return games_history
