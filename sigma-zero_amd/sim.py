"""Self-play of /root/reference/sim.py:31-123 — same functions, same return dict — with all games of a
call advanced concurrently on the GPU (one board per game, lock-step searches, one batched network call
per simulation)."""
import os

import numpy as np
import torch

from .chess_tensor import ChessTensor, Move, index_to_move, QUEEN
from .selfplay import SelfPlayEngine, unpack_planes

device = "cuda" if torch.cuda.is_available() else "cpu"


def _moves_for_record(action_idx, colour_white, packed_root):
    """Decode child action indices of one root into Move objects.  A sliding-plane move of a pawn from the
    7th to the 8th rank (mover's view rows 1 -> 0) is a queen promotion (tensorToAction + queen_promotion dict)."""
    own_pawns = unpack_planes(packed_root[0])                 # plane 0 = mover's pawns, [row][col] in the mover's view
    moves = []
    for a in action_idx:
        mv = index_to_move(int(a), colour_white)
        plane, cell = divmod(int(a), 64)
        row, col = divmod(cell, 8)
        if plane < 56 and row == 1 and own_pawns[row, col] and plane // 7 in (0, 1, 7) and plane % 7 == 0:
            mv.promotion = QUEEN
        moves.append(mv)
    return moves


def play_games(model, args, n_games, c960=False, scharnagl=None, uniforms=None, learning=True, planes_dtype=None, max_plies=100000,
               verbose=False):
    """Plays n_games concurrently.  Returns a list of per-game history dicts (sim.py:38-43 layout).
    scharnagl: start index per game (default: python `random.randint(0,959)` per game like chess_tensor.py:69).
    uniforms(game, ply) -> float: the np.random.random_sample() draw of sim.py:68 (default: global numpy RNG, drawn per ply in game order)."""
    import random
    model = model.to(device)
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
    eng = SelfPlayEngine(model, args, n_games, chess960=c960, learning=learning, planes_dtype=planes_dtype)
    eng.new_games(scharnagl)
    games = [dict(states=[], actions=[], rewards=[], colours=[], result=None) for _ in range(n_games)]
    alive = np.ones(n_games, dtype=bool)
    ply = 0

    def absorb(rec, was_alive):
        """host-side bookkeeping of one ply's records (sim.py:71-73); runs while the GPU searches the next ply"""
        for g in np.nonzero(was_alive & rec["active"].astype(bool))[0]:
            k = int(rec["n_child"][g])
            white = bool(rec["colour"][g])
            acts = rec["action"][g, :k]
            vis = rec["visits"][g, :k].astype(np.int64)
            total = int(vis.sum())
            moves = _moves_for_record(acts, white, rec["packed"][g])
            games[g]["states"].append(torch.from_numpy(unpack_planes(rec["packed"][g])))
            games[g]["actions"].append({m: int(v) / total for m, v in zip(moves, vis)})
            games[g]["colours"].append(white)
            if rec["game_over"][g]:
                games[g]["result"] = {1: "1-0", -1: "0-1", 0: "1/2-1/2"}[int(rec["result"][g])]

    pending = None
    while alive.any() and ply < max_plies:
        eng.search()                                       # enqueues num_searches x (network + tree step); returns before the GPU is done
        if pending is not None:
            absorb(*pending)                               # previous ply's records, overlapped with this ply's search
            pending = None
        eng.check_errors()
        u = np.zeros(n_games, dtype=np.float64)
        for g in range(n_games):
            if alive[g]:
                u[g] = uniforms(g, ply) if uniforms is not None else np.random.random_sample()
        eng.play(u)
        rec = eng.fetch_ply()
        st = eng.stats()
        if st["boards_error"]:
            eng.check_errors()
        pending = (rec, alive.copy())
        alive &= ~(rec["game_over"].astype(bool) & rec["active"].astype(bool))
        ply += 1
        if verbose:
            print("ply %d: %d games alive" % (ply, int(alive.sum())))
    if pending is not None:
        absorb(*pending)
    for g in range(n_games):
        reward = {"1-0": 1, "0-1": -1}.get(games[g]["result"], 0)
        games[g]["rewards"] = [reward if i % 2 == 0 else -reward for i in range(len(games[g]["actions"]))]   # sim.py:94-97
    eng.close()
    return games


def play_game(model, args, c960=False):
    """One self-play game (sim.py:31-99): {'states','actions','rewards','colours'}."""
    g = play_games(model, args, 1, c960=c960)[0]
    return {k: g[k] for k in ("states", "actions", "rewards", "colours")}


def generate_training_data(model, num_games=1, args=None, return_dict=None, c960=False):
    """sim.py:102-123: concatenated histories of num_games games; also stored under return_dict[os.getpid()]."""
    games_history = {"states": [], "actions": [], "rewards": [], "colours": []}
    for g in play_games(model, args, num_games, c960=c960):
        for key in games_history:
            games_history[key] += g[key]
    if return_dict is not None:
        return_dict[os.getpid()] = games_history
    return games_history
