"""Model-vs-model gating matches on the GPU engine (the role of /root/reference/test_update.py:12-97 and the
search-then-argmax usage of eval.py:92-94, play.py:40-41): both networks play greedy (`max` visit count) with
learning=False; every game is one board of a batch, colours alternate over the boards."""
import numpy as np
import torch

from .selfplay import SelfPlayEngine


@torch.no_grad()
def play_match(model_a, model_b, args, n_games, chess960=False, scharnagl=None, max_plies=512, planes_dtype=None, device="cuda:0"):
    """Returns {'a_wins', 'b_wins', 'draws', 'unfinished', 'results'}; model_a has white on even boards, black on odd ones."""
    import random
    if planes_dtype is None:
        planes_dtype = ("bits128" if getattr(model_a, "w16", False) and getattr(model_b, "w16", False) else "nhwc128") if hasattr(model_a, "tower") else torch.float32
    eng = SelfPlayEngine(None, args, n_games, chess960=chess960, learning=False, device=device, planes_dtype=planes_dtype)
    if chess960 and scharnagl is None:
        scharnagl = [random.randint(0, 959) for _ in range(n_games)]
    eng.new_games(scharnagl if chess960 else [-1] * n_games)
    a_is_white = (np.arange(n_games) % 2 == 0)
    results = np.full(n_games, 2, dtype=np.int8)            # +1 white won, -1 black won, 0 draw, 2 unfinished
    idx_dev = torch.arange(n_games, device=eng.device)
    for ply in range(max_plies):
        white_to_move = (ply % 2 == 0)                      # all games start together: one side to move per ply
        a_moves = torch.as_tensor(a_is_white == white_to_move, device=eng.device)
        ia, ib = idx_dev[a_moves], idx_dev[~a_moves]

        def evaluator(planes):
            policy = torch.empty(n_games, 4672, dtype=torch.float32, device=eng.device)
            value = torch.empty(n_games, dtype=torch.float32, device=eng.device)
            for model, ii in ((model_a, ia), (model_b, ib)):
                if ii.numel():
                    p, v = model(planes[ii].contiguous(), inference=True)
                    policy[ii] = p.float()
                    value[ii] = v.float().reshape(-1)
            return policy, value

        eng.search(evaluator)
        eng.check_errors()
        eng.play(np.full(n_games, -1.0))                    # greedy
        rec = eng.fetch_ply()
        over = rec["game_over"].astype(bool) & rec["active"].astype(bool)
        results[over] = rec["result"][over]
        if not (results == 2).any():
            break
    eng.close()
    a_score = np.where(a_is_white, results, -results)
    return {"a_wins": int(((a_score == 1) & (results != 2)).sum()), "b_wins": int(((a_score == -1) & (results != 2)).sum()),
            "draws": int((results == 0).sum()), "unfinished": int((results == 2).sum()), "results": results.tolist()}
