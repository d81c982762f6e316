"""MCTS0 of /root/reference/mcts.py:24-122 with the same constructor and search() signature; the tree
lives in HBM and every step of the search runs in the HIP engine (one wavefront per board)."""
import torch

from .chess_tensor import ChessTensor, Move
from .selfplay import SelfPlayEngine, NOISE_REFERENCE

device = "cuda" if torch.cuda.is_available() else "cpu"


class MCTS0:
    """Monte Carlo Tree Search modified for AlphaZero (single position front-end of the batched engine)."""

    def __init__(self, game=None, args=None, model=None):
        self.game = game
        self.args = args
        self.model = model.to(device)                         # mcts.py:36
        self._engines = {}
        self.last_root = None

    def _engine(self, learning):
        key = (bool(learning), bool(self.game.chess960))
        if key not in self._engines:
            dtype = next(self.model.parameters()).dtype if hasattr(self.model, "parameters") else torch.float32
            if dtype not in (torch.float32, torch.bfloat16):
                dtype = torch.float32
            self._engines[key] = SelfPlayEngine(self.model, self.args, 1, chess960=key[1], learning=key[0], planes_dtype=dtype,
                                                noise_value=self.args.get("noise_value", NOISE_REFERENCE))
        return self._engines[key]

    @torch.no_grad()
    def search(self, state=None, verbose=True, learning=False):
        """Returns {Move: visit_count / total} over the root's children in ascending action-index order (mcts.py:113-122).
        `state` (the chess.Board of the reference signature) is implied by self.game, which search() uses un-copied (mcts.py:43)."""
        if not isinstance(self.game, ChessTensor):
            raise TypeError("MCTS0.search needs the ChessTensor game object it was constructed with")
        eng = self._engine(learning)
        eng.upload_game(0, self.game)
        eng.search()
        eng.check_errors()
        action, visits, n_child, prior, wsum = eng.root_children()
        k = int(n_child[0])
        self.last_root = dict(action=action[0, :k].copy(), visits=visits[0, :k].copy(), prior=prior[0, :k].copy(), value_sum=wsum[0, :k].copy())
        total = int(visits[0, :k].sum())
        if k and total == 0:
            raise ZeroDivisionError("division by zero")          # num_searches == 1, mcts.py:118-120
        return {self.game.move_from_index(int(a)): int(v) / total for a, v in zip(action[0, :k], visits[0, :k])}
