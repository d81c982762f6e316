"""MCTS0 of /root/reference/mcts.py:24-122 with the same constructor and search() signature; the tree
lives in HBM and every step of the search runs in the HIP engine (one wavefront per board)."""
import torch

from .chess_tensor import ChessTensor
from .mctsnode import Node
from .selfplay import SelfPlayEngine, NOISE_REFERENCE, model_device

device = "cuda" if torch.cuda.is_available() else "cpu"


class MCTS0:
    """Monte Carlo Tree Search modified for AlphaZero (single position front-end of the batched engine)."""

    def __init__(self, game=None, args=None, model=None):
        self.game = game
        self.args = args
        # mcts.py:36 `model.to(device)`: a CPU model moves to this process's current GPU, a model already on cuda:N stays THERE
        self.device = model_device(model) if torch.cuda.is_available() else torch.device("cpu")
        self.model = model.to(self.device)
        self._engines = {}
        self._tree = None
        self._root = None
        self._root_game = None

    def _engine(self, learning):
        key = (bool(learning), bool(self.game.chess960))
        if key not in self._engines:
            dtype = next(self.model.parameters()).dtype if hasattr(self.model, "parameters") else torch.float32
            if dtype not in (torch.float32, torch.bfloat16):
                dtype = torch.float32
            self._engines[key] = SelfPlayEngine(self.model, self.args, 1, chess960=key[1], learning=key[0], planes_dtype=dtype,
                                                noise_value=self.args.get("noise_value", NOISE_REFERENCE), device=self.device)
        return self._engines[key]

    @torch.no_grad()
    def search(self, state=None, verbose=True, learning=False):
        """Returns {Move: visit_count / total} over the root's children in ascending action-index order (mcts.py:113-122).
        `state` is the chess.Board of the reference signature: the reference reads only `state.turn` from it (root colour,
        mcts.py:43) and searches self.game, un-copied; so does this — a `state` whose side to move contradicts self.game is refused."""
        if not isinstance(self.game, ChessTensor):
            raise TypeError("MCTS0.search needs the ChessTensor game object it was constructed with")
        if state is not None and hasattr(state, "turn") and bool(state.turn) != bool(self.game.board.turn):
            raise ValueError("state.turn contradicts the game MCTS0 was constructed with (mcts.py:43 takes the root colour from state.turn)")
        eng = self._engine(learning)
        eng.upload_game(0, self.game)
        eng.search()
        eng.check_errors()
        action, visits, n_child, prior, wsum = eng.root_children()
        k = int(n_child[0])
        # the Node view is built lazily (`.root`): the whole-tree readback costs more than a small search and most callers only want the dict
        # (sim.py:63-68, eval.py:92-94).  The engines are this object's own, so the tree stays in HBM untouched until its next search.
        self._tree, self._tree_engine = None, eng
        self._root_game = self.game.copy()                    # the caller may have moved self.game by the time the view is asked for (sim.py:76)
        self._root = None
        total = int(visits[0, :k].sum())
        if k and total == 0:
            raise ZeroDivisionError("division by zero")          # num_searches == 1, mcts.py:118-120
        return {self.game.move_from_index(int(a)): int(v) / total for a, v in zip(action[0, :k], visits[0, :k])}

    @property
    def root(self):
        """The finished tree as reference-style Node objects (mctsnode.py:7-18 fields), built on first access from the engine's store."""
        if self._root is None and getattr(self, "_tree_engine", None) is not None:
            if self._tree is None:
                self._tree = self._tree_engine.debug_tree(0)
            self._root = Node.from_engine_tree(self._root_game, self.args, self._tree)
        return self._root

    @property
    def last_root(self):
        """arrays of the root's children (ascending action index): action, visits, prior, value_sum"""
        r = self.root
        if r is None:
            return None
        import numpy as np
        return dict(action=np.array([c.action_index for c in r.children], np.int32), visits=np.array([c.visit_count for c in r.children], np.int32),
                    prior=np.array([c.prior for c in r.children], np.float32), value_sum=np.array([c.value_sum for c in r.children], np.float64))
