"""Node of /root/reference/mctsnode.py:7-63 — same constructor, fields and methods — as the host-side VIEW of the engine's tree.

In the engine a node is not a Python object: N (visit_count), W (value_sum), P (prior) and the child spans live in a
Structure-of-Arrays store in HBM and select / get_ucb / expand / backpropagate run in the HIP kernels
(csrc/sz_engine.hip).  After MCTS0.search() the whole tree is read back once (sz_debug_tree) and materialised as Node
objects with the reference's field names: `mcts.root`, `root.children[i].visit_count / value_sum / prior / action_taken /
color / parent`, recursively.  The methods below are the reference's arithmetic on those objects (float32 tensors in
get_ucb exactly like mctsnode.py:23-37), so code written against the reference's Node — walking the tree, re-scoring
children, growing it by hand with expand()/backpropagate() — keeps working on the view.
"""
import math

import torch

from .chess_tensor import WHITE


class Node:
    def __init__(self, game=None, args=None, state=None, parent=None, action_taken=None, prior=0, color=WHITE, search_scope_game=None):
        self.game = game
        self.args = args
        self.parent = parent
        self.action_taken = action_taken
        self.prior = prior
        self.color = color
        self.children = []
        self.visit_count = 0
        self.value_sum = .0
        self.value = .0  # state value
        self.action_index = None        # extra (not in the reference): index of action_taken in the 73x8x8 action space of the parent's colour

    def is_fully_expanded(self):
        return len(self.children)

    def select(self):
        vc = torch.tensor([child.visit_count for child in self.children])
        vsum = torch.tensor([child.value_sum for child in self.children])
        prior = torch.tensor([child.prior for child in self.children])
        ucb = self.get_ucb(vc, vsum, prior)
        return self.children[torch.argmax(ucb).item()]

    def get_ucb(self, vc, vsum, prior):
        q_value = 1 - (vsum / (vc + 1e-6) + 1) / 2
        return q_value + self.args['C'] * (math.sqrt(self.visit_count) / (vc + 1)) * prior

    def expand(self, policy):
        """policy: list of (action, prob) tuples; prob a 0-d / 1-element tensor or a float (mctsnode.py:39-54)"""
        for action, prob in policy:
            child = Node(game=None, args=self.args, state=None, parent=self, action_taken=action,
                         prior=prob.item() if hasattr(prob, "item") else float(prob), color=not self.color)
            self.children.append(child)

    def backpropagate(self, value):
        self.value_sum += value
        self.visit_count += 1
        value = self.game.get_opponent_value(value) if self.game is not None else -value
        if self.parent is not None:
            self.parent.backpropagate(value)

    # ------------------------------------------------------------------ view construction
    def _ensure_game(self):
        """node.game = deepcopy(parent.game); node.game.move_piece(action_taken)   (mcts.py:57-59, done for every visited node)"""
        if self.game is None and self.parent is not None:
            self.parent._ensure_game()
            self.game = self.parent.game.copy()
            self.game.push_action(self.action_index)
        return self.game

    @classmethod
    def from_engine_tree(cls, game, args, tree):
        """tree = SelfPlayEngine.debug_tree(board): (depth, action, visits, value_sum, prior), row 0 the root, then depth-first in
        child order.  game: the root's ChessTensor, kept un-copied on the root like mcts.py:43.  As in the reference, a node that the
        search visited owns its position (`game`), an unvisited child has game=None (mctsnode.py:45)."""
        depth, action, visits, wsum, prior = tree
        root = cls(game, args, None, color=bool(game.board.turn))
        root.visit_count, root.value_sum = int(visits[0]), float(wsum[0])
        stack = [root]                                     # stack[d + 1] = the latest node at depth d (stack[0] = root)
        for k in range(1, len(depth)):
            d = int(depth[k])
            parent = stack[d]
            mv = parent._ensure_game().move_from_index(int(action[k]))      # the position knows queen promotions and castling spelling
            child = cls(None, args, None, parent=parent, action_taken=mv, prior=float(prior[k]), color=not parent.color)
            child.visit_count, child.value_sum, child.action_index = int(visits[k]), float(wsum[k]), int(action[k])
            parent.children.append(child)
            if child.visit_count > 0:
                child._ensure_game()
            del stack[d + 1:]
            stack.append(child)
        return root
