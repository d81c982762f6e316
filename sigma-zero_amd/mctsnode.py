"""Node of /root/reference/mctsnode.py:7-63 as a read-only VIEW.

In the engine a node is not a Python object: N (visit_count), W (value_sum), P (prior) and the child
spans live in a Structure-of-Arrays store in HBM (csrc/sz_engine.hip).  After a search the root and its
children can be inspected through this class with the reference's field names; the arithmetic of
select / get_ucb / expand / backpropagate runs in the HIP kernels.
"""
import math

import torch


class Node:
    def __init__(self, game=None, args=None, state=None, parent=None, action_taken=None, prior=0, color=True, search_scope_game=None):
        self.game = game
        self.args = args
        self.parent = parent
        self.action_taken = action_taken
        self.prior = prior
        self.color = color
        self.children = []
        self.visit_count = 0
        self.value_sum = .0
        self.value = .0

    def is_fully_expanded(self):
        return len(self.children)

    def get_ucb(self, vc, vsum, prior):
        """Same expression as mctsnode.py:33-37, for inspection of a finished tree (the search does not call this)."""
        q_value = 1 - (vsum / (vc + 1e-6) + 1) / 2
        return q_value + self.args['C'] * (math.sqrt(self.visit_count) / (vc + 1)) * prior

    def select(self):
        vc = torch.tensor([c.visit_count for c in self.children])
        vsum = torch.tensor([c.value_sum for c in self.children])
        prior = torch.tensor([c.prior for c in self.children])
        return self.children[torch.argmax(self.get_ucb(vc, vsum, prior)).item()]
