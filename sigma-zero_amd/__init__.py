"""sigma_zero_amd — MI355X-native batched MCTS self-play engine behind the Python surface of
DidItWork/Sigma-Zero's train_RL.py loop (mcts.py / mctsnode.py / chess_tensor.py / sim.py / network.py).

The search runs only on the GPU (hand-written HIP, csrc/sz_engine.hip); the host mirror objects
(ChessTensor, Move, codecs) run the same rules code on the CPU for the per-game API the reference exposes.
"""
from .chess_tensor import (WHITE, BLACK, Move, Board, ChessTensor, actionsToTensor, actionToTensor, tensorToAction)
from .network import policyNN
from .mcts import MCTS0
from .mctsnode import Node
from .sim import play_game, generate_training_data
from .selfplay import SelfPlayEngine

__all__ = ["WHITE", "BLACK", "Move", "Board", "ChessTensor", "actionsToTensor", "actionToTensor", "tensorToAction",
           "policyNN", "MCTS0", "Node", "play_game", "generate_training_data", "SelfPlayEngine"]
