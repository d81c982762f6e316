"""mcts.py of the reference, served by the MI355X engine."""
from sigma_zero_amd.mcts import MCTS0, device  # noqa: F401
