"""chess_tensor.py of the reference: ChessTensor + the three action codecs."""
from sigma_zero_amd.chess_tensor import ChessTensor, actionsToTensor, actionToTensor, tensorToAction  # noqa: F401
