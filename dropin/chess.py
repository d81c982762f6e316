"""The names of python-chess that the reference's hot path touches; rules live in the engine."""
from sigma_zero_amd.chess_tensor import WHITE, BLACK, PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING, Move, Board  # noqa: F401
Color = bool
SQUARES = list(range(64))
