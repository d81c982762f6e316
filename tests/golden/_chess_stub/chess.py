"""Name-only stand-in for the `chess` namespace, used ONLY by gen_reference_fixtures.py in the
build container so that the reference's pure-arithmetic modules (mctsnode.py, mcts.py, network.py
and the three codec functions of chess_tensor.py) can be imported unmodified.

It contains NO chess rules: constants, an empty Board, and a Move value class (squares +
promotion + UCI text).  Anything in the reference that needs real python-chess behaviour
(ChessTensor, sim.play_game) is NOT exercised through this file; those parts stay
"parity unpinned" by the reference (see DESIGN.md / SURVEY.md §8(c)).
This file never travels into the product path and is never imported by tests.
"""
WHITE, BLACK = True, False
Color = bool
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = range(1, 7)
SQUARES = list(range(64))
_FILES, _RANKS, _PIECES = "abcdefgh", "12345678", " pnbrqk"


class Board:
    pass


class Move:
    def __init__(self, from_square, to_square, promotion=None):
        self.from_square, self.to_square, self.promotion = from_square, to_square, promotion

    def uci(self):
        s = _FILES[self.from_square % 8] + _RANKS[self.from_square // 8] + _FILES[self.to_square % 8] + _RANKS[self.to_square // 8]
        return s + (_PIECES[self.promotion] if self.promotion else "")

    @classmethod
    def from_uci(cls, u):
        f = _FILES.index(u[0]) + 8 * _RANKS.index(u[1])
        t = _FILES.index(u[2]) + 8 * _RANKS.index(u[3])
        return cls(f, t, _PIECES.index(u[4]) if len(u) > 4 else None)

    def __eq__(self, o):
        return (self.from_square, self.to_square, self.promotion) == (o.from_square, o.to_square, o.promotion)

    def __hash__(self):
        return hash((self.from_square, self.to_square, self.promotion))

    def __repr__(self):
        return "Move.from_uci(%r)" % self.uci()
