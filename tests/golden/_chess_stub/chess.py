"""Stand-in for the `chess` namespace (python-chess 1.10.0 is pinned by the reference at environment.yml:21 but is
not installed in this image and cannot be).  Used ONLY by the fixture generators in tests/golden/ in the build
container, so that the reference's own modules can be imported UNMODIFIED from /root/reference and run:

    mctsnode.py, mcts.py, network.py, chess_tensor.py (ChessTensor + codecs), sim.py (play_game, generate_training_data)

What is in here:
  * constants and the `Move` / `Piece` value classes (our own few lines);
  * `Board`: an adapter with the ~18 python-chess calls the hot path makes (SURVEY.md §8(c) call sites: piece_at,
    from_chess960_pos, legal_moves, push, is_repetition, move_stack, has_*_castling_rights, halfmove_clock, turn,
    is_game_over, outcome, result) whose RULES come from oracle/oc_chess.c — the repo's own C restatement of
    python-chess, pinned by public perft tables (tests/test_oracle_chess.py).

So fixtures generated through this file pin everything the REFERENCE's code does on top of the rules — the
119-plane stacks, history shift, black/white views, flips, saturating counters, the search on real positions,
play_game's sampling / records / rewards — to the reference's own source.  They do NOT pin the rules themselves
to python-chess: that stays "parity unpinned by the reference" (perft-pinned), as DESIGN.md §4 says.

This file never travels into the product path and is never imported by tests.
"""
import ctypes as _C
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _ROOT not in _sys.path:
    _sys.path.append(_ROOT)
from oracle import oracle as _O          # noqa: E402  (test infrastructure; this stub is test infrastructure too)

WHITE, BLACK = True, False
Color = bool
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = range(1, 7)
SQUARES = list(range(64))
_FILES, _RANKS, _PIECES = "abcdefgh", "12345678", " pnbrqk"


class Move:
    def __init__(self, from_square, to_square, promotion=None):
        self.from_square, self.to_square, self.promotion = int(from_square), int(to_square), (int(promotion) if promotion else None)

    def uci(self):
        s = _FILES[self.from_square % 8] + _RANKS[self.from_square // 8] + _FILES[self.to_square % 8] + _RANKS[self.to_square // 8]
        return s + (_PIECES[self.promotion] if self.promotion else "")

    @classmethod
    def from_uci(cls, u):
        f = _FILES.index(u[0]) + 8 * _RANKS.index(u[1])
        t = _FILES.index(u[2]) + 8 * _RANKS.index(u[3])
        return cls(f, t, _PIECES.index(u[4]) if len(u) > 4 else None)

    def __eq__(self, o):
        return isinstance(o, Move) and (self.from_square, self.to_square, self.promotion) == (o.from_square, o.to_square, o.promotion)

    def __hash__(self):
        return hash((self.from_square, self.to_square, self.promotion))

    def __repr__(self):
        return "Move.from_uci(%r)" % self.uci()

    def __str__(self):
        return self.uci()

    def _native(self):
        return _O.Move(self.from_square, self.to_square, self.promotion or 0)


class Piece:
    def __init__(self, piece_type, color):
        self.piece_type, self.color = piece_type, color

    def symbol(self):
        s = _PIECES[self.piece_type]
        return s.upper() if self.color else s


class Outcome:
    def __init__(self, termination, winner):
        self.termination, self.winner = termination, winner

    def result(self):
        return "1/2-1/2" if self.winner is None else ("1-0" if self.winner else "0-1")


class _LegalMoves:
    """board.legal_moves: iterable + `move in ...` (chess_tensor.py:91,146,158)"""

    def __init__(self, board):
        self._b = board

    def _list(self):
        buf = (_O.Move * _O.MAX_MOVES)()
        n = _O.lib().oc_legal_moves(self._b._p, buf)
        return [Move(buf[k].from_square, buf[k].to_square, buf[k].promotion or None) for k in range(n)]

    def __iter__(self):
        return iter(self._list())

    def __len__(self):
        return len(self._list())

    def count(self):
        return len(self._list())

    def __contains__(self, move):
        return bool(_O.lib().oc_is_legal(self._b._p, move._native()))


class _MoveStack:
    """board.move_stack: only its length is read (chess_tensor.py:113)"""

    def __init__(self, board):
        self._b = board

    def __len__(self):
        return int(_O.lib().oc_board_ply(self._b._p))


class Board:
    def __init__(self, fen=None, chess960=False, _ptr=None):
        if _ptr is not None:
            self._p = _ptr
        elif fen is None:
            self._p = _O.lib().oc_board_new()
        else:
            self._p = _O.lib().oc_board_from_fen(fen.encode(), int(chess960))

    @classmethod
    def from_chess960_pos(cls, scharnagl):
        return cls(_ptr=_O.lib().oc_board_new_960(int(scharnagl)))

    def __del__(self):
        try:
            if self._p:
                _O.lib().oc_board_free(self._p)
                self._p = None
        except Exception:
            pass

    def __deepcopy__(self, memo):
        return Board(_ptr=_O.lib().oc_board_copy(self._p))

    def copy(self):
        return self.__deepcopy__({})

    # -- state
    @property
    def turn(self):
        return bool(_O.lib().oc_board_turn(self._p))

    @property
    def halfmove_clock(self):
        return int(_O.lib().oc_board_halfmove_clock(self._p))

    @property
    def chess960(self):
        return bool(_O.lib().oc_board_is_chess960(self._p))

    @property
    def move_stack(self):
        return _MoveStack(self)

    @property
    def legal_moves(self):
        return _LegalMoves(self)

    def piece_at(self, square):
        col = _C.c_int(0)
        pt = _O.lib().oc_piece_at(self._p, int(square), _C.byref(col))
        return Piece(pt, bool(col.value)) if pt else None

    # -- rules
    def push(self, move):
        _O.lib().oc_push(self._p, move._native())

    def is_repetition(self, count=3):
        return bool(_O.lib().oc_is_repetition(self._p, int(count)))

    def has_kingside_castling_rights(self, color):
        return bool(_O.lib().oc_has_kingside_castling_rights(self._p, int(bool(color))))

    def has_queenside_castling_rights(self, color):
        return bool(_O.lib().oc_has_queenside_castling_rights(self._p, int(bool(color))))

    def is_check(self):
        return bool(_O.lib().oc_is_check(self._p))

    def outcome(self, claim_draw=False):
        w = _C.c_int(-1)
        o = _O.lib().oc_outcome(self._p, _C.byref(w))
        if not o:
            return None
        return Outcome(o, None if w.value < 0 else bool(w.value))

    def is_game_over(self, claim_draw=False):
        return self.outcome() is not None

    def result(self, claim_draw=False):
        o = self.outcome()
        return o.result() if o else "*"

    def board_fen(self):
        buf = _C.create_string_buffer(80)
        _O.lib().oc_board_fen_pieces(self._p, buf)
        return buf.value.decode()

    def __str__(self):
        rows = []
        for r in range(7, -1, -1):
            row = []
            for f in range(8):
                p = self.piece_at(r * 8 + f)
                row.append(p.symbol() if p else ".")
            rows.append(" ".join(row))
        return "\n".join(rows)
