"""Host mirror of /root/reference/chess_tensor.py (ChessTensor :30-188, codecs :190-410).

python-chess is not a dependency: `Move` and `Board` below carry the small part of its API that the
reference's hot path touches (SURVEY.md §8(c) call sites); the rules are the engine's own
(csrc/sz_chess.h, the same code the HIP kernels execute, reached through the szh_* C ABI).
"""
import ctypes as C
import random
from typing import Dict, List, Union

import numpy as np
import torch

from . import _native as N

WHITE, BLACK = True, False
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = range(1, 7)
_FILES, _RANKS, _PIECE_LETTERS = "abcdefgh", "12345678", " pnbrqk"


class Move:
    """chess.Move stand-in: from_square, to_square, promotion (None or KNIGHT..QUEEN), uci()."""
    __slots__ = ("from_square", "to_square", "promotion")

    def __init__(self, from_square, to_square, promotion=None):
        self.from_square, self.to_square, self.promotion = int(from_square), int(to_square), (promotion or None)

    def uci(self):
        s = _FILES[self.from_square & 7] + _RANKS[self.from_square >> 3] + _FILES[self.to_square & 7] + _RANKS[self.to_square >> 3]
        return s + (_PIECE_LETTERS[self.promotion] if self.promotion else "")

    @classmethod
    def from_uci(cls, uci):
        f = _FILES.index(uci[0]) + 8 * _RANKS.index(uci[1])
        t = _FILES.index(uci[2]) + 8 * _RANKS.index(uci[3])
        return cls(f, t, _PIECE_LETTERS.index(uci[4]) if len(uci) > 4 else None)

    def __eq__(self, other):
        return isinstance(other, Move) and (self.from_square, self.to_square, self.promotion) == (other.from_square, other.to_square, other.promotion)

    def __hash__(self):
        return hash((self.from_square, self.to_square, self.promotion))

    def __repr__(self):
        return "Move.from_uci(%r)" % self.uci()

    def __str__(self):
        return self.uci()


# ----------------------------------------------------------------------------- action codec
_DIRS = ((0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1))       # clockwise from "up", chess_tensor.py:225-234
_KNIGHTS = ((1, -2), (2, -1), (2, 1), (1, 2), (-1, 2), (-2, 1), (-2, -1), (-1, -2))  # chess_tensor.py:236-245


def action_index(move: Move, color=WHITE) -> int:
    """Index of `move` in the 73x8x8 action space seen from `color` (actionToTensor, chess_tensor.py:221-306)."""
    idx = N.lib().szh_move_to_action(move.from_square, move.to_square, move.promotion or 0, 1 if color else 0)
    if idx < 0:
        raise KeyError("move %s is neither a queen-line nor a knight move" % move.uci())
    return idx


def actionToTensor(move: Move, color=WHITE, prob: float = 1) -> torch.Tensor:
    out = torch.zeros(N.SZ_ACTIONS)
    out[action_index(move, color)] = prob
    return out


def actionsToTensor(valid_moves: Union[List[Move], Dict[Move, float]], color=WHITE):
    """Mask (list) or probability vector (dict) over 4672 actions + dict of queen promotions (chess_tensor.py:190-218)."""
    out = torch.zeros(N.SZ_ACTIONS)
    queen_promotion = {}
    weighted = isinstance(valid_moves, dict)
    for mv in valid_moves:
        if mv.promotion == QUEEN:
            queen_promotion[mv.uci()] = True
        out[action_index(mv, color)] += (valid_moves[mv] if weighted else 1)
    return out, queen_promotion


def index_to_move(idx: int, color=WHITE, queen_promotion=None) -> Move:
    plane, cell = divmod(int(idx), 64)
    row, col = divmod(cell, 8)
    promo = None
    if plane < 56:
        d, n = divmod(plane, 7)
        tcol, trow = col + _DIRS[d][0] * (n + 1), row + _DIRS[d][1] * (n + 1)
    elif plane < 64:
        tcol, trow = col + _KNIGHTS[plane - 56][0], row + _KNIGHTS[plane - 56][1]
    else:
        q = plane - 64
        trow = row - 1
        tcol = col + (1 if q % 3 == 1 else (-1 if q % 3 == 2 else 0))
        promo = (KNIGHT, BISHOP, ROOK)[q // 3]
    if not (0 <= trow < 8 and 0 <= tcol < 8):
        raise IndexError("action index %d leaves the board" % idx)
    flip = 56 if color else 7
    mv = Move((row * 8 + col) ^ flip, (trow * 8 + tcol) ^ flip, promo)
    if plane < 56 and queen_promotion and queen_promotion.get(mv.uci() + "q", False):
        mv.promotion = QUEEN
    return mv


def tensorToAction(moves: torch.Tensor, color=WHITE, queen_promotion: dict = {}) -> List[Move]:
    """Moves of the non-zero entries, ascending index (chess_tensor.py:309-410)."""
    return [index_to_move(i, color, queen_promotion) for i in moves.flatten().nonzero().flatten().tolist()]


# ----------------------------------------------------------------------------- board / game
class _Outcome:
    def __init__(self, kind, winner):
        self.termination, self.winner = kind, winner


class Board:
    """The slice of chess.Board the hot path uses: turn, legal_moves, is_game_over(), outcome(), result(),
    halfmove_clock, move_stack length.  A view onto the owning ChessTensor's native game."""

    def __init__(self, owner):
        self._o = owner

    def _st(self):
        out = (C.c_int32 * 12)()
        N.lib().szh_status(self._o._g, out)
        return list(out)

    @property
    def turn(self):
        return bool(self._st()[0])

    @property
    def halfmove_clock(self):
        return self._st()[2]

    @property
    def ply(self):
        return self._st()[1]

    @property
    def legal_moves(self):
        return self._o.get_moves()

    def is_check(self):
        return bool(self._st()[6])

    def is_game_over(self):
        return bool(self._st()[4])

    def outcome(self):
        st = self._st()
        if not st[4]:
            return None
        winner = (not bool(st[0])) if st[9] == 1 else None
        return _Outcome(st[9], winner)

    def result(self):
        o = self.outcome()
        if o is None:
            return "*"
        return "1/2-1/2" if o.winner is None else ("1-0" if o.winner else "0-1")

    def bitboards(self):
        out = (C.c_uint64 * 10)()
        N.lib().szh_bitboards(self._o._g, out)
        return [int(x) for x in out]

    def __repr__(self):
        st = self._st()
        return "Board(turn=%s, ply=%d, halfmove=%d)" % ("white" if st[0] else "black", st[1], st[2])


class ChessTensor:
    """chess_tensor.py:30-188.  One self-play game: 8-step history, 119-plane encoder, terminal test."""

    def __init__(self, chess960=False, scharnagl=None, fen=None):
        self.M, self.T, self.L = 14, 8, 7
        self._g = None
        self.start_board(chess960=chess960, scharnagl=scharnagl, fen=fen)

    def start_board(self, chess960=False, scharnagl=None, fen=None):
        if self._g:
            N.lib().szh_game_free(self._g)
        self.chess960 = bool(chess960)
        if fen is not None:
            self._g = N.lib().szh_game_from_fen(fen.encode(), int(chess960))
            self.scharnagl = None
        elif chess960:
            # chess_tensor.py:69: chess.Board.from_chess960_pos(random.randint(0, 959)) — python's global `random`
            self.scharnagl = random.randint(0, 959) if scharnagl is None else int(scharnagl)
            self._g = N.lib().szh_game_new(1, self.scharnagl)
        else:
            self.scharnagl = None
            self._g = N.lib().szh_game_new(0, -1)
        self.board = Board(self)

    def __del__(self):
        try:
            if self._g:
                N.lib().szh_game_free(self._g)
                self._g = None
        except Exception:
            pass

    def __deepcopy__(self, memo):
        return self.copy()

    def copy(self):
        other = ChessTensor.__new__(ChessTensor)
        other.M, other.T, other.L = self.M, self.T, self.L
        other.chess960, other.scharnagl = self.chess960, self.scharnagl
        other._g = N.lib().szh_game_copy(self._g)
        other.board = Board(other)
        return other

    # -- moves
    def legal_action_indices(self) -> List[int]:
        buf = (C.c_int32 * N.SZ_MAX_MOVES)()
        n = N.lib().szh_legal_actions(self._g, buf)
        return [buf[i] for i in range(n)]

    def move_from_index(self, idx) -> Move:
        f, t, p = C.c_int32(), C.c_int32(), C.c_int32()
        N.check(N.lib().szh_action_to_move(self._g, int(idx), C.byref(f), C.byref(t), C.byref(p)), "szh_action_to_move")
        return Move(f.value, t.value, p.value or None)

    def get_moves(self) -> List[Move]:
        return [self.move_from_index(i) for i in self.legal_action_indices()]

    def get_valid_moves(self, state=None) -> List[Move]:
        return self.get_moves()

    def move_piece(self, move: Move):
        if isinstance(move, str):
            move = Move.from_uci(move)
        if N.lib().szh_push_move(self._g, move.from_square, move.to_square, move.promotion or 0) != N.SZ_OK:
            raise ValueError("Invalid move")

    def push_action(self, idx: int):
        if N.lib().szh_push_action(self._g, int(idx)) != N.SZ_OK:
            raise ValueError("Invalid move")

    # -- encoder / terminal
    def get_representation(self) -> torch.Tensor:
        out = np.zeros((N.SZ_PLANES, 8, 8), dtype=np.uint8)
        N.lib().szh_planes(self._g, out.ctypes.data_as(C.c_void_p))
        return torch.from_numpy(out).to(torch.bool)

    def get_value_and_terminated(self):
        st = self.board._st()
        if st[4]:
            return (-1 if st[5] else 0), True
        return 0, False

    # -- trivial helpers kept for API compatibility (chess_tensor.py:149-188)
    def get_initial_state(self):
        return self.board

    def get_opponent(self, player):
        return -player

    def get_opponent_value(self, value):
        return -value

    def change_perspective(self, state, player):
        return state * player

    def export_ring(self):
        ring = (C.c_uint8 * (N.SZ_RING * N.SZ_POS_BYTES))()
        ply, c960 = C.c_int32(), C.c_int32()
        N.check(N.lib().szh_export(self._g, ring, C.byref(ply), C.byref(c960)), "szh_export")
        return ring, ply.value, bool(c960.value)

    def perft(self, depth):
        return int(N.lib().szh_perft(self._g, int(depth)))
