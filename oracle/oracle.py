"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (sigma_zero_amd) never does.

The C sources restate the reference's CPU algorithm:
  oc_chess.c   python-chess 1.10.0 rules (environment.yml:21; library not in /root/reference)
  oc_tensor.c  /root/reference/chess_tensor.py
  oc_mcts.c    /root/reference/mcts.py, /root/reference/mctsnode.py, sim.py:68 sampler
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

PLANES, ACTIONS, MAX_MOVES = 119, 4672, 256
WHITE, BLACK = 1, 0
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = 1, 2, 3, 4, 5, 6


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("oc_chess.c", "oc_tensor.c", "oc_mcts.c",
                                             "oc_chess.h", "oc_tensor.h", "oc_mcts.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Move(C.Structure):
    _fields_ = [("from_square", C.c_int8), ("to_square", C.c_int8), ("promotion", C.c_int8)]

    def uci(self):
        s = "abcdefgh"[self.from_square & 7] + str((self.from_square >> 3) + 1) + \
            "abcdefgh"[self.to_square & 7] + str((self.to_square >> 3) + 1)
        return s + (" pnbrqk"[self.promotion] if self.promotion else "")

    @staticmethod
    def from_uci(u):
        f = "abcdefgh".index(u[0]) + 8 * (int(u[1]) - 1)
        t = "abcdefgh".index(u[2]) + 8 * (int(u[3]) - 1)
        p = " pnbrqk".index(u[4]) if len(u) > 4 else 0
        return Move(f, t, p)

    def key(self):
        return (self.from_square, self.to_square, self.promotion)

    def __repr__(self):
        return "Move(%s)" % self.uci()


class _GameVT(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("copy", "release", "turn", "move_piece",
                                          "value_and_terminated", "legal_actions", "representation")]


class TableGame(C.Structure):
    _fields_ = [("n_states", C.c_int)] + [(n, C.POINTER(C.c_int)) for n in (
        "terminal", "term_value", "turn", "move_off", "move_from", "move_to", "move_promo", "move_child")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i, u64, dbl, flt = C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_float
    pi, pl, pu8, pf, pd = C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)
    pm = C.POINTER(Move)
    sigs = {
        "oc_board_new": (vp, []), "oc_board_new_960": (vp, [i]), "oc_board_from_fen": (vp, [C.c_char_p, i]),
        "oc_board_copy": (vp, [vp]), "oc_board_free": (None, [vp]),
        "oc_legal_moves": (i, [vp, pm]), "oc_is_legal": (i, [vp, Move]), "oc_push": (None, [vp, Move]), "oc_pop": (None, [vp]),
        "oc_is_check": (i, [vp]), "oc_is_repetition": (i, [vp, i]), "oc_has_legal_en_passant": (i, [vp]),
        "oc_has_kingside_castling_rights": (i, [vp, i]), "oc_has_queenside_castling_rights": (i, [vp, i]),
        "oc_is_insufficient_material": (i, [vp]), "oc_outcome": (i, [vp, pi]), "oc_piece_at": (i, [vp, i, pi]),
        "oc_perft": (u64, [vp, i]), "oc_board_fen_pieces": (None, [vp, C.c_char_p]),
        "oc_board_turn": (i, [vp]), "oc_board_halfmove_clock": (i, [vp]), "oc_board_ply": (i, [vp]),
        "oc_board_ep_square": (i, [vp]), "oc_board_castling_rights": (u64, [vp]), "oc_board_is_chess960": (i, [vp]),
        "oc_board_bitboards": (None, [vp, C.POINTER(C.c_uint64)]),
        "oc_ct_new": (vp, [i, i]), "oc_ct_from_board": (vp, [vp]), "oc_ct_copy": (vp, [vp]), "oc_ct_free": (None, [vp]),
        "oc_ct_move_piece": (i, [vp, Move]), "oc_ct_get_representation": (None, [vp, pu8]),
        "oc_ct_get_value_and_terminated": (i, [vp, pi]),
        "oc_action_to_index": (i, [Move, i]), "oc_index_to_action": (i, [i, i, pm, i, pm]),
        "oc_legal_action_indices": (i, [vp, i, pi, pm]),
        "oc_ucb": (flt, [C.c_long, flt, flt, C.c_long, dbl]),
        "oc_search_begin": (vp, [vp, vp, dbl, i, i, flt]), "oc_search_advance": (i, [vp]),
        "oc_search_leaf_planes": (None, [vp, pu8]), "oc_search_leaf_actions": (i, [vp, pi]),
        "oc_search_pending_game": (vp, [vp]), "oc_search_trace": (i, [vp, pi]),
        "oc_search_feed": (None, [vp, pf, flt]),
        "oc_search_root_children": (i, [vp, pi, pl, pm]), "oc_search_root_stats": (None, [vp, pf, pd]),
        "oc_search_root_visits": (C.c_long, [vp]), "oc_search_root_value_sum": (dbl, [vp]),
        "oc_search_counters": (C.c_long, [vp, i]), "oc_search_free": (None, [vp]),
        "oc_priors_from_policy": (i, [pf, pi, i, i, flt, pf, pi]),
        "oc_search_dump_tree": (i, [vp, i, pi, pi, pl, pd, pf]),
        "oc_table_state_new": (vp, [C.POINTER(TableGame), i]), "oc_table_state_id": (i, [vp]),
        "oc_sample_move": (i, [pl, i, dbl]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.CHESS_VT = C.addressof(_GameVT.in_dll(L, "OC_CHESS_VT"))
    L.TABLE_VT = C.addressof(_GameVT.in_dll(L, "OC_TABLE_VT"))
    _lib = L
    return L


NOISE_REFERENCE = float(np.float32(1.0) - np.float32(2.0 ** -24))   # SURVEY §8(a) A19 / fixture noise_probe


class Board:
    """chess.Board restatement (python-chess 1.10.0 semantics)."""

    def __init__(self, ptr=None, own=True):
        self._p = ptr if ptr is not None else lib().oc_board_new()
        self._own = own

    @classmethod
    def from_chess960_pos(cls, n):
        return cls(lib().oc_board_new_960(int(n)))

    @classmethod
    def from_fen(cls, fen, chess960=False):
        return cls(lib().oc_board_from_fen(fen.encode(), int(chess960)))

    def __del__(self):
        try:
            if getattr(self, "_own", False) and self._p:
                lib().oc_board_free(self._p)
                self._p = None
        except Exception:
            pass

    def legal_moves(self):
        buf = (Move * MAX_MOVES)()
        n = lib().oc_legal_moves(self._p, buf)
        return [Move(buf[k].from_square, buf[k].to_square, buf[k].promotion) for k in range(n)]

    def push(self, m): lib().oc_push(self._p, m)
    def pop(self): lib().oc_pop(self._p)
    def perft(self, d): return int(lib().oc_perft(self._p, d))
    def is_check(self): return bool(lib().oc_is_check(self._p))
    def is_repetition(self, n): return bool(lib().oc_is_repetition(self._p, n))

    def outcome(self):
        w = C.c_int(-1)
        o = lib().oc_outcome(self._p, C.byref(w))
        return o, w.value

    @property
    def turn(self): return lib().oc_board_turn(self._p)
    @property
    def halfmove_clock(self): return lib().oc_board_halfmove_clock(self._p)
    @property
    def ply(self): return lib().oc_board_ply(self._p)
    @property
    def ep_square(self): return lib().oc_board_ep_square(self._p)
    @property
    def castling_rights(self): return int(lib().oc_board_castling_rights(self._p))
    @property
    def chess960(self): return bool(lib().oc_board_is_chess960(self._p))

    def bitboards(self):
        out = (C.c_uint64 * 8)()
        lib().oc_board_bitboards(self._p, out)
        return [int(x) for x in out]

    def has_legal_en_passant(self): return bool(lib().oc_has_legal_en_passant(self._p))
    def has_kingside_castling_rights(self, c): return bool(lib().oc_has_kingside_castling_rights(self._p, int(c)))
    def has_queenside_castling_rights(self, c): return bool(lib().oc_has_queenside_castling_rights(self._p, int(c)))
    def is_insufficient_material(self): return bool(lib().oc_is_insufficient_material(self._p))

    def board_fen(self):
        buf = C.create_string_buffer(80)
        lib().oc_board_fen_pieces(self._p, buf)
        return buf.value.decode()


class ChessTensor:
    """chess_tensor.py:30-188 restatement."""

    def __init__(self, chess960=False, scharnagl=518, ptr=None):
        self._p = ptr if ptr is not None else lib().oc_ct_new(int(chess960), int(scharnagl))

    @classmethod
    def from_fen(cls, fen, chess960=False):
        return cls(ptr=lib().oc_ct_from_board(lib().oc_board_from_fen(fen.encode(), int(chess960))))

    def __del__(self):
        try:
            if self._p:
                lib().oc_ct_free(self._p)
                self._p = None
        except Exception:
            pass

    @property
    def board(self):
        return Board(C.cast(self._p, C.POINTER(C.c_void_p))[0], own=False)

    def copy(self):
        return ChessTensor(ptr=lib().oc_ct_copy(self._p))

    def move_piece(self, m):
        if lib().oc_ct_move_piece(self._p, m) != 0:
            raise ValueError("Invalid move")

    def get_representation(self):
        out = np.zeros(PLANES * 64, dtype=np.uint8)
        lib().oc_ct_get_representation(self._p, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out.reshape(PLANES, 8, 8)

    def get_value_and_terminated(self):
        v = C.c_int(0)
        t = lib().oc_ct_get_value_and_terminated(self._p, C.byref(v))
        return v.value, bool(t)

    def legal_action_indices(self, color=None):
        if color is None:
            color = self.turn
        idx = (C.c_int * MAX_MOVES)()
        mv = (Move * MAX_MOVES)()
        b = C.cast(self._p, C.POINTER(C.c_void_p))[0]
        n = lib().oc_legal_action_indices(b, int(color), idx, mv)
        return [idx[k] for k in range(n)], [Move(mv[k].from_square, mv[k].to_square, mv[k].promotion) for k in range(n)]

    @property
    def turn(self):
        return self.board.turn


def action_to_index(m, color):
    return lib().oc_action_to_index(m, int(color))


def index_to_action(idx, color, queen_promotions=()):
    qp = (Move * max(1, len(queen_promotions)))(*queen_promotions)
    out = Move()
    rc = lib().oc_index_to_action(int(idx), int(color), qp, len(queen_promotions), C.byref(out))
    if rc != 0:
        raise IndexError("index decodes off the board")
    return out


def ucb(vc, vsum, prior, parent_visits, c=2.0):
    return lib().oc_ucb(int(vc), float(np.float32(vsum)), float(np.float32(prior)), int(parent_visits), float(c))


def priors_from_policy(policy, idx, learning, noise_value=NOISE_REFERENCE):
    policy = np.ascontiguousarray(policy, dtype=np.float32)
    idx_a = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.zeros(len(idx), dtype=np.float32)
    pos = np.zeros(len(idx), dtype=np.int32)
    k = lib().oc_priors_from_policy(policy.ctypes.data_as(C.POINTER(C.c_float)), idx_a.ctypes.data_as(C.POINTER(C.c_int)),
                                    len(idx), int(learning), float(noise_value),
                                    out.ctypes.data_as(C.POINTER(C.c_float)), pos.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:k], pos[:k]


def sample_move(visits, u):
    v = np.ascontiguousarray(visits, dtype=np.int64)
    return lib().oc_sample_move(v.ctypes.data_as(C.POINTER(C.c_long)), len(v), float(u))


class Search:
    """MCTS0.search (mcts.py:39-122) as a stepwise coroutine: advance() -> planes -> feed(policy, value)."""

    def __init__(self, game_ptr, vt, c=2.0, num_searches=10, learning=False, noise_value=NOISE_REFERENCE, keep=None):
        self._keep = keep
        self._s = lib().oc_search_begin(game_ptr, vt, float(c), int(num_searches), int(learning), float(noise_value))

    @classmethod
    def on_chess(cls, ct, **kw):
        return cls(ct._p, lib().CHESS_VT, keep=ct, **kw)

    def __del__(self):
        try:
            if self._s:
                lib().oc_search_free(self._s)
                self._s = None
        except Exception:
            pass

    def advance(self): return bool(lib().oc_search_advance(self._s))

    def leaf_planes(self):
        out = np.zeros(PLANES * 64, dtype=np.uint8)
        lib().oc_search_leaf_planes(self._s, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out.reshape(PLANES, 8, 8)

    def leaf_actions(self):
        idx = (C.c_int * MAX_MOVES)()
        n = lib().oc_search_leaf_actions(self._s, idx)
        return [idx[k] for k in range(n)]

    def pending_table_state(self):
        return lib().oc_table_state_id(lib().oc_search_pending_game(self._s))

    def trace(self):
        buf = (C.c_int * 4096)()
        n = lib().oc_search_trace(self._s, buf)
        return [buf[k] for k in range(n)]

    def feed(self, policy, value):
        p = np.ascontiguousarray(policy, dtype=np.float32)
        assert p.size == ACTIONS
        lib().oc_search_feed(self._s, p.ctypes.data_as(C.POINTER(C.c_float)), float(np.float32(value)))

    def root_children(self):
        idx = (C.c_int * MAX_MOVES)()
        vis = (C.c_long * MAX_MOVES)()
        mv = (Move * MAX_MOVES)()
        n = lib().oc_search_root_children(self._s, idx, vis, mv)
        return ([idx[k] for k in range(n)], [vis[k] for k in range(n)],
                [Move(mv[k].from_square, mv[k].to_square, mv[k].promotion) for k in range(n)])

    def root_stats(self):
        n = len(self.root_children()[0])
        pr = np.zeros(max(n, 1), dtype=np.float32)
        ws = np.zeros(max(n, 1), dtype=np.float64)
        lib().oc_search_root_stats(self._s, pr.ctypes.data_as(C.POINTER(C.c_float)), ws.ctypes.data_as(C.POINTER(C.c_double)))
        return pr[:n], ws[:n]

    def dump_tree(self, max_nodes=1 << 20):
        n = lib().oc_search_dump_tree(self._s, 0, None, None, None, None, None)
        n = min(n, max_nodes)
        d = np.zeros(n, np.int32); a = np.zeros(n, np.int32); v = np.zeros(n, np.int64)
        w = np.zeros(n, np.float64); p = np.zeros(n, np.float32)
        lib().oc_search_dump_tree(self._s, n, d.ctypes.data_as(C.POINTER(C.c_int)), a.ctypes.data_as(C.POINTER(C.c_int)),
                                  v.ctypes.data_as(C.POINTER(C.c_long)), w.ctypes.data_as(C.POINTER(C.c_double)),
                                  p.ctypes.data_as(C.POINTER(C.c_float)))
        return d, a, v, w, p

    @classmethod
    def on_table(cls, table_arrays, root_state=0, **kw):
        """table_arrays: dict of int32 arrays terminal, term_value, turn, move_off, move_from, move_to, move_promo, move_child"""
        keep = {k: np.ascontiguousarray(table_arrays[k], dtype=np.int32) for k in
                ("terminal", "term_value", "turn", "move_off", "move_from", "move_to", "move_promo", "move_child")}
        tg = TableGame()
        tg.n_states = len(keep["terminal"])
        for k, arr in keep.items():
            setattr(tg, k, arr.ctypes.data_as(C.POINTER(C.c_int)))
        root = lib().oc_table_state_new(C.byref(tg), int(root_state))
        return cls(root, lib().TABLE_VT, keep=(keep, tg), **kw)

    def root_visits(self): return int(lib().oc_search_root_visits(self._s))
    def root_value_sum(self): return float(lib().oc_search_root_value_sum(self._s))
    def counters(self): return int(lib().oc_search_counters(self._s, 0)), int(lib().oc_search_counters(self._s, 1))

    def action_probs(self):
        """mcts.py:113-122: {action index: visit_count / sum}"""
        idx, vis, _ = self.root_children()
        tot = sum(vis)
        return {a: v / tot for a, v in zip(idx, vis)}


def search_with_evaluator(ct, evaluator, **kw):
    """Run a whole search; evaluator(planes uint8[119,8,8]) -> (policy f32[4672], value)."""
    s = Search.on_chess(ct, **kw)
    while s.advance():
        p, v = evaluator(s.leaf_planes())
        s.feed(p, v)
    return s
