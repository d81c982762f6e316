"""Oracle chess rules (python-chess 1.10.0 restatement) vs PUBLIC perft known-answer tables.

The reference holds no golden vectors for chess rules: this part of the oracle is
"parity unpinned" by the reference and pinned by these public numbers instead.
"""
import pytest

from oracle import oracle as O

PERFT = [
    ("startpos", None, False, [20, 400, 8902, 197281, 4865609]),
    ("kiwipete", "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", False, [48, 2039, 97862, 4085603]),
    ("pos3", "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", False, [14, 191, 2812, 43238, 674624, 11030083]),
    ("pos4", "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", False, [6, 264, 9467, 422333, 15833292]),
    ("pos5", "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8", False, [44, 1486, 62379, 2103487]),
    ("pos6", "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10", False, [46, 2079, 89890, 3894594]),
    ("c960", "bqnb1rkr/pp3ppp/3ppn2/2p5/5P2/P2P4/NPP1P1PP/BQ1BNRKR w HFhf - 2 9", True, [21, 528, 12189, 326672, 8146062]),
]


@pytest.mark.parametrize("name,fen,c960,want", PERFT, ids=[p[0] for p in PERFT])
def test_perft(name, fen, c960, want):
    b = O.Board() if fen is None else O.Board.from_fen(fen, chess960=c960)
    assert [b.perft(d + 1) for d in range(len(want))] == want


def test_scharnagl_kats():
    assert O.Board.from_chess960_pos(518).board_fen().split("/")[7] == "RNBQKBNR"
    assert O.Board.from_chess960_pos(0).board_fen().split("/")[7] == "BBQNNRKR"
    assert O.Board.from_chess960_pos(959).board_fen().split("/")[7] == "RKRNNQBB"
    seen = set()
    for n in range(960):
        rank = O.Board.from_chess960_pos(n).board_fen().split("/")[7]
        assert sorted(rank) == sorted("RNBQKBNR")
        k, r1, r2 = rank.index("K"), rank.index("R"), rank.rindex("R")
        assert r1 < k < r2
        b1, b2 = [i for i, c in enumerate(rank) if c == "B"]
        assert (b1 + b2) % 2 == 1
        seen.add(rank)
    assert len(seen) == 960


def _play(b, ucis):
    for u in ucis:
        m = O.Move.from_uci(u)
        assert any(m.key() == x.key() for x in b.legal_moves()), u
        b.push(m)


def test_castling_forms():
    # classical boards spell castling e1g1 / e1c1, Chess960 boards king-takes-rook (python-chess 1.10)
    b = O.Board.from_fen("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1")
    ucis = {m.uci() for m in b.legal_moves()}
    assert {"e1g1", "e1c1"} <= ucis and "e1h1" not in ucis
    b = O.Board.from_fen("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", chess960=True)
    ucis = {m.uci() for m in b.legal_moves()}
    assert {"e1h1", "e1a1"} <= ucis and "e1g1" not in ucis
    _play(b, ["e1h1"])
    assert b.board_fen().split("/")[7] == "R4RK1"
    assert not b.has_kingside_castling_rights(1) and b.has_kingside_castling_rights(0)


def test_repetition_and_fivefold():
    b = O.Board()
    shuffle = ["g1f3", "g8f6", "f3g1", "f6g8"]
    _play(b, shuffle)
    assert b.is_repetition(2) and not b.is_repetition(3)
    _play(b, shuffle)
    assert b.is_repetition(3)
    _play(b, shuffle)
    assert b.outcome()[0] == 0
    _play(b, shuffle)
    assert b.is_repetition(5) and b.outcome() == (5, -1)


def test_repetition_blocked_by_irreversible():
    b = O.Board()
    _play(b, ["e2e4", "e7e5", "g1f3", "g8f6", "f3g1", "f6g8"])
    assert b.is_repetition(2)                 # position after 1.e4 e5 repeated
    _play(b, ["g1f3", "g8f6", "f3g1", "f6g8"])
    assert b.is_repetition(3)
    # losing a castling right is irreversible
    b = O.Board()
    _play(b, ["e2e4", "e7e5", "e1e2", "e8e7", "e2e1", "e7e8"])
    assert not b.is_repetition(2)
    _play(b, ["e1e2", "e8e7", "e2e1", "e7e8"])
    assert b.is_repetition(2) and not b.is_repetition(3)


def test_terminal_kinds():
    b = O.Board()
    _play(b, ["f2f3", "e7e5", "g2g4", "d8h4"])
    assert b.outcome() == (1, 0)                                    # fool's mate, black wins
    assert O.Board.from_fen("7k/5Q2/6K1/8/8/8/8/8 b - - 0 1").outcome() == (3, -1)      # stalemate
    assert O.Board.from_fen("8/8/4k3/8/8/3K4/8/8 w - - 0 1").outcome() == (2, -1)       # K v K
    assert O.Board.from_fen("8/8/4k3/8/8/3KN3/8/8 w - - 0 1").outcome() == (2, -1)      # KN v K
    assert O.Board.from_fen("8/8/4k3/8/8/3KNN2/8/8 w - - 0 1").outcome()[0] == 0        # KNN v K is not automatic
    assert O.Board.from_fen("8/8/4kb2/8/8/3BK3/8/8 w - - 0 1").outcome()[0] == 0        # opposite-colour bishops (f6 dark, d3 light)
    assert O.Board.from_fen("8/8/4kb2/8/8/3KB3/8/8 w - - 0 1").outcome() == (2, -1)     # same-colour bishops (f6, e3 dark)
    assert O.Board.from_fen("8/8/5k2/8/8/3KR3/8/8 w - - 149 100").outcome()[0] == 0
    assert O.Board.from_fen("8/8/5k2/8/8/3KR3/8/8 w - - 150 100").outcome() == (4, -1)  # 75-move rule


def test_en_passant_legality_in_key():
    b = O.Board()
    _play(b, ["e2e4", "a7a6", "e4e5", "d7d5"])
    assert b.ep_square == 43 and b.has_legal_en_passant()
    assert "e5d6" in {m.uci() for m in b.legal_moves()}
    b = O.Board()
    _play(b, ["e2e4"])
    assert b.ep_square == 20 and not b.has_legal_en_passant()        # set after any double push
