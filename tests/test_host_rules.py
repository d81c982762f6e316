"""Product rules code (csrc/sz_chess.h through the szh_* host mirror; the same functions the HIP kernels run)
against the independent oracle (oracle/oc_chess.c, oc_tensor.c) and public perft tables.  CPU only."""
import random

import numpy as np
import pytest

import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
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
    ct = sz.ChessTensor(chess960=c960, fen=fen) if fen else sz.ChessTensor()
    assert [ct.perft(d + 1) for d in range(len(want))] == want


def test_scharnagl_matches_oracle():
    for n in list(range(0, 960, 7)) + [518, 959]:
        ct = sz.ChessTensor(chess960=True, scharnagl=n)
        ob = O.Board.from_chess960_pos(n)
        bb = ct.board.bitboards()
        assert bb[:6] == ob.bitboards()[:6] and bb[6] == ob.bitboards()[6]
        assert bb[9] == ob.castling_rights


def _compare_state(ct, oct_, tag):
    ob = oct_.board
    # legal move set in action-index order
    mine = ct.legal_action_indices()
    theirs, their_moves = oct_.legal_action_indices()
    assert mine == theirs, tag
    assert [m.uci() for m in ct.get_moves()] == [m.uci() for m in their_moves], tag
    # encoder
    assert np.array_equal(ct.get_representation().numpy().astype(np.uint8), oct_.get_representation()), tag
    # terminal
    v, t = ct.get_value_and_terminated()
    ov, ot = oct_.get_value_and_terminated()
    assert (v, t) == (ov, ot), tag
    st = ct.board._st()
    assert st[0] == ob.turn and st[2] == min(ob.halfmove_clock, 255) and st[3] == ob.ep_square, tag
    assert st[10] == int(ob.has_legal_en_passant()), tag
    assert st[7] >= 1 if ob.is_repetition(2) else st[7] == 0, tag
    assert (st[7] >= 2) == ob.is_repetition(3), tag
    if t:
        assert st[9] == ob.outcome()[0], tag
    return t


@pytest.mark.parametrize("c960", [False, True])
def test_random_games_match_oracle(c960):
    rng = random.Random(2024 + c960)
    n_plies = 0
    for game in range(12):
        n = rng.randrange(960)
        ct = sz.ChessTensor(chess960=c960, scharnagl=n)
        oct_ = O.ChessTensor(chess960=c960, scharnagl=n)
        for ply in range(400):
            if _compare_state(ct, oct_, "game %d ply %d" % (game, ply)):
                break
            idx, moves = oct_.legal_action_indices()
            k = rng.randrange(len(idx))
            # bias towards shuffling so that repetitions and the no-progress clock get exercised
            if rng.random() < 0.35:
                quiet = [i for i, m in enumerate(moves) if ct.board.bitboards()[0] >> m.from_square & 1 == 0]
                if quiet:
                    k = rng.choice(quiet)
            ct.push_action(idx[k])
            oct_.move_piece(moves[k])
            n_plies += 1
    assert n_plies > 1500


def test_fivefold_and_seventyfive():
    ct = sz.ChessTensor()
    for rep in range(4):
        for u in ("g1f3", "g8f6", "f3g1", "f6g8"):
            assert not ct.board.is_game_over()
            ct.move_piece(sz.Move.from_uci(u))
    assert ct.board.is_game_over() and ct.board.result() == "1/2-1/2" and ct.board.outcome().termination == 5
    ct = sz.ChessTensor(fen="8/8/5k2/8/8/3KR3/8/8 w - - 149 100")
    assert not ct.board.is_game_over()
    ct.move_piece(sz.Move.from_uci("e3e4"))
    assert ct.board.is_game_over() and ct.board.outcome().termination == 4


def test_invalid_move_raises():
    ct = sz.ChessTensor()
    with pytest.raises(ValueError, match="Invalid move"):
        ct.move_piece(sz.Move.from_uci("e2e5"))
    with pytest.raises(ValueError, match="Invalid move"):
        ct.move_piece(sz.Move.from_uci("e1g1"))


def test_checkmate_value():
    ct = sz.ChessTensor()
    for u in ("f2f3", "e7e5", "g2g4", "d8h4"):
        ct.move_piece(sz.Move.from_uci(u))
    assert ct.get_value_and_terminated() == (-1, True)
    assert ct.board.result() == "0-1"


def test_codec_matches_oracle_tables(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "codec_tables.npz"))
    enc, dec = z["encode"], z["decode"]
    for color, f, t, p, idx in enc:
        assert sz.chess_tensor.action_index(sz.Move(int(f), int(t), int(p) or None), bool(color)) == int(idx)
    for color, idx, f, t, p, use_qp in dec:
        qp = {sz.Move(int(f), int(t), 5).uci(): True} if use_qp else {}
        m = sz.chess_tensor.index_to_move(int(idx), bool(color), qp)
        assert (m.from_square, m.to_square, m.promotion or 0) == (int(f), int(t), int(p))


def test_library_exports_every_declared_symbol():
    import re, os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "sigmazero.h")).read()
    declared = set(re.findall(r"\b(szh?_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sz_config", "sz_stats", "sz_engine", "szh_game"}
    lib = N.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared <= set(N.EXPORTS) | {"sz_error_string"}


FENS = [
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8",
    "8/P7/8/8/8/8/7k/K7 w - - 0 1",
    "8/8/4k3/8/8/3KR3/8/8 b - - 140 100",
    "r3k2r/8/8/8/8/8/8/R3K2R b KQkq - 3 20",
]


@pytest.mark.parametrize("fen", FENS)
def test_fen_positions_match_oracle(fen):
    rng = random.Random(hash(fen) & 0xFFFF)
    ct = sz.ChessTensor(fen=fen)
    oct_ = O.ChessTensor.from_fen(fen)
    for ply in range(60):
        if _compare_state(ct, oct_, "%s ply %d" % (fen, ply)):
            break
        idx, moves = oct_.legal_action_indices()
        k = rng.randrange(len(idx))
        ct.push_action(idx[k])
        oct_.move_piece(moves[k])


def test_dropin_module_names_resolve():
    """the reference's own import lines (train_RL.py:2-6, eval.py, play.py) work against dropin/"""
    import importlib, os, sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin")
    sys.path.insert(0, d)
    try:
        for name in ("chess", "network", "chess_tensor", "mctsnode", "mcts", "sim"):
            sys.modules.pop(name, None)
        net = importlib.import_module("network")
        ct = importlib.import_module("chess_tensor")
        chess = importlib.import_module("chess")
        assert importlib.import_module("sim").generate_training_data and importlib.import_module("mcts").MCTS0
        assert net.policyNN is sz.policyNN and ct.ChessTensor is sz.ChessTensor
        mask, qp = ct.actionsToTensor([chess.Move.from_uci("e2e4"), chess.Move.from_uci("a7a8q")], chess.WHITE)
        assert mask.nonzero().flatten().tolist() == [8, 116] and qp == {"a7a8q": True}
        assert [m.uci() for m in ct.tensorToAction(mask, chess.WHITE, qp)] == ["a7a8q", "e2e4"]
    finally:
        sys.path.remove(d)
        for name in ("chess", "network", "chess_tensor", "mctsnode", "mcts", "sim"):
            sys.modules.pop(name, None)


def test_maximum_fanout_position():
    """218 legal moves — the known maximum; the engine's child span (SZ_MAX_MOVES) is sized for it"""
    fen = "R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1"
    ct = sz.ChessTensor(fen=fen)
    oct_ = O.ChessTensor.from_fen(fen)
    assert len(ct.legal_action_indices()) == 218 == len(oct_.legal_action_indices()[0])
    assert ct.legal_action_indices() == oct_.legal_action_indices()[0]


def test_promotion_and_en_passant_edge_cases():
    for fen in ("8/P6k/8/8/8/8/p6K/8 w - - 0 1", "8/P6k/8/8/8/8/p6K/8 b - - 0 1",
                "rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f6 0 3",      # two ep candidates listed, one legal square
                "8/8/8/8/k2Pp2Q/8/8/3K4 b - d3 0 1",                                    # ep capture would expose the king: illegal
                "8/8/3p4/KPp4r/1R3p1k/8/4P1P1/8 w - c6 0 2"):                          # perft-3 classic: ep pin on the rank
        ct = sz.ChessTensor(fen=fen)
        oct_ = O.ChessTensor.from_fen(fen)
        assert _compare_state(ct, oct_, fen) is False


def test_bit_parallel_plane_extraction_equals_its_definition():
    """sz_lane_plane_bits (what the kernels' move generator uses: shifts, ray masks and shift-folds per lane) against
    sz_lane_plane_bit (the coordinate arithmetic it replaces) on the legal-target sets of ~6,000 positions: random playouts from
    classical and Chess960 starts (promotions, castling, en passant included) and the KAT positions, both colours."""
    from sigma_zero_amd import _native as N
    rng = random.Random(5)
    checked = 0
    games = [sz.ChessTensor(chess960=bool(i & 1), scharnagl=rng.randrange(960) if (i & 1) else 518) for i in range(40)]
    games += [sz.ChessTensor(fen=f) for _, f, c, _ in PERFT if f and not c]
    for ct in games:
        for ply in range(150):
            assert N.lib().szh_plane_bits_mismatches(ct._g) == 0
            checked += 1
            if ct.board.is_game_over():
                break
            acts = ct.legal_action_indices()
            ct.push_action(acts[rng.randrange(len(acts))])
    assert checked > 4000
