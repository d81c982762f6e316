"""Pin the oracle (oracle/*.c) to the REFERENCE's own arithmetic through the committed golden vectors
(tests/golden/*.npz, produced by tests/golden/gen_reference_fixtures.py from /root/reference)."""
import os

import numpy as np
import pytest

from oracle import oracle as O


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}          # materialise once (NpzFile re-inflates per access)


def test_ucb_bitwise(golden_dir):
    z = _load(golden_dir, "ucb_vectors.npz")
    off = z["offsets"]
    bad = 0
    for c in range(len(off) - 1):
        lo, hi = off[c], off[c + 1]
        got = np.array([O.ucb(z["vc"][k], np.float32(z["vsum"][k]), z["prior"][k], z["parent_visits"][c], float(z["C"][c]))
                        for k in range(lo, hi)], dtype=np.float32)
        want = z["ucb"][lo:hi]
        bad += int((got.view(np.uint32) != want.view(np.uint32)).sum())
        assert int(np.argmax(got)) == int(z["argmax"][c])       # first max wins, like torch.argmax
    assert bad == 0


def test_codec_tables(golden_dir):
    z = _load(golden_dir, "codec_tables.npz")
    for color, f, t, p, idx in z["encode"]:
        assert O.action_to_index(O.Move(int(f), int(t), int(p)), int(color)) == int(idx)
    for color, idx, f, t, p, use_qp in z["decode"]:
        qp = [O.Move(int(f), int(t), O.QUEEN)] if use_qp else []
        m = O.index_to_action(int(idx), int(color), qp)
        assert m.key() == (int(f), int(t), int(p))
    # distinct indices for distinct moves of one colour
    for color in (0, 1):
        rows = z["encode"][z["encode"][:, 0] == color]
        geo = rows[rows[:, 3] == 0]
        assert len(set(geo[:, 4].tolist())) == len(geo) == 1792
    # actionsToTensor({move: prob}) training targets (train_RL.py:28)
    for moves, probs, vec in zip(z["target_moves"], z["target_probs"], z["target_vecs"]):
        mine = np.zeros(O.ACTIONS, dtype=np.float32)
        for (color, f, t, p), pr in zip(moves, probs):
            mine[O.action_to_index(O.Move(int(f), int(t), int(p)), int(color))] += np.float32(pr)
        assert np.array_equal(mine, vec)


def test_codec_kats():
    kat = {("e2e4", 1): 116, ("g1f3", 1): 4094, ("e7e5", 0): 115, ("g8f6", 0): 3641, ("e1g1", 1): 1020,
           ("e1h1", 1): 1084, ("a7a8q", 1): 8, ("a7b8n", 1): 4168, ("a2a1r", 0): 4495}
    for (u, c), want in kat.items():
        assert O.action_to_index(O.Move.from_uci(u), c) == want


def test_noise_constant(golden_dir):
    z = _load(golden_dir, "noise_probe.npz")
    assert np.all(z["draws"] == np.float32(O.NOISE_REFERENCE))


def test_sampler(golden_dir):
    z = _load(golden_dir, "sampler_golden.npz")
    off = z["offsets"]
    for c in range(len(off) - 1):
        assert O.sample_move(z["visits"][off[c]:off[c + 1]], z["u"][c]) == int(z["choice"][c])


def _cases(golden_dir):
    z = _load(golden_dir, "search_traces.npz")
    for i in range(int(z["n_cases"])):
        yield i, {k[len("c%d_" % i):]: z[k] for k in z if k.startswith("c%d_" % i)}


def _replay(case):
    s = O.Search.on_table(case, c=2.0, num_searches=int(case["S"]), learning=bool(case["learning"]))
    off = case["move_off"]
    while s.advance():
        sid = s.pending_table_state()
        pol = np.zeros(O.ACTIONS, dtype=np.float32)
        lo, hi = off[sid], off[sid + 1]
        pol[case["move_index"][lo:hi]] = case["move_policy"][lo:hi]
        assert sorted(case["move_index"][lo:hi].tolist()) == s.leaf_actions()
        s.feed(pol, case["nn_value"][sid])
    return s


def test_search_traces_match_reference(golden_dir):
    n_exact = 0
    for i, case in _cases(golden_dir):
        s = _replay(case)
        d, a, v, w, p = s.dump_tree()
        tag = "case %d S=%d learning=%d mode=%s" % (i, case["S"], case["learning"], case["mode"])
        assert len(d) == len(case["tree_depth"]), tag
        assert np.array_equal(d, case["tree_depth"]) and np.array_equal(a, case["tree_action"]), tag
        assert np.array_equal(v, case["tree_visits"]), tag
        assert s.root_visits() == int(case["root_visits"]), tag
        if str(case["mode"]) == "dyadic":
            # sums of dyadic rationals are exact in any order: priors and value sums must be BIT-identical
            assert np.array_equal(p.view(np.uint32), case["tree_prior"].view(np.uint32)), tag
            assert np.array_equal(w, case["tree_value_sum"]), tag
            assert s.root_value_sum() == float(case["root_value_sum"]), tag
        else:
            # torch.sum's reduction order differs from the oracle's fixed order: <= 1 ulp on the normaliser
            assert np.allclose(p, case["tree_prior"], rtol=3e-7, atol=0), tag
            assert np.allclose(w, case["tree_value_sum"], rtol=0, atol=1e-12), tag
        idx, vis, _ = s.root_children()
        if str(case["error"]):
            continue                               # reference raised (num_searches == 1): no readout to compare
        assert idx == case["root_actions"].tolist(), tag
        if len(vis) and sum(vis):
            probs = np.array(vis, dtype=np.float64) / sum(vis)
            assert np.array_equal(probs, case["root_probs"]), tag
        n_exact += 1
    assert n_exact >= 30


def test_num_searches_one_has_no_visits(golden_dir):
    # mcts.py:118-120 divides by zero when num_searches == 1; the oracle exposes the zero total instead
    for i, case in _cases(golden_dir):
        if str(case["error"]) == "ZeroDivisionError":
            s = _replay(case)
            assert sum(s.root_children()[1]) == 0
            return
    pytest.fail("fixture with num_searches == 1 missing")
