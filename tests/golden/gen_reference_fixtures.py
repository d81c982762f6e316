"""Generate golden vectors by running the REFERENCE's own arithmetic (build container only).

    python tests/golden/gen_reference_fixtures.py      # writes tests/golden/*.npz

What runs, unmodified, from /root/reference (read-only):
    mctsnode.Node (get_ucb / select / expand / backpropagate)         mctsnode.py:7-63
    mcts.MCTS0.search                                                  mcts.py:39-122
    chess_tensor.actionToTensor / actionsToTensor / tensorToAction     chess_tensor.py:190-410
    network.policyNN                                                   network.py:89-192
python-chess is not installed here, so `import chess` is satisfied by the name-only stand-in in
tests/golden/_chess_stub/ (no rules inside).  ChessTensor / sim.play_game therefore cannot run and
are NOT covered by these fixtures (chess-rule parity is pinned by public perft tables instead).

The fixtures are data only (inputs + expected outputs).  /root/reference does not exist on the GPU
box: tests read the .npz files, never this script's imports.
"""
import os
import sys
import random

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_chess_stub"))
sys.path.insert(1, "/root/reference")

import numpy as np
import torch

import chess                      # the stand-in
import mctsnode                   # reference
import mcts as ref_mcts           # reference
import chess_tensor as ref_ct     # reference (codec functions only)
import network as ref_net         # reference

OUT = HERE
torch.set_num_threads(1)


# --------------------------------------------------------------------------- 1. UCB vectors
def gen_ucb(n_cases=3000, seed=1234):
    rng = np.random.RandomState(seed)
    off = [0]
    vc_all, vsum_all, prior_all, ucb_all, npar, cpar, arg = [], [], [], [], [], [], []
    for case in range(n_cases):
        K = int(rng.randint(1, 60))
        N = int(rng.randint(1, 5000))
        C = [2, 2, 2, 1, 4][case % 5]
        vc = rng.randint(0, max(2, N // 2), size=K)
        if case % 7 == 0:
            vc[:] = 0
        # value_sum is a python float (double) running sum of f32-valued terms
        vsum = np.array([float(np.sum(rng.uniform(-1, 1, size=int(v)).astype(np.float32).astype(np.float64))) for v in vc])
        if case % 11 == 0:
            vsum = -vc.astype(np.float64)                       # all losses
        prior = rng.dirichlet(np.full(K, 0.5)).astype(np.float32)
        if case % 13 == 0:
            prior[:] = np.float32(1.0 / K)                      # exact ties -> first max must win
        parent = mctsnode.Node(None, {"C": C}, None)
        parent.visit_count = N
        for k in range(K):
            ch = mctsnode.Node(None, {"C": C}, None, parent=parent, prior=float(prior[k]))
            ch.visit_count = int(vc[k])
            ch.value_sum = float(vsum[k])
            parent.children.append(ch)
        u = parent.get_ucb(torch.tensor([c.visit_count for c in parent.children]),
                           torch.tensor([c.value_sum for c in parent.children]),
                           torch.tensor([c.prior for c in parent.children]))
        assert u.dtype == torch.float32
        sel = parent.select()
        arg.append(parent.children.index(sel))
        vc_all.append(vc.astype(np.int64)); vsum_all.append(vsum); prior_all.append(prior)
        ucb_all.append(u.numpy().copy()); npar.append(N); cpar.append(C)
        off.append(off[-1] + K)
    np.savez_compressed(os.path.join(OUT, "ucb_vectors.npz"),
                        offsets=np.array(off, np.int64), vc=np.concatenate(vc_all), vsum=np.concatenate(vsum_all),
                        prior=np.concatenate(prior_all), ucb=np.concatenate(ucb_all),
                        parent_visits=np.array(npar, np.int64), C=np.array(cpar, np.int64), argmax=np.array(arg, np.int64))
    print("ucb_vectors:", n_cases, "cases,", off[-1], "children")


# --------------------------------------------------------------------------- 2. codec tables
def geometric_moves():
    """all (from,to) pairs along queen lines or knight jumps"""
    out = []
    for f in range(64):
        ff, fr = f % 8, f // 8
        for t in range(64):
            if t == f:
                continue
            tf, tr = t % 8, t // 8
            dx, dy = tf - ff, tr - fr
            if dx == 0 or dy == 0 or abs(dx) == abs(dy) or sorted((abs(dx), abs(dy))) == [1, 2]:
                out.append((f, t))
    return out


def promotion_moves(color):
    out = []
    fr, tr = (6, 7) if color else (1, 0)
    for ff in range(8):
        for tf in (ff - 1, ff, ff + 1):
            if 0 <= tf < 8:
                for p in (chess.KNIGHT, chess.BISHOP, chess.ROOK, chess.QUEEN):
                    out.append((fr * 8 + ff, tr * 8 + tf, p))
    return out


def gen_codec():
    rows = []          # color, from, to, promo, index
    dec = []           # color, index, from, to, promo (tensorToAction of the one-hot, with the qp dict when promo==Q)
    for color in (True, False):
        for (f, t) in geometric_moves():
            mv = chess.Move(f, t)
            vec = ref_ct.actionToTensor(mv, color)
            nz = vec.nonzero().flatten().tolist()
            assert len(nz) == 1 and vec[nz[0]] == 1
            rows.append((int(color), f, t, 0, nz[0]))
            back = ref_ct.tensorToAction(vec, color, queen_promotion={})
            assert len(back) == 1
            dec.append((int(color), nz[0], back[0].from_square, back[0].to_square, back[0].promotion or 0, 0))
        for (f, t, p) in promotion_moves(color):
            mv = chess.Move(f, t, p)
            vec, qp = ref_ct.actionsToTensor([mv], color)
            nz = vec.nonzero().flatten().tolist()
            assert len(nz) == 1
            rows.append((int(color), f, t, p, nz[0]))
            back = ref_ct.tensorToAction(vec, color, queen_promotion=qp)
            dec.append((int(color), nz[0], back[0].from_square, back[0].to_square, back[0].promotion or 0, int(p == chess.QUEEN)))
    rows = np.array(rows, np.int64)
    dec = np.array(dec, np.int64)
    # actionsToTensor with a {move: prob} dict (training targets, train_RL.py:28)
    rng = random.Random(5)
    tgt_moves, tgt_vecs = [], []
    for color in (True, False):
        for _ in range(4):
            ms = rng.sample(geometric_moves(), 12)
            probs = np.random.RandomState(rng.randrange(1 << 30)).dirichlet(np.ones(12))
            d = {chess.Move(f, t): float(p) for (f, t), p in zip(ms, probs)}
            vec, _ = ref_ct.actionsToTensor(d, color)
            tgt_moves.append(np.array([(int(color), f, t, 0) for (f, t) in ms], np.int64))
            tgt_vecs.append((np.array(list(d.values()), np.float64), vec.numpy().copy()))
    np.savez_compressed(os.path.join(OUT, "codec_tables.npz"), encode=rows, decode=dec,
                        target_moves=np.stack(tgt_moves), target_probs=np.stack([a for a, _ in tgt_vecs]),
                        target_vecs=np.stack([b for _, b in tgt_vecs]))
    # KATs quoted in SURVEY §8(a) A17
    kat = {("e2e4", True): 116, ("g1f3", True): 4094, ("e7e5", False): 115, ("g8f6", False): 3641, ("e1g1", True): 1020,
           ("e1h1", True): 1084, ("a7a8q", True): 8, ("a7b8n", True): 4168, ("a2a1r", False): 4495}
    for (u, c), want in kat.items():
        got = ref_ct.actionToTensor(chess.Move.from_uci(u), c).nonzero().item()
        assert got == want, (u, c, got, want)
    print("codec_tables:", len(rows), "encodes,", len(dec), "decodes")


# --------------------------------------------------------------------------- 3. search traces on table-driven toy games
class ToyTable:
    """Procedural finite game: states are ints; structure drawn lazily from a seeded RNG."""

    def __init__(self, seed, mode):
        self.rng = random.Random(seed)
        self.nrng = np.random.RandomState(seed)
        self.mode = mode
        self.states = []           # dict(turn, terminal, term_value, moves=[(f,t,p,child)], value, policy(np f32 4672), depth)
        self.geo = geometric_moves()
        self.new_state(True if seed % 2 == 0 else False, 0)

    def new_state(self, turn, depth):
        sid = len(self.states)
        term = depth >= 1 and self.rng.random() < 0.12
        rec = dict(turn=turn, terminal=term, term_value=(self.rng.choice([0, -1]) if term else 0), moves=None,
                   value=np.float32(self.nrng.uniform(-1, 1)), policy=None, depth=depth)
        self.states.append(rec)
        return sid

    def materialise(self, sid):
        rec = self.states[sid]
        if rec["moves"] is not None:
            return rec
        K = self.rng.randint(1, 14) if self.rng.random() < 0.8 else self.rng.randint(15, 40)
        pairs = self.rng.sample(self.geo, K)
        moves, seen = [], set()
        color = rec["turn"]
        fr, tr = (6, 7) if color else (1, 0)
        for (f, t) in pairs:
            if (f, t) in seen:
                continue
            seen.add((f, t))
            is_promo_geom = (f // 8 == fr and t // 8 == tr and abs(t % 8 - f % 8) <= 1)
            if is_promo_geom:
                for p in self.rng.sample([chess.KNIGHT, chess.BISHOP, chess.ROOK, chess.QUEEN], self.rng.randint(1, 4)):
                    moves.append((f, t, p))
            else:
                moves.append((f, t, 0))
        rec["moves"] = [(f, t, p, self.new_state(not color, rec["depth"] + 1)) for (f, t, p) in moves]
        if self.mode == "dyadic":
            pol = (self.nrng.randint(1, 65, size=4672).astype(np.float32) / np.float32(1024.0))
            for (f, t, p, _) in rec["moves"]:
                if self.rng.random() < 0.04 and len(rec["moves"]) > 2:      # exact-zero legal probability -> child dropped
                    idx = ref_ct.actionToTensor(chess.Move(f, t, p or None), color).nonzero().item()
                    pol[idx] = 0.0
        else:
            logits = torch.tensor(self.nrng.normal(0, 2.0, size=4672).astype(np.float32))
            pol = torch.softmax(logits, 0).numpy().copy()
        rec["policy"] = pol
        return rec


class ToyBoard:
    def __init__(self, turn):
        self.turn = turn

    def __repr__(self):
        return "ToyBoard(turn=%r)" % self.turn


class ToyGame:
    """Duck-typed stand-in for ChessTensor as used by MCTS0.search (mcts.py:43-106)."""
    table = None

    def __init__(self, sid):
        self.sid = sid
        self.board = ToyBoard(ToyGame.table.states[sid]["turn"])

    def move_piece(self, move):
        rec = ToyGame.table.materialise(self.sid)
        for (f, t, p, child) in rec["moves"]:
            if (f, t, p or None) == (move.from_square, move.to_square, move.promotion):
                self.sid = child
                self.board = ToyBoard(ToyGame.table.states[child]["turn"])
                return
        raise ValueError("Invalid move")

    def get_value_and_terminated(self):
        rec = ToyGame.table.states[self.sid]
        return (rec["term_value"], True) if rec["terminal"] else (0, False)

    def get_valid_moves(self, board):
        rec = ToyGame.table.materialise(self.sid)
        return [chess.Move(f, t, p or None) for (f, t, p, _) in rec["moves"]]

    def get_representation(self):
        x = torch.zeros(119, 8, 8)
        x[0, 0, 0] = float(self.sid)
        return x

    def get_opponent_value(self, v):
        return -v


class ToyModel:
    def to(self, device):
        return self

    def __call__(self, x, inference=False):
        assert inference
        sid = int(x[0, 0, 0, 0].item())
        rec = ToyGame.table.materialise(sid)
        return torch.tensor(rec["policy"]).unsqueeze(0), torch.tensor([[rec["value"]]], dtype=torch.float32)


class RecordingNode(mctsnode.Node):
    created = []

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        RecordingNode.created.append(self)


def dump_tree(root, color_root):
    """DFS in child order: (depth, action_index, visits, value_sum, prior)"""
    rows = []

    def rec(node, depth, parent_color):
        for ch in node.children:
            idx = ref_ct.actionToTensor(ch.action_taken, parent_color).nonzero().item()
            rows.append((depth, idx, ch.visit_count, ch.value_sum, np.float32(ch.prior)))
            rec(ch, depth + 1, not parent_color)
    rec(root, 0, color_root)
    return rows


def gen_search():
    ref_mcts.Node = RecordingNode
    cases = []
    spec = [(s, l, m, seed) for s in (2, 5, 30, 200) for l in (False, True) for m in ("dyadic", "softmax") for seed in (0, 1)]
    spec += [(600, True, "dyadic", 7), (1, False, "dyadic", 3)]
    noise_seen = set()
    for ci, (S, learning, mode, seed) in enumerate(spec):
        table = ToyTable(seed * 1000 + S, mode)
        ToyGame.table = table
        RecordingNode.created = []
        game = ToyGame(0)
        torch.manual_seed(seed)
        engine = ref_mcts.MCTS0(game=game, args={"C": 2, "num_searches": S}, model=ToyModel())
        err = None
        try:
            probs = engine.search(game.board, verbose=False, learning=learning)
        except ZeroDivisionError as e:       # num_searches == 1 (mcts.py:118-120)
            probs, err = {}, "ZeroDivisionError"
        root = RecordingNode.created[0]
        tree = dump_tree(root, game.board.turn)
        # pack the explored table
        n = len(table.states)
        terminal = np.array([s["terminal"] for s in table.states], np.int32)
        term_value = np.array([s["term_value"] for s in table.states], np.int32)
        turn = np.array([int(s["turn"]) for s in table.states], np.int32)
        value = np.array([s["value"] for s in table.states], np.float32)
        off, mf, mt, mp, mc, pol_idx, pol_val = [0], [], [], [], [], [], []
        for s in table.states:
            for (f, t, p, c) in (s["moves"] or []):
                mf.append(f); mt.append(t); mp.append(p); mc.append(c)
                idx = ref_ct.actionToTensor(chess.Move(f, t, p or None), s["turn"]).nonzero().item()
                pol_idx.append(idx); pol_val.append(s["policy"][idx])
            off.append(len(mf))
        root_actions = [ref_ct.actionToTensor(m, game.board.turn).nonzero().item() for m in probs.keys()]
        cases.append(dict(S=S, learning=int(learning), mode=mode, seed=seed, error=err or "",
                          terminal=terminal, term_value=term_value, turn=turn, nn_value=value,
                          move_off=np.array(off, np.int32), move_from=np.array(mf, np.int32), move_to=np.array(mt, np.int32),
                          move_promo=np.array(mp, np.int32), move_child=np.array(mc, np.int32),
                          move_index=np.array(pol_idx, np.int32), move_policy=np.array(pol_val, np.float32),
                          root_actions=np.array(root_actions, np.int32), root_probs=np.array(list(probs.values()), np.float64),
                          root_visits=np.int64(root.visit_count), root_value_sum=np.float64(root.value_sum),
                          tree_depth=np.array([r[0] for r in tree], np.int32), tree_action=np.array([r[1] for r in tree], np.int32),
                          tree_visits=np.array([r[2] for r in tree], np.int64), tree_value_sum=np.array([r[3] for r in tree], np.float64),
                          tree_prior=np.array([r[4] for r in tree], np.float32)))
        if learning:
            for node in RecordingNode.created[1:]:
                pass
        print("search case %2d: S=%d learning=%d mode=%s seed=%d -> %d states, %d tree nodes %s"
              % (ci, S, learning, mode, seed, n, len(tree), err or ""))
    flat = {"n_cases": np.int64(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat["c%d_%s" % (i, k)] = np.array(v) if not isinstance(v, np.ndarray) else v
    np.savez_compressed(os.path.join(OUT, "search_traces.npz"), **flat)
    # the degenerate Dirichlet draw of mcts.py:91-96
    torch.manual_seed(0)
    draws = torch.cat([torch.distributions.dirichlet.Dirichlet(torch.full((k, 1), 0.3)).sample().flatten() for k in (1, 5, 30, 218)])
    np.savez_compressed(os.path.join(OUT, "noise_probe.npz"), draws=draws.numpy(), torch_version=np.array(torch.__version__))
    print("noise draws: unique values", np.unique(draws.numpy()).tolist())


# --------------------------------------------------------------------------- 4. network golden
def gen_network():
    torch.manual_seed(0)
    net = ref_net.policyNN({})
    net.eval()
    keys = list(net.state_dict().keys())
    n_params = sum(p.numel() for p in net.parameters())
    g = torch.Generator().manual_seed(123)
    x = (torch.rand(3, 119, 8, 8, generator=g) < 0.15).float()
    with torch.no_grad():
        p_inf, v = net(x, inference=True)
        p_raw, _ = net(x, inference=False)
    sd = net.state_dict()
    probe = {k: sd[k].flatten()[:4].numpy().copy() for k in ("conv1.weight", "conv_p2.bias", "fc_v1.weight", "fc_v2.bias",
                                                              "resnet_blocks.0.conv1.weight", "resnet_blocks.18.conv2.weight")}
    np.savez_compressed(os.path.join(OUT, "network_golden.npz"), x=x.numpy().astype(np.uint8), policy_softmax=p_inf.numpy(),
                        policy_logits=p_raw.numpy(), value=v.numpy(), n_params=np.int64(n_params),
                        keys=np.array(keys), **{"probe_" + k.replace(".", "_"): v_ for k, v_ in probe.items()})
    # train-mode loss golden (train_RL.py:103-112 arithmetic) on a fixed mini-batch
    torch.manual_seed(0)
    net = ref_net.policyNN({})
    net.train()
    xb = (torch.rand(8, 119, 8, 8, generator=g) < 0.15).float()
    pt = torch.softmax(torch.randn(8, 4672, generator=g) * 3, 1)
    vt = torch.tensor([1., -1., 0., 1., -1., 0., 1., -1.])
    p, vv = net(xb)
    mse = torch.nn.functional.mse_loss(vv.squeeze(-1), vt)
    ce = torch.nn.functional.cross_entropy(p, pt)
    np.savez_compressed(os.path.join(OUT, "train_loss_golden.npz"), x=xb.numpy().astype(np.uint8), p_target=pt.numpy(), v_target=vt.numpy(),
                        mse=np.float64(mse.item()), ce=np.float64(ce.item()))
    print("network: %d params, %d keys, mse %.6f ce %.6f" % (n_params, len(keys), mse.item(), ce.item()))


# --------------------------------------------------------------------------- 5. sampler (sim.py:68)
def gen_sampler():
    rows = []
    rng = np.random.RandomState(99)
    for seed in range(200):
        K = int(rng.randint(1, 40))
        visits = rng.randint(0, 50, size=K)
        visits[rng.randint(K)] += 1
        tot = int(visits.sum())
        p = [int(v) / tot for v in visits]
        np.random.seed(seed)
        u = np.random.random_sample()
        np.random.seed(seed)
        choice = int(np.random.choice(np.arange(K), p=p))
        rows.append((seed, K, u, choice, visits.copy()))
    off = np.cumsum([0] + [r[1] for r in rows])
    np.savez_compressed(os.path.join(OUT, "sampler_golden.npz"), offsets=off.astype(np.int64),
                        visits=np.concatenate([r[4] for r in rows]).astype(np.int64),
                        u=np.array([r[2] for r in rows]), choice=np.array([r[3] for r in rows], np.int64))
    print("sampler:", len(rows), "cases")


if __name__ == "__main__":
    gen_ucb()
    gen_codec()
    gen_search()
    gen_network()
    gen_sampler()
