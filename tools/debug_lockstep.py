import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_parity import make_boards, random_evaluator
from sigma_zero_amd.selfplay import SelfPlayEngine, NOISE_REFERENCE
from oracle import oracle as O

learning = True
boards = make_boards(24, seed=7)
S = 64
B = len(boards)
eng = SelfPlayEngine(None, {"C": 2, "num_searches": S}, B, learning=learning)
for b, m in enumerate(boards):
    eng.upload_game(b, m.ct)
    m.search = O.Search.on_chess(m.oct, c=2.0, num_searches=S, learning=learning, noise_value=NOISE_REFERENCE)
ev = random_evaluator(11 + learning)
eng.begin()
for step in range(S + 1):
    torch.cuda.synchronize()
    mask, depth, n_nodes, n_edges, status = eng.debug_pending()
    pend = [m.search.advance() for m in boards]
    action, visits, n_child, prior, wsum = eng.root_children()
    for b, m in enumerate(boards):
        idx, vis, _ = m.search.root_children()
        pr, ws = m.search.root_stats()
        k = int(n_child[b])
        if k != len(idx) or action[b,:k].tolist() != idx or visits[b,:k].tolist() != vis or not np.array_equal(prior[b,:k].view(np.uint32), pr.view(np.uint32)) or not np.array_equal(wsum[b,:k], ws):
            print("STEP", step, "board", b, "k", k, len(idx))
            print(" eng visits", visits[b,:k].tolist()); print(" ora visits", vis)
            print(" prior diff idx", np.nonzero(prior[b,:k].view(np.uint32) != pr.view(np.uint32))[0].tolist() if k == len(idx) else None)
            print(" eng prior", prior[b,:k].tolist()); print(" ora prior", pr.tolist())
            print(" eng wsum", wsum[b,:k].tolist()); print(" ora wsum", ws.tolist())
            print(" oracle root visits", m.search.root_visits(), "pending", pend[b], status[b])
            sys.stdout.flush(); os._exit(0)
    policy, value = ev(eng.planes, step)
    ph, vh = policy.cpu().numpy(), value.cpu().numpy()
    for b, m in enumerate(boards):
        if pend[b]:
            m.search.feed(ph[b], vh[b])
    eng.step(policy, value)
print("no difference"); sys.stdout.flush(); os._exit(0)
