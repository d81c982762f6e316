import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, contextlib
import sigma_zero_amd as sz
from sigma_zero_amd import train_rl as T
import sigma_zero_amd.trainconv as TC
from sigma_zero_amd.trainconv import split_convs
g = torch.Generator().manual_seed(5)
def batch(n):
    return {"states": (torch.rand(n, 119, 8, 8, generator=g) < 0.15).float().cuda(), "actions": torch.softmax(torch.randn(n, 4672, generator=g) * 3, 1).cuda(),
            "rewards": torch.randint(-1, 2, (n,), generator=g).float().cuda()}
all32 = [batch(32) for _ in range(8)]
b16 = batch(16)
def run(name, batches, convs=True, with_ref=True):
    torch.manual_seed(0)
    ref = sz.policyNN({}).cuda().train()
    net = sz.policyNN({}).cuda().train()
    net.load_state_dict(ref.state_dict())
    bad = 0
    with (split_convs(ref) if convs else contextlib.nullcontext()), (split_convs(net) if convs else contextlib.nullcontext()):
        gs = T.GraphedStep(net, "cuda", None)
        for i, b in enumerate(batches):
            if with_ref:
                ref.zero_grad()
                loss, mse, ce = T.loss_fn(ref, b, "cuda")
                loss.backward()
            else:
                with torch.no_grad():
                    ce = torch.nn.functional.cross_entropy(ref(b["states"])[0], b["actions"])
            m2, c2 = gs.step(b)
            ok = abs(float(c2) - float(ce.detach())) < 1e-3
            bad += (not ok)
            print("%-36s step %d n=%d: ce %.6f / %.6f %s" % (name, i, len(b["rewards"]), float(c2), float(ce.detach()), "" if ok else "  <-- WRONG"), flush=True)
    return bad
for pos in (3, 5):
    odd = all32[:pos] + [b16] + all32[pos:pos + 2]
    run("split, odd at %d" % pos, odd)
    run("split, odd at %d, ref without grad" % pos, odd, with_ref=False)
    run("torch, odd at %d" % pos, odd, convs=False)
