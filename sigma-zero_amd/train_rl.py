"""RL training step of /root/reference/train_RL.py (chessDataset :14-49, train :77-154, main :156-275), re-built for
one-process-per-GPU data parallelism: every rank trains on the games it generated itself and gradients are averaged
with a bucketed all-reduce (RCCL over xGMI on MI355X: backend "nccl"; gloo on CPU for tests) that overlaps backward.

Reference behaviour kept: loss = mse(v.squeeze(-1), z) + cross_entropy(logits, pi) with soft targets (:108-111),
Adam(lr 1e-4, weight_decay 1e-4) (:187), StepLR(step 500, gamma 0.95) stepped per batch (:199, :121-122), 7 passes per
cycle (range(0, total_steps+1), :93), batch 128, drop_last (:246-253), states stored bit-packed (119,8) uint8 with
bit j = column j (generate_training_supervised.py:91) and unpacked like collatefn (:42).
Reference defects fixed by intent: no hard-coded weight path, no test(None) call, states are packed.
Divergence: BatchNorm uses per-rank batch statistics (the reference trains on one GPU).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _native as N


# ----------------------------------------------------------------------------- data
class SelfPlayDataset(torch.utils.data.Dataset):
    """chessDataset (train_RL.py:14-33) over engine records: packed states, (action indices, visit fractions), rewards."""

    def __init__(self, packed_states, action_idx, action_prob, rewards):
        self.states = packed_states            # list/array of (119,8) uint8
        self.action_idx = action_idx           # list of int arrays
        self.action_prob = action_prob         # list of float arrays (visit fractions, sum 1)
        self.rewards = rewards                 # list of +1/-1/0

    def __len__(self):
        return len(self.states)

    def __getitem__(self, i):
        target = torch.zeros(N.SZ_ACTIONS)
        target[torch.as_tensor(np.asarray(self.action_idx[i]), dtype=torch.long)] = torch.as_tensor(np.asarray(self.action_prob[i]), dtype=torch.float32)
        return torch.as_tensor(np.asarray(self.states[i])), target, torch.tensor(float(self.rewards[i]))

    @staticmethod
    def collate(batch):
        """collatefn (train_RL.py:35-49): unpack bit j of each byte into column j."""
        states, actions, rewards = zip(*batch)
        idx = torch.arange(8).view(1, 1, 8)
        s = torch.stack(states, 0).to(torch.uint8)
        s = ((s.unsqueeze(-1) >> idx) % 2 == 1).to(torch.float)
        return {"states": s, "actions": torch.stack(actions, 0), "rewards": torch.stack(rewards, 0)}


class DeviceBatches:
    """Device-resident equivalent of DataLoader(SelfPlayDataset(...), batch_size, shuffle=True, drop_last=True,
    collate_fn=SelfPlayDataset.collate): the packed states, padded (action index, visit fraction) rows and rewards live on the
    training device; a batch is a gather + bit unpack + scatter there (the per-sample Python of chessDataset/collatefn costs
    30 ms per batch of 128 on the host, more than twice the forward+backward).  Yields the same dicts as the collate."""

    def __init__(self, packed_states, action_idx, action_prob, rewards, batch_size=128, device="cpu", shuffle=True, generator=None):
        self.device = torch.device(device)
        n = len(packed_states)
        self.n, self.batch_size, self.shuffle, self.generator = n, int(batch_size), shuffle, generator
        self.states = torch.from_numpy(np.stack([np.asarray(p, dtype=np.uint8) for p in packed_states]) if n else np.zeros((0, N.SZ_PLANES, 8), np.uint8)).to(self.device)
        k = max((len(a) for a in action_idx), default=1)
        idx = np.zeros((n, k), dtype=np.int64)
        prob = np.zeros((n, k), dtype=np.float32)                      # padding adds 0 to action 0
        for i, (a, pr) in enumerate(zip(action_idx, action_prob)):
            idx[i, :len(a)] = a
            prob[i, :len(a)] = np.asarray(pr, dtype=np.float64).astype(np.float32)
        self.idx, self.prob = torch.from_numpy(idx).to(self.device), torch.from_numpy(prob).to(self.device)
        self.rewards = torch.tensor([float(r) for r in rewards], dtype=torch.float32, device=self.device)
        self._shift = torch.arange(8, device=self.device, dtype=torch.uint8).view(1, 1, 1, 8)

    def __len__(self):
        return self.n // self.batch_size                               # drop_last

    def __iter__(self):
        if self.shuffle:
            gdev = self.generator.device if self.generator is not None else self.device
            perm = torch.randperm(self.n, generator=self.generator, device=gdev).to(self.device)
        else:
            perm = torch.arange(self.n, device=self.device)
        for b in range(len(self)):
            sel = perm[b * self.batch_size:(b + 1) * self.batch_size]
            states = ((self.states[sel].unsqueeze(-1) >> self._shift) & 1).to(torch.float)          # bit j of a byte = column j
            target = torch.zeros(sel.numel(), N.SZ_ACTIONS, dtype=torch.float32, device=self.device)
            target.scatter_add_(1, self.idx[sel], self.prob[sel])
            yield {"states": states, "actions": target, "rewards": self.rewards[sel]}


def records_from_games(games):
    """sim.play_games() output -> arrays for SelfPlayDataset (states re-packed to the (119,8) uint8 format)."""
    from .chess_tensor import action_index
    packed, aidx, aprob, rew = [], [], [], []
    w = (1 << np.arange(8)).astype(np.uint8)
    for g in games:
        pre = g.get("packed_states")                       # play_games() keeps the engine's own bit-packed record next to the bool tensor
        for i, (st, act, r, col) in enumerate(zip(g["states"], g["actions"], g["rewards"], g["colours"])):
            packed.append(pre[i] if pre else (st.numpy().astype(np.uint8) * w).sum(-1).astype(np.uint8))
            aidx.append(np.array([action_index(m, col) for m in act], dtype=np.int64))
            aprob.append(np.array(list(act.values()), dtype=np.float64))
            rew.append(r)
    return packed, aidx, aprob, rew


# ----------------------------------------------------------------------------- gradient sync
class GradSync:
    """Flat gradient buffer + bucketed asynchronous all-reduce launched from backward hooks.

    On MI355X xGMI is point-to-point (7 links x ~153 GB/s): a ring all-reduce moves 2(n-1)/n * S through every link,
    so a few large buckets (default 4 x ~23 MB for the 91 MB fp32 gradient) keep the per-link pipeline full while the
    first buckets still overlap the tail of backward; tiny buckets would be launch/latency bound."""

    def __init__(self, model, process_group=None, n_buckets=4, always_sync=False):
        """always_sync: issue the collectives even in a 1-rank group (exercises the RCCL path on a single GPU; results unchanged)"""
        import torch.distributed as dist
        self.dist = dist
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (bool(always_sync) and dist.is_initialized())
        params = [p for p in model.parameters() if p.requires_grad]
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        # gradients become views into the flat buffer; buckets are contiguous slices in REVERSE parameter order
        # (backward produces the last layers' gradients first)
        off = 0
        self.slices = {}
        for p in params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self.slices[p] = (off, off + p.numel())
            off += p.numel()
        bounds = [round(total * k / n_buckets) for k in range(n_buckets + 1)]
        self.buckets = [(bounds[k], bounds[k + 1]) for k in range(n_buckets)]
        self.pending = [0] * n_buckets
        self.bucket_params = [[] for _ in range(n_buckets)]
        for p in params:
            lo, hi = self.slices[p]
            for b, (blo, bhi) in enumerate(self.buckets):
                if lo < bhi and hi > blo:
                    self.bucket_params[b].append(p)
        self.handles = []
        self._left = None
        if self.collective:
            for p in params:
                p.register_post_accumulate_grad_hook(self._hook)

    def common_batches(self, n_local):
        """Number of batches EVERY rank can run in one pass: MIN over ranks of the local batch count.  Ranks hold different numbers
        of samples (their games end at different plies), and every batch issues one all-reduce per bucket: a rank that ran more
        batches than its peers would wait forever.  A rank with more samples than the minimum sees a different random subset of them
        in each of the 7 passes (the batches are reshuffled per pass)."""
        if not self.collective:
            return int(n_local)
        t = torch.tensor([int(n_local)], dtype=torch.int64, device=self.flat.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def begin_step(self):
        self.handles = []
        self._left = [len(ps) for ps in self.bucket_params]

    def _hook(self, p):
        if self._left is None:
            return
        lo, hi = self.slices[p]
        for b, (blo, bhi) in enumerate(self.buckets):
            if lo < bhi and hi > blo:
                self._left[b] -= 1
                if self._left[b] == 0:
                    self.handles.append(self.dist.all_reduce(self.flat[blo:bhi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish_step(self):
        if self.collective:
            for h in self.handles:
                h.wait()
            if self.world > 1:
                self.flat.div_(self.world)
        self._left = None

    def zero(self):
        self.flat.zero_()

    def reduce_all(self):
        """All buckets at once, after a backward that ran without the hooks (a replayed HIP graph: train(graph=True))."""
        if self.collective:
            for h in [self.dist.all_reduce(self.flat[blo:bhi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True) for blo, bhi in self.buckets]:
                h.wait()
            if self.world > 1:
                self.flat.div_(self.world)


# ----------------------------------------------------------------------------- training
def loss_fn(model, batch, device):
    p, v = model(batch["states"].to(device))
    v = v.squeeze(-1)
    mse = F.mse_loss(v, batch["rewards"].to(device))
    ce = F.cross_entropy(p, batch["actions"].to(device))
    return mse + ce, mse, ce


class GraphedStep:
    """Gradient zeroing + forward + backward of one optimiser step as ONE HIP-graph replay (train(graph=True); opt-in).  With the convolutions on the matrix cores a
    step is 7.2 ms of kernels in ~1,100 launches, which a host issues in 6-10 ms: on a slow host the eager loop is host-bound (tools/train_loop_probe.py), the replay
    (7.6-8.0 ms) is not.  The first WARMUP steps run eagerly on
    the capture stream (real steps: nothing is computed twice), the second one is then captured with static batch buffers; every later batch of the same shape is copied
    in and replayed; a batch of another shape drops the graph (eager steps, then a new capture).  Gradients are static tensors (GradSync's flat buffer, or per-parameter tensors allocated here) zeroed inside the
    graph; the optimiser step, the learning-rate schedule and the gradient all-reduce stay outside."""
    WARMUP = 2

    def __init__(self, model, device, sync=None):
        self.model, self.device, self.sync = model, torch.device(device), sync
        self.graph, self.static, self.out, self.seen = None, None, None, 0
        self.side = torch.cuda.Stream(self.device)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if sync is None:
            for p in self.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        self.grads = [p.grad for p in self.params]

    def _zero(self):
        if self.sync is not None:
            self.sync.zero()
        else:
            torch._foreach_zero_(self.grads)

    def _forward_backward(self, batch):
        self._zero()
        loss, mse, ce = loss_fn(self.model, batch, self.device)
        loss.backward()
        return mse.detach(), ce.detach()

    def step(self, batch):
        """-> (mse, ce) device scalars of this batch; the gradients are in place when it returns (ordered on the current stream)."""
        b = {k: batch[k].to(self.device) for k in ("states", "actions", "rewards")}
        if self.graph is not None and all(b[k].shape == self.static[k].shape and b[k].dtype == self.static[k].dtype for k in b):
            for k in b:
                self.static[k].copy_(b[k])
            self.graph.replay()
            return self.out
        if self.graph is not None:
            # A batch of another shape: the graph is dropped and captured again two steps later.  (Keeping it does not work on ROCm 7.2: the first eager step of a shape
            # the process has not run before — MIOpen loads kernels for it — leaves every later replay of an EXISTING graph with a wrong multi-block reduction (the
            # cross-entropy sum comes out as one block's partial; everything else, gradients included, stays right); a graph captured afterwards is fine.
            # tests/test_gpu_round3.py holds the case.)
            self.graph, self.static, self.out, self.seen = None, None, None, 0
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            out = self._forward_backward(b)
        cur.wait_stream(self.side)
        self.seen += 1
        if self.graph is None and self.seen >= self.WARMUP:
            self.static = {k: v.clone() for k, v in b.items()}
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self.side):
                self.out = self._forward_backward(self.static)
            self.graph = graph
        return out


def train(model, dataloader, optimiser, total_steps=6, lr_scheduler=None, sync=None, device=None, log=None, start_epoch=0, split_convs=None, graph=False):
    """train() of train_RL.py:77-154 without the test()/checkpoint side effects; returns the list of (mse, ce).
    split_convs (default None = on for an fp32 model on a GPU; False = torch/MIOpen): the 38 3x3 convolutions of the tower run forward, backward-data and weight
    gradient on the matrix cores with hi + lo f16 operands (22 bits of mantissa, exact power-of-two scaling) and f32 accumulation (trainconv.py): one convolution
    5.0e-7 relative L2 from fp64 — fp32 through torch is at 4.9e-7 — independent of the tensor's magnitude; the whole-network gradient at batch 128 is 3.45e-3 from an
    fp64 step where MIOpen's fp32 step is 3.37e-3; same loss.  12.9 -> 7.3-8.6 ms per optimiser step at batch 128 (host-bound when eager).
    graph (default False): True = gradient zeroing + forward + backward as one HIP-graph replay per step (GraphedStep): the same kernels on the same data, issued by
    one call.  Off by default because it does not pay on a host that keeps up: with the matrix-core convolutions a step is 7.2 ms of kernels and a fast host issues
    them in 6.0-7.1 ms, so eager is GPU-bound at 7.25-7.7 ms where the replay takes 7.6-8.0 ms (profiles/r03zt_train_loop_probe.txt, r03zu_cycle_graph_ab.txt); it is
    the switch for a slow or contended host (8-10 ms eager).  With several ranks graph=True reduces all gradient buckets after the replay instead of overlapping them
    with backward from hooks."""
    import itertools
    device = device or next(model.parameters()).device
    on_gpu = torch.device(device).type == "cuda"
    eligible = on_gpu and next(model.parameters()).dtype == torch.float32
    split_convs = eligible if split_convs is None else (bool(split_convs) and eligible)
    if split_convs:
        from .trainconv import enable_split_convs, disable_split_convs
        enable_split_convs(model)
        try:
            return train(model, dataloader, optimiser, total_steps, lr_scheduler, sync, device, log, start_epoch, split_convs=False, graph=graph)
        finally:
            disable_split_convs(model)
    graph = bool(graph) and on_gpu
    history = []
    model.train()
    n_batches = sync.common_batches(len(dataloader)) if sync is not None else len(dataloader)     # identical on every rank
    graphed = GraphedStep(model, device, sync) if graph else None
    # BatchNorm's step counters: every train-mode forward launches one `num_batches_tracked += 1` per layer (41 launches of a few microseconds in a step of ~510).
    # They only feed momentum=None layers, which this network has none of: the counters are set aside for the loop and advanced once at its end.
    counters = [(m, m.num_batches_tracked) for m in model.modules()
                if on_gpu and isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None and m.momentum is not None]
    for m, _ in counters:
        m.num_batches_tracked = None
    try:
        _train_loop(model, dataloader, optimiser, total_steps, lr_scheduler, sync, device, log, start_epoch, graphed, n_batches, history)
    finally:
        for m, t in counters:
            m.num_batches_tracked = t
        if counters and history:
            torch._foreach_add_([t for _, t in counters], len(history))
    if not history:
        return []
    return [(float(m), float(c)) for m, c in torch.stack(history).cpu().tolist()]


def _train_loop(model, dataloader, optimiser, total_steps, lr_scheduler, sync, device, log, start_epoch, graphed, n_batches, history):
    import itertools
    for step in range(start_epoch, total_steps + 1):
        for batch in itertools.islice(iter(dataloader), n_batches):
            if graphed is not None:
                mse, ce = graphed.step(batch)
                if sync is not None:
                    sync.reduce_all()
            else:
                if sync is not None:
                    sync.zero()
                    sync.begin_step()
                else:
                    optimiser.zero_grad()
                loss, mse, ce = loss_fn(model, batch, device)
                loss.backward()
                if sync is not None:
                    sync.finish_step()
            optimiser.step()
            if lr_scheduler is not None:
                lr_scheduler.step()
            if log:
                log(step, float(mse.detach()), float(ce.detach()))
            history.append(torch.stack((mse.detach(), ce.detach())))  # stays on the device: no host sync per step


def aggregate_throughput(counts, seconds, device="cpu", force=False):
    """Whole-job aggregation used by bench.py: SUM of per-rank unit counts, MAX of per-rank wall time (no-op on 1 rank unless
    force=True, which runs the two all-reduces in a 1-rank group too)."""
    import torch.distributed as dist
    t = torch.tensor([float(c) for c in counts] + [float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t[:-1]], float(mx[-1])
    return [float(x) for x in t[:-1]], float(t[-1])


def make_optimiser(model):
    on_gpu = next(model.parameters()).is_cuda
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=on_gpu)     # train_RL.py:187 (one fused launch on the GPU: 13.2 -> 12.8 ms per step)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=500, gamma=0.95)          # train_RL.py:199
    return opt, sched


def save_cycle(model, optimiser, cycle, out_dir="saves"):
    """train_RL.py:151-154: state_dicts with the reference's key names."""
    os.makedirs(out_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(out_dir, "RL_%d.pt" % cycle))
    torch.save(optimiser.state_dict(), os.path.join(out_dir, "RL_opt_%d.pt" % cycle))


def load_cycle(model, optimiser, cycle, out_dir="saves", device=None):
    """Resume of train_RL.py:189-197 by intent: model weights from RL_{cycle}.pt AND optimiser state from RL_opt_{cycle}.pt (the
    reference loads the model file into both, which cannot work).  Returns False — leaving model and optimiser untouched — when either
    file is missing, like the reference's `except: start_epoch = 1`."""
    mp, op = os.path.join(out_dir, "RL_%d.pt" % cycle), os.path.join(out_dir, "RL_opt_%d.pt" % cycle)
    if not (os.path.exists(mp) and os.path.exists(op)):
        return False
    device = device or next(model.parameters()).device
    msd = torch.load(mp, map_location=device, weights_only=True)
    osd = torch.load(op, map_location=device, weights_only=True)
    model.load_state_dict(msd)
    optimiser.load_state_dict(osd)
    return True


def run_cycle(model, optimiser, lr_scheduler, args, n_games, chess960=True, sync=None, batch_size=128, total_steps=6, fast_inference=True, train_convs="split", train_graph=False):
    """One epoch of train_RL.main (:205-264) on this rank: self-play n_games on this GPU, then 7 passes of training.
    fast_inference — the self-play network, fastest first (measured on MI355X at 4096 boards x 800 searches; fidelity = the same 64 positions
    searched with the fp32 module, tests/test_gpu_train_and_precision.py):
      "bf16"          FastPolicyNet on bf16 operands: the throughput configuration BASELINE.json names (1.00x); 61-62 of 64 boards with the fp32
                      network's exact visit counts, the others differ by one visit (max |delta fraction| 1.3e-3 at 800 searches);
      True / "fp16"   THE DEFAULT: the same kernels on f16 operands (11 bits of mantissa instead of 8; same cycles, the chip holds a 5 % lower
                      clock): 0.95x, 64 of 64 boards with the fp32 network's exact visit counts at 100 and at 800 searches;
      "split"         SplitPolicyNet (hi + lo bf16 operands, 3 MFMAs per product): the reference's precision class by construction (logits
                      within 6e-6 of fp64, fp32 itself is at 4e-7), 0.40x; reproduces the reference's own CPU game ply for ply;
      False / "fp32"  the torch module itself (MIOpen), 0.04x."""
    from .sim import play_games
    from .fastnet import FastPolicyNet, SplitPolicyNet
    device = next(model.parameters()).device
    model.eval()
    if device.type != "cuda" or fast_inference in (False, "fp32"):
        player = model
    elif fast_inference == "split":
        player = SplitPolicyNet(model, device=device)
    elif fast_inference == "bf16":
        player = FastPolicyNet(model, device=device)
    elif fast_inference in (True, "fp16"):
        player = FastPolicyNet(model, device=device, operands="fp16")
    else:
        raise ValueError("fast_inference: %r" % (fast_inference,))
    games = play_games(player, args, n_games, c960=chess960, max_plies=args.get("max_plies", 100000))
    packed, aidx, aprob, rew = records_from_games(games)
    dl = DeviceBatches(packed, aidx, aprob, rew, batch_size=batch_size, device=device, shuffle=True)       # same batches as DataLoader + collate
    return train(model, dl, optimiser, total_steps=total_steps, lr_scheduler=lr_scheduler, sync=sync, device=device,
                 split_convs=None if train_convs == "split" else False, graph=train_graph), games


# ----------------------------------------------------------------------------- train_RL.main (:156-275), one process per GPU
def sync_module_state(model, average_buffers=True):
    """Rank 0's parameters to every rank (start of training); BatchNorm running statistics averaged over ranks (end of a
    cycle) so that every rank plays its next self-play cycle with the same inference network."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    for p in model.parameters():
        dist.broadcast(p.data, src=0)
    for name, buf in model.named_buffers():
        if not buf.dtype.is_floating_point:
            dist.broadcast(buf, src=0)
        elif average_buffers:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            buf.div_(world)
        else:
            dist.broadcast(buf, src=0)


def pack_games_for_save(games):
    """games/RL_960_{epoch}.pt of train_RL.py:241, with states in the (119,8) uint8 bit-packed format and UCI move keys."""
    packed, aidx, aprob, rew = records_from_games(games)
    return {"states": [torch.from_numpy(p) for p in packed],
            "actions": [{m.uci(): float(v) for m, v in act.items()} for g in games for act in g["actions"]],
            "rewards": rew, "colours": [c for g in games for c in g["colours"]]}


def merge_game_files(games_dir, epoch, world):
    """train_RL.py:229-241 merges every worker's games into ONE games/RL_960_{epoch}.pt.  Here every rank writes its own file; this concatenates
    them key-wise (rank order) into the reference's file name and removes the per-rank files.  Meant for small runs: at 8 x 4096 games per cycle
    the merged pickle is several GB through one process (`--merge-games`, off by default)."""
    base = os.path.join(games_dir, "RL_960_%d.pt" % epoch)
    merged = torch.load(base, weights_only=True)
    for r in range(1, world):
        part_path = os.path.join(games_dir, "RL_960_%d.rank%d.pt" % (epoch, r))
        part = torch.load(part_path, weights_only=True)
        for k in ("states", "actions", "rewards", "colours"):
            merged[k] = list(merged[k]) + list(part[k])
        os.remove(part_path)
    torch.save(merged, base)
    return len(merged["rewards"])


def write_step_log(path, epoch, hist, lr_scheduler, append=True):
    """One JSON line per optimiser step (train_RL.py:124-125 logs loss, mse and cross-entropy per step): written after the passes from the
    device-side history, so that logging costs no host synchronisation inside the training loop."""
    import json
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    first = (lr_scheduler.last_epoch - len(hist)) if lr_scheduler is not None else 0
    with open(path, "a" if append else "w") as f:
        for i, (mse, ce) in enumerate(hist):
            rec = {"epoch": epoch, "step": first + i, "loss": mse + ce, "mse": mse, "cross_entropy": ce}
            if lr_scheduler is not None:
                rec["lr"] = lr_scheduler.base_lrs[0] * lr_scheduler.gamma ** ((first + i) // lr_scheduler.step_size)
            f.write(json.dumps(rec) + "\n")


def main(argv=None):
    """`python -m torch.distributed.run --nproc-per-node N -m sigma_zero_amd.train_rl ...` (or plain python for one GPU):
    every rank self-plays its own games on its own GPU (no communication), then all ranks train with averaged gradients."""
    import argparse
    import torch.distributed as dist
    ap = argparse.ArgumentParser(description="RL loop of train_RL.py on the MI355X engine")
    ap.add_argument("--epochs", type=int, default=1)                  # train_RL.py:174 num_epochs (cycles)
    ap.add_argument("--start-epoch", type=int, default=1)             # :175
    ap.add_argument("--games-per-rank", default="40", help="train_RL.py:165 num_games, per GPU here; one number, or a comma list with one entry per rank")
    ap.add_argument("--searches", type=int, default=100)              # :170
    ap.add_argument("--batch-size", type=int, default=128)            # :173
    ap.add_argument("--total-steps", type=int, default=6)             # :261 -> 7 passes
    ap.add_argument("--chess960", type=int, default=1)                # :176
    ap.add_argument("--max-plies", type=int, default=100000)
    ap.add_argument("--init", default=None, help="state_dict to start from (reference checkpoints load: same keys)")
    ap.add_argument("--init-opt", default=None, help="optimiser state_dict to start from (RL_opt_N.pt)")
    ap.add_argument("--save-dir", default="saves")
    ap.add_argument("--games-dir", default="games")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--log-dir", default="logs", help="per-step loss log logs/RL_train.jsonl (rank 0); empty string = off")
    ap.add_argument("--merge-games", action="store_true", help="concatenate the ranks' game files into the reference's single games/RL_960_{epoch}.pt (small runs)")
    ap.add_argument("--train-convs", default="split", choices=["split", "torch"],
                    help="3x3 convolutions of the train step: split = the matrix-core kernels on hi+lo f16 operands (forward, backward-data, weight gradient; fp32's "
                         "accuracy class, 35-45 %% faster step; default), torch = MIOpen fp32")
    ap.add_argument("--train-graph", default="off", choices=["on", "off"],
                    help="gradient zeroing + forward + backward of a step as one HIP-graph replay: makes the step independent of the host's launch rate (7.6-8.0 ms); "
                         "off by default, a host that keeps up runs the eager loop GPU-bound at 7.3-7.7 ms")
    ap.add_argument("--inference", default="fp16", choices=["fp16", "bf16", "split", "fp32"],
                    help="self-play network (run_cycle): fp16 = MFMA tower on f16 operands (default: fp32's visit counts on every tested position, 0.95x of bf16), "
                         "bf16 = fastest (single visits move), split = hi+lo bf16 operands (fp32-class by construction, 0.40x), fp32 = torch module")
    a = ap.parse_args(argv)
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    gpr = [int(x) for x in str(a.games_per_rank).split(",")]
    a.games_per_rank = gpr[rank] if len(gpr) > 1 else gpr[0]
    if not torch.cuda.is_available():
        raise SystemExit("train_rl needs an MI355X: self-play has no CPU path")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    if world > 1:
        dist.init_process_group(backend=a.backend)
    from .network import policyNN
    torch.manual_seed(0)
    model = policyNN({}).to(device)
    if a.init:
        model.load_state_dict(torch.load(a.init, map_location=device, weights_only=True))
    optimiser = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)            # train_RL.py:187
    start_epoch = a.start_epoch
    if start_epoch > 1 and not a.init:                                                   # train_RL.py:189-197
        if load_cycle(model, optimiser, start_epoch - 1, a.save_dir, device):
            print("rank %d: resumed model + optimiser from cycle %d" % (rank, start_epoch - 1), flush=True)
        else:
            print("No saved weights from epoch %d found!" % (start_epoch - 1), flush=True)
            start_epoch = 1
    if a.init_opt:
        optimiser.load_state_dict(torch.load(a.init_opt, map_location=device, weights_only=True))
    sched = torch.optim.lr_scheduler.StepLR(optimiser, step_size=500, gamma=0.95)           # created after the resume, like train_RL.py:199
    sync_module_state(model, average_buffers=False)
    sync = GradSync(model) if world > 1 else None
    args = {"C": 2, "num_searches": a.searches, "max_plies": a.max_plies}
    import random
    random.seed(1000 + rank)
    np.random.seed(1000 + rank)
    for epoch in range(start_epoch, start_epoch + a.epochs):
        hist, games = run_cycle(model, optimiser, sched, args, a.games_per_rank, chess960=bool(a.chess960), sync=sync,
                                batch_size=a.batch_size, total_steps=a.total_steps, fast_inference=a.inference, train_convs=a.train_convs,
                                train_graph=(a.train_graph == "on"))
        sync_module_state(model, average_buffers=True) if world > 1 else None
        n_samples = sum(len(g["actions"]) for g in games)
        # games/RL_960_{epoch}.pt (train_RL.py:229-241 merges every worker's games into one file): rank 0 writes its games under the
        # reference's name, every other rank writes games/RL_960_{epoch}.rank{r}.pt next to it (same layout; a merge is a key-wise
        # concatenation — not done here: at 8 x 4096 games per cycle the merged pickle would be several GB through one process)
        os.makedirs(a.games_dir, exist_ok=True)
        torch.save(pack_games_for_save(games), os.path.join(a.games_dir, ("RL_960_%d.pt" % epoch) if rank == 0 else ("RL_960_%d.rank%d.pt" % (epoch, rank))))
        if a.merge_games and world > 1:
            dist.barrier()                                               # every rank's file is on disk
            if rank == 0:
                n_all = merge_game_files(a.games_dir, epoch, world)
                print("epoch %d: merged %d samples of %d ranks into %s" % (epoch, n_all, world, os.path.join(a.games_dir, "RL_960_%d.pt" % epoch)), flush=True)
        if rank == 0:
            save_cycle(model, optimiser, epoch, a.save_dir)
            if a.log_dir:
                write_step_log(os.path.join(a.log_dir, "RL_train.jsonl"), epoch, hist, sched)
            last = hist[-1] if hist else (float("nan"), float("nan"))
            print("epoch %d: %d ranks x %d games, %d samples on rank 0, %d optimiser steps, last mse %.4f ce %.4f"
                  % (epoch, world, a.games_per_rank, n_samples, len(hist), last[0], last[1]), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
