"""One train_RL cycle on this rank's GPU, phase by phase (BASELINE config 4 on one GPU; under torch.distributed.run every rank
does the same with GradSync over RCCL): inference-network build, self-play, record packing, optimiser steps.  One JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_zero_amd as sz
from sigma_zero_amd import train_rl as T
from sigma_zero_amd.fastnet import FastPolicyNet, SplitPolicyNet
from sigma_zero_amd.sim import play_games

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=512)
ap.add_argument("--searches", type=int, default=100)
ap.add_argument("--max-plies", type=int, default=40)
ap.add_argument("--batch-size", type=int, default=128)
ap.add_argument("--total-steps", type=int, default=6)
ap.add_argument("--backend", default="nccl")
ap.add_argument("--slots", type=int, default=0, help="board slots (default: one per game); fewer slots than games = refill + compaction")
ap.add_argument("--train-convs", default="split", choices=["split", "torch"], help="3x3 convolutions of the train step (train_rl.train split_convs)")
ap.add_argument("--train-graph", default="off", choices=["on", "off"], help="train_rl.train graph")
ap.add_argument("--inference", default="fp16", choices=["fp16", "bf16", "split"], help="self-play network (run_cycle's default is fp16)")
a = ap.parse_args()
rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", local_rank % torch.cuda.device_count())
torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group(backend=a.backend)
torch.manual_seed(0); np.random.seed(rank); import random; random.seed(rank)
model = sz.policyNN({}).to(dev)
T.sync_module_state(model, average_buffers=False)
opt, sched = T.make_optimiser(model)
sync = T.GradSync(model) if world > 1 else None
def sync_t():
    torch.cuda.synchronize(dev); return time.perf_counter()
t0 = sync_t()
model.eval()
player = SplitPolicyNet(model, device=dev) if a.inference == "split" else FastPolicyNet(model, device=dev, operands=a.inference)
t1 = sync_t()
args = {"C": 2, "num_searches": a.searches}
st = {}
games = play_games(player, args, a.games, c960=True, max_plies=a.max_plies, n_boards=a.slots or None, stats=st, verbose=True)
t2 = sync_t()
packed, aidx, aprob, rew = T.records_from_games(games)
dl = T.DeviceBatches(packed, aidx, aprob, rew, batch_size=a.batch_size, device=dev, shuffle=True)
t3 = sync_t()
hist = T.train(model, dl, opt, total_steps=a.total_steps, lr_scheduler=sched, sync=sync, device=dev, split_convs=None if a.train_convs == "split" else False, graph=(a.train_graph == "on"))
t4 = sync_t()
T.sync_module_state(model, average_buffers=True)
t5 = sync_t()
plies = sum(len(g["actions"]) for g in games)
lens = sorted(len(g["actions"]) for g in games)
out = {"workload": "train_RL cycle: %d Chess960 games/rank on %d slots x <=%d plies x %d searches (%s self-play network), then %d passes of batch %d (fp32, Adam)"
                   % (a.games, a.slots or a.games, a.max_plies, a.searches, a.inference, a.total_steps + 1, a.batch_size),
       "game_plies_min_median_max": [lens[0], lens[len(lens) // 2], lens[-1]], "self_play_work": st,
       "ranks": world, "samples_rank0": plies, "optimiser_steps": len(hist),
       "seconds": {"inference_net_build (BN fold + weight pack)": t1 - t0, "self_play": t2 - t1, "records + dataset": t3 - t2, "train": t4 - t3, "state sync": t5 - t4},
       "self_play_simulations_per_s": plies * a.searches / (t2 - t1), "train_share_of_cycle": (t4 - t3) / (t5 - t0), "train_samples_per_s": len(hist) * a.batch_size / max(t4 - t3, 1e-9),
       "train_ms_per_step": 1e3 * (t4 - t3) / max(len(hist), 1), "last_loss": list(hist[-1]) if hist else None}
if rank == 0:
    print(json.dumps(out))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
