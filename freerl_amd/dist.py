"""Multi-GPU: one process per GPU, independent seeds per rank, ONE small collective.

The reference is single-process / single-device and none of its algorithms exchanges gradients
or replay data, so the path shards by independent units — (seed, env-instance
set, learner) — with no data-path collective; the only exchange is the all-reduce of a short
metrics vector (env steps, updates, sum of returns, episodes, sum of losses) per reporting
interval.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" runs the same code on CPU tensors
(tests).  At <= 16 floats the collective is latency-bound; link bandwidth is irrelevant.
"""
import os

import torch
import torch.distributed as dist

METRIC_FIELDS = ("env_steps", "updates", "return_sum", "episodes", "loss_sum", "wall_s_max")


def init(backend=None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    MASTER_ADDR/PORT).  Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_seeds(seeds, rank, world):
    """Seed s (or env-instance block s) -> rank s mod world: every rank owns
    its learners outright (own replay ring, weights, RNG stream)."""
    return [s for i, s in enumerate(seeds) if i % world == rank]


def shard_count(n_units, rank, world):
    """How many of n_units independent learners rank `rank` hosts (remainder to the low ranks)."""
    return n_units // world + (1 if rank < n_units % world else 0)


def allreduce_metrics(env_steps, updates, return_sum, episodes, loss_sum, wall_s, device=None):
    """Sum the counters over ranks and take the MAX of the wall-clock (the job's time is the
    slowest rank's).  Returns a dict keyed by METRIC_FIELDS; works without a process group."""
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() and dist.is_initialized()
                                             and dist.get_backend() == "nccl" else "cpu")
    sums = torch.tensor([env_steps, updates, return_sum, episodes, loss_sum], dtype=torch.float64, device=dev)
    tmax = torch.tensor([wall_s], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    vals = sums.tolist() + tmax.tolist()
    return dict(zip(METRIC_FIELDS, vals))


def throughput(metrics):
    """Whole-job env-steps/s and updates/s from an all-reduced metrics dict."""
    t = max(metrics["wall_s_max"], 1e-12)
    return dict(env_steps_per_sec=metrics["env_steps"] / t, updates_per_sec=metrics["updates"] / t)
