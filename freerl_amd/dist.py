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
    """Initialise the default process group from the launcher's environment (RANK, WORLD_SIZE,
    MASTER_ADDR/PORT).  Returns (rank, world_size, local_rank).  A process started without a
    launcher (WORLD_SIZE unset) is the degenerate single-rank case and creates no group; under a
    launcher the group is created even for one rank, so a 1-GPU run exercises RCCL's init and
    all-reduce too."""
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn(n_ranks, script, argv, extra_env=None, timeout=None):
    """Run `script argv...` as n_ranks processes of ONE node under torch.distributed.run (one rank
    per GPU, rendezvous on 127.0.0.1) and return its exit code: what `bench.py --gpus N` does when
    it is started without a launcher.  The child ranks see RANK / LOCAL_RANK / WORLD_SIZE and call
    init()."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL / cross-process device memory on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    if extra_env:
        env.update(extra_env)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def finalize():
    """Barrier + destroy the process group (no-op without one)."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def shard_seeds(seeds, rank, world):
    """Seed s (or env-instance block s) -> rank s mod world: every rank owns
    its learners outright (own replay ring, weights, RNG stream)."""
    return [s for i, s in enumerate(seeds) if i % world == rank]


def shard_count(n_units, rank, world):
    """How many of n_units independent learners rank `rank` hosts (remainder to the low ranks)."""
    return n_units // world + (1 if rank < n_units % world else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def allreduce_metrics(env_steps, updates, return_sum, episodes, loss_sum, wall_s, device=None, extra_max=()):
    """Sum the counters over ranks and take the MAX of the wall-clock (the job's time is the
    slowest rank's) and of any `extra_max` values (returned as a list under "extra_max").  Returns a
    dict keyed by METRIC_FIELDS; works without a process group."""
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() and dist.is_initialized()
                                             and dist.get_backend() == "nccl" else "cpu")
    sums = torch.tensor([env_steps, updates, return_sum, episodes, loss_sum], dtype=torch.float64, device=dev)
    tmax = torch.tensor([wall_s] + [float(x) for x in extra_max], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized():        # also with one rank: the collective still runs (RCCL smoke)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tm = tmax.tolist()
    out = dict(zip(METRIC_FIELDS, sums.tolist() + tm[:1]))
    out["extra_max"] = tm[1:]
    return out


def throughput(metrics):
    """Whole-job env-steps/s and updates/s from an all-reduced metrics dict."""
    t = max(metrics["wall_s_max"], 1e-12)
    return dict(env_steps_per_sec=metrics["env_steps"] / t, updates_per_sec=metrics["updates"] / t)
