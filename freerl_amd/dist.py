"""Multi-GPU: one process per GPU, independent seeds per rank, ONE small collective.

The reference is single-process / single-device and none of its algorithms exchanges gradients
or replay data, so the path shards by independent units — (seed, env-instance
set, learner) — with no data-path collective; the only exchange is the all-reduce of a short
metrics vector (env steps, updates, sum of returns, episodes, sum of losses) per reporting
interval: `frl_metrics_allreduce` of the C ABI (include/freerl_hip.h) — ncclAllReduce over RCCL / xGMI on the
engine's own communicator — with this module as its binding and bootstrap; a gloo process group is the host-side
control plane and runs the same Python path on CPU tensors in the tests.  At <= 64 doubles the collective is
latency-bound; link bandwidth is irrelevant.
"""
import os

import torch
import torch.distributed as dist

METRIC_FIELDS = ("env_steps", "updates", "return_sum", "episodes", "loss_sum", "wall_s_max")


_comm = None          # frl_comm* (ctypes void pointer) of this rank, or None
_comm_note = "none (single process)"


def _native_comm_create(rank, world, local_rank):
    """The C ABI's RCCL communicator (include/freerl_hip.h: frl_comm_*): rank 0 makes the 128-byte id, the launcher's
    process group carries it to the other ranks, every rank joins.  Returns the handle or raises FrlError."""
    import ctypes as C
    from . import _native as N
    L = N.lib()
    # Joining is collective (ncclCommInitRank blocks until every rank has called it): a rank that cannot get as far as the call
    # — no librccl, no device — would leave the others waiting for ever.  Every rank probes first (making an id loads RCCL and
    # touches the device) and the ranks agree over the process group before anyone joins.
    buf = (C.c_uint8 * N.FRL_COMM_ID_BYTES)()
    err = None
    try:
        N.check(L.frl_comm_unique_id(buf))
    except Exception as ex:      # noqa: BLE001 - agreed on below
        err = ex
    if not _all_ok(err is None):
        raise err if err else N.FrlError("another rank cannot create the RCCL communicator")
    uid = torch.zeros(N.FRL_COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        uid = torch.tensor(list(buf), dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        uid = uid.cuda()
    dist.broadcast(uid, src=0)
    raw = (C.c_uint8 * N.FRL_COMM_ID_BYTES)(*[int(x) for x in uid.cpu().tolist()])
    h = C.c_void_p()
    err = None
    # Known limit of the agreement rounds: they bracket the join, they cannot reach INSIDE it.  A rank that dies or errors inside
    # ncclCommInitRank after its peers have entered it leaves those peers blocked in RCCL's own bootstrap, before the second
    # _all_ok — nothing at this level can interrupt a blocked native call.  What bounds that wait is the launcher: torch.distributed.run
    # tears the whole job down when one worker exits non-zero, and RCCL's bootstrap gives up on its socket timeout.  The first
    # round removes the failures that can be seen from outside (no librccl, no device, an id that cannot be made).
    try:
        N.check(L.frl_comm_create(raw, int(rank), int(world), int(local_rank), C.byref(h)))
    except Exception as ex:      # noqa: BLE001 - agreed on below
        err, h = ex, None
    # The join can still fail on ONE rank (device_id out of range, an allocation after ncclCommInitRank, an RCCL error on one
    # GPU).  The communicator is used by every rank or by none: a second agreement round, and the ranks that did join tear
    # theirs down if anybody did not — otherwise barrier() / allreduce_metrics() would run ncclAllReduce on some ranks and
    # gloo on the others and the job would hang.
    if not _all_ok(err is None):
        if h is not None:
            L.frl_comm_destroy(h)
        raise err if err else N.FrlError("another rank failed to join the RCCL communicator")
    return h


def _all_ok(mine):
    """True iff every rank of the process group reports success (all_reduce MIN of a flag)."""
    ok = torch.tensor([1 if mine else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        ok = ok.cuda()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return int(ok.item()) == 1


def init(backend=None):
    """Initialise from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR/PORT).  Returns (rank, world_size,
    local_rank).  A process started without a launcher (WORLD_SIZE unset) is the degenerate single-rank case and creates
    nothing.  Under a launcher — even with one rank — two things are set up:
      * a torch.distributed process group (gloo) as the host-side control plane: rendezvous, barrier, and the channel the
        RCCL bootstrap id travels over;
      * on a GPU box, the engine's own RCCL communicator (frl_comm_create) — the metric all-reduce is the C ABI's
        frl_metrics_allreduce over it.  backend="gloo" skips this (CPU tests: the same Python path, gloo tensors).
    If the communicator cannot be created the reduce falls back to the process group with a warning on stderr: the metric
    exchange is bookkeeping, not the product path, and a bench line over gloo beats no line."""
    global _comm, _comm_note
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost") and os.path.exists("/sys/class/net/lo"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # the container hostname may not resolve
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")        # RCCL's bootstrap sockets: one node, loopback is enough
        want_native = backend != "gloo" and torch.cuda.is_available()
        if want_native:
            torch.cuda.set_device(local_rank)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        except Exception:
            if not want_native:
                raise
            dist.init_process_group("nccl", rank=rank, world_size=world)
        _comm_note = "torch.distributed %s (%d rank%s)" % (dist.get_backend(), world, "" if world == 1 else "s")
        if want_native:
            try:
                _comm = _native_comm_create(rank, world, local_rank)
                _comm_note = "frl_metrics_allreduce over RCCL (%d rank%s; bootstrap via torch.distributed %s)" % (
                    world, "" if world == 1 else "s", dist.get_backend())
            except Exception as ex:      # noqa: BLE001 - reported, then the process group carries the metrics
                import sys
                print("freerl_amd.dist: RCCL communicator unavailable (%s); metrics go over torch.distributed %s"
                      % (ex, dist.get_backend()), file=sys.stderr, flush=True)
    return rank, world, local_rank


def collective_name():
    """What carries the metric all-reduce in this process (for the bench line)."""
    return _comm_note


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
    """Barrier + destroy the communicator and the process group (no-op without them)."""
    global _comm
    barrier()
    if _comm is not None:
        from . import _native as N
        N.lib().frl_comm_destroy(_comm)
        _comm = None
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def shard_seeds(seeds, rank, world):
    """Seed s (or env-instance block s) -> rank s mod world: every rank owns
    its learners outright (own replay ring, weights, RNG stream)."""
    return [s for i, s in enumerate(seeds) if i % world == rank]


def shard_count(n_units, rank, world):
    """How many of n_units independent learners rank `rank` hosts (remainder to the low ranks)."""
    return n_units // world + (1 if rank < n_units % world else 0)


def barrier():
    """All ranks have arrived: a one-word all-reduce on the RCCL communicator when there is one (the GPU ranks' own
    channel), the process group's barrier otherwise."""
    if _comm is not None:
        import ctypes as C
        from . import _native as N
        one = (C.c_double * 1)(1.0)
        N.check(N.lib().frl_metrics_allreduce(_comm, one, 1, None, 0))
    elif dist.is_available() and dist.is_initialized():
        dist.barrier()


def allreduce_metrics(env_steps, updates, return_sum, episodes, loss_sum, wall_s, device=None, extra_max=()):
    """Sum the counters over ranks and take the MAX of the wall-clock (the job's time is the slowest rank's) and of any
    `extra_max` values (returned as a list under "extra_max").  Returns a dict keyed by METRIC_FIELDS.  With the RCCL
    communicator of init() this is ONE call of the C ABI's frl_metrics_allreduce (float64, ncclAllReduce sum + max);
    without it the process group's all_reduce on CPU tensors (gloo tests), and without either the single-process identity."""
    sums = [float(env_steps), float(updates), float(return_sum), float(episodes), float(loss_sum)]
    tm = [float(wall_s)] + [float(x) for x in extra_max]
    if _comm is not None:
        import ctypes as C
        from . import _native as N
        cs, cm = (C.c_double * len(sums))(*sums), (C.c_double * len(tm))(*tm)
        N.check(N.lib().frl_metrics_allreduce(_comm, cs, len(sums), cm, len(tm)))
        sums, tm = list(cs), list(cm)
    elif dist.is_available() and dist.is_initialized():        # also with one rank: the collective still runs
        dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        ts = torch.tensor(sums, dtype=torch.float64, device=dev)
        tt = torch.tensor(tm, dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sums, tm = ts.tolist(), tt.tolist()
    out = dict(zip(METRIC_FIELDS, sums + tm[:1]))
    out["extra_max"] = tm[1:]
    return out


def throughput(metrics):
    """Whole-job env-steps/s and updates/s from an all-reduced metrics dict."""
    t = max(metrics["wall_s_max"], 1e-12)
    return dict(env_steps_per_sec=metrics["env_steps"] / t, updates_per_sec=metrics["updates"] / t)
