"""N > 1 path on CPU: world_size-2 gloo processes exercise seed sharding and the metric
all-reduce (the only collective of the path)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    from freerl_amd import dist as fd
    rank, world, local = fd.init("gloo")
    assert world == 2
    seeds = fd.shard_seeds([0, 10, 100, 7, 8], rank, world)
    n_mine = fd.shard_count(5, rank, world)
    assert len(seeds) == n_mine
    m = fd.allreduce_metrics(env_steps=100.0 * (rank + 1), updates=10.0 * len(seeds), return_sum=-5.0 * rank,
                             episodes=rank + 1, loss_sum=0.25, wall_s=1.0 + rank)
    thr = fd.throughput(m)
    print(json.dumps(dict(rank=rank, seeds=seeds, metrics=m, thr=thr)))
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
''') % ROOT


def test_two_rank_seed_sharding_and_metric_allreduce(tmp_path):
    import json
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 500)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["seeds"] == [0, 100, 8] and outs[1]["seeds"] == [10, 7]        # s -> rank s mod G
    for d in outs:                                                                # identical on every rank
        m = d["metrics"]
        assert m["env_steps"] == 300.0 and m["updates"] == 50.0 and m["return_sum"] == -5.0
        assert m["episodes"] == 3.0 and m["loss_sum"] == 0.5 and m["wall_s_max"] == 2.0
        assert d["thr"]["env_steps_per_sec"] == 150.0 and d["thr"]["updates_per_sec"] == 25.0


def test_single_process_is_the_degenerate_case():
    sys.path.insert(0, ROOT)
    from freerl_amd import dist as fd
    assert fd.shard_seeds([3, 4, 5], 0, 1) == [3, 4, 5] and fd.shard_count(7, 0, 1) == 7
    m = fd.allreduce_metrics(10, 5, 1.0, 2, 0.5, 2.0, device="cpu")
    assert m["env_steps"] == 10 and m["wall_s_max"] == 2.0


SPAWN_WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    from freerl_amd import dist as fd
    rank, world, local = fd.init("gloo")            # the launcher's env: group exists even for one rank
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_world_size() == world == int(sys.argv[1])
    m = fd.allreduce_metrics(env_steps=7.0, updates=float(rank + 1), return_sum=0.0, episodes=1.0, loss_sum=0.0,
                             wall_s=1.0 + rank, extra_max=[10.0 * (rank + 1)])
    fd.finalize()
    assert not dist.is_initialized()
    if rank == 0:
        open(sys.argv[2], "w").write(json.dumps(m))
''') % ROOT


def _spawn(tmp_path, n):
    import json
    sys.path.insert(0, ROOT)
    from freerl_amd import dist as fd
    script, out = tmp_path / "sw.py", tmp_path / "out.json"
    script.write_text(SPAWN_WORKER)
    rc = fd.respawn(n, str(script), [str(n), str(out)], timeout=300)
    assert rc == 0
    return json.loads(out.read_text())


def test_respawn_path_two_ranks(tmp_path):
    """What `bench.py --gpus N` does without a launcher: re-exec under torch.distributed.run, one rank per unit,
    init from the env, the metric all-reduce, finalize."""
    m = _spawn(tmp_path, 2)
    assert m["env_steps"] == 14.0 and m["updates"] == 3.0 and m["wall_s_max"] == 2.0 and m["extra_max"] == [20.0]


def test_respawn_path_single_rank_still_runs_the_collective(tmp_path):
    m = _spawn(tmp_path, 1)
    assert m["env_steps"] == 7.0 and m["wall_s_max"] == 1.0 and m["extra_max"] == [10.0]


def test_bench_refuses_more_gpus_than_devices():
    """`python bench.py --gpus 2` on a node with fewer HIP devices must fail loudly, never print n_gpus: 1."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 HIP devices" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout


JOIN_FAIL_WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from freerl_amd import dist as fd
    from freerl_amd import _native as N
    rank, world, local = fd.init("gloo")
    destroyed = []

    class FakeLib:                       # frl_comm_* of the C ABI: the id probe succeeds everywhere, the JOIN fails on rank 1 only
        def frl_comm_unique_id(self, buf):
            return 0
        def frl_comm_create(self, raw, r, w, dev, out):
            if r == 1:
                return 1
            out._obj.value = 1234
            return 0
        def frl_comm_destroy(self, h):
            destroyed.append(int(h.value))
            return 0

    N.lib = lambda: FakeLib()
    def check(rc):
        if rc:
            raise N.FrlError("injected join failure")
    N.check = check
    try:
        fd._native_comm_create(rank, world, local)
        outcome = "joined"
    except N.FrlError as ex:
        outcome = "raised: " + str(ex)
    # every rank is back on the process group: the same collective completes on both
    m = fd.allreduce_metrics(1.0, 1.0, 0.0, 0.0, 0.0, 1.0)
    print(json.dumps(dict(rank=rank, outcome=outcome, destroyed=destroyed, env_steps=m["env_steps"])))
    dist.barrier(); dist.destroy_process_group()
''') % ROOT


def test_join_failure_on_one_rank_leaves_no_rank_with_a_communicator(tmp_path):
    """frl_comm_create failing on ONE rank: the ranks that joined destroy their communicator and every rank raises, so
    init() falls back to the process group everywhere (a communicator on some ranks only would hang the next collective)."""
    import json
    script = tmp_path / "jf.py"
    script.write_text(JOIN_FAIL_WORKER)
    port = 30100 + (os.getpid() % 500)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["outcome"].startswith("raised") and outs[1]["outcome"].startswith("raised")
    assert outs[0]["destroyed"] == [1234] and outs[1]["destroyed"] == []
    assert outs[0]["env_steps"] == 2.0 and outs[1]["env_steps"] == 2.0
