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
