"""The path's one collective on the GPU: frl_metrics_allreduce over the engine's RCCL communicator — one rank in process and through the
launcher path, and test_n_rank_communicator with one rank per visible GPU (skipped on a one-GPU box such as gpurun's; N > 1 is covered
on CPU by tests/test_dist_gloo.py through the same freerl_amd.dist code)."""
import ctypes as C
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_communicator_in_process():
    from freerl_amd import _native as N
    L = N.lib()
    assert N.device_count() > 0
    uid = (C.c_uint8 * N.FRL_COMM_ID_BYTES)()
    N.check(L.frl_comm_unique_id(uid))
    assert any(uid), "ncclGetUniqueId left the id empty"
    h = C.c_void_p()
    N.check(L.frl_comm_create(uid, 0, 1, 0, C.byref(h)))
    rank, world = C.c_int(-1), C.c_int(-1)
    N.check(L.frl_comm_info(h, C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    sums = (C.c_double * 5)(1e15 + 1.0, 2.0, -3.5, 4.0, 0.25)          # counters stay exact in float64
    mx = (C.c_double * 2)(1.5, -7.0)
    for _ in range(3):
        N.check(L.frl_metrics_allreduce(h, sums, 5, mx, 2))
    assert list(sums) == [1e15 + 1.0, 2.0, -3.5, 4.0, 0.25] and list(mx) == [1.5, -7.0]
    N.check(L.frl_comm_destroy(h))


WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    from freerl_amd import dist as fd
    rank, world, local = fd.init()
    m = fd.allreduce_metrics(env_steps=7.0, updates=3.0, return_sum=-1.0, episodes=1.0, loss_sum=0.5, wall_s=2.0, extra_max=[9.0])
    fd.barrier()
    name = fd.collective_name()
    fd.finalize()
    open(sys.argv[1], "w").write(json.dumps(dict(m=m, name=name)))
''') % ROOT


def test_launcher_path_uses_the_native_collective(tmp_path):
    """What `bench.py --spawn` does: one rank under torch.distributed.run; dist.init() must bring up frl_comm_create and
    carry the metrics through frl_metrics_allreduce (not the process-group fallback)."""
    sys.path.insert(0, ROOT)
    from freerl_amd import dist as fd
    script, out = tmp_path / "w.py", tmp_path / "o.json"
    script.write_text(WORKER)
    assert fd.respawn(1, str(script), [str(out)], timeout=600) == 0
    r = json.loads(out.read_text())
    assert "frl_metrics_allreduce over RCCL" in r["name"], r["name"]
    assert r["m"]["env_steps"] == 7.0 and r["m"]["wall_s_max"] == 2.0 and r["m"]["extra_max"] == [9.0]


WORKER_N = textwrap.dedent('''
    import ctypes as C, json, os, sys
    sys.path.insert(0, %r)
    from freerl_amd import _native as N, dist as fd
    rank, world, local = fd.init()
    r, w = C.c_int(-1), C.c_int(-1)
    N.check(N.lib().frl_comm_info(fd._comm, C.byref(r), C.byref(w)))
    import torch
    m = fd.allreduce_metrics(env_steps=7.0 + rank, updates=3.0, return_sum=-1.0, episodes=1.0, loss_sum=0.5, wall_s=2.0 + rank,
                             extra_max=[float(torch.cuda.current_device())])
    fd.barrier()
    name = fd.collective_name()
    fd.finalize()
    open(sys.argv[1] + ".%%d" %% rank, "w").write(json.dumps(dict(m=m, name=name, info=[r.value, w.value], device=torch.cuda.current_device(),
                                                                   native=fd._comm is None)))
''') % ROOT


def test_n_rank_communicator(tmp_path):
    """Every GPU of the node as one rank (north_star: seeds sharded over the GPUs, RCCL only for the metric all-reduce): skipped on a
    one-GPU box, evidence on the first multi-GPU one.  frl_comm_info reports `world` ranks on distinct devices, the summed counters are
    world x one rank's (+ the rank offsets), the maxed wall-clock is the slowest rank's, and bench.py --gpus world --headline-only prints
    a line whose collective is RCCL with `world` ranks and whose value counts every rank's learners."""
    sys.path.insert(0, ROOT)
    from freerl_amd import _native as N
    from freerl_amd import dist as fd
    world = min(N.device_count(), 8)
    if world < 2:
        pytest.skip("one HIP device: N > 1 ranks need a multi-GPU node (tests/test_dist_gloo.py covers the code on CPU)")
    script, out = tmp_path / "w.py", tmp_path / "o.json"
    script.write_text(WORKER_N)
    assert fd.respawn(world, str(script), [str(out)], timeout=900) == 0
    rs = [json.loads((tmp_path / ("o.json.%d" % r)).read_text()) for r in range(world)]
    assert sorted(r["device"] for r in rs) == list(range(world)), "ranks do not sit on distinct devices"
    for rank, r in enumerate(rs):
        assert r["info"] == [rank, world]
        assert "frl_metrics_allreduce over RCCL (%d ranks" % world in r["name"], r["name"]
        assert r["m"]["env_steps"] == 7.0 * world + world * (world - 1) / 2 and r["m"]["updates"] == 3.0 * world
        assert r["m"]["wall_s_max"] == 2.0 + world - 1 and r["m"]["extra_max"] == [float(world - 1)]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--headline-only", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and line["scaling"] == "weak"
    assert "RCCL (%d ranks" % world in line["config"]["collective"], line["config"]["collective"]
