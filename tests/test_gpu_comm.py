"""The path's one collective on the GPU: frl_metrics_allreduce over the engine's RCCL communicator (one rank — gpurun
leases one GPU; N > 1 is covered on CPU by tests/test_dist_gloo.py through the same freerl_amd.dist code)."""
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
