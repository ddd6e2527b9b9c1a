"""draw_kernel's duplicate check for batches of 257 .. 2048 rows (MADDPG's 1024) is a hash table in LDS — every entry inserted under its
row, a slot keeps the smallest position that holds the row — instead of the scan of every entry's predecessors (FRL_DRAW_SCAN=1: 92 us
for config 5's 3 x 1024 rows).  Same rule (the LATER of two equal entries is redrawn, round by round), same Philox streams: the rows
must be identical, also from a ring barely twice the batch, where a draw collides hundreds of times and takes several rounds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from freerl_amd import _native
    assert _native.device_count() > 0
    return _native


def _rows(N, monkeypatch, scan, B, size, P, family):
    from freerl_amd.engine import Engine
    monkeypatch.setenv("FRL_DRAW_SCAN", "1" if scan else "0")
    if family is None:
        monkeypatch.delenv("FRL_CRITIC_V2", raising=False)
    else:
        monkeypatch.setenv("FRL_CRITIC_V2", str(family))
    e = Engine(N.ALGO_MADDPG, [6, 5, 7], [2, 3, 2], 4096, n_learners=P, batch_max=B, seed=23)
    e.fill_synthetic(size, seed=2)
    out = []
    for k in range(6):
        e.learn(B, gamma=0.95, tau=0.01, actor_lr=1e-3, critic_lr=1e-3)
        out.append(e.last_indices(B).copy())
    e.close()
    return np.stack(out)                       # [calls][P][agents][B]


@pytest.mark.parametrize("B,size,P,family", [(1024, 4096, 1, None), (1000, 2003, 2, 0), (300, 4096, 3, 0), (2048, 4096, 1, 0)])
def test_table_and_scan_draw_the_same_rows(N, monkeypatch, B, size, P, family):
    a = _rows(N, monkeypatch, False, B, size, P, family)
    b = _rows(N, monkeypatch, True, B, size, P, family)
    np.testing.assert_array_equal(a, b)
    assert a.min() >= 0 and a.max() < size
    for k in range(a.shape[0]):
        for p in range(P):
            for j in range(3):
                assert len(np.unique(a[k, p, j])) == B, "a batch holds a row twice"
    assert not np.array_equal(a[0, 0, 0], a[0, 0, 1]) and not np.array_equal(a[0, 0, 0], a[1, 0, 0])
