"""CPU-side checks (no GPU, no compute calls): the C-ABI library builds, loads and exports every
symbol include/freerl_hip.h declares; the ctypes struct mirrors match the C structs; host logic
fails loudly without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "freerl_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(frl_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def native():
    from freerl_amd import _native
    _native.build()
    return _native


def test_header_symbols_are_all_exported_and_bound(native):
    fns = declared_functions()
    assert len(fns) >= 30
    L = native.lib()
    for f in fns:
        assert hasattr(L, f), "libfreerl_hip.so does not export %s" % f
        assert f in native.SIGNATURES, "freerl_amd/_native.py does not bind %s" % f
    assert sorted(native.SIGNATURES) == fns, "binding lists symbols the header does not declare"
    assert L.frl_version() >= 100


def test_ctypes_structs_match_c_layout(native, tmp_path):
    """sizeof/offsetof of the C structs, measured by compiling a probe against the header."""
    probe = tmp_path / "probe.c"
    probe.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "freerl_hip.h"\nint main(){'
                     'printf("%zu %zu %zu %zu ", sizeof(frl_config), sizeof(frl_record_layout), sizeof(frl_learn_args), sizeof(frl_ppo_args));'
                     'printf("%zu %zu ", sizeof(frl_rollout_args), sizeof(frl_rollout_stats));'
                     'printf("%zu %zu %zu %zu\\n", offsetof(frl_config, seed), offsetof(frl_learn_args, idx), offsetof(frl_learn_args, stats_out), offsetof(frl_ppo_args, perms));'
                     'return 0;}')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(native.Config), C.sizeof(native.RecordLayout), C.sizeof(native.LearnArgs), C.sizeof(native.PpoArgs),
            C.sizeof(native.RolloutArgs), C.sizeof(native.RolloutStats),
            native.Config.seed.offset, native.LearnArgs.idx.offset, native.LearnArgs.stats_out.offset,
            native.PpoArgs.perms.offset]
    assert got == want


def _gcc_layout(tmp_path, structs):
    """{struct: (sizeof, {field: offset})} measured by compiling a probe against include/freerl_hip.h."""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "freerl_hip.h"', 'int main(void){']
    for name, fields in structs:
        lines.append('printf("%s %%zu", sizeof(%s));' % (name, name))
        for f, _ in fields:
            lines.append('printf(" %s=%%zu", offsetof(%s, %s));' % (f, name, f))
        lines.append('printf("\\n");')
    lines.append('return 0;}')
    probe = tmp_path / "layout.c"
    probe.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    out = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        parts = line.split()
        out[parts[0]] = (int(parts[1]), {kv.split("=")[0]: int(kv.split("=")[1]) for kv in parts[2:]})
    return out


def _load_stub_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_ctypes_stub", os.path.join(ROOT, "tools", "gen_ctypes_stub.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_integration_md_stub_matches_the_header(tmp_path):
    """INTEGRATION.md's ctypes block is what a maintainer copies: execute it and compare every struct's size and every
    field's offset with the C compiler's view of include/freerl_hip.h (round 2's hand-written block was 24 bytes short)."""
    tool = _load_stub_tool()
    assert tool.main(["--check"]) == 0, "INTEGRATION.md is stale: python tools/gen_ctypes_stub.py"
    structs, _ = tool.parse_structs()
    assert {n for n, _ in structs} >= {"frl_config", "frl_learn_args", "frl_ppo_args", "frl_rollout_args", "frl_record_layout"}
    ns = {}
    exec(tool.extract(), ns)
    want = _gcc_layout(tmp_path, structs)
    for name, fields in structs:
        cls = ns[name]
        assert C.sizeof(cls) == want[name][0], name
        for f, _ in fields:
            assert getattr(cls, f).offset == want[name][1][f], (name, f)


def test_native_binding_field_offsets(native, tmp_path):
    """freerl_amd/_native.py's mirrors, field by field (the size-only test above this one would miss two swapped ints)."""
    tool = _load_stub_tool()
    structs, _ = tool.parse_structs()
    want = _gcc_layout(tmp_path, structs)
    mirror = {"frl_config": native.Config, "frl_record_layout": native.RecordLayout, "frl_learn_args": native.LearnArgs,
              "frl_ppo_args": native.PpoArgs, "frl_explore_args": native.ExploreArgs, "frl_rollout_args": native.RolloutArgs,
              "frl_rollout_stats": native.RolloutStats, "frl_ppo_rollout_args": native.PpoRolloutArgs}
    assert set(mirror) == {n for n, _ in structs}
    for name, fields in structs:
        cls = mirror[name]
        assert C.sizeof(cls) == want[name][0], name
        assert [f for f, _ in cls._fields_] == [f for f, _ in fields], name
        for f, _ in fields:
            assert getattr(cls, f).offset == want[name][1][f], (name, f)


def test_metrics_allreduce_abi_without_device(native):
    """The collective's C entry point: NULL communicator = the single-process identity; argument checks; a communicator
    cannot be created without a HIP device (no CPU fallback for the RCCL path either)."""
    L = native.lib()
    sums = (C.c_double * 3)(1.0, 2.0, 3.0)
    mx = (C.c_double * 2)(4.0, 5.0)
    assert L.frl_metrics_allreduce(None, sums, 3, mx, 2) == 0
    assert list(sums) == [1.0, 2.0, 3.0] and list(mx) == [4.0, 5.0]
    assert L.frl_metrics_allreduce(None, sums, native.FRL_COMM_MAX_VALUES + 1, mx, 2) == 1
    assert L.frl_metrics_allreduce(None, None, 3, mx, 2) == 1
    rank, world = C.c_int(-1), C.c_int(-1)
    assert L.frl_comm_info(None, C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
    import torch
    if not torch.cuda.is_available():
        uid = (C.c_uint8 * native.FRL_COMM_ID_BYTES)()
        h = C.c_void_p()
        assert L.frl_comm_create(uid, 0, 1, 0, C.byref(h)) == 3                 # FRL_ERR_NO_DEVICE
        assert b"no HIP device" in L.frl_last_error()
        assert L.frl_comm_create(uid, 2, 2, 0, C.byref(h)) == 1                 # rank outside world
    assert L.frl_comm_destroy(None) == 0


def test_header_is_plain_c(tmp_path):
    probe = tmp_path / "c.c"
    probe.write_text('#include "freerl_hip.h"\nint main(void){return FRL_OK;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(probe),
                           "-o", str(tmp_path / "c.o")])


def test_no_device_fails_loudly(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert native.device_count() == 0
    from freerl_amd.engine import Engine
    with pytest.raises(native.FrlError, match="no HIP device|no CPU fallback"):
        Engine(native.ALGO_DQN, 8, 4, 100, discrete=True)
    from freerl_amd.DQN import DQN
    with pytest.raises(native.FrlError, match="no CPU fallback"):
        DQN([8, 4], False, 1e-3, 100, torch.device("cpu"))
    from freerl_amd.Buffer import Buffer
    with pytest.raises(native.FrlError):
        Buffer(100, 3, 1, "cpu")


def test_argument_validation_without_device(native):
    L = native.lib()
    h = C.c_void_p()
    assert L.frl_create(None, C.byref(h)) == 1                      # FRL_ERR_INVALID
    assert b"NULL" in L.frl_last_error()
    cfg = native.Config()
    cfg.algo, cfg.n_learners, cfg.n_agents, cfg.capacity = native.ALGO_DQN, 0, 1, 10
    assert L.frl_create(C.byref(cfg), C.byref(h)) == 1 and b"n_learners" in L.frl_last_error()
    cfg.n_learners, cfg.n_agents = 1, 9
    assert L.frl_create(C.byref(cfg), C.byref(h)) == 1
    assert L.frl_sync(None) == 1 and L.frl_learn(None, None) == 1
    assert L.frl_destroy(None) == 0


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under freerl_amd/ may import it."""
    pkg = os.path.join(ROOT, "freerl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_normalization_mirror_matches_oracle():
    from freerl_amd import normalization as P
    from oracle import normalization as O
    g = np.random.default_rng(3)
    xs = g.standard_normal((9, 4)) * 3 + 1
    a, b = P.Normalization(4), O.Normalization(4)
    for x in xs:
        np.testing.assert_allclose(a(x.copy()), b(x.copy()), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a(xs[0], update=False), b(xs[0], update=False), rtol=1e-12)
    ra, rb = P.RewardScaling(1, 0.99), O.RewardScaling(1, 0.99)
    for r in g.standard_normal(7):
        np.testing.assert_allclose(ra(r), rb(r), rtol=1e-12)
    ra.reset(); rb.reset()
    np.testing.assert_allclose(ra(0.5), rb(0.5), rtol=1e-12)
