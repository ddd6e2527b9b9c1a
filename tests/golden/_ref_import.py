"""Import the FreeRL reference classes in THIS container only (golden-vector generation).

Test infrastructure.  Nothing here is imported by the product (`freerl_amd/`), by the
`-m gpu` tests, by `bench.py` or by `__graft_entry__.smoke()`: `/root/reference` does not
exist on the GPU box.  Only `tests/golden/make_golden.py` (run by hand, here) and the
optional here-only cross-check test use it.

Recipe: stub the three absent third-party modules, push the reference
directory on sys.path, pop the same-named helper modules between directories, and import the
script as a module (its ``__main__`` guard keeps the training loop from running).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("FREERL_REFERENCE", "/root/reference")

_HELPERS = ("Buffer", "normalization", "c_adamw", "Noisy_net")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "DQN_file"))


def _install_stubs():
    sys.dont_write_bytecode = True
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")

        class Box:  # noqa: D401 - dummy
            pass

        class Discrete:
            pass

        spaces.Box, spaces.Discrete = Box, Discrete
        gym.spaces = spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        import torch.utils

        tb = types.ModuleType("torch.utils.tensorboard")
        tb.SummaryWriter = object
        sys.modules["torch.utils.tensorboard"] = tb
        torch.utils.tensorboard = tb
    if "pettingzoo" not in sys.modules:
        sys.modules["pettingzoo"] = types.ModuleType("pettingzoo")


def import_reference(directory, script):
    """Return the module object of /root/reference/<directory>/<script>.py."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    path = os.path.join(REF_ROOT, directory)
    for name in _HELPERS + (script,):
        sys.modules.pop(name, None)
    sys.path.insert(0, path)
    try:
        mod = importlib.import_module(script)
    finally:
        sys.path.remove(path)
    # detach the helper modules again so that the next directory gets its own copies
    helpers = {n: sys.modules.pop(n) for n in _HELPERS if n in sys.modules}
    mod._helpers = helpers
    sys.modules.pop(script, None)
    return mod
