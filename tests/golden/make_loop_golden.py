#!/usr/bin/env python3
"""Golden runs of the reference's OWN training loops (the `__main__` blocks).

Run by hand in the build container only:   python -m tests.golden.make_loop_golden
The reference scripts are executed in memory from /root/reference (never copied): their source
is compiled under a temporary file name (so `make_dir` writes its results/ tree into a temp dir
instead of the read-only reference tree) with `__name__ == "__main__"` and the reference's own
argparse flags in sys.argv.  The absent third-party packages are replaced by stubs:
  gymnasium          -> freerl_amd.envs (make / spec / spaces; in-repo Pendulum, CartPole, SynLinear)
  pettingzoo.mpe.*   -> freerl_amd.envs.SpreadEnv
  tensorboard        -> a no-op SummaryWriter
Every env is wrapped in a recorder, so the fixture holds the per-step env actions and rewards,
the per-episode returns the script saved, and digests of the checkpoint it wrote.
`tests/test_gpu_loops.py` replays the same flags through freerl_amd.train on the GPU.
"""
import contextlib
import glob
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from freerl_amd import envs as E  # noqa: E402
from tests.golden import synth  # noqa: E402
from tests.golden._ref_import import REF_ROOT  # noqa: E402

# name -> (directory, script, env kind, flags).  Short runs: the loop is a chaotic system, the
# comparison is step-by-step at fp32 tolerance.
LOOPS = {
    "loop_dqn_cartpole": ("DQN_file", "DQN", "--env_name CartPole-v1 --seed 0 --max_episodes 12 --save_freq 100 "
                          "--start_steps 64 --batch_size 32 --buffer_size 2000 --device cpu"),
    "loop_dqn_pendulum": ("DQN_file", "DQN", "--env_name PendulumShort-v1 --seed 0 --max_episodes 4 --save_freq 2 "
                          "--start_steps 100 --batch_size 64 --buffer_size 1000 --device cpu"),
    "loop_ddpg_pendulum": ("DDPG_file", "DDPG_simple", "--env_name PendulumShort-v1 --seed 0 --max_episodes 4 --save_freq 2 "
                           "--start_steps 100 --batch_size 64 --buffer_size 1000 --device cpu"),
    "loop_td3_pendulum": ("TD3_file", "TD3", "--env_name PendulumShort-v1 --seed 0 --max_episodes 4 --save_freq 2 "
                          "--start_steps 100 --batch_size 64 --buffer_size 1000 --gauss_sigma 0.1 --policy_noise 0.2 --device cpu"),
    "loop_sac_pendulum": ("SAC_file", "SAC", "--env_name PendulumShort-v1 --seed 0 --max_episodes 4 --save_freq 2 "
                          "--random_steps 30 --start_steps 100 --batch_size 64 --buffer_size 1000 --device cpu"),
    "loop_ppo_pendulum": ("PPO_file", "PPO_with_tricks", "--env_name PendulumShort-v1 --seed 0 --max_episodes 4 --save_freq 2 "
                          "--horizon 64 --minibatch_size 32 --K_epochs 2 --device cpu"),
    "loop_ppo_cartpole": ("PPO_file", "PPO_with_tricks", "--env_name CartPole-v1 --seed 0 --max_episodes 8 --save_freq 4 "
                          "--horizon 64 --minibatch_size 32 --K_epochs 2 --device cpu"),
    "loop_maddpg_spread": ("MADDPG_file", "MADDPG_simple", "--env_name simple_spread_v3 --N 3 --seed 100 --max_episodes 6 "
                           "--save_freq 100 --start_steps 50 --batch_size 32 --buffer_size 500 --device cpu"),
}


class Recorder:
    """Env proxy that logs every env action and reward (single-agent or parallel API)."""

    def __init__(self, env, log):
        self._env, self._log = env, log

    def __getattr__(self, k):
        return getattr(self._env, k)

    def step(self, action):
        out = self._env.step(action)
        if isinstance(action, dict):
            self._log["actions"].append(np.concatenate([np.asarray(action[a], np.float64).reshape(-1) for a in self._env.possible_agents]))
            self._log["rewards"].append(np.array([out[1][a] for a in self._env.possible_agents], np.float64))
        else:
            self._log["actions"].append(np.asarray(action, np.float64).reshape(-1))
            self._log["rewards"].append(np.float64(out[1]))
        return out


def install_stubs(log):
    gym = types.ModuleType("gymnasium")
    gym.make = lambda name, **kw: Recorder(E.make(name, prefer_gymnasium=False), log)
    gym.spec = E.spec
    gym.spaces = E.spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = E.spaces
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    import torch.utils
    torch.utils.tensorboard = tb
    pz = types.ModuleType("pettingzoo")
    mpe = types.ModuleType("pettingzoo.mpe")
    spread = types.ModuleType("pettingzoo.mpe.simple_spread_v3")
    spread.parallel_env = lambda max_cycles=25, continuous_actions=True, N=3: Recorder(E.SpreadEnv(N, max_cycles), log)
    sys.modules.update({"pettingzoo": pz, "pettingzoo.mpe": mpe, "pettingzoo.mpe.simple_spread_v3": spread})


@contextlib.contextmanager
def torch_dtype_tolerant_zeros():
    """PPO_with_tricks.py:302 passes dtype=torch.float32 to np.zeros (TypeError as committed):
    map it to np.float32 for the duration of the run (the evident intent)."""
    orig = np.zeros

    def zeros(shape, dtype=float, *a, **k):
        return orig(shape, np.float32 if dtype is torch.float32 else dtype, *a, **k)
    np.zeros = zeros
    try:
        yield
    finally:
        np.zeros = orig


def run_reference(directory, script, flags):
    log = dict(actions=[], rewards=[])
    install_stubs(log)
    path = os.path.join(REF_ROOT, directory, script + ".py")
    src = open(path, encoding="utf-8").read()
    tmp = tempfile.mkdtemp(prefix="frl_loop_")
    fake = os.path.join(tmp, script + ".py")
    for m in ("Buffer", "normalization", "c_adamw", "Noisy_net"):
        sys.modules.pop(m, None)
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF_ROOT, directory))
    argv, sys.argv = sys.argv, [fake] + flags.split()
    torch.set_num_threads(1)
    try:
        with torch_dtype_tolerant_zeros(), contextlib.redirect_stdout(io.StringIO()):
            exec(compile(src, fake, "exec"), {"__name__": "__main__", "__file__": fake})
    finally:
        sys.argv = argv
        sys.path.remove(os.path.join(REF_ROOT, directory))
        for m in ("Buffer", "normalization", "c_adamw", "Noisy_net"):
            sys.modules.pop(m, None)
    npys = [f for f in glob.glob(os.path.join(tmp, "results", "*", "*", "*.npy")) if "running_mean" not in f]
    ckpt = glob.glob(os.path.join(tmp, "results", "*", "*", "*.pt")) + glob.glob(os.path.join(tmp, "results", "*", "*", "*.pth"))
    assert len(npys) == 1 and len(ckpt) == 1, (npys, ckpt)
    return log, np.load(npys[0]), torch.load(ckpt[0]), os.path.basename(npys[0]), os.path.basename(ckpt[0])


def main():
    only = sys.argv[1:]
    for name, (directory, script, flags) in LOOPS.items():
        if only and name not in only:
            continue
        log, returns, sd, npy_name, ckpt_name = run_reference(directory, script, flags)
        out = {"flags": np.array(flags), "returns": np.asarray(returns, np.float64),
               "actions": np.stack(log["actions"]), "rewards": np.stack(log["rewards"]),
               "npy_name": np.array(npy_name), "ckpt_name": np.array(ckpt_name)}
        if all(isinstance(v, dict) for v in sd.values()):          # MADDPG.pth: {agent: state_dict}
            for a, d in sd.items():
                synth.pack_digest("ckpt/" + a, {k: v.numpy() for k, v in d.items()}, out, full_limit=0)
        else:
            synth.pack_digest("ckpt", {k: v.numpy() for k, v in sd.items()}, out, full_limit=0)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-20s steps %5d episodes %3d returns[:3] %s -> %.1f KB" % (
            name, len(log["actions"]), np.asarray(returns).shape[-1], np.round(np.asarray(returns).reshape(-1)[:3], 3),
            os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
