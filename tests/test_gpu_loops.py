"""The caller side: freerl_amd.train's loops against golden runs of the
reference's OWN `__main__` loops (tests/golden/make_loop_golden.py executed the reference scripts
with their argparse flags on the in-repo envs).  Same flags, same envs, same seeds -> the env
actions and rewards must agree step by step, the per-episode returns and the final checkpoint
within fp32 tolerance.  This pins the action-selection rule, the RNG draw order per step, add
before learn, the learn trigger, the noise schedules and the results layout.

Tolerance: the closed loop amplifies fp32 differences (Adam turns 1-ulp gradient differences into
lr-sized parameter differences, the env feeds action differences back): measured against the
reference on Pendulum the per-step actions agree to 1e-5 for the first ~100 learner updates and
drift apart afterwards, as any two fp32 implementations do.  The golden runs are therefore short
(<= 100 updates, several episode boundaries, 40-step Pendulum); env actions are compared step by
step at 2e-3 absolute, returns at 1e-3 relative; discrete actions (DQN on CartPole) must be identical."""
import os

import numpy as np
import pytest
import torch

from tests.golden import synth
from tests.golden.make_loop_golden import LOOPS, Recorder

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALGO = {"DQN": "dqn", "DDPG_simple": "ddpg", "TD3": "td3", "SAC": "sac", "PPO_with_tricks": "ppo", "MADDPG_simple": "maddpg"}


@pytest.mark.parametrize("name", sorted(LOOPS))
def test_training_loop_follows_the_reference(name, tmp_path):
    from freerl_amd import envs as E
    from freerl_amd import train
    directory, script, flags = LOOPS[name]
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    algo = ALGO[script]
    argv = flags.replace("--device cpu", "--device cuda").split() + ["--results_root", str(tmp_path / "results")]
    log = dict(actions=[], rewards=[])
    env_name = argv[argv.index("--env_name") + 1]
    if algo == "maddpg":
        env = Recorder(E.SpreadEnv(int(argv[argv.index("--N") + 1]), 25), log)
    else:
        env = Recorder(E.make(env_name, prefer_gymnasium=False), log)
    assert env_name in ("CartPole-v1", "PendulumShort-v1", "simple_spread_v3")
    out = train.run(algo, argv, env=env, log=lambda *a: None)
    acts, rews = np.stack(log["actions"]), np.stack(log["rewards"])
    assert acts.shape == fx["actions"].shape, (acts.shape, fx["actions"].shape)     # same number of env steps
    if "CartPole" in env_name:
        np.testing.assert_array_equal(acts, fx["actions"])                         # discrete actions: identical
    else:
        worst = float(np.max(np.abs(acts - fx["actions"])))
        assert worst < 2e-3, "env actions diverge from the reference loop: max abs diff %g" % worst
    np.testing.assert_allclose(rews, fx["rewards"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(out["returns"], fx["returns"], rtol=1e-3, atol=1e-3)
    # results layout: same file names as the reference script wrote
    files = sorted(os.listdir(out["model_dir"]))
    assert str(fx["npy_name"]) in files and str(fx["ckpt_name"]) in files, files
    assert os.path.basename(out["model_dir"]).startswith(out["args"].policy_name + "_")
    saved = np.load(os.path.join(out["model_dir"], str(fx["npy_name"])))
    np.testing.assert_allclose(saved, fx["returns"], rtol=1e-3, atol=1e-3)
    sd = torch.load(os.path.join(out["model_dir"], str(fx["ckpt_name"])))
    if algo == "maddpg":
        for a, d in sd.items():
            synth.check_digest("ckpt/" + a, {k: v.numpy() for k, v in d.items()}, fx, 5e-3, 5e-4, name)
    else:
        synth.check_digest("ckpt", {k: v.numpy() for k, v in sd.items()}, fx, 5e-3, 5e-4, name)


def test_make_dir_numbering_and_trick_prefix(tmp_path):
    from freerl_amd.train import make_dir
    root = str(tmp_path)
    assert os.path.basename(make_dir(root, "Pendulum-v1", "TD3", None)) == "TD3_1"
    assert os.path.basename(make_dir(root, "Pendulum-v1", "TD3", None)) == "TD3_2"
    d = make_dir(root, "Pendulum-v1", "SAC", {"ObsNorm": False, "OUNoise": True, "GaussNoise": False})
    assert os.path.basename(d) == "SAC_OUNoise_1"
