"""Long-horizon parity cases: the inputs of the reference-generated loss CURVES (`long_*.npz`).

north_star's "TD-loss curve matching reference seed=0 to 1e-4" is a statement about hundreds of consecutive learn() calls;
the injected-draw cases of `cases.py` are 2-8 calls long.  These cases run the imported reference for 500 (DQN,
DDPG, TD3, SAC) calls, 100 SAC calls at config 4's shape, 200 TD3 calls at hidden width 256, 150 MADDPG calls at config 5's shape and one full-size PPO learn() (320 + 320 minibatch steps) on seeded inputs with every legacy-RNG
draw injected, and store the per-call LOSSES only (a few KB each).  `tests/test_gpu_longrun.py` compares the HIP engine
DIRECTLY with these curves on every kernel family; `tests/test_oracle_golden.py` holds the oracle to them on CPU.

Inputs are regenerated from PCG64 seeds (the same draws round 2's HIP-vs-oracle long runs used).
"""
import numpy as np

from . import cases, synth

H = cases.H


def _idx(seed, n_calls, n_table, batch):
    g = np.random.default_rng(seed)
    return [g.choice(n_table, batch, replace=False).astype(np.int64) for _ in range(n_calls)]


LONG = {
    # DQN.learn (DQN_file/DQN.py:104-128) at the SYN-D shape, 500 calls
    "long_dqn": dict(kind="dqn", obs_dim=8, n_actions=4, batch=256, n_table=2048, capacity=4096, n_calls=500,
                     gamma=0.99, tau=0.01, lr=1e-3),
    # DDPG_simple.learn (DDPG_file/DDPG_simple.py:137-156), 500 calls
    "long_ddpg": dict(kind="ddpg", obs_dim=8, act_dim=2, batch=256, n_table=2048, capacity=4096, n_calls=500,
                      gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3, max_action=1.0),
    # TD3.learn (TD3_file/TD3.py:189-233) at BASELINE config 2's dims (Pendulum: obs 3, act 1, max_action 2), batch 256, 500 calls
    "long_td3_c2": dict(kind="td3", obs_dim=3, act_dim=1, batch=256, n_table=2048, capacity=4096, n_calls=500,
                        gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, policy_noise=0.2, noise_clip=0.5,
                        max_action=2.0, policy_freq=2, policy_noise_scale=1.0),
    # SAC.learn + Alpha (SAC_file/SAC.py:222-260,154-169), 500 calls
    "long_sac": dict(kind="sac", obs_dim=8, act_dim=2, batch=256, n_table=2048, capacity=4096, n_calls=500,
                     gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3),
    # PPO_with_tricks.learn (PPO_file/PPO_with_tricks.py:290-354) at BASELINE config 3's full shape: one learn() = 320 + 320 steps
    "long_ppo_c3": dict(cases.CASES["ppo"], obs_dim=17, act_dim=6, horizon=2048, minibatch=64, k_epochs=10, table_seed=531,
                        param_seed=532, perm_seed=533, actor_lr=3e-4, critic_lr=3e-4),
    # MADDPG_simple.learn (MADDPG_file/MADDPG_simple.py:165-195) at BASELINE config 5's full shape (simple_spread_v3: 3 agents,
    # obs 18, act 5, batch 1024), 150 calls = 450 critic + 450 actor updates; one index set per agent and call (:169)
    "long_maddpg_c5": dict(kind="maddpg", dims={"agent_0": [18, 5], "agent_1": [18, 5], "agent_2": [18, 5]}, batch=1024, n_table=2048,
                           capacity=4096, n_calls=150, n_learn=150, gamma=0.95, tau=0.01, actor_lr=1e-3, critic_lr=1e-3,
                           table_seed=541, param_seed=5420, idx_seed=54300),
    # SAC.learn + Alpha (SAC_file/SAC.py:222-260,154-169) at BASELINE config 4's shape (Humanoid-v4: obs 376, act 17, batch 256), 100 calls:
    # the curve the K-sliced chained kernels (kernels_criticw / _actorw, *_wide_h2a2) are held to
    "long_sac_c4": dict(kind="sac", obs_dim=376, act_dim=17, batch=256, n_table=1024, capacity=2048, n_calls=100,
                        gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3),
    # TD3.learn with the reference's Actor / Critic_TD3 constructed at hidden_1 = hidden_2 = 256 (TD3_file/TD3.py:53,67,86 take the
    # widths as constructor arguments): north_star's "dense 256 x 256" shape, 200 calls — the x-stationary kernels' curve
    "long_td3_h256": dict(kind="td3", obs_dim=8, act_dim=2, hidden=256, batch=256, n_table=2048, capacity=4096, n_calls=200,
                          gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, policy_noise=0.2, noise_clip=0.5,
                          max_action=1.0, policy_freq=2, policy_noise_scale=1.0),
}


def dqn_inputs(c):
    tab = synth.transitions(501, c["n_table"], c["obs_dim"], 1, n_discrete=c["n_actions"])
    params = synth.mlp_params(502, [("l1", H, c["obs_dim"]), ("l2", c["n_actions"], H)])
    return dict(table=tab, params=dict(Qnet=params), idx=_idx(503, c["n_calls"], c["n_table"], c["batch"]))


def ac_inputs(c):
    """DDPG / TD3 / SAC: table, parameters, per-call index sets and the two N(0,1) draws of a call (TD3 uses the first for
    its target-policy noise, TD3.py:197; SAC the first for actor_target's rsample and the second for the actor's, SAC.py:227,244)."""
    O, A = c["obs_dim"], c["act_dim"]
    gaussian, twin = c["kind"] == "sac", c["kind"] != "ddpg"
    hidden = c.get("hidden", H)
    tab = synth.transitions(511, c["n_table"], O, A)
    actor = synth.mlp_params(512, cases.actor_layers(O, A, head="mean_layer" if gaussian else "l3", hidden=hidden))
    if gaussian:
        actor = dict([("log_std", np.zeros((1, A), np.float32))] + list(actor.items()))
    critic = synth.mlp_params(513, cases.critic_layers(O + A, twin=twin, hidden=hidden))
    idx = _idx(514, c["n_calls"], c["n_table"], c["batch"])
    g = np.random.default_rng(515)
    noise = [(g.standard_normal((c["batch"], A)).astype(np.float32), g.standard_normal((c["batch"], A)).astype(np.float32))
             for _ in range(c["n_calls"])]
    return dict(table=tab, params=dict(actor=actor, critic=critic), idx=idx, noise=noise)


def ppo_inputs(c):
    return cases.ppo_inputs(c)


def maddpg_inputs(c):
    """tables / params per agent as cases.maddpg_inputs; idx[k][j] = agent j's index set in call k (drawn without replacement)."""
    inp = cases.maddpg_inputs(dict(c, n_learn=0))
    g = np.random.default_rng(c["idx_seed"])
    inp["idx"] = [[g.choice(c["n_table"], c["batch"], replace=False).astype(np.int64) for _ in inp["ids"]] for _ in range(c["n_calls"])]
    return inp
