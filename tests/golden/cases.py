"""Parity cases: dimensions, hyper-parameters and seeds.

One definition, three consumers: `make_golden.py` drives the imported REFERENCE through a
case and stores its outputs; `tests/test_oracle_golden.py` drives `oracle/` through the same
case and compares with the stored outputs; `tests/test_gpu_parity.py` drives the HIP engine
and compares with the oracle (live) and with the stored outputs.

Inputs are regenerated from seeds by `synth.py` — fixtures hold outputs only.
"""
import numpy as np

from . import synth

H = 128  # the reference's hidden width (DQN.py:38, TD3.py:53, SAC.py:61 ...)


def actor_layers(obs_dim, act_dim, head="l3", hidden=H):
    return [("l1", hidden, obs_dim), ("l2", hidden, hidden), (head, act_dim, hidden)]


def critic_layers(in_dim, twin=False, hidden=H):
    ls = [("l1", hidden, in_dim), ("l2", hidden, hidden), ("l3", 1, hidden)]
    if twin:
        ls += [("l4", hidden, in_dim), ("l5", hidden, hidden), ("l6", 1, hidden)]
    return ls


CASES = {
    # Buffer ring + sample (TD3_file/Buffer.py:11-61)
    "buffer": dict(kind="buffer", obs_dim=5, act_dim=2, capacity=300, n_add=450, batch=64,
                   table_seed=11, idx_seed=12),
    # PER_Buffer + SumTree (DQN_file/Buffer.py:66-194): add past the capacity, stratified samples, priority updates
    "per_buffer": dict(kind="per_buffer", obs_dim=5, act_dim=1, capacity=300, n_add=450, batch=64, n_rounds=4,
                       table_seed=15, u_seed=16, td_seed=17),
    # DQN_with_tricks.learn with trick Double + PER + N_Step (DQN_with_tricks.py:242-284, N_Step_PER_Buffer Buffer.py:333-399)
    "dqn_tricks": dict(kind="dqn_tricks", obs_dim=8, n_actions=4, capacity=2048, n_table=700, batch=128, n_learn=5,
                       gamma=0.99, n_step=3, tau=0.01, lr=1e-3, table_seed=133, param_seed=1010, u_seed=2010),
    # DQN_with_tricks.learn with trick Dueling + Double (DQN_with_tricks.py:60-79,263-265), uniform replay
    "dqn_dueling": dict(kind="dqn_dueling", obs_dim=8, n_actions=4, capacity=2048, n_table=600, batch=128, n_learn=4,
                        gamma=0.99, tau=0.01, lr=1e-3, table_seed=134, param_seed=1020, idx_seed=2020),
    # DQN_with_tricks.learn with Noisy + Dueling + Double (Noisy_net.py:17-76; DQN_with_tricks.py:60-79,263-265)
    "dqn_noisy": dict(kind="dqn_noisy", obs_dim=8, n_actions=4, capacity=2048, n_table=600, batch=128, n_learn=4,
                      gamma=0.99, tau=0.01, lr=1e-3, table_seed=135, param_seed=1030, idx_seed=2030, noise_seed=3030),
    # full Rainbow = the reference's default trick set: Double + Dueling + PER + Noisy + N_Step + Categorical
    "dqn_rainbow": dict(kind="dqn_rainbow", obs_dim=8, n_actions=4, capacity=2048, n_table=500, batch=64, n_learn=4,
                        gamma=0.99, n_step=3, tau=0.01, lr=1e-3, atoms=51, vmin=-100.0, vmax=100.0,
                        table_seed=136, param_seed=1040, u_seed=2040, noise_seed=3040),
    # Categorical alone (plain l2 head, uniform replay, no Double): DQN_with_tricks.py:82-158,248-260
    "dqn_c51": dict(kind="dqn_c51", obs_dim=8, n_actions=3, capacity=2048, n_table=500, batch=64, n_learn=4,
                    gamma=0.99, tau=0.01, lr=1e-3, atoms=51, vmin=-100.0, vmax=100.0,
                    table_seed=137, param_seed=1050, idx_seed=2050),
    # DQN.learn (DQN_file/DQN.py:104-128); SYN-D shape of SURVEY §8(d)
    "dqn": dict(kind="dqn", obs_dim=8, n_actions=4, capacity=4096, n_table=1024, batch=256,
                n_learn=5, gamma=0.99, tau=0.01, lr=1e-3, table_seed=123, param_seed=1000,
                idx_seed=2000),
    # DDPG_simple.learn (DDPG_file/DDPG_simple.py:137-156)
    "ddpg": dict(kind="ddpg", obs_dim=8, act_dim=2, capacity=4096, n_table=1024, batch=256,
                 n_learn=4, gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3,
                 table_seed=123, param_seed=1100, idx_seed=2100),
    # TD3.learn (TD3_file/TD3.py:189-233), all `realize` flags on, max_action 2 (Pendulum)
    "td3": dict(kind="td3", obs_dim=8, act_dim=2, capacity=4096, n_table=1024, batch=256,
                n_learn=4, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3,
                policy_noise=0.2, noise_clip=0.5, max_action=2.0, policy_freq=2,
                policy_noise_scale=1.0, table_seed=123, param_seed=1200, idx_seed=2200,
                noise_seed=3200),
    # TD3 with batch not a multiple of the 64-row tile and Pendulum's dims (config 2 shape)
    "td3_pendulum": dict(kind="td3", obs_dim=3, act_dim=1, capacity=1000, n_table=700, batch=100,
                         n_learn=3, gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3,
                         policy_noise=0.1, noise_clip=0.5, max_action=2.0, policy_freq=2,
                         policy_noise_scale=1.0, table_seed=124, param_seed=1210, idx_seed=2210,
                         noise_seed=3210),
    # SAC.learn + Alpha (SAC_file/SAC.py:222-260,154-169)
    "sac": dict(kind="sac", obs_dim=8, act_dim=2, capacity=4096, n_table=1024, batch=256,
                n_learn=3, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3,
                table_seed=123, param_seed=1300, idx_seed=2300, noise_seed=3300),
    # MADDPG_simple.learn (MADDPG_file/MADDPG_simple.py:165-186), heterogeneous agents
    "maddpg": dict(kind="maddpg", dims={"agent_0": [6, 2], "agent_1": [5, 3], "agent_2": [7, 2]},
                   capacity=512, n_table=256, batch=64, n_learn=2, gamma=0.95, tau=0.01,
                   actor_lr=1e-3, critic_lr=1e-3, table_seed=125, param_seed=1400, idx_seed=2400),
    # MADDPG.py with its default supplements (weight_decay, net_init, per-agent Batch_ObsNorm; MADDPG.py:60-228)
    "maddpg_full": dict(kind="maddpg", dims={"agent_0": [6, 2], "agent_1": [5, 3], "agent_2": [7, 2]},
                        capacity=512, n_table=256, batch=64, n_learn=3, gamma=0.95, tau=0.01,
                        actor_lr=1e-3, critic_lr=1e-3, table_seed=125, param_seed=1470, idx_seed=2470),
    # MATD3_simple.learn (MADDPG_file/MATD3_simple.py:217-262): twin critics, per-agent target smoothing, delayed policy
    "matd3": dict(kind="matd3", dims={"agent_0": [6, 2], "agent_1": [5, 3], "agent_2": [7, 2]},
                  capacity=512, n_table=256, batch=64, n_learn=4, gamma=0.95, tau=0.01,
                  actor_lr=1e-3, critic_lr=1e-3, policy_noise=0.2, noise_clip=0.5, max_action=1.0, policy_freq=2,
                  policy_noise_scale=1.0, table_seed=125, param_seed=1450, idx_seed=2450, noise_seed=3450),
    # PPO_with_tricks.learn (PPO_file/PPO_with_tricks.py:290-354), Gaussian actor, no tricks
    "ppo": dict(kind="ppo", obs_dim=8, act_dim=2, horizon=256, minibatch=64, k_epochs=2,
                gamma=0.99, lmbda=0.95, clip=0.2, ent=0.01, actor_lr=1e-3, critic_lr=1e-3,
                trick=dict(adv_norm=False, ObsNorm=False, reward_norm=False, reward_scaling=False,
                           orthogonal_init=False, adam_eps=False, lr_decay=False, tanh=False,
                           Batch_ObsNorm=False),
                table_seed=126, param_seed=1500, perm_seed=2500),
    # PPO.py (PPO_file/PPO.py:213-286): no tricks, ONE cautious AdamW (c_adamw.py) over actor + critic, lr = actor_lr
    "ppo_py": dict(kind="ppo", obs_dim=8, act_dim=2, horizon=256, minibatch=64, k_epochs=2,
                   gamma=0.99, lmbda=0.95, clip=0.2, ent=0.01, actor_lr=1e-3, critic_lr=3e-3,
                   trick=dict(adv_norm=False, ObsNorm=False, reward_norm=False, reward_scaling=False,
                              orthogonal_init=False, adam_eps=False, lr_decay=False, tanh=False,
                              Batch_ObsNorm=False),
                   table_seed=129, param_seed=1530, perm_seed=2530),
    # Buffer_for_PPO with and without trick['decaystd'] (PPO_file/Buffer.py:266-323); 44 adds into 32 rows: wraps
    "ppo_buffer": dict(kind="buffer", obs_dim=5, act_dim=3, capacity=32, n_add=44, table_seed=141),
    # PPO_advance/PPO_2.py:152-292: values stored at rollout time, stable-baselines3-style float64 GAE, two torch Adams
    "ppo_2": dict(kind="ppo", obs_dim=8, act_dim=2, horizon=256, minibatch=64, k_epochs=2,
                  gamma=0.99, lmbda=0.95, clip=0.2, ent=0.005, actor_lr=1e-3, critic_lr=2e-3, last_value=0.37,
                  trick=dict(adv_norm=False, ObsNorm=False, reward_norm=False, reward_scaling=False,
                             orthogonal_init=False, adam_eps=False, lr_decay=False, tanh=False,
                             Batch_ObsNorm=False),
                  table_seed=133, param_seed=1560, perm_seed=2560),
    # PPO with Actor_Beta (PPO_with_tricks.py:120-151,240-251,325-332): actions in (0,1), alpha/beta heads, adv_norm
    "ppo_beta": dict(kind="ppo_beta", obs_dim=8, act_dim=2, horizon=256, minibatch=64, k_epochs=2,
                     gamma=0.99, lmbda=0.95, clip=0.2, ent=0.01, actor_lr=1e-3, critic_lr=1e-3,
                     trick=dict(adv_norm=True, ObsNorm=False, reward_norm=False, reward_scaling=False,
                                orthogonal_init=False, adam_eps=False, lr_decay=False, tanh=False,
                                Batch_ObsNorm=False),
                     table_seed=130, param_seed=1540, perm_seed=2540),
    # PPO with adv_norm + tanh hidden activations + Adam eps 1e-5
    "ppo_tricks": dict(kind="ppo", obs_dim=8, act_dim=2, horizon=256, minibatch=64, k_epochs=2,
                       gamma=0.99, lmbda=0.95, clip=0.2, ent=0.01, actor_lr=1e-3, critic_lr=1e-3,
                       trick=dict(adv_norm=True, ObsNorm=False, reward_norm=False,
                                  reward_scaling=False, orthogonal_init=False, adam_eps=True,
                                  lr_decay=False, tanh=True, Batch_ObsNorm=False),
                       table_seed=127, param_seed=1510, perm_seed=2510),
    # DDPG.py (DDPG_file/DDPG.py:150-222) with supplements weight_decay + Batch_ObsNorm
    "ddpg_full": dict(kind="ddpg", obs_dim=8, act_dim=2, capacity=4096, n_table=1024, batch=256,
                      n_learn=4, gamma=0.99, tau=0.01, actor_lr=1e-3, critic_lr=1e-3,
                      table_seed=123, param_seed=1130, idx_seed=2130),
    # SAC with trick['Batch_ObsNorm'] (SAC.py:181-182,194-195,215-217)
    "sac_bn": dict(kind="sac", obs_dim=8, act_dim=2, capacity=4096, n_table=1024, batch=256,
                   n_learn=3, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3,
                   table_seed=123, param_seed=1330, idx_seed=2330, noise_seed=3330),
    # PPO discrete: Actor_discrete + Categorical (PPO_with_tricks.py:110-121,333-336), CartPole-like dims
    "ppo_discrete": dict(kind="ppo_discrete", obs_dim=4, n_actions=3, horizon=192, minibatch=64, k_epochs=2,
                         gamma=0.99, lmbda=0.95, clip=0.2, ent=0.01, actor_lr=1e-3, critic_lr=1e-3,
                         trick=dict(adv_norm=True, ObsNorm=False, reward_norm=False, reward_scaling=False,
                                    orthogonal_init=False, adam_eps=False, lr_decay=False, tanh=False,
                                    Batch_ObsNorm=False),
                         table_seed=128, param_seed=1520, perm_seed=2520),
    # PPO_file/PPO.py discrete: raw logits into Categorical(logits=...) (PPO.py:78-90,176,257) + the one cautious AdamW.  The head is
    # scaled (logit_gain) so that some class probabilities fall below float eps: there the clamp of PPO_with_tricks.py's
    # Categorical(probs=softmax) and the unclamped log-softmax of this variant give different log-probs / entropies
    "ppo_py_discrete": dict(kind="ppo_discrete", obs_dim=4, n_actions=3, horizon=192, minibatch=64, k_epochs=2,
                            gamma=0.99, lmbda=0.95, clip=0.2, ent=0.01, actor_lr=1e-3, critic_lr=1e-3, logit_gain=60.0,
                            trick={}, table_seed=129, param_seed=1530, perm_seed=2530),
}


def dqn_inputs(c):
    tab = synth.transitions(c["table_seed"], c["n_table"], c["obs_dim"], 1, n_discrete=c["n_actions"])
    params = synth.mlp_params(c["param_seed"], [("l1", H, c["obs_dim"]), ("l2", c["n_actions"], H)])
    idx = [synth.indices(c["idx_seed"] + k, c["n_table"], c["batch"]) for k in range(c["n_learn"])]
    return dict(table=tab, params=dict(Qnet=params), idx=idx)


def per_buffer_inputs(c):
    tab = synth.transitions(c["table_seed"], c["n_add"], c["obs_dim"], 1, n_discrete=3)
    g = np.random.default_rng(c["u_seed"])
    us = [g.random(c["batch"]) for _ in range(c["n_rounds"])]                    # float64 in [0,1)
    g2 = np.random.default_rng(c["td_seed"])
    tds = [(g2.standard_normal((c["batch"], 1)) * 2).astype(np.float32) for _ in range(c["n_rounds"])]
    return dict(table=tab, uniforms=us, td=tds)


def dqn_dueling_inputs(c):
    tab = synth.transitions(c["table_seed"], c["n_table"], c["obs_dim"], 1, n_discrete=c["n_actions"])
    params = synth.mlp_params(c["param_seed"], [("l1", H, c["obs_dim"]), ("V", 1, H), ("A", c["n_actions"], H)])
    idx = [synth.indices(c["idx_seed"] + k, c["n_table"], c["batch"]) for k in range(c["n_learn"])]
    return dict(table=tab, params=dict(Qnet=params), idx=idx)


def noisy_f(x):
    """scale_noise (Noisy_net.py:72-76): sign(x) * sqrt(|x|) in float32."""
    x = np.asarray(x, dtype=np.float32)
    return (np.sign(x) * np.sqrt(np.abs(x))).astype(np.float32)


def dqn_noisy_inputs(c):
    """Noisy + Dueling: params l1, V (NoisyLinear hidden->1), A (NoisyLinear hidden->n_actions).  raw[k][j][head] = the
    (randn(in), randn(out)) pair of forward j of learn call k (j: Qnet(next_obs), Qnet_target(next_obs), Qnet(obs)), plus one
    pair set for a select_action probe."""
    nA, O = c["n_actions"], c["obs_dim"]
    tab = synth.transitions(c["table_seed"], c["n_table"], O, 1, n_discrete=nA)
    g = np.random.default_rng(c["param_seed"])
    p = dict(synth.mlp_params(c["param_seed"], [("l1", H, O)]))
    for name, rows in (("V", 1), ("A", nA)):
        r = 1 / np.sqrt(H)
        p[name + ".weight_mu"] = g.uniform(-r, r, (rows, H)).astype(np.float32)
        p[name + ".weight_sigma"] = g.uniform(0.02, 0.08, (rows, H)).astype(np.float32)      # distinct values: exercises d/d sigma
        p[name + ".bias_mu"] = g.uniform(-r, r, rows).astype(np.float32)
        p[name + ".bias_sigma"] = g.uniform(0.02, 0.08, rows).astype(np.float32)
    idx = [synth.indices(c["idx_seed"] + k, c["n_table"], c["batch"]) for k in range(c["n_learn"])]
    gn = np.random.default_rng(c["noise_seed"])
    draw = lambda: {"V": (gn.standard_normal(H).astype(np.float32), gn.standard_normal(1).astype(np.float32)),
                    "A": (gn.standard_normal(H).astype(np.float32), gn.standard_normal(nA).astype(np.float32))}
    raw = [[draw() for _ in range(3)] for _ in range(c["n_learn"])]
    probe = draw()
    return dict(table=tab, params=dict(Qnet=p), idx=idx, raw=raw, probe=probe)


def noisy_eps(raw_one):
    """{head: (randn_in, randn_out)} -> {head: (eps_in, eps_out)}"""
    return {h: (noisy_f(a), noisy_f(b)) for h, (a, b) in raw_one.items()}


def dqn_c51_inputs(c):
    tab = synth.transitions(c["table_seed"], c["n_table"], c["obs_dim"], 1, n_discrete=c["n_actions"])
    tab["rew"] = (tab["rew"] * 30).astype(np.float32)          # spread the projected targets over the [-100, 100] support
    params = synth.mlp_params(c["param_seed"], [("l1", H, c["obs_dim"]), ("l2", c["n_actions"] * c["atoms"], H)])
    idx = [synth.indices(c["idx_seed"] + k, c["n_table"], c["batch"]) for k in range(c["n_learn"])]
    return dict(table=tab, params=dict(Qnet=params), idx=idx)


def dqn_rainbow_inputs(c):
    nA, O, atoms = c["n_actions"], c["obs_dim"], c["atoms"]
    tab = synth.transitions(c["table_seed"], c["n_table"], O, 1, n_discrete=nA)
    tab["rew"] = (tab["rew"] * 30).astype(np.float32)
    g = np.random.default_rng(c["param_seed"])
    p = dict(synth.mlp_params(c["param_seed"], [("l1", H, O)]))
    rows = dict(V=atoms, A=nA * atoms)
    for name in ("V", "A"):
        r = 1 / np.sqrt(H)
        p[name + ".weight_mu"] = g.uniform(-r, r, (rows[name], H)).astype(np.float32)
        p[name + ".weight_sigma"] = g.uniform(0.02, 0.08, (rows[name], H)).astype(np.float32)
        p[name + ".bias_mu"] = g.uniform(-r, r, rows[name]).astype(np.float32)
        p[name + ".bias_sigma"] = g.uniform(0.02, 0.08, rows[name]).astype(np.float32)
    us = [np.random.default_rng(c["u_seed"] + k).random(c["batch"]) for k in range(c["n_learn"])]
    gn = np.random.default_rng(c["noise_seed"])
    draw = lambda: {h: (gn.standard_normal(H).astype(np.float32), gn.standard_normal(rows[h]).astype(np.float32)) for h in ("V", "A")}
    raw = [[draw() for _ in range(3)] for _ in range(c["n_learn"])]
    return dict(table=tab, params=dict(Qnet=p), uniforms=us, raw=raw, probe=draw())


def dqn_tricks_inputs(c):
    tab = synth.transitions(c["table_seed"], c["n_table"], c["obs_dim"], 1, n_discrete=c["n_actions"])
    params = synth.mlp_params(c["param_seed"], [("l1", H, c["obs_dim"]), ("l2", c["n_actions"], H)])
    g = np.random.default_rng(c["u_seed"])
    us = [g.random(c["batch"]) for _ in range(c["n_learn"])]
    return dict(table=tab, params=dict(Qnet=params), uniforms=us)


def ac_inputs(c, twin, gaussian=False):
    """DDPG / TD3 / SAC inputs."""
    O, A = c["obs_dim"], c["act_dim"]
    tab = synth.transitions(c["table_seed"], c["n_table"], O, A)
    actor = synth.mlp_params(c["param_seed"], actor_layers(O, A, head="mean_layer" if gaussian else "l3"))
    if gaussian:
        g = np.random.default_rng(c["param_seed"] + 7)
        # reference initialises log_std to zeros (SAC.py:66); a non-zero value makes the
        # fixture exercise d(loss)/d(log_std) with distinct per-dimension std
        actor = dict([("log_std", g.uniform(-0.5, 0.3, (1, A)).astype(np.float32))] + list(actor.items()))
    critic = synth.mlp_params(c["param_seed"] + 1, critic_layers(O + A, twin=twin))
    idx = [synth.indices(c["idx_seed"] + k, c["n_table"], c["batch"]) for k in range(c["n_learn"])]
    out = dict(table=tab, params=dict(actor=actor, critic=critic), idx=idx)
    if "noise_seed" in c:
        n_draw = 2 if gaussian else 1
        out["noise"] = [[synth.normal(c["noise_seed"] + 10 * k + j, (c["batch"], A)) for j in range(n_draw)]
                        for k in range(c["n_learn"])]
    return out


def maddpg_inputs(c, twin=False):
    dims = c["dims"]
    ids = list(dims.keys())
    total = sum(o + a for o, a in dims.values())
    tabs = {}
    for j, aid in enumerate(ids):
        o, a = dims[aid]
        tabs[aid] = synth.transitions(c["table_seed"] + 100 * j, c["n_table"], o, a)
    params = {}
    for j, aid in enumerate(ids):
        o, a = dims[aid]
        params[aid] = dict(actor=synth.mlp_params(c["param_seed"] + 10 * j, actor_layers(o, a)),
                           critic=synth.mlp_params(c["param_seed"] + 10 * j + 1, critic_layers(total, twin=twin)))
    # MADDPG_simple.learn re-samples once PER AGENT per learn call (MADDPG_simple.py:169)
    idx = [[synth.indices(c["idx_seed"] + 10 * k + j, c["n_table"], c["batch"]) for j in range(len(ids))]
           for k in range(c["n_learn"])]
    out = dict(tables=tabs, params=params, idx=idx, ids=ids)
    if "noise_seed" in c:      # MATD3: noise[k][i][j] = the draw for agent j's target action inside agent i's sample()
        out["noise"] = [[[synth.normal(c["noise_seed"] + 100 * k + 10 * i + j, (c["batch"], dims[aj][1]))
                          for j, aj in enumerate(ids)] for i in range(len(ids))] for k in range(c["n_learn"])]
    return out


def ppo_inputs(c):
    O, A, T = c["obs_dim"], c["act_dim"], c["horizon"]
    tab = synth.transitions(c["table_seed"], T, O, A)
    g = np.random.default_rng(c["table_seed"] + 1)
    tab["act"] = g.standard_normal((T, A)).astype(np.float32) * 0.7   # un-squashed Gaussian actions
    tab["logp"] = (-0.5 * g.standard_normal((T, A)) ** 2 - 0.9).astype(np.float32)  # stored per dim
    # adv_done = terminated or truncated (PPO_with_tricks.py:291-295)
    tab["adv_done"] = np.logical_or(tab["done"], g.random(T) < 0.03)
    actor = synth.mlp_params(c["param_seed"], actor_layers(O, A, head="mean_layer"))
    actor = dict([("log_std", g.uniform(-0.5, 0.3, (1, A)).astype(np.float32))] + list(actor.items()))
    critic = synth.mlp_params(c["param_seed"] + 1, critic_layers(O))
    perms = [synth.permutation(c["perm_seed"] + k, T) for k in range(c["k_epochs"])]
    if "last_value" in c:                      # PPO_2: the critic's value of each step as select_action returned it
        tab["value"] = (0.8 * g.standard_normal(T)).astype(np.float32)
    return dict(table=tab, params=dict(actor=actor, critic=critic), perms=perms)


def ppo_buffer_inputs(c):
    O, A, n = c["obs_dim"], c["act_dim"], c["n_add"]
    tab = synth.transitions(c["table_seed"], n, O, A)
    g = np.random.default_rng(c["table_seed"] + 1)
    tab["logp"] = (-np.abs(g.standard_normal((n, A))) - 0.5).astype(np.float32)
    tab["adv_done"] = np.logical_or(tab["done"], g.random(n) < 0.1)
    return dict(table=tab)


def ppo_beta_inputs(c):
    O, A, T = c["obs_dim"], c["act_dim"], c["horizon"]
    tab = synth.transitions(c["table_seed"], T, O, A)
    g = np.random.default_rng(c["table_seed"] + 1)
    tab["act"] = g.uniform(0.03, 0.97, (T, A)).astype(np.float32)        # Beta samples live in (0, 1)
    tab["logp"] = (0.4 * g.standard_normal((T, A)) + 0.1).astype(np.float32)
    tab["adv_done"] = np.logical_or(tab["done"], g.random(T) < 0.03)
    H_ = H
    actor = synth.mlp_params(c["param_seed"], [("l1", H_, O), ("l2", H_, H_), ("alpha_layer", A, H_), ("beta_layer", A, H_)])
    critic = synth.mlp_params(c["param_seed"] + 1, critic_layers(O))
    perms = [synth.permutation(c["perm_seed"] + k, T) for k in range(c["k_epochs"])]
    return dict(table=tab, params=dict(actor=actor, critic=critic), perms=perms)


def ppo_discrete_inputs(c):
    O, nA, T = c["obs_dim"], c["n_actions"], c["horizon"]
    tab = synth.transitions(c["table_seed"], T, O, 1, n_discrete=nA)
    g = np.random.default_rng(c["table_seed"] + 1)
    tab["logp"] = (-np.abs(g.standard_normal((T, 1))) - 0.7).astype(np.float32)    # one stored log-prob per step
    tab["adv_done"] = np.logical_or(tab["done"], g.random(T) < 0.03)
    actor = synth.mlp_params(c["param_seed"], actor_layers(O, nA, head="l3"))
    if c.get("logit_gain"):
        actor["l3.weight"] = (actor["l3.weight"] * np.float32(c["logit_gain"])).astype(np.float32)
        actor["l3.bias"] = (actor["l3.bias"] * np.float32(c["logit_gain"])).astype(np.float32)
    critic = synth.mlp_params(c["param_seed"] + 1, critic_layers(O))
    perms = [synth.permutation(c["perm_seed"] + k, T) for k in range(c["k_epochs"])]
    return dict(table=tab, params=dict(actor=actor, critic=critic), perms=perms)


def buffer_inputs(c):
    tab = synth.transitions(c["table_seed"], c["n_add"], c["obs_dim"], c["act_dim"])
    size = min(c["n_add"], c["capacity"])
    idx = synth.indices(c["idx_seed"], size, c["batch"])
    return dict(table=tab, idx=idx)
