"""Soak of the pre-armed rollout launches (frl_api_rollout.inc: two streams, doorbell + device word): the same long run with
FRL_ROLLOUT_PREARM=1 and =0 in two processes must end in bit-identical rings, nets, targets and Adam moments — a lost update, a launch
that read a block or a net too early, or a doorbell that was missed shows up as a difference (or as one of the 2 s time-outs).
    python tools/prearm_soak.py [steps] [kind ...]          kinds: dqn1 dqn4 td3 sac ddpg2 (default: all)"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = {"dqn1": ("dqn", 1, 1), "dqn4": ("dqn", 4, 3), "td3": ("td3", 1, 1), "sac": ("sac", 1, 2), "ddpg2": ("ddpg", 2, 1),
         # full-chip populations of the sixteen-workgroup kernels (every CU busy: the flag hand-overs under load); no launch is armed at
         # these sizes, so the two runs are the same program twice — a stale read or a race shows as a run-to-run difference
         "td3x16": ("td3", 16, 1), "sacx12": ("sac", 12, 2),
         # where the default population bounds of the arming sit (armed vs plain, us per step in the last column of each run)
         "dqn8": ("dqn", 8, 1), "dqn16": ("dqn", 16, 1), "dqn32": ("dqn", 32, 1), "td3p4": ("td3", 4, 1), "td3p8": ("td3", 8, 1)}


def worker(kind, steps):
    sys.path.insert(0, ROOT)
    import numpy as np
    from freerl_amd import _native as N
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout
    algo, P, Ev = KINDS[kind]
    if algo == "dqn":
        e = Engine(N.ALGO_DQN, 8, 4, 5000, discrete=True, batch_max=64, n_learners=P, seed=11)
        pool = EnvPool("SynLinearDiscrete-v0", P * Ev, n_threads=1, seed=5)
        kw = dict(envs_per_learner=Ev, start_steps=200, learn_every=1, epsilon=0.2, batch=64, critic_lr=1e-3, tau=0.05)
        nets = (0,)
    else:
        aid = dict(td3=N.ALGO_TD3, sac=N.ALGO_SAC, ddpg=N.ALGO_DDPG)[algo]
        e = Engine(aid, 8, 2, 5000, twin_critic=algo != "ddpg", batch_max=64, n_learners=P, seed=11)
        if algo == "sac":
            for p in range(P):
                e.set_alpha_state([np.log(0.05), 0, 0, 0.05], learner=p)
        pool = EnvPool("SynLinear-v0", P * Ev, n_threads=1, seed=5)
        kw = dict(envs_per_learner=Ev, start_steps=200, learn_every=1, batch=64, actor_lr=1e-4, critic_lr=1e-4, tau=0.01, policy_freq=2)
        nets = (0, 1)
    rng = np.random.default_rng(12)
    for net in nets:
        for p in range(P):
            flat = (rng.standard_normal(e.num_params(net)) * 0.1).astype(np.float32)
            e.set_params(net, flat, N.PARAM_ONLINE, learner=p); e.set_params(net, flat, N.PARAM_TARGET, learner=p)
    t0 = time.perf_counter()
    out = rollout(e, pool, steps, **kw)
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    for p in range(P):
        h.update(e.read_rows(p, 0, 5000).tobytes())
        for net in nets:
            for k in (N.PARAM_ONLINE, N.PARAM_TARGET, N.PARAM_ADAM_M, N.PARAM_ADAM_V):
                h.update(e.get_params(net, k, learner=p).tobytes())
    print("%s %d %.6f %s %.1f" % (h.hexdigest()[:16], out["updates"], out["return_sum"], "finite" if np.all(np.isfinite(e.stats())) else "NONFINITE", dt / steps * 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    bad = 0
    for kind in (sys.argv[2:] or ["dqn1", "dqn4", "td3", "sac", "ddpg2"]):
        res = []
        for arm in ("1", "0"):
            env = dict(os.environ, FRL_ROLLOUT_PREARM=arm)
            if "x" in kind:                    # (the default: nothing armed at these sizes)
                env.pop("FRL_ROLLOUT_PREARM")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", kind, str(steps)], env=env, capture_output=True, text=True)
            res.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED: " + r.stderr[-300:])
        same = res[0].split()[:3] == res[1].split()[:3] and not res[0].startswith("FAILED")
        bad += not same
        print("%-6s %d steps  pre-armed: %s | plain: %s  -> %s" % (kind, steps, res[0], res[1], "identical" if same else "DIFFERENT"), flush=True)
    sys.exit(1 if bad else 0)
