#!/usr/bin/env python3
"""bench.py — learner updates/s of the fused replay-sample + batched-update hot path.

Workload (BASELINE.json configs[1], the config the metric is quoted on): TD3.learn()
(TD3_file/TD3.py:189-233, all `realize` flags on, policy_freq 2) with replay capacity 1e6
(filled) and batch 256, at the north_star's synthetic shape obs_dim 8 / act_dim 2, hidden 128.
One "step" = one pass of the hot path over one batch for EVERY learner resident on the GPU:
index draw (device Philox, without replacement) -> record gather from the HBM ring -> target
forward -> twin-critic forward/backward -> clip -> Adam [-> actor forward/backward -> clip ->
Adam -> soft updates on every 2nd step].  Learners are independent seeds: P per
GPU in one launch, sharded over ranks with no data-path collective (scaling "weak"); the only
collective is the RCCL all-reduce of the metric vector.

Contract: W untimed warmup steps, then exactly K timed steps between barrier +
torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OBS, ACT, BATCH, CAP, HIDDEN = 8, 2, 256, 1_000_000, 128
DQN_LOOP_ROWS = (10_000, 100_000, 300_000, 1_000_000)      # ring sizes (full) of the DQN-loop comparison, CPU and GPU legs alike
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak
HBM_PEAK_GBS = 8000.0


def td3_kwargs(k):
    return dict(gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, do_actor=(k % 2 == 1),
                use_policy_noise=True, policy_noise=0.2, noise_clip=0.5, max_action=1.0)


def make_engine(N, Engine, learners, device_id, seed):
    e = Engine(N.ALGO_TD3, OBS, ACT, CAP, n_learners=learners, twin_critic=True, batch_max=BATCH, hidden=HIDDEN,
               device_id=device_id, seed=seed)
    rng = np.random.default_rng(seed)
    # random-init weights of the reference architecture: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    def init(net, dims):
        parts = []
        for out_d, in_d in dims:
            b = 1.0 / np.sqrt(in_d)
            parts += [rng.uniform(-b, b, out_d * in_d), rng.uniform(-b, b, out_d)]
        return np.concatenate(parts).astype(np.float32)
    a_dims = [(HIDDEN, OBS), (HIDDEN, HIDDEN), (ACT, HIDDEN)]
    c_dims = [(HIDDEN, OBS + ACT), (HIDDEN, HIDDEN), (1, HIDDEN)] * 2
    for p in range(learners):
        fa, fc = init(0, a_dims), init(1, c_dims)
        for kind in (N.PARAM_ONLINE, N.PARAM_TARGET):
            e.set_params(0, fa, kind, learner=p)
            e.set_params(1, fc, kind, learner=p)
    e.fill_synthetic(CAP, seed=seed + 17)
    e.sync()
    return e


# ------------------------------------------------------------------------------------------ CPU baselines (same box)
# The oracle (oracle/: the NumPy fp32 restatement of the reference's arithmetic, validated against the reference's own
# outputs in tests/) timed on THIS box's host cores, in worker subprocesses of this script (`--cpu-worker`; a fork of a
# process that holds a HIP context is not safe): one core, and all cores as independent learners — one process per core,
# each its own seed, the same unit of parallelism the GPU engine batches ("P learners").  BLAS threads are pinned to 1 per
# process (the 256 x 128 matrices of this workload do not scale across threads; SURVEY §8c: identical at 1 and 8 threads).
def _cpu_worker(kind, budget_s, seed):
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    from oracle import algos
    from tests.golden import cases, synth
    g = np.random.default_rng(3 + seed)
    np.random.seed(seed)

    def prefill(b, rows, act_cols, discrete):
        b.obs[:rows] = g.standard_normal((rows, OBS)); b.next_obs[:rows] = g.standard_normal((rows, OBS))
        b.actions[:rows] = g.integers(0, 4, (rows, act_cols)) if discrete else g.uniform(-1, 1, (rows, act_cols))
        b.rewards[:rows] = g.standard_normal(rows)
        b.dones[:rows] = g.random(rows) < 0.05
        b._size, b._index = rows, rows % b.capacity

    if kind.startswith("td3"):          # "td3" (the bench's full 1e6-row ring) or "td3:<rows>"
        cap = int(kind.split(":")[1]) if ":" in kind else CAP
        actor = synth.mlp_params(1 + seed, cases.actor_layers(OBS, ACT))
        critic = synth.mlp_params(2 + seed, cases.critic_layers(OBS + ACT, twin=True))
        pol = algos.TD3(actor, critic, OBS, ACT, 1e-3, 1e-3, cap)
        prefill(pol.buffer, cap, ACT, False)
        noise = g.standard_normal((BATCH, ACT)).astype(np.float32)
        step = lambda: pol.learn(BATCH, 0.99, 0.005, 0.2, 0.5, 1.0, 2, 1.0, noise=noise)
    else:                               # "dqn_loop:<rows>": the reference's DQN loop (DQN.py:294-343) on the synthetic discrete env
        rows = int(kind.split(":")[1])
        from freerl_amd.envs import LinearGaussianEnv
        env = LinearGaussianEnv(discrete=True)
        pol = algos.DQN(synth.mlp_params(5 + seed, [("l1", HIDDEN, OBS), ("l2", 4, HIDDEN)]), OBS, 4, 1e-3, rows)
        prefill(pol.buffer, rows, 1, True)      # a FULL ring of `rows` rows (the steady state of a run): the GPU leg uses the same
        state = {"obs": env.reset(seed=seed)[0]}

        def step():
            obs = state["obs"]
            a = pol.select_action(obs)
            if np.random.rand() < 0.1:                                         # DQN.py:307-310
                a = np.random.randint(4)
            nobs, r, term, trunc, _ = env.step(a)
            pol.add(obs, a, r, nobs, term)
            state["obs"] = env.reset(seed=seed)[0] if (term or trunc) else nobs
            pol.learn(BATCH, 0.99, 0.01)                                       # np.random.choice(len(buffer), 256, replace=False) inside
    for _ in range(3):
        step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        step()
        n += 1
    print(json.dumps({"n": n, "dt": time.perf_counter() - t0}), flush=True)


def _run_cpu_workers(kind, n_procs, budget_s):
    """n_procs worker processes side by side -> (aggregate units/s, per-process counts)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", kind, "--cpu-budget", str(budget_s),
                               "--cpu-seed", str(i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for i in range(n_procs)]
    rate, counts = 0.0, []
    for pr in procs:
        o, e = pr.communicate(timeout=600)
        if pr.returncode != 0:
            raise RuntimeError("cpu worker failed: " + e[-2000:])
        r = json.loads(o.strip().splitlines()[-1])
        rate += r["n"] / r["dt"]
        counts.append(r["n"])
    return rate, counts


def cpu_baseline(budget_s=6.0):
    """TD3.learn() of the bench's workload (1 learner, replay 1e6 full, batch 256, the reference's np.random.choice index
    draw included) on one host core and on all of them; plus the DQN loop's env-steps/s (see dqn_single_learner_loop)."""
    cores = os.cpu_count() or 1
    n_all = min(cores, 64)              # 64 x ~250 MB of float64 ring is enough to characterise the box
    one, c1 = _run_cpu_workers("td3", 1, budget_s)
    allc, call = _run_cpu_workers("td3", n_all, budget_s)
    # the same learn() from a 1e4-row ring: np.random.choice(size, 256, replace=False) permutes the whole ring per call (TD3.py:183), so
    # this point is the port's own arithmetic (~the reference's 5 ms) and the 1e6 one above mostly the permutation
    small, cs = _run_cpu_workers("td3:10000", 1, budget_s * 0.5)
    loops = {}
    for rows in DQN_LOOP_ROWS:
        r1, _ = _run_cpu_workers("dqn_loop:%d" % rows, 1, budget_s * 0.6)
        loops["replay %d rows" % rows] = {"one_core": r1}
        if rows in (DQN_LOOP_ROWS[0], DQN_LOOP_ROWS[-1]):
            ra, _ = _run_cpu_workers("dqn_loop:%d" % rows, n_all, budget_s * 0.6)
            loops["replay %d rows" % rows].update({"all_cores": ra, "processes": n_all})
    return {"value": one, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": "%d oracle TD3.learn() calls (1 learner, replay 1e6 full, batch 256, np.random.choice index draw "
                      "included) in %.1f s on one core" % (c1[0], budget_s),
            "ring_1e4_rows": {"value": small, "unit": "updates/s", "cores": 1,
                              "sample": "%d oracle TD3.learn() calls from a full 1e4-row ring in %.1f s on one core (separates the port's arithmetic "
                                        "from np.random.choice's O(ring) permutation)" % (cs[0], budget_s * 0.5)},
            "all_cores": {"value": allc, "unit": "updates/s", "cores": n_all, "host_cpus": cores,
                          "sample": "%d independent oracle learners, one process per core, %d calls in %.1f s" % (n_all, sum(call), budget_s)},
            "dqn_loop_env_steps_per_sec": loops}


def dropin_classes():
    """learn() through the reference-shaped CLASSES (freerl_amd.TD3 / .DQN: what an unchanged *_file training script calls),
    one learner, replay 1e6 rows full, batch 256: `rng="host"` consumes the reference's legacy NumPy stream — its
    np.random.choice(1e6, 256, replace=False) permutes the whole buffer per call (TD3.py:183), which is also what bounds the
    reference itself (22-30 ms) — `rng="device"` draws on the GPU; the default "auto" is "device" from 32768
    rows on.  updates/s of the blocking Python call sequence, losses not read back."""
    import torch
    from freerl_amd import _native as N
    from freerl_amd.DQN import DQN
    from freerl_amd.TD3 import TD3
    out = {}
    for name, mk, call in (
            ("TD3", lambda rng: TD3([OBS, ACT], True, 1e-3, 1e-3, CAP, "cuda", rng=rng, batch_max=BATCH),
             lambda p: p.learn(BATCH, 0.99, 0.005, 0.2, 0.5, 1.0, 2, 1.0)),
            ("DQN", lambda rng: DQN([OBS, 4], False, 1e-3, CAP, "cuda", rng=rng, batch_max=BATCH),
             lambda p: p.learn(BATCH, 0.99, 0.01))):
        for rng in ("host", "device", "auto"):
            pol = mk(rng)
            pol._e.fill_synthetic(CAP, seed=3)
            n = 12 if rng == "host" else 300
            for _ in range(3):
                call(pol)
            pol._e.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                call(pol)
            pol._e.sync()
            out["%s.learn rng=%s" % (name, rng)] = n / (time.perf_counter() - t0)
            pol._e.close()
    return {"unit": "updates/s (one learner, replay 1e6 full, batch 256, through the Python class)", **out}


def dqn_single_learner_loop(P=512):
    """north_star's env-steps/s target is quoted on the DQN loop (DQN.py:294-339: select_action -> epsilon-greedy -> env.step ->
    add -> learn per env step) of ONE learner.  LunarLander-v2 cannot be built here (no Box2D), so the env is the synthetic
    discrete task at its dims (obs 8, 4 actions).  The loop runs on FULL rings of the same sizes as the CPU leg
    (cpu_baseline.dqn_loop_env_steps_per_sec): the reference's np.random.choice(len(buffer), 256, replace=False) permutes the
    whole buffer per learn(), so its cost grows with the ring, the device draw's does not."""
    from freerl_amd import _native as N
    from freerl_amd.engine import Engine
    from freerl_amd.envpool import EnvPool, rollout

    def one(rows, E, n_learners=1, steps=400, threads=1):
        e = Engine(N.ALGO_DQN, 8, 4, rows, discrete=True, batch_max=BATCH, n_learners=n_learners, seed=1)
        g = np.random.default_rng(0)
        for p in range(n_learners):
            flat = (g.standard_normal(e.num_params(0)) * 0.05).astype(np.float32)
            e.set_params(0, flat, N.PARAM_ONLINE, learner=p); e.set_params(0, flat, N.PARAM_TARGET, learner=p)
        e.fill_synthetic(rows, seed=5)            # full ring: the run's steady state
        pool = EnvPool("SynLinearDiscrete-v0", n_learners * E, n_threads=threads, seed=2)
        kw = dict(envs_per_learner=E, start_steps=0, learn_every=1, epsilon=0.1, batch=BATCH, gamma=0.99, tau=0.01, critic_lr=1e-3)
        rollout(e, pool, 20, **kw)
        r = rollout(e, pool, steps, **kw)
        pool.close(); e.close()
        return r
    out = {"by_ring_rows": {}}
    for rows in DQN_LOOP_ROWS:                    # one learner x one env: the reference's own loop shape
        r = one(rows, 1)
        out["by_ring_rows"]["replay %d rows" % rows] = r["env_steps"] / r["seconds"]
    for E in (8, 64):                             # ... and with vectorised envs behind the same learner (ring 1e6)
        r = one(DQN_LOOP_ROWS[-1], E)
        out["%d env(s)" % E] = r["env_steps"] / r["seconds"]
    # the same loop for a population (BASELINE configs[0]'s algorithm at the bench's learner count, one env per learner)
    r = one(100_000, 1, n_learners=P, steps=300, threads=8)
    return {"unit": "env-steps/s, one DQN learner, one learn() per vector step, full replay ring", **out,
            "reference_cpu_env_steps_per_sec": "measured on this box: cpu_baseline.dqn_loop_env_steps_per_sec (BASELINE.md's 560 / 45 "
                                               "were taken in the survey container)",
            "population": {"learners": P, "envs_per_learner": 1, "ring_rows": 100_000, "env_steps_per_sec": r["env_steps"] / r["seconds"],
                           "updates_per_sec": r["updates"] / r["seconds"]}}


def baseline_configs_one_learner(device_id=0):
    """BASELINE.json's configs 4 and 5 as the reference runs them — ONE learner per GPU, one learn() per vector step: SAC at Humanoid-v4's
    376 + 17 columns (batch 256), MADDPG_simple on simple_spread's 3 x (18, 5) (batch 1024).  us per learn() with device-drawn rows,
    the kernel family that served it (round 6: kernels_solow.hip, sixteen workgroups per (learner, agent) unit with the first layer
    streamed from its block; 64 per unit at batch 1024), random weights, synthetic ring.  Not bench lines: a detail of the report."""
    from freerl_amd import _native as N
    from freerl_amd.engine import Engine
    out = {}
    for name, algo, obs, act, B, kw in (("config4_sac_humanoid_dims", N.ALGO_SAC, 376, 17, 256, dict(alpha_lr=1e-4, target_entropy=-17.0)),
                                        ("config5_maddpg_simple_spread_dims", N.ALGO_MADDPG, [18] * 3, [5] * 3, 1024, {})):
        try:
            e = Engine(algo, obs, act, 20_000, n_learners=1, twin_critic=(algo == N.ALGO_SAC), batch_max=B, seed=1, device_id=device_id)
            rng = np.random.default_rng(0)
            for net in range(e.n_nets):
                flat = (rng.standard_normal(e.num_params(net)) * 0.05).astype(np.float32)
                e.set_params(net, flat, N.PARAM_ONLINE); e.set_params(net, flat, N.PARAM_TARGET)
            if algo == N.ALGO_SAC:
                e.set_alpha_state([np.log(0.01), 0, 0, 0.01])
            e.fill_synthetic(20_000, seed=5)
            for k in range(10):
                e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, **kw)
            e.sync()
            t0 = time.perf_counter()
            n = 300
            for k in range(n):
                e.learn(B, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, **kw)
            e.sync()
            dt = (time.perf_counter() - t0) / n
            path = e.learn_path(B)
            out[name] = {"us_per_learn": dt * 1e6, "updates_per_sec": 1.0 / dt, "batch": B,
                         "kernel_family": "sixteen workgroups per unit, first layer streamed (kernels_solow.hip)" if path[0] and path[2] == 16 else ("chained" if path[0] else "row-chunk")}
            e.close()
        except Exception as ex:                       # a detail of the report, never the reason a bench line is missing
            out[name] = {"error": str(ex)[:200]}
    return out


def dqn_roofline(device_id=0):
    """The DQN update (north_star's own target algorithm: DQN_file/DQN.py:104-118) against BOTH rooflines.  One launch of
    dqn_fused_kernel (kernels_dqn2.hip) is the whole learn() of every resident learner — index draw, target + online forward,
    TD loss, backward, Adam, soft update: SURVEY 8(d)'s figures per learn() (frl_learn_work: 3.1 MFLOP, 72.8 KB at obs 8 / 4 actions /
    batch 256) x the launch's learners / the launch's average duration from HIP events on the engine's stream.  What bounds it is
    neither: the 128 -> 4 head and the 8 -> 128 input layer are padded to 16-wide MFMA tiles (the kernel ISSUES ~2.7x its useful
    flops) and a learner's chain of dependent phases (draw -> gather -> forward -> TD -> backward -> exchange -> Adam) is latency:
    `issued_frac` prices the padded tiles, `resident_learners` says how many chains a CU overlaps."""
    from freerl_amd import _native as N
    from freerl_amd.engine import Engine
    out = {"kernel": "dqn_fused_kernel", "unit": "per launch = one learn() of every learner; HIP events on the engine's stream",
           "shape": "obs 8, 4 actions, batch 256, hidden 128, replay 1e5 rows filled, device-drawn indices", "by_population": {}}
    for P in (512, 4096):
        e = Engine(N.ALGO_DQN, 8, 4, 100_000, discrete=True, batch_max=BATCH, n_learners=P, device_id=device_id, seed=1)
        g = np.random.default_rng(0)
        flat = (g.standard_normal(e.num_params(0)) * 0.05).astype(np.float32)
        for p in range(P):
            e.set_params(0, flat, N.PARAM_ONLINE, learner=p); e.set_params(0, flat, N.PARAM_TARGET, learner=p)
        e.fill_synthetic(100_000, seed=5)
        kw = dict(gamma=0.99, tau=0.01, critic_lr=1e-3, clip_norm=0.0)
        for _ in range(5):
            e.learn(BATCH, **kw)
        e.sync()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            e.learn(BATCH, **kw)
        e.sync()
        wall = (time.perf_counter() - t0) / n
        e.profile(True)
        for _ in range(n):
            e.learn(BATCH, **kw)
        prof = e.profile_read()
        e.profile(False)
        launch_s = prof["grad_critic"][0] / prof["grad_critic"][1] * 1e-3
        fl, by = e.learn_work(BATCH, False)                # whole launch: P learners
        issued = 2.0 * BATCH * (16 * 128 + 128 * 16) * (2 + 2) * P      # padded tiles: two forwards, backward ~ two more
        out["by_population"]["%d learners" % P] = {
            "avg_launch_ms": launch_s * 1e3, "updates_per_sec": P / wall, "flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
            "mfma": {"achieved": fl / launch_s / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / launch_s / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                     "issued_frac": issued / launch_s / 1e12 / FP32_MFMA_PEAK_TFLOPS},
            "hbm": {"achieved": by / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / launch_s / 1e9 / HBM_PEAK_GBS},
            "lds_bytes": e.learn_path(BATCH)[1]}
        e.close()
    out["bound"] = ("latency of one learner's dependent phases x resident learners per CU (two 77 KB workgroups); of the MFMA issue, "
                    "the padded head / input tiles are ~63 %: DESIGN.md 5")
    return out


def fifty_x_statement(gpu_by_rows, cpu_loops):
    """north_star: ">= 50x the reference CPU env-steps/s on DQN".  ONE statement from like-for-like points (same loop, same box,
    same full ring on both sides, one learner x one env against one core): the ratio per ring size and — interpolated
    log-log between the measured sizes — the ring size from which it is >= 50."""
    import math
    pts = []
    for rows in DQN_LOOP_ROWS:
        k = "replay %d rows" % rows
        pts.append((rows, gpu_by_rows[k] / cpu_loops[k]["one_core"]))
    cross = None
    if pts[0][1] >= 50:
        cross = pts[0][0]
    else:
        for (n0, r0), (n1, r1) in zip(pts, pts[1:]):
            if r0 < 50 <= r1:
                t = (math.log(50) - math.log(r0)) / (math.log(r1) - math.log(r0))
                cross = int(round(math.exp(math.log(n0) + t * (math.log(n1) - math.log(n0))), -3))
                break
    ratios = ", ".join("%.0fx at %.0e rows" % (r, n) for n, r in pts)
    text = ("one DQN learner stepping one env on the GPU vs the same loop on one host core of this box, same full replay ring: %s; "
            % ratios) + ("the ratio reaches 50x at a ring of ~%d rows" % cross if cross else "the ratio stays below 50x up to 1e6 rows")
    return {"ratio_by_ring_rows": {("%d" % n): r for n, r in pts}, "ring_rows_for_50x": cross, "statement": text}


def traffic_figure():
    """HBM bytes per ac_critic_kernel launch from the PMC passes (profiles/traffic.json, written by tools/pmc_summary.py
    --traffic together with a digest of the kernel sources it was measured on).  Returns (bytes | None, stale)."""
    pf = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(pf):
        return None, None
    try:
        t = json.load(open(pf))
    except Exception:
        return None, None
    from tools.pmc_summary import kernel_source_digest
    return t.get("hbm_bytes_per_launch"), (t.get("kernel_source_digest") != kernel_source_digest())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--learners", type=int, default=int(os.environ.get("FRL_BENCH_LEARNERS", "512")),
                    help="independent learners (seeds) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)       # internal: one CPU-baseline worker process
    ap.add_argument("--cpu-budget", type=float, default=6.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seed", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--spawn", action="store_true",
                    help="go through the launcher even for --gpus 1 (RCCL init + the metric all-reduce with one rank)")
    ap.add_argument("--headline-only", action="store_true",
                    help="profiling runs: only the P-learner engine (no P = 1 engine, no CPU baseline), so that rocprofv3's "
                         "per-kernel averages are averages over the headline launches")
    args = ap.parse_args()
    if args.cpu_worker:
        return _cpu_worker(args.cpu_worker, args.cpu_budget, args.cpu_seed)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    from freerl_amd import _native as N
    from freerl_amd import dist as fdist

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not launched and (args.gpus > 1 or args.spawn):
        # started without a launcher: become the launcher — N ranks of this script, one per GPU, under torch.distributed.run
        have = N.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d needs %d HIP devices on this node, found %d (the engine has no CPU "
                             "fallback and ranks do not share a GPU)" % (args.gpus, args.gpus, have))
        sys.exit(fdist.respawn(args.gpus, os.path.abspath(__file__), [a for a in sys.argv[1:] if a != "--spawn"]))

    # ONE JSON line on stdout: RCCL and gloo print banners to fd 1 at init ("RCCL version : ...", "[Gloo] Rank 0 is connected
    # ..."), so the real stdout is kept aside and fd 1 points at stderr for the rest of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    from freerl_amd.engine import Engine

    if not torch.cuda.is_available() or N.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    rank, world, local_rank = fdist.init()      # under a launcher: the engine's RCCL communicator (frl_comm_create) + a gloo control plane
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if local_rank >= N.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d HIP devices visible" % (rank, local_rank, N.device_count()))
    torch.cuda.set_device(local_rank)

    P = args.learners
    e = make_engine(N, Engine, P, local_rank, seed=1000 + rank)           # seeds sharded by rank

    def barrier():
        fdist.barrier()
        torch.cuda.synchronize()
        e.sync()

    for k in range(args.warmup):
        e.learn(BATCH, **td3_kwargs(k))
    barrier()
    e.timer_start()
    t0 = time.perf_counter()
    for k in range(args.steps):
        e.learn(BATCH, **td3_kwargs(k))
    kernel_ms = e.timer_stop()            # HIP events on the engine's stream around the K launches (synchronises)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0         # this rank's K steps, device drained; the job's time is the MAX over ranks (below)
    barrier()
    stats = e.stats()
    assert np.all(np.isfinite(stats)), "non-finite losses"

    # per-kernel durations of the launch chain (HIP events around every launch), outside the timed region
    e.profile(True)
    for k in range(args.steps):
        e.learn(BATCH, **td3_kwargs(k))
    prof = e.profile_read()
    e.profile(False)

    # env-steps/s: the full rollout-and-update loop (act + exploration on the device -> env.step on the host pool ->
    # add -> learn, one env per learner = the reference's UTD 1) on the synthetic obs-8/act-2 task
    from freerl_amd.envpool import EnvPool, rollout
    pool = EnvPool("SynLinear-v0", P, n_threads=8, seed=1000 + rank)
    rollout(e, pool, max(args.warmup, 20), start_steps=0, batch=BATCH)
    barrier()
    # the loop is host-paced (env workers, launch latency): one short sample is noise (537 k / 615 k / 661 k for one build in round 3),
    # so the reported figure is the MEDIAN of five rollouts of >= 200 vector steps each, with the spread next to it
    ro_runs = [rollout(e, pool, max(200, args.steps), start_steps=0, batch=BATCH) for _ in range(5)]
    ro_runs.sort(key=lambda r: r["env_steps"] / r["seconds"])
    ro = ro_runs[len(ro_runs) // 2]
    ro_rates = [r["env_steps"] / r["seconds"] for r in ro_runs]
    pool.close()

    # the path's only collective: the metric vector, summed over ranks / wall-clock maxed (RCCL over xGMI)
    learn_m = fdist.allreduce_metrics(env_steps=0.0, updates=float(P * args.steps), return_sum=0.0, episodes=0.0,
                                      loss_sum=float(np.mean(stats[:, 0, N.STAT_CRITIC_LOSS])), wall_s=dt, extra_max=[kernel_ms])
    roll_m = fdist.allreduce_metrics(env_steps=float(ro["env_steps"]), updates=float(ro["updates"]), return_sum=ro["return_sum"],
                                     episodes=float(ro["episodes"]), loss_sum=0.0, wall_s=ro["seconds"])
    dt_max, total_updates, kernel_ms = learn_m["wall_s_max"], learn_m["updates"], learn_m["extra_max"][0]
    env_sps = fdist.throughput(roll_m)["env_steps_per_sec"]
    ro_ups = fdist.throughput(roll_m)["updates_per_sec"]
    backend = fdist.collective_name()
    fdist.finalize()          # every rank leaves the collective window here; what follows is rank 0's own work

    if rank == 0:
        fl_a, by_a = e.learn_work(BATCH, True)
        fl_c, by_c = e.learn_work(BATCH, False)
        fx_a, fx_c = e.learn_work_executed(BATCH, True), e.learn_work_executed(BATCH, False)
        n_act = sum(1 for k in range(args.steps) if k % 2 == 1)
        step_s = kernel_ms * 1e-3 / args.steps
        kern = {k: {"avg_ms": v[0] / v[1], "launches": v[1]} for k, v in prof.items()}
        # dominant kernel: the critic stage.  At this population it is ac_critic_v2_twin_nv_kernel (kernels_critic2.hip; _nv = the
        # record layout allows 16-byte row loads): targets, twin-critic forward / backward, clip + Adam + soft update of one learner per
        # eight-wave workgroup in ONE launch ("adam_critic" absent from the per-kernel times); up to 128 learners ac_critic_kernel +
        # adam_fused_kernel.  Its flops are the same
        # algorithmic figure either way (frl_learn_work): the Adam / soft-update phase adds HBM bytes, not flops.
        fused = "adam_critic" not in kern
        dominant = "ac_critic_v2_twin_nv_kernel" if fused else "ac_critic_kernel"
        launch_s = kern["grad_critic"]["avg_ms"] * 1e-3
        stage_s = launch_s + (0.0 if fused else kern["adam_critic"]["avg_ms"] * 1e-3)
        achieved = fl_c / launch_s / 1e12
        flops, abytes = fl_c, by_c
        chained, lds, rc = e.learn_path(BATCH)
    e.close()
    if rank == 0:
        # single-learner latency (P = 1): the reference's own case — one learn() per env step (TD3.py:403-450).  One learner of this
        # shape runs kernels_solo.hip (sixteen workgroups per learner, slab sum + Adam behind a grid barrier)
        single, single_detail = None, None
        if not args.headline_only:
            single_detail = {"unit": "us per learn(), one learner, asynchronous calls", "batch": BATCH}
            for name, algo, twin, kw in (("td3", N.ALGO_TD3, True, None), ("ddpg", N.ALGO_DDPG, False, {}),
                                         ("sac", N.ALGO_SAC, True, dict(alpha_lr=1e-4, target_entropy=-float(ACT)))):
                if name == "td3":
                    e1 = make_engine(N, Engine, 1, local_rank, seed=7)
                else:
                    e1 = Engine(algo, OBS, ACT, 100_000, n_learners=1, twin_critic=twin, batch_max=BATCH, hidden=HIDDEN, device_id=local_rank, seed=7)
                    g1 = np.random.default_rng(7)
                    for net in range(2):
                        flat = (g1.standard_normal(e1.num_params(net)) * 0.05).astype(np.float32)
                        e1.set_params(net, flat, N.PARAM_ONLINE); e1.set_params(net, flat, N.PARAM_TARGET)
                    if algo == N.ALGO_SAC:
                        e1.set_alpha_state([np.log(0.01), 0, 0, 0.01])
                    e1.fill_synthetic(100_000, seed=5)
                call = (lambda k: e1.learn(BATCH, **td3_kwargs(k))) if name == "td3" else \
                    (lambda k: e1.learn(BATCH, gamma=0.99, tau=0.005, actor_lr=1e-3, critic_lr=1e-3, **kw))
                for k in range(20):
                    call(k)
                e1.sync()
                t1 = time.perf_counter()
                n1 = 1000
                for k in range(n1):
                    call(k)
                e1.sync()
                dt1 = (time.perf_counter() - t1) / n1
                path = e1.learn_path(BATCH)
                single_detail[name] = dt1 * 1e6
                single_detail["kernel_family"] = "solo: 16 workgroups per learner (kernels_solo.hip)" if path[2] == 16 and path[0] else ("chained" if path[0] else "row-chunk")
                if name == "td3":
                    single = 1.0 / dt1
                e1.close()
            # ... and sixteen learners: the largest population of the same kernels (16 x 16 workgroups = every CU of the chip)
            try:
                e16 = make_engine(N, Engine, 16, local_rank, seed=7)
                for k in range(20):
                    e16.learn(BATCH, **td3_kwargs(k))
                e16.sync()
                t1 = time.perf_counter()
                for k in range(400):
                    e16.learn(BATCH, **td3_kwargs(k))
                e16.sync()
                dt16 = (time.perf_counter() - t1) / 400
                p16 = e16.learn_path(BATCH)
                single_detail["td3_16_learners"] = {"us_per_learn": dt16 * 1e6, "updates_per_sec": 16 / dt16,
                                                    "kernel_family": "solo" if p16[2] == 16 and p16[0] else ("chained" if p16[0] else "row-chunk")}
                e16.close()
            except Exception as ex:                       # a detail of the report, never the reason a bench line is missing
                single_detail["td3_16_learners"] = {"error": str(ex)[:200]}
        traffic, traffic_stale = traffic_figure()
        line = {
            "metric": "learner_updates_per_sec", "value": total_updates / dt_max, "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "TD3.learn() (BASELINE configs[1] algorithm; north_star synthetic shape): obs_dim 8, "
                                   "act_dim 2, batch 256, replay 1e6 rows filled, hidden 128, policy_freq 2, "
                                   "device-drawn indices/noise",
                       "learners_per_gpu": P, "updates_per_step": P * world, "kernel_family": "chained (one eight-wave workgroup per learner, two waves per SIMD)" if chained else "row-chunk",
                       "rows_per_workgroup": rc, "lds_bytes": lds,
                       "parallelism": "seeds sharded over %d GPU(s), no data-path collective" % world,
                       "collective": "metric all-reduce: %s" % backend},
            "env_steps_per_sec": env_sps,
            "rollout": {"env": "SynLinear-v0 (obs 8, act 2)", "envs_per_learner": 1, "env_workers": 8,
                        "updates_per_sec_in_loop": ro_ups,
                        "sample": "median of 5 rollouts of %d vector steps (this rank: min %.0f, median %.0f, max %.0f env-steps/s)"
                                  % (max(200, args.steps), ro_rates[0], ro_rates[len(ro_rates) // 2], ro_rates[-1]),
                        "loop": "act kernel with device-side exploration -> D2H actions -> host env pool step -> staged add "
                                "(one H2D) -> learn"},
            "single_learner_updates_per_sec": single, "single_learner": single_detail,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_stale": traffic_stale,
                         "kernel": dominant, "avg_launch_ms": launch_s * 1e3,
                         "critic_stage_ms": stage_s * 1e3, "critic_stage_tflops": fl_c / stage_s / 1e12,
                         "note": "round 6: eight-wave workgroups (two waves per SIMD, every wave inside 256 registers) - FRL_CHAIN_WAVES=4 runs "
                                 "rounds 2-5's four-wave kernels on the same box; the clip + Adam phase (~17 % of the launch) streams "
                                 "theta / m / v at the chip's HBM rate with every CU in it at once and is not overlapped with MFMA work",
                         "flops_per_launch": flops, "algorithmic_bytes_per_launch": abytes,
                         "flops_convention": "achieved / frac: SURVEY 8(d)'s formula, 2 B sum(in x out) x (#fwd + 2 x #bwd) (frl_learn_work); "
                                             "*_executed: the flops autograd and these kernels execute - no first-layer dX of a trained net, "
                                             "action columns only for dQ/da (frl_learn_work_executed)",
                         "flops_executed_per_launch": fx_c, "achieved_executed": fx_c / launch_s / 1e12,
                         "frac_executed": fx_c / launch_s / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                         "step_tflops_executed": (fx_a * n_act + fx_c * (args.steps - n_act)) / args.steps / step_s / 1e12,
                         "hbm_bound_frac": abytes / launch_s / 1e9 / HBM_PEAK_GBS,
                         "step_flops_avg": (fl_a * n_act + fl_c * (args.steps - n_act)) / args.steps,
                         "step_ms_avg": step_s * 1e3,
                         "step_tflops": (fl_a * n_act + fl_c * (args.steps - n_act)) / args.steps / step_s / 1e12,
                         "kernels": kern},
            "baseline_configs_one_learner": None if args.headline_only else baseline_configs_one_learner(local_rank),
            "roofline_dqn": None if args.headline_only else dqn_roofline(local_rank),
            "dropin_classes": None if args.headline_only else dropin_classes(),
            "dqn_single_learner_loop": None if args.headline_only else dqn_single_learner_loop(args.learners),
            "cpu_baseline": None if (args.no_cpu_baseline or args.headline_only) else cpu_baseline(),
        }
        cb, dl = line["cpu_baseline"], line["dqn_single_learner_loop"]
        if cb and dl:     # north_star's ">= 50x the reference CPU env-steps/s": same loop, same box, same ring, GPU engine / CPU port
            dl["vs_cpu_port_same_box"] = fifty_x_statement(dl["by_ring_rows"], cb["dqn_loop_env_steps_per_sec"])
            small, full = cb["dqn_loop_env_steps_per_sec"]["replay %d rows" % DQN_LOOP_ROWS[0]], cb["dqn_loop_env_steps_per_sec"]["replay %d rows" % DQN_LOOP_ROWS[-1]]
            dl["vs_cpu_port_same_box"]["population_vs_all_cores"] = {
                "ring 1e5 rows on the GPU / 1e4 on the CPU (its fastest)": dl["population"]["env_steps_per_sec"] / small["all_cores"],
                "ring 1e5 rows on the GPU / 1e6 on the CPU": dl["population"]["env_steps_per_sec"] / full["all_cores"]}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
