"""Deterministic synthetic inputs shared by the golden generator, the oracle tests and the
GPU parity tests.  Everything is derived from ``np.random.default_rng`` (PCG64) raw
streams (`random`, `standard_normal`, `integers`, `uniform`), which are stable across NumPy
releases, so fixtures only need to store OUTPUTS — the inputs are regenerated from seeds.

Shapes follow SURVEY.md §8(c)/(d): obs ~ N(0,1) f32, actions uniform(-1,1) f32 or
integers, reward ~ N(0,1) f32, done ~ Bernoulli(0.05).
"""
import numpy as np

F32 = np.float32


def transitions(seed, n, obs_dim, act_dim, n_discrete=None):
    """Transition table in the survey's draw order: obs, next_obs, act, rew, done."""
    g = np.random.default_rng(seed)
    obs = g.standard_normal((n, obs_dim)).astype(F32)
    next_obs = g.standard_normal((n, obs_dim)).astype(F32)
    if n_discrete is not None:
        act = g.integers(0, n_discrete, (n, 1)).astype(F32)
    else:
        act = g.uniform(-1, 1, (n, act_dim)).astype(F32)
    rew = g.standard_normal(n).astype(F32)
    done = g.random(n) < 0.05
    return dict(obs=obs, act=act, rew=rew, next_obs=next_obs, done=done)


def linear_params(g, prefix, out_dim, in_dim, scale=1.0):
    """U(-1/sqrt(in), 1/sqrt(in)) weight and bias (the distribution nn.Linear's default
    init produces), drawn from PCG64 instead of torch's generator."""
    bound = scale / np.sqrt(in_dim)
    w = g.uniform(-bound, bound, (out_dim, in_dim)).astype(F32)
    b = g.uniform(-bound, bound, (out_dim,)).astype(F32)
    return {prefix + ".weight": w, prefix + ".bias": b}


def mlp_params(seed, layers):
    """layers: list of (name, out_dim, in_dim).  Returns an ordered dict name->array."""
    g = np.random.default_rng(seed)
    out = {}
    for name, o, i in layers:
        out.update(linear_params(g, name, o, i))
    return out


def indices(seed, size, batch):
    """`batch` unique indices in [0,size): argsort of a uniform stream (no Generator.choice)."""
    g = np.random.default_rng(seed)
    return np.argsort(g.random(size), kind="stable")[:batch].astype(np.int64)


def permutation(seed, n):
    g = np.random.default_rng(seed)
    return np.argsort(g.random(n), kind="stable").astype(np.int64)


def normal(seed, shape):
    g = np.random.default_rng(seed)
    return g.standard_normal(shape).astype(F32)


def digest(x, n_sample=64):
    """Small fingerprint of a tensor: sum, abs-sum (float64) and a strided sample."""
    x = np.asarray(x)
    flat = x.reshape(-1).astype(np.float64)
    step = max(1, flat.size // n_sample)
    return dict(sum=flat.sum(), abssum=np.abs(flat).sum(), sample=flat[::step][:n_sample].astype(F32),
                shape=np.array(x.shape, dtype=np.int64))


def pack_digest(prefix, tensors, out, full_limit=4000):
    """Store digests (and the full tensors when small) of a name->array dict into `out`."""
    total = sum(int(np.asarray(t).size) for t in tensors.values())
    for name, t in tensors.items():
        d = digest(t)
        out["%s/%s/sum" % (prefix, name)] = np.float64(d["sum"])
        out["%s/%s/abssum" % (prefix, name)] = np.float64(d["abssum"])
        out["%s/%s/sample" % (prefix, name)] = d["sample"]
        if total <= full_limit:
            out["%s/%s/full" % (prefix, name)] = np.asarray(t, dtype=F32)


def check_digest(prefix, tensors, fixture, rtol, atol, label=""):
    """Compare a name->array dict with the digests stored by pack_digest.  Returns the
    worst relative error seen (for reporting)."""
    worst = 0.0
    for name, t in tensors.items():
        t = np.asarray(t)
        d = digest(t)
        ref_s = np.asarray(fixture["%s/%s/sample" % (prefix, name)])
        np.testing.assert_allclose(d["sample"], ref_s, rtol=rtol, atol=atol,
                                   err_msg="%s %s/%s sample" % (label, prefix, name))
        ref_abs = float(fixture["%s/%s/abssum" % (prefix, name)])
        ref_sum = float(fixture["%s/%s/sum" % (prefix, name)])
        tol = rtol * ref_abs + atol * t.size
        assert abs(d["sum"] - ref_sum) <= tol, "%s %s/%s sum %r vs %r" % (label, prefix, name, d["sum"], ref_sum)
        assert abs(d["abssum"] - ref_abs) <= tol, "%s %s/%s abssum" % (label, prefix, name)
        key = "%s/%s/full" % (prefix, name)
        if key in fixture:
            ref = np.asarray(fixture[key])
            np.testing.assert_allclose(t, ref.reshape(t.shape), rtol=rtol, atol=atol,
                                       err_msg="%s %s/%s full" % (label, prefix, name))
            denom = np.maximum(np.abs(ref.reshape(t.shape)), 1e-3)
            worst = max(worst, float(np.max(np.abs(t - ref.reshape(t.shape)) / denom)))
    return worst
