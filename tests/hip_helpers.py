"""Helpers shared by the GPU parity tests: pack records / parameters for the C ABI."""
import numpy as np

F32 = np.float32


def flat_params(p, layer_names, extra=None):
    parts = []
    for n in layer_names:
        parts += [np.asarray(p[n + ".weight"], F32).reshape(-1), np.asarray(p[n + ".bias"], F32).reshape(-1)]
    if extra:
        parts.append(np.asarray(p[extra], F32).reshape(-1))
    return np.concatenate(parts)


def unflat_params(flat, template, layer_names, extra=None):
    out, o = {}, 0
    for n in layer_names:
        for suffix in (".weight", ".bias"):
            shp = np.asarray(template[n + suffix]).shape
            sz = int(np.prod(shp))
            out[n + suffix] = flat[o:o + sz].reshape(shp).copy()
            o += sz
    if extra:
        shp = np.asarray(template[extra]).shape
        sz = int(np.prod(shp))
        out[extra] = flat[o:o + sz].reshape(shp).copy()
        o += sz
    assert o == flat.size
    return out


def records(tabs, extra=None):
    """tabs: list (one per agent) of transition dicts -> [n, width] records in the engine's
    column order [obs_all | act_all | rew_all | done_all | next_obs_all | extra]."""
    n = len(tabs[0]["rew"])
    cols = [t["obs"] for t in tabs] + [t["act"].reshape(n, -1) for t in tabs]
    cols += [t["rew"].reshape(n, 1) for t in tabs] + [t["done"].astype(F32).reshape(n, 1) for t in tabs]
    cols += [t["next_obs"] for t in tabs]
    if extra is not None:
        cols.append(extra)
    return np.concatenate([np.asarray(c, F32) for c in cols], axis=1)


def rel_err(a, b, floor=1e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0
