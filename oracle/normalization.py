"""Oracle normalisers: restates PPO_file/normalization.py:17-101 (identical copies in
PPO_advance/ and MAPPO_file/).  Test infrastructure (see oracle/__init__.py)."""
import numpy as np

F32 = np.float32


class RunningMeanStd:
    """normalization.py:17-35: per-element Welford on host float64.  The FIRST update sets
    std = x (not 0) — kept as in the reference."""

    def __init__(self, shape):
        self.n = 0
        self.mean = np.zeros(shape)
        self.S = np.zeros(shape)
        self.std = np.sqrt(self.S)

    def update(self, x):
        x = np.array(x)
        self.n += 1
        if self.n == 1:
            self.mean = x
            self.std = x
        else:
            old = self.mean.copy()
            self.mean = old + (x - old) / self.n
            self.S = self.S + (x - old) * (x - self.mean)
            self.std = np.sqrt(self.S / self.n)


class Normalization:
    """normalization.py:38-48: (x-mean)/(std+1e-8), statistics updated unless update=False."""

    def __init__(self, shape):
        self.running_ms = RunningMeanStd(shape)

    def __call__(self, x, update=True):
        if update:
            self.running_ms.update(x)
        return (x - self.running_ms.mean) / (self.running_ms.std + 1e-8)


class RunningMeanStdBatch:
    """normalization.py:53-71: the same recurrence on the BATCH MEAN, float32 tensors."""

    def __init__(self, shape):
        self.n = 0
        self.mean = np.zeros(shape, dtype=F32)
        self.S = np.zeros(shape, dtype=F32)
        self.std = np.sqrt(self.S)

    def update(self, x):
        x = x.mean(axis=0, keepdims=True, dtype=F32)
        self.n += 1
        if self.n == 1:
            self.mean = x
            self.std = x
        else:
            old = self.mean
            self.mean = (old + (x - old) / F32(self.n)).astype(F32)
            self.S = (self.S + (x - old) * (x - self.mean)).astype(F32)
            self.std = np.sqrt(self.S / F32(self.n)).astype(F32)


class NormalizationBatch:
    """normalization.py:74-84."""

    def __init__(self, shape):
        self.running_ms = RunningMeanStdBatch(shape)

    def __call__(self, x, update=True):
        if update:
            self.running_ms.update(x)
        return ((x - self.running_ms.mean) / (self.running_ms.std + F32(1e-8))).astype(F32)


class RewardScaling:
    """normalization.py:87-101: divide the reward by the running std of the discounted return."""

    def __init__(self, shape, gamma):
        self.shape, self.gamma = shape, gamma
        self.running_ms = RunningMeanStd(shape)
        self.R = np.zeros(shape)

    def __call__(self, x):
        self.R = self.gamma * self.R + x
        self.running_ms.update(self.R)
        return x / (self.running_ms.std + 1e-8)

    def reset(self):
        self.R = np.zeros(self.shape)
