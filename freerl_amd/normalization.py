"""Host-side normalisers with the reference's surface (PPO_file/normalization.py:17-101; inline
copies in SAC.py:334-421, DDPG.py:305-403): `Normalization`, `RewardScaling` run once per env
step in the CALLER's loop on single observations/rewards, so they stay host code here too.
`Normalization_batch_size`, the per-sampled-batch variant that runs inside `sample()` on device
tensors, lives in the engine (`frl_obsnorm_*`, enabled by trick/supplement `Batch_ObsNorm`)."""
import numpy as np


class RunningMeanStd:
    def __init__(self, shape):
        self.n = 0
        self.mean = np.zeros(shape)
        self.S = np.zeros(shape)
        self.std = np.sqrt(self.S)

    def update(self, x):
        x = np.array(x)
        self.n += 1
        if self.n == 1:          # the reference's first update sets std = x (normalization.py:27-29)
            self.mean, self.std = x, x
            return
        delta = x - self.mean
        self.mean = self.mean + delta / self.n
        self.S = self.S + delta * (x - self.mean)
        self.std = np.sqrt(self.S / self.n)


class Normalization:
    def __init__(self, shape):
        self.running_ms = RunningMeanStd(shape=shape)

    def __call__(self, x, update=True):
        if update:
            self.running_ms.update(x)
        return (x - self.running_ms.mean) / (self.running_ms.std + 1e-8)


class RewardScaling:
    def __init__(self, shape, gamma):
        self.shape, self.gamma = shape, gamma
        self.running_ms = RunningMeanStd(shape=self.shape)
        self.R = np.zeros(self.shape)

    def __call__(self, x):
        self.R = self.gamma * self.R + x
        self.running_ms.update(self.R)
        return x / (self.running_ms.std + 1e-8)

    def reset(self):
        self.R = np.zeros(self.shape)
