"""freerl_amd — MI355X-native replay-sample + batched-update engine behind FreeRL's class surface.

The hot path of FreeRL's DQN / DDPG / TD3 / SAC / PPO / MADDPG scripts (`Buffer.sample` ->
`Agent.learn` -> soft update, plus `select_action` and PPO's GAE) runs in hand-written HIP
kernels for gfx950 behind the C ABI of `include/freerl_hip.h`.  This package is the
reference-side binding: same class names, constructor arguments, method signatures and
checkpoint layout as the reference scripts, no CPU fallback.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
