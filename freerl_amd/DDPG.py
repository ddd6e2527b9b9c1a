"""`from freerl_amd.DDPG import DDPG` — see freerl_amd/TD3.py (DDPG shares TD3's engine path)."""
from .TD3 import DDPG, Agent  # noqa: F401
