"""MADDPG_file/MATD3_simple.py's class, see MADDPG.py."""
from .MADDPG import Agent, MATD3  # noqa: F401
