"""rocprofv3 target: the Double + PER DQN rollout loop at 512 learners x 8 envs (4096-row adds), for per_add_kernel's duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rollout_bench
rollout_bench.run("per", 512, 8, False, steps=100)
