#!/usr/bin/env python3
"""Driver for profiling: runs of cz_step_device_ring as fused launches with the outputs in place (cz_set_ring_fused), 4096 envs of
the bench workload.   python3 tools/ring_fused_run.py [K=2000] [runs=5]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, A, period = 4096, 2, 64
env = CookingVecEnv(N, "coop_test", "example", A, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
env.reset(return_obs=False)
d_ring = env.alloc((period, N, A), np.int32)
d_ring.from_host(np.random.default_rng(0).integers(0, 5, size=(period, N, A), dtype=np.int32))
d_obs, d_rew = env.alloc((N, A, env.F), np.float64), env.alloc((N, A), np.float64)
d_t, d_u = env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
env.set_ring_fused(True)
for r in range(runs):
    env.step_device_ring(K, d_ring, N * A, period, 0, d_obs, d_rew, d_t, d_u)
env.sync()
print("fused ring steps issued:", env.ring_fused_steps(), "in", runs, "runs of", K)
