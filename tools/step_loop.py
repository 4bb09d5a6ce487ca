#!/usr/bin/env python3
"""Minimal driver for profiling: N launches of cz_step_device on the bench workload (no timing, no fused pass)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd.vec_env import CookingVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                    num_layouts=256, auto_reset=True)
env.reset(return_obs=False)
rng = np.random.default_rng(0)
d_act = env.alloc((64, N, 2), np.int32); mode = sys.argv[3] if len(sys.argv) > 3 else 'random'
acts = rng.integers(0, 5, size=(64, N, 2), dtype=np.int32)
if mode == 'zero':
    acts[:] = 0
d_act.from_host(acts)
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
from cooking_zoo_amd import _native
L, h = _native.lib(), env._h
# warm the world up with real steps first (objects get picked up) using the shipped code path semantics
for t in range(n):
    L.cz_step_device(h, d_act.ptr + (t % 64) * N * 8, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
env.sync()
