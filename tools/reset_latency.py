#!/usr/bin/env python3
"""Latency of reset() on the single-env drop-in surface and of cz_reset on small batches (run from the tree to measure)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from cooking_zoo_amd.environment.cooking_env import parallel_env
from cooking_zoo_amd.vec_env import CookingVecEnv
pe = parallel_env(level="coop_test", meta_file="example", num_agents=2, max_steps=400, recipes=["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3")
for _ in range(20):
    pe.reset()
t0 = time.perf_counter()
for _ in range(300):
    pe.reset()
print(f"parallel_env.reset() (level instantiation in Python + layout upload + cz_reset + observation): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us")
for n in (1, 64, 4096):
    env = CookingVecEnv(n, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=8)
    for _ in range(10):
        env.reset()
    t0 = time.perf_counter()
    for _ in range(200):
        env.reset()
    a = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    for _ in range(200):
        env.reset(return_obs=False)
    b = (time.perf_counter() - t0) / 200 * 1e6
    print(f"CookingVecEnv.reset() N={n}: {a:.1f} us with observations, {b:.1f} us without")
    env.close()
