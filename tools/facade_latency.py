#!/usr/bin/env python3
"""Latency of the single-env drop-in surface: cz_step through ctypes (host pointers) and the parallel_env facade."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd.environment.cooking_env import parallel_env  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

for n in (1, 16, 256, 1024, 4096):
    env = CookingVecEnv(n, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                        num_layouts=8, auto_reset=True)
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 5, size=(2000, n, 2), dtype=np.int32)
    for i in range(200):
        env.step(acts[i])
    t0 = time.perf_counter()
    for i in range(2000):
        env.step(acts[i])
    dt = time.perf_counter() - t0
    print(f"CookingVecEnv.step (host arrays, obs included) N={n}: {dt / 2000 * 1e6:.1f} us/step  ({n * 2000 / dt:.0f} env-steps/s)")
    env.close()

pe = parallel_env(level="coop_test", meta_file="example", num_agents=2, max_steps=400,
                  recipes=["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3")
pe.reset()
rng = np.random.default_rng(0)
k = 0
t0 = time.perf_counter()
for i in range(4000):
    if not pe.agents:
        pe.reset()
    pe.step({a: int(rng.integers(5)) for a in pe.agents})
    k += 1
dt = time.perf_counter() - t0
print(f"parallel_env facade (dict API, resets included): {dt / k * 1e6:.1f} us/step  ({k / dt:.0f} steps/s)")

v = pe._vec
t0 = time.perf_counter()
for i in range(2000):
    v.get_state()
print(f"get_state (1 env): {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us")
