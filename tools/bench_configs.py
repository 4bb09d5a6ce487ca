#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configs (3: 65536 envs mixed levels/recipes; 5: 65536 envs, 4 agents,
16x16, max density) and the config-2 workload at larger batches.  One JSON line per case.  GPU box only."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402


def measure(name, env, steps=300, warm=30, fused_T=32):
    N, A = env.num_envs, env.num_agents
    L, h = _native.lib(), env._h
    env.reset(return_obs=False)
    rng = np.random.default_rng(0)
    chunk = 32
    d_act = env.alloc((chunk, N, A), np.int32)
    d_act.from_host(rng.integers(0, env.n_actions, size=(chunk, N, A), dtype=np.int32))
    d_obs = env.alloc((N, A, env.F), np.float64)
    d_rew = env.alloc((N, A), np.float64)
    d_t = env.alloc((N, A), np.uint8)
    d_u = env.alloc((N, A), np.uint8)
    sb = N * A * 4

    def run(k, first):
        for t in range(first, first + k):
            L.cz_step_device(h, d_act.ptr + (t % chunk) * sb, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
    run(warm, 0)
    env.sync()
    s0 = env.stats()["env_steps"]
    t0 = time.perf_counter()
    run(steps, warm)
    env.sync()
    dt = time.perf_counter() - t0
    per_step = (env.stats()["env_steps"] - s0) / dt
    # fused
    d_traj = env.alloc((fused_T, N, A, env.F), np.float64)
    env.rollout(fused_T, 1, 0, d_traj)
    env.sync()
    reps = max(2, steps // fused_T)
    s0 = env.stats()["env_steps"]
    t0 = time.perf_counter()
    for r in range(reps):
        env.rollout(fused_T, 1, (r + 1) * fused_T, d_traj)
    env.sync()
    dtf = time.perf_counter() - t0
    fused = (env.stats()["env_steps"] - s0) / dtf
    obs_bytes = A * env.F * 8
    print(json.dumps({"case": name, "envs": N, "agents": A, "F": env.F, "per_step_env_steps_per_s": per_step,
                      "per_step_us": dt / steps * 1e6, "fused_env_steps_per_s": fused, "fused_us_per_step": dtf / (reps * fused_T) * 1e6,
                      "obs_GBps_per_step_api": per_step * obs_bytes / 1e9, "obs_GBps_fused": fused * obs_bytes / 1e9}))
    env.close()


def main():
    which = sys.argv[1:] or ["cfg2", "cfg2_16k", "cfg4_shard", "cfg2_64k", "cfg3", "cfg5"]
    L = os.path.join(REPO, "cooking_zoo_amd", "utils", "level")
    if "cfg2" in which:
        measure("cfg2: 4096 envs coop_test 2 agents", CookingVecEnv(4096, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256))
    if "cfg2_16k" in which:
        measure("cfg2 workload at 16384 envs", CookingVecEnv(16384, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256), steps=200)
    if "cfg4_shard" in which:
        # config 4 = 262 144 envs over 8 GPUs: what ONE of its ranks runs (global env ids 98 304..131 071 = rank 3)
        measure("cfg4: one rank's shard of 262144 envs over 8 GPUs (32768 envs, env_id_base 98304)",
                CookingVecEnv(32768, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                              num_layouts=256, env_id_base=98304), steps=150, fused_T=16)
    if "cfg2_64k" in which:
        measure("cfg2 workload at 65536 envs", CookingVecEnv(65536, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256), steps=100, fused_T=16)
    if "cfg3" in which:
        n = 65536
        rid = np.array([[e % 8, (e + 1) % 8] for e in range(n)])
        measure("cfg3: 65536 envs, levels e%3 of coop/coexistence/switch, recipes cycling over the book",
                CookingVecEnv(n, ["coop_test", "coexistence_test", "switch_test"], "example", 2, 400, rid, action_scheme="scheme3", num_layouts=256), steps=100, fused_T=16)
    if "cfg5" in which:
        measure("cfg5: 65536 envs, 4 agents competing, large_16x16, max object density",
                CookingVecEnv(65536, "large_16x16", "large_16x16", 4, 400, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], action_scheme="scheme3", num_layouts=64), steps=40, fused_T=4)


if __name__ == "__main__":
    main()
