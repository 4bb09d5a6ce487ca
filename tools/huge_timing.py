#!/usr/bin/env python3
"""GPU box: step rate of the three kernel instances (one launch per step, events around 200 launches).
usage: python tools/huge_timing.py [num_envs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

CASES = [("coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"]),
         ("large_16x16", "large_16x16", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"]),
         ("huge_20x20", "huge_20x20", 3, ["TomatoLettuceSalad", "MashedCarrotBanana", "TomatoSalad"]),
         ("huge_objs_16x16", "huge_objs_16x16", 3, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon"]),
         ("huge_32x32", "huge_32x32", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"])]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    for level, meta, agents, recipes in CASES:
        env = CookingVecEnv(n, level, meta, agents, 400, recipes, action_scheme="scheme3", num_layouts=16)
        env.reset()
        rng = np.random.default_rng(0)
        d_act = env.alloc((n, agents), np.int32)
        d_act.from_host(rng.integers(0, env.n_actions, size=(n, agents), dtype=np.int32))
        d_obs = env.alloc((n, agents, env.F), np.float64)
        d_rew = env.alloc((n, agents), np.float64)
        d_t = env.alloc((n, agents), np.uint8)
        d_u = env.alloc((n, agents), np.uint8)
        for _ in range(20):
            env.step_device(d_act, d_obs, d_rew, d_t, d_u)
        env.sync()
        K = 200
        t0 = time.perf_counter()
        for _ in range(K):
            env.step_device(d_act, d_obs, d_rew, d_t, d_u)
        env.sync()
        us = (time.perf_counter() - t0) / K * 1e6
        print(f"{level:18s} {env.dims.W}x{env.dims.H} D={env.dims.D:3d} A={agents} F={env.F:5d}: {us:8.2f} us/step  "
              f"{n / us:7.2f} M env-steps/s  obs {n * agents * env.F * 8 / us / 1e6:6.2f} TB/s", flush=True)
        env.close()


if __name__ == "__main__":
    main()
