#!/usr/bin/env python3
"""Which waves set the duration of a launch?  Times the bench step (4096 envs, one launch per step) with episodes that
end every 400 steps (about ten envs re-instantiate themselves per launch) and with episodes that never end."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for max_steps, mode in ((400, "random"), (1 << 30, "random"), (1 << 30, "stay")):
    env = CookingVecEnv(N, "coop_test", "example", 2, max_steps, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                        num_layouts=256, auto_reset=True)
    env.reset(return_obs=False)
    L, h = _native.lib(), env._h
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 5, size=(256, N, 2), dtype=np.int32)
    if mode == "stay":
        acts[:] = 0
    d_act = env.alloc((256, N, 2), np.int32)
    d_act.from_host(acts)
    d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
    d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
    run = lambda k: L.cz_step_device_many(h, k, d_act.ptr, N * 2, 256, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
    run(600)
    env.sync()
    L.cz_timer_start(h)
    run(2000)
    ms = C.c_float()
    L.cz_timer_stop(h, C.byref(ms))
    print(f"max_steps={max_steps:<10d} actions={mode:6s}: {ms.value * 1e3 / 2000:.2f} us per launch, episodes finished {env.stats()['episodes']}")
    env.close()
