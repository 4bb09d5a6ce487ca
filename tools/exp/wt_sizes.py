#!/usr/bin/env python3
"""Experiment (round 6): launch time of the one-step kernel by observation-store flavour (CZ_WT = 0 plain, 1 write-through sc1, 2 streaming nt)
for a level / agent count / batch size - where the host's automatic choice should switch.

    python3 tools/exp/wt_sizes.py large 1024 2048 4096 8192        # large_16x16, 4 agents, F = 840
    python3 tools/exp/wt_sizes.py coop 8192 12288                  # coop_test, 2 agents, F = 278
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

CFG = {"coop": ("coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"]),
       "large": ("large_16x16", "large_16x16", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"])}
level, meta, A, recipes = CFG[sys.argv[1]]
L = _native.lib()
for n in [int(x) for x in sys.argv[2:]]:
    row = []
    for wt in ("auto", "0", "1", "2"):
        if wt == "auto":
            os.environ.pop("CZ_WT", None)
        else:
            os.environ["CZ_WT"] = wt
        env = CookingVecEnv(n, level, meta, A, 400, recipes, action_scheme="scheme3", num_layouts=64, auto_reset=True)
        env.reset(return_obs=False)
        P = 16
        d_act = env.alloc((P, n, A), np.int32)
        d_act.from_host(np.random.default_rng(0).integers(0, 5, size=(P, n, A), dtype=np.int32))
        outs = (env.alloc((n, A, env.F), np.float64).ptr, env.alloc((n, A), np.float64).ptr, env.alloc((n, A), np.uint8).ptr, env.alloc((n, A), np.uint8).ptr)
        K = 400
        _native.check(env._h, L.cz_step_device_ring(env._h, K, d_act.ptr, n * A, P, 0, *outs))
        env.sync()
        best = 1e9
        for _ in range(3):
            ms = C.c_float()
            L.cz_timer_start(env._h)
            _native.check(env._h, L.cz_step_device_ring(env._h, K, d_act.ptr, n * A, P, 0, *outs))
            L.cz_timer_stop(env._h, C.byref(ms))
            best = min(best, ms.value * 1e3 / K)
        row.append(f"{wt} {best:8.2f}")
        mib = n * A * env.F * 8 / (1 << 20)
        env.close()
    print(f"{sys.argv[1]} {n:7d} envs ({mib:7.1f} MiB of observations per launch): us per launch  " + "   ".join(row), flush=True)
