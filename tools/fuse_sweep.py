#!/usr/bin/env python3
"""cz_rollout with T = 1, 2, 4, 8, 16 steps per launch (python3 tools/fuse_sweep.py [N [T,T,.. [steps [noobs]]]]): how much of the fused kernel's advantage comes from
amortising the launch (end-of-kernel write-back, cold start, ramp)?"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from cooking_zoo_amd import _native
from cooking_zoo_amd.vec_env import CookingVecEnv
L = _native.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
h = env._h
env.reset(return_obs=False)
Ts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
with_obs = not (len(sys.argv) > 4 and sys.argv[4] == "noobs")
Tmax = max(Ts)
d_traj = env.alloc((Tmax, N, 2, env.F), np.float64) if with_obs else None
for T in Ts:
    env.rollout(T, 1, 0, d_traj); env.sync()
    ms = C.c_float()
    L.cz_timer_start(h)
    for r in range(steps // T):
        env.rollout(T, 1, r * T, d_traj)
    L.cz_timer_stop(h, C.byref(ms))
    print(f"N={N} T={T:2d}: {ms.value * 1e3 / steps:8.2f} us per step, {N * steps / (ms.value * 1e3):7.1f} M env-steps/s", flush=True)
