#!/usr/bin/env python3
"""Launch duration of the one-step kernel under different action streams (boundary-ordered launches, HIP events around
400 back-to-back launches): stay (nobody acts: every wave runs the same path), random (the bench workload),
rich (random actions, episodes never end: after 3000 warm-up steps most agents carry something and the worlds are full of
chopped / plated objects)."""
import ctypes as C, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.isdir(os.path.join(os.getcwd(), "cooking_zoo_amd")):
    REPO = os.getcwd()            # run from another tree (tools/ab_trees.sh style A/B)
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = _native.lib()
rng = np.random.default_rng(0)
for mode in ("stay", "random", "rich"):
    env = CookingVecEnv(N, "coop_test", "example", 2, (1 << 30) if mode == "rich" else 400, ["TomatoLettuceSalad", "CarrotBanana"],
                        action_scheme="scheme3", num_layouts=256, auto_reset=True)
    h = env._h
    env.reset(return_obs=False)
    P = 64
    acts = rng.integers(0, 5, size=(P, N, 2), dtype=np.int32)
    if mode == "stay":
        acts[:] = 0
    d_act = env.alloc((P, N, 2), np.int32); d_act.from_host(acts)
    d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
    d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
    outs = (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
    warm = 3000 if mode == "rich" else 300
    _native.check(h, L.cz_step_device_ring(h, warm, d_act.ptr, N * 2, P, 0, *outs))
    env.sync()
    res = []
    for rep in range(5):
        ms = C.c_float()
        L.cz_timer_start(h)
        _native.check(h, L.cz_step_device_ring(h, 400, d_act.ptr, N * 2, P, 0, *outs))
        L.cz_timer_stop(h, C.byref(ms))
        res.append(ms.value * 1e3 / 400)
    print(f"{mode:7s} N={N}: {min(res):.3f} us per launch (median {sorted(res)[2]:.3f})")
    env.close()
