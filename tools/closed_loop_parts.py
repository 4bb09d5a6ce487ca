#!/usr/bin/env python3
"""Where a closed-loop step's time goes (4096 envs, bench workload): the step kernel alone writing float64 rows / codes only /
nothing, issued back to back from Python (direct launches), and the two closed loops of bench.py."""
import ctypes as C, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = _native.lib()
env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
h = env._h
env.reset(return_obs=False)
A, F, Fp = 2, env.F, env.codes_pitch
d_act = env.alloc((N, A), np.int32); d_act.from_host(np.random.default_rng(0).integers(0, 5, size=(N, A), dtype=np.int32))
d_obs, d_codes = env.alloc((N, A, F), np.float64), env.alloc((N, A, Fp), np.uint8)
d_rew, d_t, d_u = env.alloc((N, A), np.float64), env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
ms = C.c_float()


def timed(fn, n=3000):
    for _ in range(200):
        fn()
    env.sync()
    best = 1e9
    for rep in range(3):
        L.cz_timer_start(h)
        for _ in range(n):
            fn()
        L.cz_timer_stop(h, C.byref(ms))
        best = min(best, ms.value * 1e3 / n)
    return best


print("step, float64 rows      : %.3f us per launch (direct launches from Python)" % timed(lambda: L.cz_step_device(h, d_act.ptr, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)))
print("step, codes only        : %.3f us" % timed(lambda: L.cz_step_device_compact(h, d_act.ptr, d_codes.ptr, None, d_rew.ptr, d_t.ptr, d_u.ptr)))
print("step, codes + float64   : %.3f us" % timed(lambda: L.cz_step_device_compact(h, d_act.ptr, d_codes.ptr, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)))
print("step, no observation    : %.3f us" % timed(lambda: L.cz_step_device(h, d_act.ptr, None, d_rew.ptr, d_t.ptr, d_u.ptr)))
us = C.c_float()
os.environ["CZ_PROBE_NO_POLICY"] = "1"
for rep in range(2):
    _native.check(h, L.cz_probe_closed_loop(h, 200, 10, d_act.ptr, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr, C.byref(us)))
    a = us.value
    _native.check(h, L.cz_probe_closed_loop_compact(h, 200, 10, d_act.ptr, d_codes.ptr, d_rew.ptr, d_t.ptr, d_u.ptr, C.byref(us)))
    b = us.value
    _native.check(h, L.cz_probe_closed_loop(h, 200, 10, d_act.ptr, None, d_rew.ptr, d_t.ptr, d_u.ptr, C.byref(us))) if False else None
    print("graph of 200 step kernels WITHOUT the policy kernel: float64 rows %.3f us per step, codes only %.3f us per step" % (a, b))
del os.environ["CZ_PROBE_NO_POLICY"]
for rep in range(2):
    _native.check(h, L.cz_probe_closed_loop(h, 200, 10, d_act.ptr, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr, C.byref(us)))
    a = us.value
    _native.check(h, L.cz_probe_closed_loop_compact(h, 200, 10, d_act.ptr, d_codes.ptr, d_rew.ptr, d_t.ptr, d_u.ptr, C.byref(us)))
    print("closed loop: float64 %.3f us per step, codes %.3f us per step" % (a, us.value))
env.close()
