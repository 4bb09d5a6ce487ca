import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from cooking_zoo_amd import _native
from cooking_zoo_amd.vec_env import CookingVecEnv
L = _native.lib()
for N in (65536, 131072):
    env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
    h = env._h
    env.reset(return_obs=False)
    d_act = env.alloc((16, N, 2), np.int32); d_act.from_host(np.random.default_rng(0).integers(0, 5, size=(16, N, 2), dtype=np.int32))
    S = 16
    d_obs = env.alloc((S, N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64); d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
    stride = N * 2 * env.F * 8
    for mode in ("same buffer", "arena of 16 buffers"):
        for rep in range(2):
            ms = C.c_float()
            L.cz_timer_start(h)
            for k in range(64):
                off = (k % S) * stride if mode.startswith("arena") else 0
                L.cz_step_device(h, d_act.ptr + (k % 16) * N * 8, d_obs.ptr + off, d_rew.ptr, d_t.ptr, d_u.ptr)
            L.cz_timer_stop(h, C.byref(ms))
        print(f"N={N} {mode}: {ms.value * 1e3 / 64:.2f} us per launch", flush=True)
    env.close()
