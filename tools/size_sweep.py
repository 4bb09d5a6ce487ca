#!/usr/bin/env python3
"""One-launch-per-step rate of the config-2 workload against the batch size (where is the cliff, and does it follow the
observation buffer's size?): python3 tools/size_sweep.py [obs=1|0] sizes..."""
import ctypes as C, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.isdir(os.path.join(os.getcwd(), "cooking_zoo_amd")):
    REPO = os.getcwd()
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
with_obs = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
sizes = [int(v) for v in sys.argv[2:]] or [16384, 32768, 40960, 49152, 57344, 65536, 98304, 131072]
L = _native.lib()
for N in sizes:
    env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
    h = env._h
    env.reset(return_obs=False)
    d_ring = env.alloc((16, N, 2), np.int32); d_ring.from_host(np.random.default_rng(0).integers(0, 5, size=(16, N, 2), dtype=np.int32))
    d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64); d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
    outs = (d_obs.ptr if with_obs else None, d_rew.ptr, d_t.ptr, d_u.ptr)
    K = 64
    _native.check(h, L.cz_step_device_ring(h, K, d_ring.ptr, N * 2, 16, 0, *outs)); env.sync()
    best = 1e9
    ms = C.c_float()
    for rep in range(3):
        L.cz_timer_start(h)
        _native.check(h, L.cz_step_device_ring(h, K, d_ring.ptr, N * 2, 16, 0, *outs))
        L.cz_timer_stop(h, C.byref(ms)); best = min(best, ms.value)
    us = best * 1e3 / K
    print(f"N={N:7d} obs={int(with_obs)} obs buffer {N * 2 * env.F * 8 / 2**20:7.1f} MiB: {us:8.2f} us per launch, {N / us:7.1f} M env-steps/s, {us * 1e3 / N:6.3f} ns per env", flush=True)
    env.close()
