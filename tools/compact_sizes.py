#!/usr/bin/env python3
"""One launch per step (ring, graph replay, boundary-ordered) with float64 rows / compact codes only / no observation, by batch
size: where the observation's BYTES bound the launch (large batches) the codes buy what the roofline says; at 4096 envs the
launch is bound by the latency chain of the encode, not by its output bytes."""
import ctypes as C, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
L = _native.lib()
sizes = [int(a) for a in sys.argv[1:]] or [4096, 16384, 65536]
for N in sizes:
    env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
    h = env._h
    env.reset(return_obs=False)
    P, A = 32, 2
    d_ring = env.alloc((P, N, A), np.int32); d_ring.from_host(np.random.default_rng(0).integers(0, 5, size=(P, N, A), dtype=np.int32))
    d_obs, d_codes = env.alloc((N, A, env.F), np.float64), env.alloc((N, A, env.codes_pitch), np.uint8)
    d_rew, d_t, d_u = env.alloc((N, A), np.float64), env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
    K = max(64, min(2000, (1 << 23) // N))
    ms = C.c_float()
    res = {}
    for name, obs, codes in (("float64 rows", d_obs.ptr, None), ("codes only", None, d_codes), ("no observation", None, None), ("float64 + codes", d_obs.ptr, d_codes)):
        env.set_compact_output(codes)
        outs = (obs, d_rew.ptr, d_t.ptr, d_u.ptr)
        _native.check(h, L.cz_ring_prepare(h, K, d_ring.ptr, N * A, P, 0, *outs))
        _native.check(h, L.cz_step_device_ring(h, P, d_ring.ptr, N * A, P, 0, *outs))
        best = 1e9
        for rep in range(3):
            L.cz_timer_start(h)
            _native.check(h, L.cz_step_device_ring(h, K, d_ring.ptr, N * A, P, 0, *outs))
            L.cz_timer_stop(h, C.byref(ms))
            best = min(best, ms.value * 1e3 / K)
        res[name] = best
    print("N=%6d: " % N + "  ".join("%s %.2f us (%.3f ns/env)" % (k, v, v * 1e3 / N) for k, v in res.items()))
    env.close()
