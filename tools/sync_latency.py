#!/usr/bin/env python3
"""Host <-> GPU round trip of a short run: wall time of [K launches of the step kernel + stream synchronisation] from an
idle stream, for K = 1 (direct launch), 20 (graph replay) and 20 launched directly (CZ_GRAPHS=0 in the environment).
Run it under different runtime settings (ROC_ACTIVE_WAIT_TIMEOUT, HSA_ENABLE_INTERRUPT, ...) to see what the fixed cost
of a K-step timed region is made of."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = 4096
env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
env.reset(return_obs=False)
L, h = _native.lib(), env._h
d_act = env.alloc((240, N, 2), np.int32)
d_act.from_host(np.random.default_rng(0).integers(0, 5, size=(240, N, 2), dtype=np.int32))
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
outs = (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
res = {}
for K in (1, 20, 200):
    L.cz_ring_prepare(h, K, d_act.ptr, N * 2, 240, 0, *outs)
    ts = []
    for rep in range(60):
        env.sync()
        time.sleep(0.0005)
        t0 = time.perf_counter()
        L.cz_step_device_ring(h, K, d_act.ptr, N * 2, 240, 0, *outs)
        t1 = time.perf_counter()
        env.sync()
        t2 = time.perf_counter()
        if rep >= 10:
            ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    a = np.median(np.array(ts), axis=0)
    res[K] = a
    print(f"K={K:4d}: submit {a[0]:7.1f} us   submit+sync {a[1]:7.1f} us   -> {a[1] / K:6.2f} us/step")
fixed = res[20][1] - 20 * (res[200][1] - res[20][1]) / 180
print(f"fixed cost of a 20-step region ~ {fixed:.1f} us (per-step slope {(res[200][1] - res[20][1]) / 180:.2f} us)")
