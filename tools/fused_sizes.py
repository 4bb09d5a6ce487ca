#!/usr/bin/env python3
"""Fused rollouts by batch size (bench workload, 32 steps per launch): on-device actions with a float64 trajectory (cz_rollout),
with a compact trajectory (cz_rollout_compact), and ring runs fused with the float64 rows
written in place (cz_set_ring_fused).  env-steps/s and ns per env-step.
    python3 tools/fused_sizes.py [N ...]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536]
T = 32
for N in sizes:
    env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
    env.reset(return_obs=False)
    out = []
    for compact in (False, True):
        if compact:
            buf = env.alloc((T, N, 2, env.codes_pitch), np.uint8)
            run = lambda r: env.rollout_compact(T, 1, r * T, buf)
        else:
            if T * N * 2 * env.F * 8 > (6 << 30):
                out.append(None); continue
            buf = env.alloc((T, N, 2, env.F), np.float64)
            rew, te, tr = env.alloc((T, N, 2), np.float64), env.alloc((T, N, 2), np.uint8), env.alloc((T, N, 2), np.uint8)
            run = lambda r: env.rollout(T, 1, r * T, buf, rew, te, tr)
        run(0); env.sync()
        reps = max(3, 2000 // T // max(1, N // 4096))
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for r in range(reps):
                run(r + 1)
            env.sync()
            best = min(best, (time.perf_counter() - t0) / (reps * T))
        out.append(best)
        buf.free()
    # ring runs fused, outputs in place (cz_set_ring_fused): 64-slot ring, runs of 512 steps
    period = 64
    d_ring = env.alloc((period, N, 2), np.int32)
    d_ring.from_host(np.random.default_rng(1).integers(0, 5, size=(period, N, 2), dtype=np.int32))
    o = (env.alloc((N, 2, env.F), np.float64), env.alloc((N, 2), np.float64), env.alloc((N, 2), np.uint8), env.alloc((N, 2), np.uint8))
    env.set_ring_fused(True)
    K = 512
    env.step_device_ring(K, d_ring, N * 2, period, 0, *o); env.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        env.step_device_ring(K, d_ring, N * 2, period, 0, *o)
        env.sync()
        best = min(best, (time.perf_counter() - t0) / K)
    out.append(best)
    f = lambda b: "n/a" if b is None else "%7.2f us per step = %5.2f G env-steps/s (%.3f ns per env-step)" % (b * 1e6, N / b / 1e9, b * 1e9 / N)
    print("N=%6d  float64 trajectory: %s   compact trajectory: %s   ring run fused, float64 rows in place: %s" % (N, f(out[0]), f(out[1]), f(out[2])))
    env.close()
