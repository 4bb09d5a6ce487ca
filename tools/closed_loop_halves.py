#!/usr/bin/env python3
"""A closed loop over part-batches that take turns: S handles of N / S envs, each running its own closed loop (step kernel ->
policy kernel -> step kernel ..., cz_probe_closed_loop: a 200-step HIP graph replayed `reps` times) on its own stream, all at
the same time (one host thread per handle; ctypes releases the GIL).  While one part's policy kernel and launch boundaries
pass, another part's step kernel has the device.  Prints wall-clock microseconds per step of ALL parts together.
    python3 tools/closed_loop_halves.py [N=4096]"""
import ctypes as C, os, sys, threading, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = _native.lib()
K, REPS = 200, 20
for compact in (False, True):
    for S in (1, 2, 3, 4):
        n = (N // S) // 8 * 8
        parts = []
        for s in range(S):
            env = CookingVecEnv(n, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
            env.reset(return_obs=False)
            d_act = env.alloc((n, 2), np.int32); d_act.from_host(np.random.default_rng(s).integers(0, 5, size=(n, 2), dtype=np.int32))
            d_o = env.alloc((n, 2, env.codes_pitch), np.uint8) if compact else env.alloc((n, 2, env.F), np.float64)
            bufs = (env.alloc((n, 2), np.float64), env.alloc((n, 2), np.uint8), env.alloc((n, 2), np.uint8))
            parts.append((env, d_act, d_o, bufs))
        fn = L.cz_probe_closed_loop_compact if compact else L.cz_probe_closed_loop
        res = [0.0] * S

        def run(i, reps):
            env, d_act, d_o, bufs = parts[i]
            us = C.c_float()
            _native.check(env._h, fn(env._h, K, reps, d_act.ptr, d_o.ptr, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, C.byref(us)))
            res[i] = us.value

        for i in range(S):
            run(i, 2)                                     # graph capture + warm-up, one after the other
        best = 1e9
        for rep in range(3):
            th = [threading.Thread(target=run, args=(i, REPS)) for i in range(S)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            best = min(best, (time.perf_counter() - t0) * 1e6 / (K * REPS))
        print("%s  S=%d x %d envs: %.2f us per closed-loop step of all %d envs (wall clock) -> %.0f M env-steps/s; each part alone reports %s us" % (
            "codes  " if compact else "float64", S, n, best, S * n, S * n / best, ", ".join("%.2f" % r for r in res)))
        for env, *_ in parts:
            env.close()
