#!/usr/bin/env python3
"""What would S independent part-batches on S streams give?  S handles of N / S envs each run K boundary-ordered steps
(graph replay, no overlapped mode) at the same time; the host issues all of them, then waits for all.
    python3 tools/split_streams.py [N=4096] [K=2048]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
L = _native.lib()
P = 64
for S in (1, 2, 4, 8):
    n = N // S
    parts = []
    for s in range(S):
        env = CookingVecEnv(n, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                            num_layouts=256, auto_reset=True, env_id_base=s * n) if "env_id_base" in CookingVecEnv.__init__.__code__.co_varnames else \
            CookingVecEnv(n, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
        env.reset(return_obs=False)
        acts = np.random.default_rng(s).integers(0, 5, size=(P, n, 2), dtype=np.int32)
        d_act = env.alloc((P, n, 2), np.int32); d_act.from_host(acts)
        bufs = (env.alloc((n, 2, env.F), np.float64), env.alloc((n, 2), np.float64), env.alloc((n, 2), np.uint8), env.alloc((n, 2), np.uint8))
        parts.append((env, d_act, bufs))

    def run(k):
        for env, d_act, bufs in parts:
            _native.check(env._h, L.cz_step_device_ring(env._h, k, d_act.ptr, n * 2, P, 0, *(b.ptr for b in bufs)))
    run(256)
    for env, _, _ in parts:
        env.sync()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        run(K)
        for env, _, _ in parts:
            env.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"S={S}: {S} x {n} envs, {K} steps: {best * 1e6 / K:.3f} us per step of all parts -> {N * K / best / 1e6:.0f} M env-steps/s")
    for env, _, _ in parts:
        env.close()
