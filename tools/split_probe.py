#!/usr/bin/env python3
"""GPU box experiment: the 4096-env batch of config 2 stepped as S independent sub-batches on S streams.

Envs are independent, so a batch may be cut into sub-batches whose step chains overlap: while one sub-batch's kernel
drains and the next is dispatched (the ~1.6 us launch boundary), the other sub-batches' kernels keep the CUs busy.
Prints the wall time of K steps of the WHOLE batch for S = 1, 2, 4 (same total work, same results).
usage: python tools/split_probe.py [K]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = 4096


def build(S, K):
    L = _native.lib()
    parts = []
    period = min(K, 256)
    for s in range(S):
        n = N // S
        env = CookingVecEnv(n, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                            num_layouts=256, layout_seed=0, auto_reset=True, env_id_base=s * n)
        env.reset(return_obs=False)
        rng = np.random.default_rng(1234 + s)
        d_act = env.alloc((period, n, 2), np.int32)
        d_act.from_host(rng.integers(0, 5, size=(period, n, 2), dtype=np.int32))
        d_obs = env.alloc((n, 2, env.F), np.float64)
        d_rew = env.alloc((n, 2), np.float64)
        d_t = env.alloc((n, 2), np.uint8)
        d_u = env.alloc((n, 2), np.uint8)
        ring = (d_act.ptr, n * 2, period)
        outs = (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
        _native.check(env._h, L.cz_ring_prepare(env._h, K, *ring, 0, *outs))
        parts.append((env, ring, outs))
    return parts


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    L = _native.lib()
    for S in (1, 2, 4, 1, 2, 4):
        parts = build(S, K)
        def run():
            for env, ring, outs in parts:
                _native.check(env._h, L.cz_step_device_ring(env._h, K, *ring, 0, *outs))
            for env, _, _ in parts:
                env.sync()
        run()
        ts = []
        for _ in range(7):
            for env, _, _ in parts:
                env.sync()
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[len(ts) // 2]
        print(f"S={S}: {med / K * 1e6:7.3f} us per whole-batch step  ({N * K / med / 1e6:7.1f} M env-steps/s)  min {min(ts) / K * 1e6:.3f}", flush=True)
        for env, _, _ in parts:
            env.close()


if __name__ == "__main__":
    main()
