#!/usr/bin/env python3
"""GPU box: how long the host takes to ISSUE a run of K step launches (cz_step_device_ring returns when everything is
queued) against how long the GPU takes to execute it, for graph replay, direct launches and overlapped launches."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402


def main():
    K, N = 2000, 4096
    L = _native.lib()
    for mode in ("graphs", "direct"):
        if mode == "direct":
            os.environ["CZ_GRAPHS"] = "0"
        env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
        os.environ.pop("CZ_GRAPHS", None)
        env.reset(return_obs=False)
        rng = np.random.default_rng(0)
        d_act = env.alloc((256, N, 2), np.int32)
        d_act.from_host(rng.integers(0, 5, size=(256, N, 2), dtype=np.int32))
        d_obs, d_rew = env.alloc((N, 2, env.F), np.float64), env.alloc((N, 2), np.float64)
        d_t, d_u = env.alloc((N, 2), np.uint8), env.alloc((N, 2), np.uint8)
        args = (env._h, K, d_act.ptr, N * 2, 256, 0, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
        _native.check(env._h, L.cz_step_device_ring(*args)); env.sync()
        rows = []
        for _ in range(7):
            t0 = time.perf_counter()
            _native.check(env._h, L.cz_step_device_ring(*args))
            t1 = time.perf_counter()
            env.sync()
            t2 = time.perf_counter()
            rows.append(((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
        rows.sort()
        print(f"{mode:10s}: host issues a launch in {rows[3][0]:.3f} us (min {rows[0][0]:.3f}); run complete after {sorted(r[1] for r in rows)[3]:.3f} us per launch", flush=True)
        env.close()


if __name__ == "__main__":
    main()
