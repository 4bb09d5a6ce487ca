#!/usr/bin/env python3
"""Host-array step (cz_step: actions from host memory, observations / rewards / flags copied back) at BASELINE config 2's
size, with fresh pageable output arrays and with the env's pinned output buffers.  PCIe-inclusive: never the bench value."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
acts = rng.integers(0, 5, size=(64, N, 2), dtype=np.int32)
for pinned in (False, True):
    env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                        num_layouts=256, pinned_outputs=pinned)
    env.reset(return_obs=False)
    for i in range(10):
        env.step(acts[i])
    t0 = time.perf_counter()
    K = 200
    for i in range(K):
        obs, rew, term, trunc = env.step(acts[i % 64])
    dt = (time.perf_counter() - t0) / K
    bytes_back = obs.nbytes + rew.nbytes + term.nbytes + trunc.nbytes
    print(f"{N} envs, pinned_outputs={pinned}: {dt * 1e6:7.1f} us per host-array step  ({N / dt / 1e6:.1f} M env-steps/s, {bytes_back / dt / 1e9:.1f} GB/s of outputs)")
    for i in range(10):
        env.step_compact(acts[i])
    t0 = time.perf_counter()
    for i in range(K):
        codes, rew, term, trunc = env.step_compact(acts[i % 64])
    dt = (time.perf_counter() - t0) / K
    print(f"{N} envs, pinned_outputs={pinned}, COMPACT observation (uint8 codes, {codes.nbytes / 1e6:.1f} MB instead of {obs.nbytes / 1e6:.1f} MB): "
          f"{dt * 1e6:7.1f} us per host-array step  ({N / dt / 1e6:.1f} M env-steps/s)")
    t0 = time.perf_counter()
    for i in range(K):
        env.step(acts[i % 64], return_obs=False)
    dt = (time.perf_counter() - t0) / K
    print(f"{N} envs, pinned_outputs={pinned}, no observation copy: {dt * 1e6:7.1f} us per step")
    env.close()
