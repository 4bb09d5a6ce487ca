#!/usr/bin/env python3
"""GPU box: many short overlapped runs at the overlap limit (4096 envs of the config-2 workload: the two kernels in flight
fill the device), run lengths 2..64 with both parities, start slots all over the ring, no synchronisation between most
runs; every 25 runs the records, the last outputs and the episode statistics are compared with the oracle.
usage: python tools/overlap_stress.py [runs]"""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from cooking_zoo_amd import _native, soa  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
from oracle_binding import ShardedOracle  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n, A, period = 4096, 2, 64
    env = CookingVecEnv(n, "coop_test", "example", A, 40, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=64)
    env.set_overlap(True)
    assert n <= env.overlap_limit()
    orc = ShardedOracle(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    L = _native.lib()
    rng = np.random.default_rng(2024)
    ring_host = rng.integers(0, env.n_actions, size=(period, n, A), dtype=np.int32)
    d_ring = env.alloc((period, n, A), np.int32)
    d_ring.from_host(ring_host)
    d_obs, d_rew = env.alloc((n, A, env.F), np.float64), env.alloc((n, A), np.float64)
    d_t, d_u = env.alloc((n, A), np.uint8), env.alloc((n, A), np.uint8)
    steps, t0 = 0, time.time()
    pending = []
    for r in range(runs):
        K, first = int(rng.integers(2, 65)), int(rng.integers(period))
        _native.check(env._h, L.cz_step_device_ring(env._h, K, d_ring.ptr, n * A, period, first, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr))
        pending.append((K, first))
        steps += K
        if rng.random() < 0.3:
            env.sync()
        if (r + 1) % 25 == 0 or r == runs - 1:
            env.sync()
            for K2, f2 in pending:
                for k in range(K2):
                    oo, ro, to, uo = orc.step(ring_host[(f2 + k) % period], False)
            pending = []
            recs = env.get_state()
            recs[:, soa.RET_WORD0:soa.RET_WORD0 + 8] = 0
            assert np.array_equal(recs, orc.records), f"records differ after run {r}"
            assert np.array_equal(bits(d_rew.to_host()), bits(ro)) and np.array_equal(d_t.to_host(), to) and np.array_equal(d_u.to_host(), uo), r
            print(f"run {r + 1}: {steps} steps so far, records / rewards / flags identical to the oracle", flush=True)
    c = C.c_int64()
    L.cz_chain_counts(env._h, C.byref(c), 0)
    st = env.stats()
    assert st["episodes"] == int(orc.records[:, soa.W_EPISODE].sum()) + int((orc.records[:, soa.W_STATUS] & 1).sum())
    print(f"overlap_stress OK: {runs} runs, {steps} steps x {n} envs, {c.value} overlapped launches, {st['episodes']} episodes, {time.time() - t0:.0f} s")
    env.close()


if __name__ == "__main__":
    main()
