#!/usr/bin/env python3
"""Experiment (round 6): the bench workload's 4096 envs as 1 / 2 / 4 shards on ONE device - ShardedVecEnv(device_ids=[0] * s): one
handle, one stream, one host thread per shard - so that one stream's launch boundary is covered by the other streams' kernels.
Prints us per env step of the whole batch for ring runs of K launches (graph replay), median of R regions.

    python3 tools/exp/halves_on_streams.py [N=4096] [K=2000] [R=10]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cooking_zoo_amd import ShardedVecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
WORKLOAD = ("coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"])

for s in (1, 2, 4, 1, 2, 4):
    senv = ShardedVecEnv(N, *WORKLOAD, action_scheme="scheme3", num_layouts=256, auto_reset=True, device_ids=[0] * s, comm="host")
    senv.reset(return_obs=False)
    period = 200
    ring = senv.alloc((2,), np.int32, leading=(period,))
    ring.from_host(np.random.default_rng(1).integers(0, 5, size=(period, N, 2), dtype=np.int32))
    outs = (senv.alloc((2, senv.F), np.float64), senv.alloc((2,), np.float64), senv.alloc((2,), np.uint8), senv.alloc((2,), np.uint8))
    senv.ring_prepare(K, ring, period, 0, *outs)
    senv.step_device_ring(K, ring, period, 0, *outs); senv.sync()
    ts = []
    for _ in range(R):
        senv.sync()
        t0 = time.perf_counter()
        senv.step_device_ring(K, ring, period, 0, *outs)
        senv.sync()
        ts.append(time.perf_counter() - t0)
    us = float(np.median(ts)) * 1e6 / K
    print(f"{s} shard(s) of {N // s} envs on device 0: {us:.3f} us per step of the whole batch = {N / us:.1f} M env-steps/s "
          f"(min {min(ts) * 1e6 / K:.3f}, max {max(ts) * 1e6 / K:.3f})", flush=True)
    senv.close()
