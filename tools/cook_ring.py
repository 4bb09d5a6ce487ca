#!/usr/bin/env python3
"""The cooking-policy workload of bench.py (every env replays the reference heuristic agent's actions) as a plain program for
counter passes: `rocprofv3 --pmc ... -- python3 tools/cook_ring.py [random]` (boundary-ordered launches; `random` = the same
batch under uniform random actions, for the difference)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["CZ_GRAPHS"] = "0"
import bench  # noqa: E402
from cooking_zoo_amd import _native  # noqa: E402

K = 512
env, ring, P, T = bench.cooking_policy_workload(0, K)
if len(sys.argv) > 1 and sys.argv[1] == "random":
    ring = np.random.default_rng(0).integers(0, 5, size=ring.shape, dtype=np.int32)
L, h, N, A = _native.lib(), env._h, env.num_envs, env.num_agents
d_ring = env.alloc((K, N, A), np.int32)
d_ring.from_host(ring)
outs = (env.alloc((N, A, env.F), np.float64).ptr, env.alloc((N, A), np.float64).ptr, env.alloc((N, A), np.uint8).ptr, env.alloc((N, A), np.uint8).ptr)
_native.check(h, L.cz_step_device_ring(h, K, d_ring.ptr, N * A, K, 0, *outs))
env.sync()
print("done", env.stats())
