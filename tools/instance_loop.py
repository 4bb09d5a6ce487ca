#!/usr/bin/env python3
"""Profiling driver: a handful of launches of ONE launch form of the step path on the bench workload (4096 envs, BASELINE config 2),
so that a rocprofv3 --pmc pass sees exactly one kernel instance with a known number of env steps per launch.

    python3 tools/instance_loop.py FORM [ENVS]
    FORM: step | step_compact | rollout | rollout_actions | rollout_compact | ring_in_place | cooking
Prints `FORM instance=<key of issue_per_env_step.json> steps_per_launch=<T>` for tools/pmc_instances.py."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

form = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
A, T, P = 2, 32, 64
if form == "cooking":
    import bench
    env, ring, _, _ = bench.cooking_policy_workload(0, 256)
    P = 256
else:
    env = CookingVecEnv(N, "coop_test", "example", A, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
    env.reset(return_obs=False)
    ring = np.random.default_rng(0).integers(0, 5, size=(P, N, A), dtype=np.int32)
d_ring = env.alloc((P, N, A), np.int32)
d_ring.from_host(ring)
d_obs, d_rew = env.alloc((N, A, env.F), np.float64), env.alloc((N, A), np.float64)
d_t, d_u = env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
os.environ.setdefault("CZ_GRAPHS", "0")
steps = 1
if form in ("step", "cooking"):
    env.rollout(300, 3, 0) if form == "step" else None            # (worlds a few hundred steps old, like the bench's timed regions)
    for k in range(P):
        env.step_device(d_ring.ptr + k * N * A * 4, d_obs, d_rew, d_t, d_u)
    inst = "k_step<1,1,2,3,0>" + ("/cooking" if form == "cooking" else "")
elif form == "step_compact":
    d_codes = env.alloc((N, A, env.codes_pitch), np.uint8)
    env.rollout(300, 3, 0)
    for k in range(P):
        env.step_device_compact(d_ring.ptr + k * N * A * 4, d_codes, d_rew, d_t, d_u)
    inst = "k_step<1,1,2,3,3>"
elif form == "rollout":
    d_traj = env.alloc((T, N, A, env.F), np.float64)
    for r in range(12):
        env.rollout(T, 1, r * T, d_traj)
    inst, steps = "k_step<1,1,2,3,1>", T
elif form == "rollout_actions":
    d_traj = env.alloc((T, N, A, env.F), np.float64)
    for r in range(12):
        env.rollout_actions(d_ring, T, d_traj)
    inst, steps = "k_step<1,1,2,3,2>", T
elif form == "rollout_compact":
    d_codes = env.alloc((T, N, A, env.codes_pitch), np.uint8)
    for r in range(12):
        env.rollout_compact(T, 1, r * T, d_codes)
    inst, steps = "k_step<1,1,2,3,5>", T
elif form == "ring_in_place":
    env.set_ring_fused(True)
    for r in range(12):
        env.step_device_ring(P, d_ring, N * A, P, 0, d_obs, d_rew, d_t, d_u)
    inst, steps = "k_step<1,1,2,3,2>/in_place", P
else:
    raise SystemExit(f"unknown form {form}")
env.sync()
print(f"{form} instance={inst} steps_per_launch={steps}")
env.close()
