#!/usr/bin/env python3
"""GPU box, driver of tools/phase_cut.sh: launches of the one-step kernel from a WARM batch state with the library CZ_LIB names.

    python3 tools/phase_cut_run.py warm STATE.npy        # the full (marker) library: 300 steps from reset, saves the records
    python3 tools/phase_cut_run.py run STATE.npy [N]     # any library (also a cut one): set_state, then N launches over ring actions
A cut library never stores its records, so every one of its launches starts from the same warm state with different actions."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["CZ_GRAPHS"] = "0"
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 400
N, A, P = 4096, 2, 64
# (the ABI check compares the library with the header: the marker / cut libraries are built from this tree)
env = CookingVecEnv(N, "coop_test", "example", A, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
env.reset(return_obs=False)
if mode == "warm":
    env.rollout(300, 3, 0)
    env.sync()
    np.save(path, env.get_state())
    sys.exit(0)
env.set_state(np.load(path))
d_ring = env.alloc((P, N, A), np.int32)
d_ring.from_host(np.random.default_rng(0).integers(0, 5, size=(P, N, A), dtype=np.int32))
d_obs, d_rew = env.alloc((N, A, env.F), np.float64), env.alloc((N, A), np.float64)
d_t, d_u = env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
L, h = _native.lib(), env._h
for k in range(n):
    L.cz_step_device(h, d_ring.ptr + (k % P) * N * A * 4, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
env.sync()
env.close()
