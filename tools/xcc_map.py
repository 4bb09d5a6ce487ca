#!/usr/bin/env python3
"""Which XCD does workgroup w of launch k run on?  (timeline build, see tools/timeline.py)  python3 tools/xcc_map.py [overlap=1]"""
import ctypes as C, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CZ_LIB", os.path.join(REPO, "cooking_zoo_amd", "csrc", "libcookingzoo_hip_tl.so"))
os.environ["CZ_GRAPHS"] = "0"
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
overlap = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
N, K, P = 4096, 400, 64
L = _native.lib()
env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
h = env._h
env.reset(return_obs=False)
acts = np.random.default_rng(0).integers(0, 5, size=(P, N, 2), dtype=np.int32)
d_act = env.alloc((P, N, 2), np.int32); d_act.from_host(acts)
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
if overlap:
    env.set_overlap(True)
tl = env.alloc((K, N, 2), np.uint64)
_native.check(h, L.cz_debug_set_timeline(h, tl.ptr, K))
_native.check(h, L.cz_step_device_ring(h, K, d_act.ptr, N * 2, P, 0, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr))
env.sync()
t = tl.to_host()
xcc = ((t[:, :, 1] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
wg = xcc[:, ::8]                                   # one value per workgroup (its eight waves share an XCD)
print("waves of a workgroup on one XCD:", bool((xcc.reshape(K, N // 8, 8) == wg[:, :, None]).all()))
for k in list(range(100, 108)):
    print("launch", k, "workgroups 0..23 ->", " ".join(str(v) for v in wg[k, :24]))
rot = (wg - wg[:, 0:1]) % 8
print("xcc(w) - xcc(0) mod 8 identical in every launch:", bool((rot == rot[0:1]).all()), " = w mod 8:", bool((rot == (np.arange(N // 8) % 8)[None, :]).all()))
print("xcc of workgroup 0 over launches 100..131:", " ".join(str(v) for v in wg[100:132, 0]))
for par in (0, 1):
    v = wg[100 + par:K:2, 0]
    print("  launches of parity", par, ": distinct xcc of workgroup 0:", sorted(set(int(x) for x in v)))
