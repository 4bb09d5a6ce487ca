#!/usr/bin/env python3
"""Phase stamps of the one-step kernel under the heuristic agent's action stream (bench.py `cooking_policy`): which waves are
the slowest of a launch - auto-reset passes, waves with interactions, waves that re-evaluate a recipe graph?  Needs the
diagnostic build (make -C cooking_zoo_amd/csrc prof).  python3 tools/cook_profile.py [launches]"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.isdir(os.path.join(os.getcwd(), "cooking_zoo_amd")):
    REPO = os.getcwd()
sys.path.insert(0, REPO)
os.environ.setdefault("CZ_LIB", os.path.join(REPO, "cooking_zoo_amd", "csrc", "libcookingzoo_hip_prof.so"))
import bench  # noqa: E402
from cooking_zoo_amd import _native, soa  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
env, ring, P, T = bench.cooking_policy_workload(0, K)
L, h, N, A = _native.lib(), env._h, env.num_envs, env.num_agents
stamps = env.alloc((N, 16), np.uint64)
zeros = np.zeros((N, 16), np.uint64)
L.cz_debug_set_stamps(h, stamps.ptr)
d_act = env.alloc((N, A), np.int32)
d_obs, d_rew = env.alloc((N, A, env.F), np.float64), env.alloc((N, A), np.float64)
d_t, d_u = env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
names = ["prologue+loads", "agents", "progress", "rewards/flags", "outputs", "observe", "store"]
rows = {"reset pass": [], "step": []}
worst_kind, worst_rows, evt, sub = [], [], [], []
for k in range(K):
    was_done = (env.get_state()[:, soa.W_STATUS] & 1).astype(bool)          # these envs take their auto-reset pass now
    d_act.from_host(np.ascontiguousarray(ring[k]))
    stamps.from_host(zeros)
    ms = C.c_float()
    L.cz_timer_start(h)
    env.step_device(d_act, d_obs, d_rew, d_t, d_u)
    L.cz_timer_stop(h, C.byref(ms))
    s16 = stamps.to_host().astype(np.int64)
    s = s16[:, :8]
    if k < 20:
        continue
    # inside the reward phase (only waves in which an object moved or changed write these): 3 -> 8 filter, 8 -> 9 evaluation
    # of the recipe graphs (0 when the filter says no), 9 -> 10 rewards, 10 -> 4 flags
    tw = (s16[:, 8] > 0) & ~was_done
    s9 = np.where(s16[:, 9] > 0, s16[:, 9], s16[:, 8])             # (no evaluation: stamp 9 is not written)
    sub.append(np.stack([s16[tw, 8] - s16[tw, 3], s9[tw] - s16[tw, 8], s16[tw, 10] - s9[tw], s16[tw, 4] - s16[tw, 10], s16[tw, 4] - s16[tw, 3]], axis=1))
    evt.append(ms.value * 1e3)
    life = s[:, 7] - s[:, 0]
    d = np.diff(s, axis=1)
    # a reset pass leaves stamps 2 and 3 untouched (step_env returns before them): take its phases as 0->1, 1->4, 4->...
    dr = d.copy()
    dr[was_done, 1] = s[was_done, 4] - s[was_done, 1]
    dr[was_done, 2] = 0
    dr[was_done, 3] = 0
    rows["reset pass"].append(dr[was_done])
    rows["step"].append(dr[~was_done])
    w = int(life.argmax())
    worst_kind.append("reset pass" if was_done[w] else "step")
    worst_rows.append(np.concatenate([dr[w], [life[w]]]))
print(f"{K - 20} launches, {N} envs; event-timed launch (diagnostic build): median {np.median(evt):.2f} us")
for kind in ("step", "reset pass"):
    a = np.concatenate(rows[kind])
    lf = a.sum(axis=1)
    print(f"{kind}: {len(a) / (K - 20):.1f} waves per launch; lifetime (10 ns ticks) median {np.median(lf):.0f} p90 {np.percentile(lf, 90):.0f} "
          f"p99 {np.percentile(lf, 99):.0f} max {lf.max():.0f}")
    for i, n in enumerate(names):
        print(f"  {n:16s} median {np.median(a[:, i]):6.0f}  p99 {np.percentile(a[:, i], 99):6.0f}" + ("   (reset pass: layout fetch + all marks)" if kind == "reset pass" and i == 1 else ""))
sb = np.concatenate(sub)
slow = sb[sb[:, 4] >= np.percentile(sb[:, 4], 95)]
print(f"reward phase of the waves with a moved / changed object ({len(sb) / (K - 20):.0f} per launch): median / slowest 5 % (10 ns ticks)")
for i, n in enumerate(["filter", "recipe graphs", "rewards", "flags", "whole phase"]):
    print(f"  {n:14s} {np.median(sb[:, i]):6.0f} {slow[:, i].mean():8.0f}")
wk = np.array(worst_kind)
wr = np.stack(worst_rows)
print(f"the slowest wave of a launch is a reset pass in {100 * (wk == 'reset pass').mean():.0f} % of the launches")
for kind in ("step", "reset pass"):
    m = wk == kind
    if m.any():
        print(f"  slowest = {kind}: mean ticks per phase " + ", ".join(f"{n} {wr[m][:, i].mean():.0f}" for i, n in enumerate(names)) + f"; lifetime {wr[m][:, 7].mean():.0f}")
env.close()
