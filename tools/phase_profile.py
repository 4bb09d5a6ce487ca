#!/usr/bin/env python3
"""Where does a wave spend its cycles?  Needs the diagnostic build (make -C cooking_zoo_amd/csrc prof) and
CZ_LIB=cooking_zoo_amd/csrc/libcookingzoo_hip_prof.so.  Prints median s_memtime deltas per phase (shares, not times)."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CZ_LIB", os.path.join(REPO, "cooking_zoo_amd", "csrc", "libcookingzoo_hip_prof.so"))
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                    num_layouts=256, auto_reset=True)
env.reset(return_obs=False)
L, h = _native.lib(), env._h
stamps = env.alloc((N, 16), np.uint64)
L.cz_debug_set_stamps(h, stamps.ptr)
rng = np.random.default_rng(0)
d_act = env.alloc((N, 2), np.int32)
d_obs = env.alloc((N, 2, env.F), np.float64)
d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8)
d_u = env.alloc((N, 2), np.uint8)
names = ["prologue+loads", "agents", "progress", "rewards/flags", "outputs", "observe", "store"]
MODE = sys.argv[2] if len(sys.argv) > 2 else "random"
TICK_NS = 10.0                      # s_memrealtime: 100 MHz
acc = []
inner, frac = [], []
spread, endspread, evt_us = [], [], []
for it in range(60):
    d_act.from_host(rng.integers(0, 5, size=(N, 2), dtype=np.int32) if MODE == "random" else np.zeros((N, 2), np.int32))
    L.cz_timer_start(h)
    env.step_device(d_act, d_obs, d_rew, d_t, d_u)
    ms = C.c_float()
    L.cz_timer_stop(h, C.byref(ms))
    if it >= 10:
        evt_us.append(ms.value * 1e3)
    s_all = stamps.to_host().astype(np.int64)
    s = s_all[:, :8]
    if it >= 10:
        acc.append(np.diff(s, axis=1))
        # inside the reward phase (stamps 8, 9, 10 are written only by waves that got that far in THIS launch)
        ev = (s_all[:, 8] > s_all[:, 3]) & (s_all[:, 9] > s_all[:, 8]) & (s_all[:, 10] > s_all[:, 9]) & (s_all[:, 10] < s_all[:, 4])
        inner.append(np.stack([s_all[ev, 8] - s_all[ev, 3], s_all[ev, 9] - s_all[ev, 8], s_all[ev, 10] - s_all[ev, 9], s_all[ev, 4] - s_all[ev, 10],
                               s_all[ev, 2] - s_all[ev, 1]], axis=1))
        touched = (s_all[:, 8] > s_all[:, 3]) & (s_all[:, 8] < s_all[:, 4])
        frac.append((touched.mean(), ev.mean()))
        spread.append(s[:, 0].max() - s[:, 0].min())
        endspread.append(s[:, 7].max() - s[:, 0].min())
d = np.concatenate(acc)
tot = np.median((np.concatenate([a.sum(axis=1) for a in acc])))
life = np.concatenate([a.sum(axis=1) for a in acc])
print(f"[{MODE}] wave lifetime in 10 ns ticks: median {tot:.0f}  p90 {np.percentile(life,90):.0f}  p99 {np.percentile(life,99):.0f}  max-per-launch median {np.median([a.sum(axis=1).max() for a in acc]):.0f}")
st = np.concatenate([ (a[:,0]*0) for a in acc])
print(f"event-timed launch (diagnostic build, incl. stamp stores): median {np.median(evt_us):.2f} us")
print("per-launch: first-start -> last-start", np.median(spread) * TICK_NS, "ns; first-start -> last-end", np.median(endspread) * TICK_NS, "ns")
for i, n in enumerate(names):
    print(f"  {n:16s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  ({100 * d[:, i].mean() / d.sum(axis=1).mean():4.1f} %)")

slow = d[life > np.percentile(life, 99)]
print("slowest 1% of waves: mean cycles per phase")
for i, n in enumerate(names):
    print(f"  {n:16s} {slow[:, i].mean():8.0f}")

print("the slowest wave of each launch: mean ticks per phase")
worst = np.stack([a[a.sum(axis=1).argmax()] for a in acc])
for i, n in enumerate(names):
    print(f"  {n:16s} {worst[:, i].mean():8.0f}")
print("  total", worst.sum(axis=1).mean())

if inner:
    I = np.concatenate(inner)
    f = np.array(frac).mean(axis=0)
    print("waves that re-evaluate recipe graphs: %.1f %% of all (an object changed: %.1f %%); their reward phase, median ticks: filter %.0f | "
          "recipe_marks_cells %.0f | marks -> rewards %.0f | flags %.0f;  their agents phase %.0f" % (
              100 * f[1], 100 * f[0], *[np.median(I[:, k]) for k in range(4)], np.median(I[:, 4])))
