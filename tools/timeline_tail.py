#!/usr/bin/env python3
"""Which waves are the last ones of a launch?  (timeline build, see tools/timeline.py; launch-boundary ordering)
    python3 tools/timeline_tail.py [N=4096] [K=2000] [mode=random|stay]"""
import ctypes as C, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CZ_LIB", os.path.join(REPO, "cooking_zoo_amd", "csrc", "libcookingzoo_hip_tl.so"))
os.environ["CZ_GRAPHS"] = "0"
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
MODE = sys.argv[3] if len(sys.argv) > 3 else "random"
P = 64
L = _native.lib()
env = CookingVecEnv(N, "coop_test", "example", 2, 1 << 30, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
h = env._h
env.reset(return_obs=False)
acts = np.random.default_rng(0).integers(0, 5, size=(P, N, 2), dtype=np.int32)
if MODE == "stay":
    acts[:] = 0
d_act = env.alloc((P, N, 2), np.int32); d_act.from_host(acts)
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
outs = (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
_native.check(h, L.cz_step_device_ring(h, 300, d_act.ptr, N * 2, P, 0, *outs)); env.sync()
tl = env.alloc((K, N, 2), np.uint64)
_native.check(h, L.cz_debug_set_timeline(h, tl.ptr, K))
_native.check(h, L.cz_step_device_ring(h, K, d_act.ptr, N * 2, P, 0, *outs)); env.sync()
t = tl.to_host()[50:]
t_in = (t[:, :, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64); t_out = (t[:, :, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
hw = ((t[:, :, 0] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64); dbg = (t[:, :, 0] >> np.uint64(48)).astype(np.int64); xcc = ((t[:, :, 1] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
# HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
first = t_in.min(axis=1, keepdims=True)
start, life, end = (t_in - first) * 0.01, (t_out - t_in) * 0.01, (t_out - first) * 0.01
print("mode %s, %d launches of %d envs; per launch: last wave out at %.2f us (median), wave lifetime median %.2f p99 %.2f" % (
    MODE, t.shape[0], N, np.median(end.max(axis=1)), np.median(life), np.percentile(life, 99)))
last = end.argmax(axis=1)
rows = np.arange(t.shape[0])
print("the LAST wave of a launch: started at %.2f us (median; all waves: %.2f), lived %.2f us (median)" % (
    np.median(start[rows, last]), np.median(start), np.median(life[rows, last])))
print("  its workgroup index / %d: quartiles %s" % (N // 8, np.percentile(last // 8, [25, 50, 75]).astype(int).tolist()))
print("  its XCD: counts %s" % np.bincount(xcc[rows, last], minlength=8).tolist())
print("  its SE: counts %s" % np.bincount(se[rows, last], minlength=8).tolist())
print("mean lifetime by XCD: %s" % [round(float(life[xcc == x].mean()), 2) for x in range(8)])
print("mean lifetime by start decile: %s" % [round(float(life[(start >= a) & (start < b)].mean()), 2) for a, b in zip(np.percentile(start, range(0, 100, 10)), list(np.percentile(start, range(10, 100, 10))) + [1e9])])
print("mean lifetime by wave-in-workgroup: %s" % [round(float(life[:, w::8].mean()), 2) for w in range(8)])
# lifetime against what the env did: how many of its agents' actions were not 'stay' is all the host knows cheaply
k_idx = (300 + 50 + np.arange(t.shape[0])) % P
nact = (acts[k_idx] != 0).sum(axis=2)
print("mean lifetime by number of acting agents: %s" % [round(float(life[nact == n].mean()), 2) if (nact == n).any() else None for n in range(3)])
# how much later than the median wave do the last waves end, and how many waves end in the last 0.5 us of a launch
tail = end.max(axis=1, keepdims=True) - end
print("waves ending within 0.25 / 0.5 / 1.0 us of the launch's last: %.1f / %.1f / %.1f (mean count per launch)" % (
    (tail < 0.25).sum(axis=1).mean(), (tail < 0.5).sum(axis=1).mean(), (tail < 1.0).sum(axis=1).mean()))
print("lifetime percentiles 50/90/99/99.9/max-per-launch-median: %s" % [round(float(x), 2) for x in list(np.percentile(life, [50, 90, 99, 99.9])) + [np.median(life.max(axis=1))]])

# what the waves did (timeline build: 1 an object moved or changed, 2 recipe graphs re-evaluated, 4 marks changed, 8 somebody interacted)
for name, bit in (("an object moved or changed", 1), ("re-evaluated its recipe graphs", 2), ("somebody interacted", 8)):
    m = (dbg & bit) != 0
    print("waves in which %-32s %5.2f %% of all, mean lifetime %.2f us (others %.2f); the LAST wave of a launch is one of them in %4.1f %% of the launches" % (
        name + ":", 100 * m.mean(), life[m].mean() if m.any() else float("nan"), life[~m].mean(), 100 * m[rows, last].mean()))
for k, lab in ((1, "last"), (5, "5 last"), (20, "20 last")):
    idx = np.argsort(end, axis=1)[:, -k:]
    ev = np.take_along_axis((dbg & 2) != 0, idx, axis=1)
    print("  of the %s waves of a launch %.1f %% re-evaluated recipe graphs" % (lab, 100 * ev.mean()))
