#!/usr/bin/env python3
"""Device-clock timeline of the step kernel's launches.  Needs `make -C cooking_zoo_amd/csrc timeline`
(libcookingzoo_hip_tl.so: the shipped kernels + an entry and an exit stamp per wave on the device-wide 100 MHz clock, no
tracer, no added waits).  Runs K launches of the bench workload in one cz_step_device_ring call, direct launches
(CZ_GRAPHS=0), either ordered by launch boundaries or overlapped (two streams, per-env sequence words), and reports per
launch: first wave in, last wave in, last wave out, and the start-to-start / end-to-end interval to the next launch.

    python3 tools/timeline.py [N=4096] [K=4000] [overlap=0|1]
"""
import ctypes as C, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CZ_LIB", os.path.join(REPO, "cooking_zoo_amd", "csrc", "libcookingzoo_hip_tl.so"))
os.environ["CZ_GRAPHS"] = "0"
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
overlap = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
MODE = sys.argv[4] if len(sys.argv) > 4 else "random"          # random | stay (nobody acts)
MAX_STEPS = int(sys.argv[5]) if len(sys.argv) > 5 else 400     # 1073741824: episodes never end (no reset passes)
WITH_OBS = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True  # 0: no observation encode at all (ablation)
TICK_NS = 10.0
L = _native.lib()
env = CookingVecEnv(N, "coop_test", "example", 2, MAX_STEPS, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                    num_layouts=256, auto_reset=True)
h = env._h
env.reset(return_obs=False)
P = 64
rng = np.random.default_rng(0)
acts = rng.integers(0, 5, size=(P, N, 2), dtype=np.int32)
if MODE == "stay":
    acts[:] = 0
d_act = env.alloc((P, N, 2), np.int32); d_act.from_host(acts)
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
outs = (d_obs.ptr if WITH_OBS else None, d_rew.ptr, d_t.ptr, d_u.ptr)
if overlap:
    raise SystemExit("overlapped launches were removed in round 5 (profiles/r04 holds their timelines)")
_native.check(h, L.cz_step_device_ring(h, 3000 if MAX_STEPS > 100000 else 300, d_act.ptr, N * 2, P, 0, *outs))      # warm-up, not recorded
env.sync()
tl = env.alloc((K, N, 2), np.uint64)
_native.check(h, L.cz_debug_set_timeline(h, tl.ptr, K))
ms = C.c_float()
L.cz_timer_start(h)
_native.check(h, L.cz_step_device_ring(h, K, d_act.ptr, N * 2, P, 0, *outs))
L.cz_timer_stop(h, C.byref(ms))
env.sync()
t = tl.to_host()
t_in = (t[:, :, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t_out = (t[:, :, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = ((t[:, :, 1] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
first_in, last_in, last_out = t_in.min(axis=1), t_in.max(axis=1), t_out.max(axis=1)
life = (t_out - t_in)
sel = slice(50, K - 1)                     # the first launches of a run start from an idle device


def us(x):
    return float(x) * TICK_NS / 1e3


res = {
    "what": "device-clock (s_memrealtime, 100 MHz) timeline of %d consecutive launches of the one-step kernel, %d envs, actions: %s, max_steps %d, obs %d, %s" % (
        K, N, MODE, MAX_STEPS, WITH_OBS, "OVERLAPPED launches (two streams, per-env sequence words)" if overlap else "launch-boundary ordering, direct launches"),
    "hip_events_us_per_launch": ms.value * 1e3 / K,
    "start_to_start_us": {"median": us(np.median(np.diff(first_in)[sel])), "mean": us(np.diff(first_in)[sel].mean()),
                          "p10": us(np.percentile(np.diff(first_in)[sel], 10)), "p90": us(np.percentile(np.diff(first_in)[sel], 90))},
    "end_to_end_us": {"median": us(np.median(np.diff(last_out)[sel])), "mean": us(np.diff(last_out)[sel].mean())},
    "whole_run_us_per_launch": us(last_out[-1] - first_in[50]) / (K - 50),
    "first_in_to_last_in_us": {"median": us(np.median((last_in - first_in)[sel]))},
    "first_in_to_last_out_us": {"median": us(np.median((last_out - first_in)[sel])), "p90": us(np.percentile((last_out - first_in)[sel], 90))},
    "gap_last_out_to_next_first_in_us": {"median": us(np.median((first_in[1:] - last_out[:-1])[sel])),
                                         "min": us((first_in[1:] - last_out[:-1])[sel].min())},
    "wave_lifetime_us": {"median": us(np.median(life[sel])), "p90": us(np.percentile(life[sel], 90)), "p99": us(np.percentile(life[sel], 99)),
                         "max_per_launch_median": us(np.median(life.max(axis=1)[sel]))},
    "launches_in_flight_at_a_wave_start": float(np.mean([(first_in[i + 1] < last_out[i]) for i in range(50, K - 1)])) + 1.0,
    "xcc_ids_seen": sorted(int(v) for v in np.unique(xcc[100])),
}
# does an env run on the same XCD in every launch?  (workgroup w of a launch goes to XCD w mod 8 when that holds)
res["envs_on_one_xcc_in_every_launch"] = float(np.mean((xcc == xcc[0:1]).all(axis=0)))
res["envs_on_xcc_of_workgroup_mod_8"] = float(np.mean((xcc == ((np.arange(N) // 8) % 8)[None, :]).all(axis=0)))
# per launch: how the wave starts spread (fraction of waves started after x us)
rel = (t_in - first_in[:, None])[sel]
res["wave_start_offset_us_percentiles"] = {str(p): us(np.percentile(rel, p)) for p in (10, 50, 90, 99, 100)}
relo = (t_out - first_in[:, None])[sel]
res["wave_end_offset_us_percentiles"] = {str(p): us(np.percentile(relo, p)) for p in (10, 50, 90, 99, 100)}
print(json.dumps(res, indent=1))
