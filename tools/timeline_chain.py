#!/usr/bin/env python3
"""The hand-off of overlapped launches on the device clock (timeline build): for env e and launch k, when did the wave
enter, when had it seen its predecessor's number, when did it leave (= publish)?
    python3 tools/timeline_chain.py [N=4096] [K=2000]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CZ_LIB", os.path.join(REPO, "cooking_zoo_amd", "csrc", "libcookingzoo_hip_tl.so"))
os.environ["CZ_GRAPHS"] = "0"
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
P = 64
L = _native.lib()
env = CookingVecEnv(N, "coop_test", "example", 2, 1 << 30, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
h = env._h
env.reset(return_obs=False)
acts = np.random.default_rng(0).integers(0, 5, size=(P, N, 2), dtype=np.int32)
d_act = env.alloc((P, N, 2), np.int32); d_act.from_host(acts)
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
outs = (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
env.set_overlap(True)
_native.check(h, L.cz_step_device_ring(h, 300, d_act.ptr, N * 2, P, 0, *outs)); env.sync()
tl = env.alloc((K, N, 2), np.uint64)
_native.check(h, L.cz_debug_set_timeline(h, tl.ptr, K))
_native.check(h, L.cz_step_device_ring(h, K, d_act.ptr, N * 2, P, 0, *outs)); env.sync()
t = tl.to_host()
t_in = (t[:, :, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t_out = (t[:, :, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
seen = t_in + ((t[:, :, 1] >> np.uint64(36)) & np.uint64(0xFFFFFFF)).astype(np.int64)
sl = slice(100, K - 1)
tick = 0.01
pub_prev = t_out[:-1]                                  # launch k - 1 of the same env leaves (its number is on its way)
ho = (seen[1:] - pub_prev)[sl] * tick                  # predecessor's exit stamp -> successor has seen the number
work = (t_out - seen)[1:][sl] * tick
spin = (seen - t_in)[1:][sl] * tick
early = (t_in[1:] < pub_prev)[sl]                      # successor was already waiting when the predecessor left
period = (t_out[1:] - t_out[:-1])[sl] * tick           # per env: publish to publish


def pct(x):
    return "median %.2f  p10 %.2f  p90 %.2f  p99 %.2f" % tuple(np.percentile(x, [50, 10, 90, 99]))


print("overlapped run, %d envs, %d launches; per env and step, us:" % (N, K))
print("  publish -> publish (the env's own period):   ", pct(period))
print("  successor already waiting when the predecessor left: %.1f %% of the hand-offs" % (100 * early.mean()))
print("  predecessor's exit stamp -> successor saw it: ", pct(ho[early]))
print("  saw it -> own exit (loads, step, encode, stores acknowledged, publish):", pct(work))
print("  entry -> saw it (waiting):                    ", pct(spin))
print("  whole run: %.3f us per launch" % ((t_out[-1].max() - t_in[100].min()) * tick / (K - 100)))
