#!/usr/bin/env python3
"""At what shader clock do the step kernels run?  Needs `make -C cooking_zoo_amd/csrc tlclock` (libcz_tlclk.so: the timeline build with -DCZ_TL_CLOCK): every wave reports its
lifetime on the 100 MHz device clock and in shader-clock cycles (s_memtime).
    python3 tools/shader_clock.py [N=4096] [K=2000] [overlap=0]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CZ_LIB", os.path.join(REPO, "cooking_zoo_amd", "csrc", "libcz_tlclk.so"))
os.environ["CZ_GRAPHS"] = "0"
from cooking_zoo_amd import _native  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
overlap = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
P = 64
L = _native.lib()
env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256, auto_reset=True)
h = env._h
env.reset(return_obs=False)
acts = np.random.default_rng(0).integers(0, 5, size=(P, N, 2), dtype=np.int32)
d_act = env.alloc((P, N, 2), np.int32); d_act.from_host(acts)
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
outs = (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
if overlap:
    raise SystemExit("overlapped launches were removed in round 5 (profiles/r04 holds their timelines)")
_native.check(h, L.cz_step_device_ring(h, 3000, d_act.ptr, N * 2, P, 0, *outs)); env.sync()
tl = env.alloc((K, N, 2), np.uint64)
_native.check(h, L.cz_debug_set_timeline(h, tl.ptr, K))
_native.check(h, L.cz_step_device_ring(h, K, d_act.ptr, N * 2, P, 0, *outs)); env.sync()
t = tl.to_host()[100:]
ticks = ((t[:, :, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64) - (t[:, :, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64))
cyc = ((t[:, :, 1] >> np.uint64(36)) & np.uint64(0xFFFFFFF)).astype(np.int64)
ok = ticks > 100                                   # >= 1 us: the 10 ns granularity is then below 1 %
ghz = cyc[ok] / (ticks[ok] * 10.0)
print("%d envs, %s launches: wave lifetime %.2f us median = %d s_memtime counts median; counts per ns: median %.3f  p1 %.3f  p99 %.3f" % (
    N, "overlapped" if overlap else "boundary-ordered", np.median(ticks) * 0.01, np.median(cyc), np.median(ghz), np.percentile(ghz, 1), np.percentile(ghz, 99)))
