#!/usr/bin/env python3
"""Where does a rotating layout pool cost time?  Timeline library (make timeline), direct launches: the gaps between
consecutive launches on the device clock, with the rotation events of the run next to them."""
import sys, time, os
sys.path.insert(0, os.getcwd())
os.environ.setdefault("CZ_LIB", os.path.join(os.getcwd(), "cooking_zoo_amd", "csrc", "libcookingzoo_hip_tl.so"))
os.environ["CZ_GRAPHS"] = "0"
import numpy as np
from cooking_zoo_amd.vec_env import CookingVecEnv
from cooking_zoo_amd import _native
if __name__ == "__main__":
    n, K, calls, period = 512, 50, 40, 64
    L = _native.lib()
    for rotate in (False, True):
        env = CookingVecEnv(n, "coop_test", "example", 2, 20, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=64, layout_seed=11)
        env.reset(return_obs=False)
        rng = np.random.default_rng(4)
        d_ring = env.alloc((period, n, 2), np.int32); d_ring.from_host(rng.integers(0, 5, size=(period, n, 2), dtype=np.int32))
        d_obs, d_rew = env.alloc((n, 2, env.F), np.float64), env.alloc((n, 2), np.float64)
        d_t, d_u = env.alloc((n, 2), np.uint8), env.alloc((n, 2), np.uint8)
        if rotate:
            env.rotate_layouts(500, groups=2, seed=5, prefetch=4)
            while env.rotation_ready() < 4: time.sleep(0.01)
        env.step_device_ring(K, d_ring, n * 2, period, 0, d_obs, d_rew, d_t, d_u)
        env.sync()
        tl = env.alloc((K * (calls - 1), n, 2), np.uint64)
        _native.check(env._h, L.cz_debug_set_timeline(env._h, tl.ptr, K * (calls - 1)))
        base_step = env._steps
        t0 = time.perf_counter()
        for c in range(1, calls):
            env.step_device_ring(K, d_ring, n * 2, period, (c * K) % period, d_obs, d_rew, d_t, d_u)
        env.sync()
        dt = time.perf_counter() - t0
        t = tl.to_host()
        t_in = (t[:, :, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64); t_out = (t[:, :, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        first_in, last_out = t_in.min(axis=1), t_out.max(axis=1)
        gaps = (first_in[1:] - last_out[:-1]) * 0.01          # us
        big = np.argsort(gaps)[-8:][::-1]
        print(f"rotate={rotate}: wall {dt*1e3:.2f} ms; device span {(last_out[-1]-first_in[0])*1e-5:.2f} ms; median gap {np.median(gaps):.2f} us; sum of gaps > 5 us: {gaps[gaps > 5].sum():.0f} us")
        print("  largest gaps (us) before step:", [(int(base_step + i + 1), round(float(gaps[i]), 1)) for i in big])
        dur = (last_out - first_in) * 0.01
        s2s = np.diff(first_in) * 0.01
        print(f"  kernel duration us: median {np.median(dur):.2f} mean {dur.mean():.2f} p99 {np.percentile(dur, 99):.2f}; start-to-start median {np.median(s2s):.2f} mean {s2s.mean():.2f}")
        print("  mean start-to-start per 100 launches:", [round(float(s2s[i:i + 100].mean()), 2) for i in range(0, len(s2s), 100)])
        print("  rotation events:", [(e[0], e[1]) for e in env.rotation_events])
        env.close()
