#!/usr/bin/env python3
"""What does an interaction cost?  Drives the bench batch with hand-made actions: every agent stays / walks randomly /
bumps into a neighbouring non-walkable cell whenever it has one (interaction every step).  Run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES` to read instructions per wave per mode (each
mode is a contiguous block of 40 launches: print order = stay, random, bump), or plain to time the launches."""
import ctypes as C, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cooking_zoo_amd import _native, soa  # noqa: E402
from cooking_zoo_amd.vec_env import CookingVecEnv  # noqa: E402

N = 4096
env = CookingVecEnv(N, "coop_test", "example", 2, 1 << 30, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256)
L, h = _native.lib(), env._h
d_act = env.alloc((N, 2), np.int32)
d_obs = env.alloc((N, 2, env.F), np.float64); d_rew = env.alloc((N, 2), np.float64)
d_t = env.alloc((N, 2), np.uint8); d_u = env.alloc((N, 2), np.uint8)
rng = np.random.default_rng(0)
DX, DY = [0, -1, 1, 0, 0], [0, 0, 0, 1, -1]


def bump_actions(recs):
    d = env.dims
    cells = recs[:, d.cells_word0:d.cells_word0 + d.CW].view(np.uint8)[:, :d.C]
    walk = np.isin(cells & 7, (soa.FLOOR, soa.SWITCH))
    acts = np.zeros((N, 2), np.int32)
    for a in range(2):
        w = recs[:, soa.AGENT_WORD0 + a]
        x, y = (w & 0xFF).astype(int), ((w >> 8) & 0xFF).astype(int)
        order = rng.permutation([1, 2, 3, 4])
        for k in order:
            tx, ty = x + DX[k], y + DY[k]
            ok = (tx >= 0) & (tx < d.W) & (ty >= 0) & (ty < d.H)
            c = np.clip(ty, 0, d.H - 1) * d.W + np.clip(tx, 0, d.W - 1)
            hit = ok & ~walk[np.arange(N), c] & (acts[:, a] == 0)
            acts[hit, a] = k
        none = acts[:, a] == 0
        acts[none, a] = rng.integers(1, 5, size=int(none.sum()))        # nobody to bump: walk somewhere
    return acts


for mode in ("stay", "random", "bump"):
    env.reset(return_obs=False)
    times = []
    for t in range(40):
        if mode == "stay":
            acts = np.zeros((N, 2), np.int32)
        elif mode == "random":
            acts = rng.integers(0, 5, size=(N, 2), dtype=np.int32)
        else:
            acts = bump_actions(env.get_state())
        d_act.from_host(acts)
        env.sync()
        L.cz_timer_start(h)
        for rep in range(1):
            env.step_device(d_act, d_obs, d_rew, d_t, d_u)
        ms = C.c_float()
        L.cz_timer_stop(h, C.byref(ms))
        times.append(ms.value * 1e3)
    st = env.get_state()
    holding = ((st[:, soa.AGENT_WORD0:soa.AGENT_WORD0 + 2] >> 24) != 0).mean()
    print(f"{mode:7s}: single launch from an idle stream, median {np.median(times[5:]):.2f} us; agents holding something at the end: {holding:.2f}")
