#!/usr/bin/env python3
"""bench.py -- env-steps/s of the CookingZoo step() hot path on MI355X (BASELINE.json metric).

A "step" is one batched env step (one launch of the step kernel) over every env of the rank: world dynamics, recipe
checks, rewards and the float64 feature-vector encode of all agents, with actions and outputs resident in HBM.
Workload at N GPUs = BASELINE config 2 per GPU (weak scaling): 4096 envs, level coop_test, 2 agents, recipes
[TomatoLettuceSalad, CarrotBanana], scheme3, max_steps 400, a pool of 256 layouts, next-step auto-reset, uniform
random actions.  Prints ONE JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W       # N > 1: starts N fresh processes itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W          # or under any launcher that sets RANK/LOCAL_RANK/WORLD_SIZE

No torch anywhere: every rank runs on the system ROCm runtime; the ranks of the node meet through a tmpfs directory
(cooking_zoo_amd.distributed.FileRendezvous) and, for the barrier of the timed region and the episode-statistics
all-gather, through the C-ABI's own RCCL communicator (cz_comm_init / cz_comm_barrier / cz_stats_allgather).

Timing: W warmup steps, then the K-step region is timed `--repeats` times; every region is bracketed by a stream
synchronisation + barrier over all ranks on both sides, its time is the MAX over ranks, and `value` / `ms_per_step` are
the MEDIAN over the regions (min and max are reported next to it).
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
COMM_DEADLINE_S = 120.0
EXIT_COMM_FAILED = 3


def algorithmic_bytes_per_env_step(env):
    """SURVEY.md 8(d):  A(8F + 8 + 1 + 1) written + 4A actions read + 2*S_dyn + W*H static cells read,
    S_dyn = 4A + 4D + action objects + linked + 4 (t) + ceil(nodes/8).  4655 B for config 2 (D = 12: 10 objects + 2
    Bread clones, 4 action objects: 3 Cutboards + 1 Blender, 8 recipe nodes)."""
    A, F, C_ = env.num_agents, env.F, env.dims.C
    D = max(l.slots_used for l in env.layouts)
    n_action = max(len(l.static_lists.get("Cutboard", [])) + len(l.static_lists.get("Blender", [])) for l in env.layouts)
    n_linked = max(len(l.static_lists.get("Switch", [])) + len(l.static_lists.get("Block", [])) for l in env.layouts)
    nodes = sum(int(env.recipe_table[r][0]) for r in env.recipe_ids[0][:env.num_recipes])
    s_dyn = 4 * A + 4 * D + n_action + n_linked + 4 + (nodes + 7) // 8
    return A * (8 * F + 10) + 4 * A + 2 * s_dyn + C_


def reference_python_timing():
    """The unmodified reference timed in the build container (tools/time_reference.py; the reference cannot travel to the
    GPU box), committed as data under profiles/."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*", "reference_python_timing.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        d["source"] = os.path.relpath(files[-1], REPO) + " (tools/time_reference.py, build container)"
        return d
    except Exception:
        return None


def cpu_baseline(env, seconds_target=15.0):
    """The oracle (a bit-exact C port of the reference step path, oracle/cz_oracle.c) timed on this box's host cores
    on a bounded sample of the same workload: every thread steps its own slice of envs through 400-step episodes with
    the same counter-based action stream and encodes the observations every step."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle_binding import VecOracle
    cores = len(os.sched_getaffinity(0))
    envs_per_thread, T = 8, 400
    workers = [VecOracle.from_vec_env(env, num_envs=envs_per_thread, env_id_base=i * envs_per_thread) for i in range(cores)]
    for w in workers:
        w.reset()

    def run_all(reps, first):
        def run(w):
            for r in range(reps):
                w.rollout(T, 0, (first + r) * T)
        ths = [threading.Thread(target=run, args=(w,)) for w in workers]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return time.perf_counter() - t0

    t0 = time.perf_counter()
    workers[0].rollout(T, 0, 0)
    single = envs_per_thread * T / (time.perf_counter() - t0)
    dt1 = run_all(1, 1)                                        # calibration pass on all cores
    reps = int(min(max(seconds_target / max(dt1, 1e-3), 1), 2000))
    dt = run_all(reps, 2)
    total = cores * envs_per_thread * T * reps
    out = {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"oracle/cz_oracle.c, {cores} threads x {envs_per_thread} envs x {T * reps} steps of the bench "
                     f"workload incl. obs encode ({total} env-steps in {dt:.1f} s); one thread alone: {single:.0f} env-steps/s"}
    ref = reference_python_timing()
    if ref is not None:
        out["reference_python"] = ref
    return out


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/rNN/traffic.json), or None."""
    best = None
    for d in sorted(os.listdir(os.path.join(REPO, "profiles"))) if os.path.isdir(os.path.join(REPO, "profiles")) else []:
        p = os.path.join(REPO, "profiles", d, "traffic.json")
        if os.path.exists(p):
            try:
                t = json.load(open(p))
                if kernel_key in t:
                    best = t[kernel_key]
            except Exception:
                pass
    return best


def roofline_block(b_alg, units, us_per_launch, kernel, note=None):
    """algorithmic bytes per env-step x env-steps per launch / launch duration, against the 8 TB/s HBM peak"""
    achieved = b_alg * units / (us_per_launch * 1e-6) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "kernel": kernel, "kernel_us": us_per_launch, "alg_bytes_per_env_step": b_alg, "units_per_launch": units}
    if note:
        out["kernel_us_from"] = note
    return out


def timed_ring(env, K, ring, outs, reps=3):
    """K boundary-ordered launches (graph replay) timed with HIP events on the handle's stream, best of `reps`; -> us per launch"""
    from cooking_zoo_amd import _native
    L, h = _native.lib(), env._h
    d_ring, stride, period = ring
    _native.check(h, L.cz_ring_prepare(h, K, d_ring, stride, period, 0, *outs))
    _native.check(h, L.cz_step_device_ring(h, min(K, period), d_ring, stride, period, 0, *outs))       # warm
    best = None
    ms = C.c_float()
    for _ in range(reps):
        _native.check(h, L.cz_timer_start(h))
        _native.check(h, L.cz_step_device_ring(h, K, d_ring, stride, period, 0, *outs))
        _native.check(h, L.cz_timer_stop(h, C.byref(ms)))
        best = ms.value if best is None else min(best, ms.value)
    return best * 1e3 / K


def leg_fused(env, K):
    """the same work fused: T = 32 steps per launch with the on-device action stream and a [T][N][A][F] trajectory buffer"""
    N, A, T = env.num_envs, env.num_agents, 32
    d_traj = env.alloc((T, N, A, env.F), np.float64)
    d_r = env.alloc((T, N, A), np.float64)
    d_te = env.alloc((T, N, A), np.uint8)
    d_tr = env.alloc((T, N, A), np.uint8)
    try:
        reps = max(2, min(K, 2000) // T)
        env.rollout(T, 1, 0, d_traj, d_r, d_te, d_tr)
        env.sync()
        f0 = env.stats()["env_steps"]
        t0 = time.perf_counter()
        for r in range(reps):
            env.rollout(T, 1, (r + 1) * T, d_traj, d_r, d_te, d_tr)
        env.sync()
        dtf = time.perf_counter() - t0
        return {"env_steps_per_s_per_gpu": (env.stats()["env_steps"] - f0) / dtf, "steps_per_launch": T,
                "ms_per_step": dtf * 1e3 / (reps * T), "api": "cz_rollout, obs trajectory [T][N][A][F] in HBM"}
    finally:
        for b in (d_traj, d_r, d_te, d_tr):
            b.free()


def leg_fused_compact(env, K):
    """T = 32 fused steps per launch with a COMPACT trajectory (cz_rollout_compact: uint8 codes [T][N][A][pitch], decodable bit for
    bit with the 256-entry table): what a learner that embeds the codes reads back, at an eighth of the trajectory's bytes"""
    N, A, T = env.num_envs, env.num_agents, 32
    d_codes = env.alloc((T, N, A, env.codes_pitch), np.uint8)
    try:
        reps = max(2, min(K, 2000) // T)
        env.rollout_compact(T, 1, 0, d_codes)
        env.sync()
        f0 = env.stats()["env_steps"]
        t0 = time.perf_counter()
        for r in range(reps):
            env.rollout_compact(T, 1, (r + 1) * T, d_codes)
        env.sync()
        dt = time.perf_counter() - t0
        b_alg = algorithmic_bytes_per_env_step(env) - A * 8 * env.F + A * env.F
        out = {"env_steps_per_s_per_gpu": (env.stats()["env_steps"] - f0) / dt, "steps_per_launch": T, "ms_per_step": dt * 1e3 / (reps * T),
               "api": "cz_rollout_compact: uint8 code trajectory [T][N][A][%d] in HBM (no float64 rows)" % env.codes_pitch}
        out["roofline"] = roofline_block(b_alg, N, out["ms_per_step"] * 1e3, "cz::k_step<1,1,2,3,4> (32 steps per launch, codes)",
                                         "wall clock around the cz_rollout_compact launches; algorithmic bytes with 1 byte per feature")
        return out
    finally:
        d_codes.free()


def leg_closed_loop(env, d_obs, d_rew, d_term, d_trunc):
    """What a reinforcement-learning loop gets: cz_step_device, then a kernel of the caller that turns the observation into the
    next actions, then the next step - every step waits for a policy that waits for the step before it.  (The policy here is
    the cheapest possible one; a real one adds its own time.)"""
    from cooking_zoo_amd import _native
    L, h = _native.lib(), env._h
    N, A = env.num_envs, env.num_agents
    d_act = env.alloc((N, A), np.int32)
    d_act.from_host(np.random.default_rng(7).integers(0, env.n_actions, size=(N, A), dtype=np.int32))
    K, reps = 200, 10
    us = C.c_float()
    s0 = env.stats()["env_steps"]
    _native.check(h, L.cz_probe_closed_loop(h, K, reps, d_act.ptr, d_obs.ptr, d_rew.ptr, d_term.ptr, d_trunc.ptr, C.byref(us)))
    stepped = (env.stats()["env_steps"] - s0) / float(N * K * (reps + 1))        # (auto-reset passes are not env steps)
    d_act.free()
    b_alg = algorithmic_bytes_per_env_step(env)
    return {"env_steps_per_s": stepped * N / (us.value * 1e-6), "us_per_step": us.value, "envs": N,
            "what": "closed loop on one stream: cz_step_device -> policy kernel (next actions = hash of the observation just written) "
                    f"-> cz_step_device ...; {K}-step HIP graph replayed {reps} times, HIP events; includes the policy kernel and its launch boundary",
            "roofline": roofline_block(b_alg, N, us.value, "cz::k_step<1,1,2,3,0> + k_probe_policy per step",
                                       "HIP events around graph replays of the closed loop; step + policy kernel + both boundaries")}


def leg_closed_loop_compact(env, d_rew, d_term, d_trunc):
    """The same closed loop for a consumer that stays on the device and reads the COMPACT observation (cz_step_device_compact:
    one byte per feature = the index of its value in a table of 256 float64, bit-exactly decodable): the step kernel writes
    N x A x pitch bytes instead of N x A x F x 8, the policy kernel reads the same four features as table indices and takes the same
    actions.  Roofline on ITS algorithmic bytes (the code bytes instead of the float64 rows)."""
    from cooking_zoo_amd import _native
    L, h = _native.lib(), env._h
    N, A = env.num_envs, env.num_agents
    d_act = env.alloc((N, A), np.int32)
    d_act.from_host(np.random.default_rng(7).integers(0, env.n_actions, size=(N, A), dtype=np.int32))
    d_codes = env.alloc((N, A, env.codes_pitch), np.uint8)
    K, reps = 200, 10
    us = C.c_float()
    s0 = env.stats()["env_steps"]
    try:
        _native.check(h, L.cz_probe_closed_loop_compact(h, K, reps, d_act.ptr, d_codes.ptr, d_rew.ptr, d_term.ptr, d_trunc.ptr, C.byref(us)))
        stepped = (env.stats()["env_steps"] - s0) / float(N * K * (reps + 1))
    finally:
        d_act.free()
        d_codes.free()
    b_alg = algorithmic_bytes_per_env_step(env) - A * 8 * env.F + A * env.F        # one byte per feature instead of eight
    return {"env_steps_per_s": stepped * N / (us.value * 1e-6), "us_per_step": us.value, "envs": N, "obs_bytes_per_env_step": A * env.codes_pitch,
            "what": "closed loop on one stream over the compact observation: cz_step_device_compact (uint8 code per feature, table of 256 "
                    f"float64 resident; no float64 rows written) -> policy kernel reading the codes -> ...; {K}-step HIP graph replayed {reps} times",
            "roofline": roofline_block(b_alg, N, us.value, "cz::k_step<1,1,2,3,0> (codes only) + k_probe_policy_codes per step",
                                       "HIP events around graph replays; algorithmic bytes with 1 byte per feature")}


def leg_fused_actions(env, K):
    """T = 32 fused steps per launch over actions of the CALLER (cz_rollout_actions: replay / open-loop search), float64
    trajectory [T][N][A][F]"""
    N, A, T = env.num_envs, env.num_agents, 32
    d_traj = env.alloc((T, N, A, env.F), np.float64)
    d_act = env.alloc((T, N, A), np.int32)
    try:
        d_act.from_host(np.random.default_rng(11).integers(0, env.n_actions, size=(T, N, A), dtype=np.int32))
        reps = max(2, min(K, 2000) // T)
        env.rollout_actions(d_act, T, d_traj)
        env.sync()
        f0 = env.stats()["env_steps"]
        t0 = time.perf_counter()
        for r in range(reps):
            env.rollout_actions(d_act, T, d_traj)
        env.sync()
        dt = time.perf_counter() - t0
        out = {"env_steps_per_s_per_gpu": (env.stats()["env_steps"] - f0) / dt, "steps_per_launch": T, "ms_per_step": dt * 1e3 / (reps * T),
               "api": "cz_rollout_actions: int32 actions [T][N][A] of the caller in HBM, obs trajectory [T][N][A][F]"}
        out["roofline"] = roofline_block(algorithmic_bytes_per_env_step(env), N, out["ms_per_step"] * 1e3, "cz::k_step<1,1,2,3,2> (32 steps per launch)",
                                         "wall clock around the cz_rollout_actions launches")
        return out
    finally:
        d_traj.free()
        d_act.free()


def leg_ring_fused(env, K, d_obs, d_rew, d_term, d_trunc):
    """The headline's call - cz_step_device_ring over a ring of action slots, outputs in place - with cz_set_ring_fused(h, 1): the K
    steps of a region go out as fused launches over the ring's own rows (one per stretch of consecutive slots) instead of one
    launch per step.  Same final state, outputs and statistics (tests/test_gpu_ring_fused.py); regions of the driver's K and of
    2000 steps, wall clock between synchronisations like the headline's regions."""
    from cooking_zoo_amd import _native
    L, h = _native.lib(), env._h
    N, A, period = env.num_envs, env.num_agents, 64
    d_ring = env.alloc((period, N, A), np.int32)
    outs = (d_obs.ptr, d_rew.ptr, d_term.ptr, d_trunc.ptr)
    was = env.set_ring_fused(True)
    try:
        d_ring.from_host(np.random.default_rng(13).integers(0, env.n_actions, size=(period, N, A), dtype=np.int32))
        out = {"api": "cz_step_device_ring after cz_set_ring_fused(h, 1): fused launches over the ring's rows, outputs written in place"}
        for k in sorted({int(K), 2000}):
            _native.check(h, L.cz_step_device_ring(h, k, d_ring.ptr, N * A, period, 0, *outs))
            env.sync()
            times = []
            for rep in range(7):
                t0 = time.perf_counter()
                _native.check(h, L.cz_step_device_ring(h, k, d_ring.ptr, N * A, period, 0, *outs))
                env.sync()
                times.append(time.perf_counter() - t0)
            dt = sorted(times)[len(times) // 2]
            out["K=%d" % k] = {"us_per_step": dt * 1e6 / k, "env_steps_per_s": N * k / dt}
        return out
    finally:
        env.set_ring_fused(was)
        d_ring.free()


def cooking_policy_workload(device_id, K=512):
    """The one-step kernel under a policy that cooks: the reference's heuristic agent's action sequences (golden fixtures
    cfg2_coop_2agents, episodes that end with a delivered dish), every env on one of those worlds at its own phase of the
    episode; when an episode ends the env resets onto a world of the same pool and follows that world's sequence."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import golden_io                                                    # (fixture loader: data, not the oracle)
    from cooking_zoo_amd import _native, soa
    from cooking_zoo_amd.vec_env import CookingVecEnv
    gs = golden_io.GoldenSet("cfg2_coop_2agents")
    eps = [ep for ep in gs.episodes if ep.policy in ("heuristic", "mixed") and ep.terms.any()]
    lays = [golden_io.layout_from_episode(ep) for ep in eps]
    N, A, P = 4096, 2, len(eps)
    env = CookingVecEnv(N, gs.cfg["level"], gs.cfg["meta_file"], A, gs.cfg["max_steps"], list(gs.cfg["recipes"]),
                        action_scheme="scheme3" if gs.scheme == 3 else "scheme1", layouts=lays, auto_reset=True, device_id=device_id)
    L, h = _native.lib(), env._h
    env.reset(return_obs=False)
    T = [len(ep.actions) for ep in eps]
    # env e: world j = the layout it drew for episode 0, phase s_e of that world's episode
    recs = env.get_state()
    rng = np.random.default_rng(3)
    lay0 = recs[:, soa.W_LAYOUT].astype(int)
    phase = np.array([rng.integers(0, T[j]) for j in lay0])
    hdr = [soa.W_LAYOUT, soa.W_RECIPES, soa.W_POOL, soa.W_EPISODE]
    for e in range(N):
        keep = recs[e, hdr].copy()
        recs[e] = eps[lay0[e]].states[phase[e]]
        recs[e, hdr] = keep
        recs[e, soa.W_STATUS] = 0
    env.set_state(recs)
    # the action ring: follow each env through its episodes (termination -> one auto-reset pass -> next world from phase 0)
    ring = np.zeros((K, N, A), dtype=np.int32)
    pool_word = int(recs[0, soa.W_POOL])
    for e in range(N):
        j, s, ep_no, k = int(lay0[e]), int(phase[e]), 0, 0
        while k < K:
            n = min(T[j] - s, K - k)
            ring[k:k + n, e] = eps[j].actions[s:s + n]
            k += n
            if k < K:
                k += 1                                                   # the auto-reset pass (its action is ignored)
                ep_no += 1
                j, s = int(L.cz_next_layout(env.env_id_base + e, ep_no, pool_word, P)), 0
    return env, ring, P, T


def leg_cooking_policy(device_id):
    from cooking_zoo_amd import _native
    K = 512
    env, ring, P, T = cooking_policy_workload(device_id, K)
    L, h, N, A = _native.lib(), env._h, env.num_envs, env.num_agents
    d_ring = env.alloc((K, N, A), np.int32)
    d_ring.from_host(ring)
    d_obs, d_rew = env.alloc((N, A, env.F), np.float64), env.alloc((N, A), np.float64)
    d_t, d_u = env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
    outs = (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr)
    env.reset_stats()
    s0 = env.stats()
    ms = C.c_float()
    _native.check(h, L.cz_ring_prepare(h, K, d_ring.ptr, N * A, K, 0, *outs))
    _native.check(h, L.cz_timer_start(h))
    _native.check(h, L.cz_step_device_ring(h, K, d_ring.ptr, N * A, K, 0, *outs))
    _native.check(h, L.cz_timer_stop(h, C.byref(ms)))
    st = env.stats()
    us = ms.value * 1e3 / K
    b_alg = algorithmic_bytes_per_env_step(env)
    out = {"env_steps_per_s": (st["env_steps"] - s0["env_steps"]) / (ms.value * 1e-3), "us_per_launch": us, "envs": N, "launches": K,
           "episodes_terminated_by_a_delivered_dish": st["terminations"] - s0["terminations"], "episodes_truncated": st["truncations"] - s0["truncations"],
           "return_sum_agent0": st["return_sum"][0],
           "what": f"one launch per step, launch-boundary ordering (graph replay); every env replays the reference heuristic agent's actions "
                   f"(golden fixtures cfg2_coop_2agents: {P} episodes of {min(T)}-{max(T)} steps that chop, plate and deliver) from its own phase",
           "roofline": roofline_block(b_alg, N, us, "cz::k_step<1,1,2,3,0>", "HIP events around one graph-replayed run of 512 launches")}
    env.close()
    return out


def leg_configs(device_id, which=("config3", "config5", "config4_shard")):
    """BASELINE configs 3 and 5 and one rank's shard of config 4: one launch per step (graph replay) and fused (cz_rollout), a few
    hundred milliseconds each."""
    from cooking_zoo_amd.vec_env import CookingVecEnv
    out = {}
    for name in which:
        if name == "config3":
            n = 65536
            rid = np.array([[e % 8, (e + 1) % 8] for e in range(n)])
            env = CookingVecEnv(n, ["coop_test", "coexistence_test", "switch_test"], "example", 2, 400, rid, action_scheme="scheme3",
                                num_layouts=256, device_id=device_id)
            what, K, T = "65536 envs, levels e % 3 of coop / coexistence / switch, recipes cycling over the book, 2 agents", 60, 16
        elif name == "config5":
            env = CookingVecEnv(65536, "large_16x16", "large_16x16", 4, 400, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"],
                                action_scheme="scheme3", num_layouts=64, device_id=device_id)
            what, K, T = "65536 envs, 4 agents, large_16x16 at maximum object density, F = %d" % env.F, 24, 4
        else:
            env = CookingVecEnv(32768, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                                num_layouts=256, env_id_base=98304, device_id=device_id)
            what, K, T = "one rank's shard of config 4: 32768 of 262144 envs (global ids from 98304), coop_test, 2 agents", 100, 16
        N, A = env.num_envs, env.num_agents
        env.reset(return_obs=False)
        period = 16
        d_ring = env.alloc((period, N, A), np.int32)
        d_ring.from_host(np.random.default_rng(5).integers(0, env.n_actions, size=(period, N, A), dtype=np.int32))
        d_obs, d_rew = env.alloc((N, A, env.F), np.float64), env.alloc((N, A), np.float64)
        d_t, d_u = env.alloc((N, A), np.uint8), env.alloc((N, A), np.uint8)
        us = timed_ring(env, K, (d_ring.ptr, N * A, period), (d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr))
        # fused: T steps per launch, trajectory buffer
        d_traj = env.alloc((T, N, A, env.F), np.float64)
        env.rollout(T, 1, 0, d_traj)
        env.sync()
        reps = max(2, K // T)
        t0 = time.perf_counter()
        for r in range(reps):
            env.rollout(T, 1, (r + 1) * T, d_traj)
        env.sync()
        us_fused = (time.perf_counter() - t0) * 1e6 / (reps * T)
        b_alg = algorithmic_bytes_per_env_step(env)
        stepping = 400.0 / 401.0                                        # one launch in 401 is an env's auto-reset pass
        kern = "cz::k_step (instance %s)" % ("1 slot / 1 cell per lane" if env.dims.D <= 64 and env.dims.C <= 64 else "2 slots / 4 cells per lane")
        out[name] = {"workload": what, "envs": N, "agents": A, "F": env.F,
                     "per_step": {"env_steps_per_s": stepping * N / (us * 1e-6), "us_per_launch": us,
                                  "roofline": roofline_block(b_alg, N, us, kern, "HIP events around graph replays of %d launches, best of 3" % K)},
                     "fused": {"env_steps_per_s": stepping * N / (us_fused * 1e-6), "us_per_step": us_fused, "steps_per_launch": T,
                               "roofline": roofline_block(b_alg, N, us_fused, kern + ", fused", "wall clock around %d cz_rollout launches of %d steps" % (reps, T))}}
        env.close()
    return out


def guarded(fn, *a, **kw):
    """an extra leg of the line: its result, or {"error": ...} - never an exception that would lose the headline"""
    try:
        return fn(*a, **kw)
    except BaseException as exc:                     # noqa: BLE001  (incl. MemoryError; KeyboardInterrupt is re-raised)
        if isinstance(exc, KeyboardInterrupt):
            raise
        import traceback
        return {"error": f"{type(exc).__name__}: {exc}", "where": traceback.format_exc(limit=3).strip().splitlines()[-3:]}


CFG4_ENVS_PER_GPU, CFG4_K = 32768, 100        # BASELINE config 4: 262 144 envs over 8 GPUs; its timed regions are 100 steps long


def config4_block(world, per_rank, b_alg4):
    """the `config4` object of the line from every rank's region timings (per_rank[r]: elapsed_s, env_steps, env_id_base, envs)"""
    a4 = aggregate(per_rank, CFG4_K)
    total = sum(p["envs"] for p in per_rank)
    out = {"workload": f"BASELINE config 4 shape: {CFG4_ENVS_PER_GPU} envs per GPU x {world} GPU(s) = {total} envs, coop_test, 2 agents, global env ids, "
                       f"statistics all-gather over RCCL; one launch per step (graph replay), {CFG4_K}-step regions between all-rank barriers",
           "envs": total, "shards": [[p["env_id_base"], p["envs"]] for p in per_rank],
           "value": a4["value"], "unit": "env-steps/s", "ms_per_step": a4["ms_per_step"], "value_min": a4["value_min"], "value_max": a4["value_max"]}
    if b_alg4 is not None:
        out["roofline"] = roofline_block(b_alg4, CFG4_ENVS_PER_GPU, a4["ms_per_step"] * 1e3, "cz::k_step<1,1,2,3,0>", "wall clock of the slowest rank per region")
    return out


def with_deadline(fn, seconds):
    """fn() on a helper thread; -> (finished, result or exception)."""
    box = {}

    def run():
        try:
            box["r"] = fn()
        except Exception as exc:                     # noqa: BLE001
            box["r"] = exc
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout=seconds)
    return (not th.is_alive()), box.get("r")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=30, help="how many times the K-step region is timed (median reported)")
    ap.add_argument("--envs", type=int, default=4096, help="env instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline measurement: skip closed_loop / cooking_policy / configs / config4 (profiling runs)")
    ap.add_argument("--no-obs", action="store_true", help="ablation only: skip the observation encode (INVALID as a result)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work at all: every rank reports made-up timings so that the launcher / rendezvous / "
                         "aggregation path can be exercised on a machine without GPUs (the line is marked INVALID)")
    return ap.parse_args(argv)


def aggregate(per_rank, K):
    """per_rank: one dict per rank with lists `elapsed_s`, `env_steps`, `kernel_us` (one entry per timed region).
    -> per-region whole-job throughput: sum of the ranks' env-steps / the slowest rank's time."""
    R = len(per_rank[0]["elapsed_s"])
    elapsed = [max(p["elapsed_s"][r] for p in per_rank) for r in range(R)]
    steps = [sum(p["env_steps"][r] for p in per_rank) for r in range(R)]
    rate = [s / e for s, e in zip(steps, elapsed)]
    order = sorted(range(R), key=lambda r: rate[r])
    med = order[R // 2] if R % 2 else None
    value = rate[med] if med is not None else 0.5 * (rate[order[R // 2 - 1]] + rate[order[R // 2]])
    ms = sorted(e * 1e3 / K for e in elapsed)
    ms_med = ms[R // 2] if R % 2 else 0.5 * (ms[R // 2 - 1] + ms[R // 2])
    return {"value": value, "value_min": min(rate), "value_max": max(rate), "ms_per_step": ms_med, "ms_per_step_min": ms[0],
            "ms_per_step_max": ms[-1], "env_steps_per_region": steps[order[R // 2]], "repeats": R}


def emit(line):
    """the one JSON line, on the process's real stdout (see worker)"""
    os.write(_REAL_STDOUT, (json.dumps(line) + "\n").encode())


_REAL_STDOUT = 1


_LAST_ENV = None


def worker(args):
    """Measures with overlapped launches first (unless CZ_CHAIN=0).  If any rank's library ever abandons a hand-off (it then
    refuses to go on: the env states are void), that rank declares the attempt's rendezvous void, every rank drops its env
    and communicator, and all of them measure again with launch-boundary ordering - the line then says so."""
    global _LAST_ENV
    _redirect_stdout()
    from cooking_zoo_amd import distributed as czd
    base = czd.FileRendezvous.from_env(timeout=600.0)
    sim = os.environ.get("CZ_BENCH_SIMULATE_HANDOFF_TIMEOUT")            # test hook: "1" = every rank, "10r" = rank r only
    want = os.environ.get("CZ_CHAIN", "1") != "0" and (not args.dry_run or bool(sim))
    note, rc = None, 1
    for attempt, overlap in enumerate([True, False] if want else [False]):
        rdzv = base.subdir(f"attempt{attempt}")
        try:
            rc = worker_body(args, rdzv, overlap, note)
            break
        except Exception as exc:                                  # noqa: BLE001
            abandoned, poisoned = "gave up waiting" in str(exc), isinstance(exc, czd.RendezvousPoisoned)
            if not overlap or not (abandoned or poisoned):
                raise
            if abandoned:
                rdzv.poison(str(exc))
            print(f"bench.py rank {rdzv.rank}: {exc}\nbench.py: measuring again with launch-boundary ordering", file=sys.stderr)
            note = f"overlapped launches were abandoned in the first attempt ({exc}); measured with launch-boundary ordering"
            if _LAST_ENV is not None:
                try:
                    _LAST_ENV.close()
                except Exception:                                 # noqa: BLE001
                    pass
                _LAST_ENV = None
    try:
        base.close()
    except Exception:                                             # noqa: BLE001
        pass
    return rc


def _redirect_stdout():
    # Native libraries print to stdout through C stdio (RCCL's version banner, for one), and whether that reaches fd 1
    # before or after this script's own line depends on who flushes when.  So fd 1 is handed to stderr for the whole
    # run, and the JSON line goes to a private duplicate of the real stdout: nothing else can appear there.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def worker_body(args, rdzv, overlap, note):
    global _LAST_ENV
    from cooking_zoo_amd import distributed as czd
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr)
    N, K, Wm, R = args.envs, args.steps, args.warmup, max(1, args.repeats)
    begin, count = czd.shard_range(N * world, world, rank)

    sim = os.environ.get("CZ_BENCH_SIMULATE_HANDOFF_TIMEOUT")
    if args.dry_run:
        if overlap and sim and (sim == "1" or int(sim) == rank + 100):
            raise RuntimeError("simulated: an overlapped launch gave up waiting for its predecessor")
        b4, n4 = czd.shard_range(CFG4_ENVS_PER_GPU * world, world, rank)
        mine = {"elapsed_s": [1e-3 * (rank + 1 + 0.01 * r) for r in range(R)], "env_steps": [K * count] * R,
                "kernel_us": [1.0] * R, "stats": {"env_steps": K * count * R},
                "device_id": local_rank, "rank": rank, "env_id_base": begin, "cfg4": {"env_id_base": b4, "envs": n4,
                "elapsed_s": [1e-2 * (rank + 1)] * 2, "env_steps": [CFG4_K * n4] * 2, "kernel_us": [0.0] * 2}}
        every = [json.loads(b) for b in rdzv.all_gather(json.dumps(mine).encode())]
        if rank == 0:
            line = {"metric": "env-steps/sec at N parallel envs (1/2/4/8 GPU) + achieved HBM GB/s", "unit": "env-steps/s",
                    "n_gpus": world, "steps": K, "warmup": Wm, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "u8/u32 state, f64 obs+reward", "data": "DRY RUN: no GPU work, made-up timings (INVALID as a result)",
                    "config": {"workload": f"dry run, {N} envs per rank x {world} rank(s)"},
                    "shards": [czd.shard_range(N * world, world, r) for r in range(world)],
                    "stats_total_env_steps": sum(e["stats"]["env_steps"] for e in every),
                    "device_ids": [e["device_id"] for e in every], "env_id_bases": [e["env_id_base"] for e in every]}
            line.update(aggregate(every, K))
            if world > 1 and not args.no_extras:
                line["config4"] = config4_block(world, [e["cfg4"] for e in every], None)
            if note:
                line["overlap_fallback"] = note
            emit(line)
        rdzv.close()
        return 0

    from cooking_zoo_amd import _native
    from cooking_zoo_amd.vec_env import CookingVecEnv
    env = CookingVecEnv(count, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"],
                        action_scheme="scheme3", num_layouts=256, layout_seed=0, auto_reset=True,
                        device_id=local_rank, env_id_base=begin)
    _LAST_ENV = env
    L, h = _native.lib(), env._h
    env.reset(return_obs=False)
    # this process drives one handle on its GPU: runs of step launches may overlap (cz_set_overlap in include/cookingzoo.h;
    # CZ_CHAIN=0 keeps the launch-boundary ordering with graph replay)
    if overlap:
        env.set_overlap(True)
        if sim and (sim == "1" or int(sim) == rank + 100):         # test hook for the second attempt
            raise RuntimeError("simulated: an overlapped launch gave up waiting for its predecessor")

    # inputs resident in HBM: a ring of int32 [N, A] action tensors, one slot per step (uniform over the 5 scheme3
    # actions); outputs: obs f64 [N, A, F], rewards f64, terminations / truncations u8.  The ring holds a whole number of
    # K-step runs so that every timed region replays one of a few graphs of exactly K launches.
    runs_in_ring = max(1, 256 // K) if K <= 256 else 1
    period = K * runs_in_ring if K <= 256 else 256
    rng = np.random.default_rng(1234 + rank)
    d_actions = env.alloc((period, N, 2), np.int32)
    d_actions.from_host(rng.integers(0, 5, size=(period, N, 2), dtype=np.int32))
    d_obs = env.alloc((N, 2, env.F), np.float64)
    d_rew = env.alloc((N, 2), np.float64)
    d_term = env.alloc((N, 2), np.uint8)
    d_trunc = env.alloc((N, 2), np.uint8)
    obs_ptr = None if args.no_obs else d_obs.ptr
    ring = (d_actions.ptr, N * 2, period)
    outs = (obs_ptr, d_rew.ptr, d_term.ptr, d_trunc.ptr)

    def run_steps(k, first_slot):
        # k launches of the step kernel, one per env step, issued from C (cz_step_device_ring: graph replay of the run)
        _native.check(h, L.cz_step_device_ring(h, k, *ring, first_slot % period, *outs))

    def first_slot_of(r):
        return (r % runs_in_ring) * K if K <= 256 else 0

    # one-off graph captures outside the measurement (nothing is stepped)
    if Wm > 0:
        _native.check(h, L.cz_ring_prepare(h, Wm, *ring, 0, *outs))
    for r in range(min(R, runs_in_ring)):
        _native.check(h, L.cz_ring_prepare(h, K, *ring, first_slot_of(r), *outs))

    # ---- the communicator: RCCL over xGMI through the C-ABI, under a deadline (helper thread; every rank learns the outcome)
    comm_ok, comm_msg = czd.comm_init_with_deadline(env, world, rank, rdzv, COMM_DEADLINE_S)

    def barrier():
        env.sync()
        rdzv.barrier()
        if comm_ok:
            _native.check(h, L.cz_comm_barrier(h))         # GPU-side: every rank's stream has drained

    if Wm > 0:
        run_steps(Wm, 0)
    L.cz_launch_counts(h, None, None, 1)
    L.cz_chain_counts(h, None, 1)
    elapsed, steps_done = [], []
    for r in range(R):
        s0 = env.stats()["env_steps"]
        barrier()                                              # stream sync + all-rank barrier
        t0 = time.perf_counter()
        run_steps(K, first_slot_of(r))
        env.sync()
        elapsed.append(time.perf_counter() - t0)
        barrier()
        steps_done.append(env.stats()["env_steps"] - s0)       # world steps executed (auto-reset passes are not counted)
    g_k, d_k, c_k = C.c_int64(), C.c_int64(), C.c_int64()
    L.cz_launch_counts(h, C.byref(g_k), C.byref(d_k), 0)
    L.cz_chain_counts(h, C.byref(c_k), 0)

    # dominant-kernel duration: HIP events on the kernels' own stream around max(R*K, 4000) launches issued back to back (the
    # host <-> GPU round trip of a synchronised region, ~20 us on this platform whatever K is, is not kernel time)
    ev_ms = C.c_float()
    n_ev = max(R * K, 4000)
    _native.check(h, L.cz_ring_prepare(h, n_ev, *ring, 0, *outs))
    barrier()
    _native.check(h, L.cz_timer_start(h))
    run_steps(n_ev, 0)                                         # the ring end to end, as few replays as possible
    _native.check(h, L.cz_timer_stop(h, C.byref(ev_ms)))       # event after the last launch, synchronised (fails if a hand-off was abandoned)
    kernel_us = [ev_ms.value * 1e3 / n_ev]
    # the same launches ordered by launch boundaries only (overlap switched off for this pass): the duration of one kernel
    # when nothing runs beside it, which is what a per-kernel trace of such a run shows
    was = max(L.cz_set_overlap(h, 0), 0)
    _native.check(h, L.cz_ring_prepare(h, n_ev, *ring, 0, *outs))
    barrier()
    _native.check(h, L.cz_timer_start(h))
    run_steps(n_ev, 0)
    _native.check(h, L.cz_timer_stop(h, C.byref(ev_ms)))
    kernel_us.append(ev_ms.value * 1e3 / n_ev)
    L.cz_set_overlap(h, was)
    overlapped = bool(c_k.value)
    mine = {"elapsed_s": elapsed, "env_steps": steps_done, "kernel_us": kernel_us, "stats": env.stats()}
    every = [json.loads(b) for b in rdzv.all_gather(json.dumps(mine).encode())]

    # ---- BASELINE config 4 when there are several ranks: 262144 envs over 8 GPUs = 32768 envs per rank, global env ids (the
    # statistics exchange below is that config's only collective); every rank times the same K-step regions between barriers
    cfg4_every = None
    if world > 1 and not args.no_extras:
        K4, R4 = CFG4_K, 5
        base4, n4 = czd.shard_range(CFG4_ENVS_PER_GPU * world, world, rank)
        env4 = CookingVecEnv(n4, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                             num_layouts=256, layout_seed=0, auto_reset=True, device_id=local_rank, env_id_base=base4)
        env4.reset(return_obs=False)
        d_ring4 = env4.alloc((16, n4, 2), np.int32)
        d_ring4.from_host(np.random.default_rng(99 + rank).integers(0, 5, size=(16, n4, 2), dtype=np.int32))
        o4 = (env4.alloc((n4, 2, env4.F), np.float64).ptr, env4.alloc((n4, 2), np.float64).ptr, env4.alloc((n4, 2), np.uint8).ptr,
              env4.alloc((n4, 2), np.uint8).ptr)
        h4 = env4._h
        _native.check(h4, L.cz_ring_prepare(h4, K4, d_ring4.ptr, n4 * 2, 16, 0, *o4))
        _native.check(h4, L.cz_step_device_ring(h4, 16, d_ring4.ptr, n4 * 2, 16, 0, *o4))
        el4, st4 = [], []
        for r in range(R4):
            s0 = env4.stats()["env_steps"]
            env4.sync(); barrier()
            t0 = time.perf_counter()
            _native.check(h4, L.cz_step_device_ring(h4, K4, d_ring4.ptr, n4 * 2, 16, 0, *o4))
            env4.sync()
            el4.append(time.perf_counter() - t0)
            barrier()
            st4.append(env4.stats()["env_steps"] - s0)
        cfg4_every = [json.loads(b) for b in rdzv.all_gather(json.dumps({"elapsed_s": el4, "env_steps": st4, "kernel_us": [0.0] * R4,
                                                                         "env_id_base": base4, "envs": n4}).encode())]
        b_alg4 = algorithmic_bytes_per_env_step(env4)
        env4.close()

    # ---- episode statistics: RCCL all-gather over xGMI of one cz_stats per rank (the path's only collective), checked
    # against the same structs exchanged over the control plane
    stats_all = {}
    rc = 0
    if comm_ok:
        done, res = with_deadline(lambda: czd.allgather_stats_rccl(env, world), COMM_DEADLINE_S)
        if not done:
            stats_all["cz_stats_allgather"] = f"timed out after {COMM_DEADLINE_S:.0f} s"
            rc = EXIT_COMM_FAILED
        elif isinstance(res, Exception):
            stats_all["cz_stats_allgather"] = "failed: " + str(res)
            rc = EXIT_COMM_FAILED
        else:
            same = res == [e["stats"] for e in every]
            stats_all["cz_stats_allgather"] = "ok, identical to the control-plane gather" if same else "ok, but DIFFERENT from the control-plane gather"
            stats_all["total"] = czd.reduce_stats(res)
            if not same:
                rc = EXIT_COMM_FAILED
    else:
        stats_all["cz_stats_allgather"] = "communicator not available: " + comm_msg
        stats_all["total"] = czd.reduce_stats([e["stats"] for e in every])
        rc = EXIT_COMM_FAILED
    flags = [int(b) for b in rdzv.all_gather(str(rc).encode())]     # every rank leaves with the same code
    rc = max(flags)

    if rank == 0:
        agg = aggregate(every, K)
        kernel_med = kernel_us[0]
        # the floor of a launch that has to emit this much output: same grid shape, nothing but the stores
        out_only_us = None
        if not args.no_obs:
            us = C.c_float()
            if L.cz_probe_output_only(h, d_obs.ptr, N * 2 * env.F * 8, 500, C.byref(us)) == 0:
                out_only_us = float(us.value)
        # secondary figure: the same work fused, T steps per launch with the on-device action stream and a trajectory buffer
        fused = None if args.no_obs else guarded(leg_fused, env, K)
        if fused is not None and "error" in fused:
            line_fused_error, fused = fused, None
        else:
            line_fused_error = None
        rccl_path, hip_path = C.create_string_buffer(512), C.create_string_buffer(512)
        L.cz_runtime_paths(rccl_path, hip_path, 512)
        b_alg = algorithmic_bytes_per_env_step(env)
        achieved = b_alg * N / (kernel_med * 1e-6) / 1e9
        total_k = g_k.value + d_k.value + c_k.value
        api = (f"cz_step_device_ring: one kernel launch per env step; of the {total_k} timed launches {c_k.value} went out as "
               f"overlapped launches (two streams alternately, each env's step ordered after that env's previous step by a "
               f"sequence word instead of a launch boundary), {g_k.value} were replayed from HIP graphs of "
               f"{K if K <= 256 else 'up to 256'} launches and {d_k.value} launched directly; "
               f"actions/obs/rewards/flags resident in HBM")
        line = {
            "metric": "env-steps/sec at N parallel envs (1/2/4/8 GPU) + achieved HBM GB/s",
            "value": agg["value"], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": agg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 state, f64 obs+reward", "data": "synthetic",
            "repeats": agg["repeats"], "value_min": agg["value_min"], "value_max": agg["value_max"],
            "ms_per_step_min": agg["ms_per_step_min"], "ms_per_step_max": agg["ms_per_step_max"],
            "timing": "median over `repeats` K-step regions, each bracketed by stream sync + all-rank barrier, MAX over ranks per region",
            "config": {"workload": f"{N} envs/GPU x {world} GPU, coop_test, 2 agents, [TomatoLettuceSalad, CarrotBanana], scheme3, "
                                   f"max_steps=400, f64 obs F={env.F}",
                       "details": "256-layout pool, on-device next-step auto-reset, feature_vector observation encoded every step, uniform random "
                                  "actions resident in HBM, device-resident outputs",
                       "envs_per_gpu": N, "parallelism": f"env-sharded x{world}, one wavefront per env, one process per GPU, no torch",
                       "api": api},
            # The dominant kernel's roofline is quoted from the launch-boundary-ordered kernel, cz::k_step<1,1,2,3,0>: its HIP-event
            # launch duration agrees with rocprofv3's per-kernel average for that kernel (profiles/r04/kernel_stats_ordered.csv).  When
            # the timed regions ran as overlapped launches, the launch-to-launch interval of those is given next to it, with
            # the device-clock timeline that shows it (a per-kernel trace cannot: two of those kernels are resident at a time).
            "roofline": {"bound": "hbm", "achieved": b_alg * N / (kernel_us[1] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": b_alg * N / (kernel_us[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(f"k_step_{N}"),
                         "kernel": "cz::k_step<1,1,2,3,0> (one wavefront per env, 8 envs per workgroup)",
                         "kernel_us": kernel_us[1],
                         "kernel_us_from": (f"HIP events on the kernels' stream around {max(R * K, 4000)} launches issued back to back (graph replay, "
                                            f"ordered by launch boundaries), divided by the number of launches; rocprofv3 --kernel-trace of the same "
                                            f"launches: profiles/r04/kernel_stats_ordered.csv"),
                         "traffic_from": "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE separately) of the same kernel: profiles/r04/traffic.json",
                         "alg_bytes_per_env_step": b_alg, "units_per_launch": N,
                         "overlapped_launch_interval_us": kernel_med if overlapped else None,
                         "overlapped_frac": (achieved / HBM_PEAK_GBS) if overlapped else None,
                         "overlapped_from": ("HIP events around the same number of OVERLAPPED launches (two streams alternately, per-env sequence words: "
                                             "cz::k_step_chain<1,1,2,3>); the device-clock timeline of such a run - start-to-start interval of "
                                             "consecutive launches - is profiles/r04/timeline_overlapped.json") if overlapped else None,
                         # measured on this box, same run: a kernel of the same grid shape that ONLY writes the observation
                         # bytes (write-through 16-byte stores); write-only traffic does not reach the 8 TB/s read+write peak
                         "output_only_launch_us": out_only_us,
                         "frac_of_output_only_launch": (out_only_us / kernel_us[1]) if out_only_us else None},
            "achieved_hbm_gbs_end_to_end": b_alg * agg["value"] / 1e9 / world,
            "runtime": {"hip": hip_path.value.decode(), "rccl": rccl_path.value.decode(), "torch_imported": "torch" in sys.modules},
            "episode_stats_allgather": stats_all,
        }
        if note:
            line["overlap_fallback"] = note
        if line_fused_error is not None:
            line["fused_rollout"] = line_fused_error
        if fused is not None:
            line["fused_rollout"] = fused
            fused["roofline"] = roofline_block(b_alg, N, fused["ms_per_step"] * 1e3, "cz::k_step<1,1,2,3,1> (32 steps per launch)",
                                               "wall clock around the cz_rollout launches")
        if cfg4_every is not None:
            line["config4"] = config4_block(world, cfg4_every, b_alg4)
        if world == 1 and not args.no_extras and not args.no_obs:
            # what users get beside the open-loop headline; each leg a fraction of a second of GPU time.
            # An extra leg must never be able to lose the headline measured above: whatever it raises is recorded under its key.
            line["fused_actions"] = guarded(leg_fused_actions, env, K)
            line["fused_compact"] = guarded(leg_fused_compact, env, K)
            line["ring_fused"] = guarded(leg_ring_fused, env, K, d_obs, d_rew, d_term, d_trunc)
            line["closed_loop"] = guarded(leg_closed_loop, env, d_obs, d_rew, d_term, d_trunc)
            line["closed_loop_compact"] = guarded(leg_closed_loop_compact, env, d_rew, d_term, d_trunc)
            line["cooking_policy"] = guarded(leg_cooking_policy, local_rank)
            line["configs"] = guarded(leg_configs, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = guarded(cpu_baseline, env)
        emit(line)
    try:
        rdzv.close()
    except Exception:
        pass
    sys.stdout.flush()
    if rc != 0:
        sys.stdout.flush()
        os._exit(rc)                             # a communicator stuck in bring-up cannot be torn down: leave, visibly failed
    env.close()
    _LAST_ENV = None
    return 0


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not started by a launcher: be the launcher.  This process never touches the GPU (it does not even load the
        # library); every rank is a fresh interpreter that pins its own device.
        from cooking_zoo_amd.distributed import launch_local
        sys.exit(launch_local(args.gpus, [os.path.abspath(__file__), *sys.argv[1:]], timeout=3600.0))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
