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

The batch, its shards, the communicator and the statistics exchange are `cooking_zoo_amd.ShardedVecEnv` - the same object a user
gets; this file only times it.

Timing: W warmup steps, then the K-step region is timed `--repeats` times; every region is bracketed by a stream
synchronisation + barrier over all ranks on both sides, its time is the MAX over ranks, and `value` / `ms_per_step` are
the MEDIAN over the regions (min and max are reported next to it).  Every timed launch is the boundary-ordered one-step kernel
`cz::k_step<1,1,2,3,0>` replayed from HIP graphs: `value`, `roofline.kernel_us` and profiles/rNN/kernel_stats.csv (newest round) describe the same
launches.
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
    cpu_model = "unknown"
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:                                             # noqa: BLE001
        pass
    out = {"value": total / dt, "unit": "env-steps/s", "cores": cores, "cpu_model": cpu_model, "kind": "port",
           "sample": f"oracle/cz_oracle.c, {cores} threads x {envs_per_thread} envs x {T * reps} steps of the bench "
                     f"workload incl. obs encode ({total} env-steps in {dt:.1f} s); one thread alone: {single:.0f} env-steps/s"}
    ref = reference_python_timing()
    if ref is not None:
        out["reference_python"] = ref
    return out


def latest_profile(name):
    """repo-relative path of the newest profiles/rNN/<name> that exists (the round's own collection once it has been made)"""
    found = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", name)))
    return os.path.relpath(found[-1], REPO) if found else f"profiles/rNN/{name} (not collected)"


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


CU_COUNT, SHADER_GHZ = 256, 2.4           # MI355X_MICROARCH.md: 256 CUs (8 XCDs x 32), 2.4 GHz


def issue_counts(instance):
    """Instructions per env-step of one kernel instance (scalar / vector / LDS), from the committed rocprofv3 --pmc passes of this
    round or the last one that has them (profiles/rNN/issue_per_env_step.json, tools/collect_round.sh), or None."""
    best = None
    for d in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "issue_per_env_step.json"))):
        try:
            t = json.load(open(d))
            if instance in t:
                best = dict(t[instance], source=os.path.relpath(d, REPO))
        except Exception:
            pass
    return best


def issue_block(instance, env_steps_per_s):
    """The instruction-issue roof next to the HBM one.  A CU has ONE scalar unit (one SALU instruction per
    cycle for all its waves) and four SIMDs whose vector ALU takes a wave64 instruction every 4 cycles:
        salu_util = SALU instructions per env-step x env-steps/s / (256 CUs x 2.4 GHz)
        valu_util = VALU instructions per env-step x 4 cycles x env-steps/s / (256 CUs x 4 SIMDs x 2.4 GHz)
    `frac` is the larger of the two: how much of the binding issue port the leg uses."""
    c = issue_counts(instance)
    if c is None:
        return None
    cap = CU_COUNT * SHADER_GHZ * 1e9
    salu, valu = c["salu"] * env_steps_per_s / cap, c["valu"] * 4.0 * env_steps_per_s / (cap * 4.0)
    return {"salu_per_env_step": c["salu"], "valu_per_env_step": c["valu"], "salu_util": salu, "valu_util": valu, "frac": max(salu, valu),
            "peak": "256 CUs x 2.4 GHz: one scalar instruction per CU and cycle; one wave64 vector instruction per SIMD and 4 cycles",
            "instance": instance, "counts_from": c["source"]}


def roofline_block(b_alg, units, us_per_launch, kernel, note=None, instance=None):
    """algorithmic bytes per env-step x env-steps per launch / launch duration, against the 8 TB/s HBM peak; with
    `instance` (a key of issue_per_env_step.json) also the instruction-issue roof, and `bound` names the one that binds."""
    achieved = b_alg * units / (us_per_launch * 1e-6) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "kernel": kernel, "kernel_us": us_per_launch, "alg_bytes_per_env_step": b_alg, "units_per_launch": units}
    if note:
        out["kernel_us_from"] = note
    if instance:
        out["issue"] = issue_block(instance, units / (us_per_launch * 1e-6))
        if out["issue"] and out["issue"]["frac"] > out["frac"]:
            out["bound"] = "issue"
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
        out["roofline"] = roofline_block(b_alg, N, out["ms_per_step"] * 1e3, "cz::k_step<1,1,2,3,5> (32 steps per launch, codes only)",
                                         "wall clock around the cz_rollout_compact launches; algorithmic bytes with 1 byte per feature", "k_step<1,1,2,3,5>")
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
                                       "HIP events around graph replays of the closed loop; step + policy kernel + both boundaries", "k_step<1,1,2,3,0>")}


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
            "roofline": roofline_block(b_alg, N, us.value, "cz::k_step<1,1,2,3,3> (codes only) + k_probe_policy_codes per step",
                                       "HIP events around graph replays; algorithmic bytes with 1 byte per feature", "k_step<1,1,2,3,3>")}


def leg_closed_loop_captured(env, d_obs, d_rew, d_term, d_trunc):
    """The closed loop as a FRAMEWORK would run it: [policy kernel -> cz_step_device] captured into a graph of the CALLER - raw
    hipStreamBeginCapture on a stream handed to cz_set_stream (what torch.cuda.graph does underneath) - and replayed; the library's own
    cz_probe_closed_loop (leg `closed_loop`) is the same loop captured inside the library.  tests/test_gpu_capture.py checks that the
    replays are bit for bit the eager loop."""
    from cooking_zoo_amd import _native
    L, h = _native.lib(), env._h
    rccl_path, hip_path = C.create_string_buffer(512), C.create_string_buffer(512)
    L.cz_runtime_paths(rccl_path, hip_path, 512)
    hip = C.CDLL(hip_path.value.decode())
    N, A, K, reps = env.num_envs, env.num_agents, 8, 250

    def ck(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with HIP error {rc}")
    stream, graph, gexec, ev0, ev1 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    d_act = env.alloc((N, A), np.int32)
    d_act.from_host(np.random.default_rng(7).integers(0, env.n_actions, size=(N, A), dtype=np.int32))
    ck(hip.hipStreamCreateWithFlags(C.byref(stream), 1), "hipStreamCreateWithFlags")
    try:
        env.observe_device(d_obs)
        env.sync()
        env.set_stream(stream)
        ck(hip.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")            # global mode: the strictest
        for _ in range(K):
            _native.check(h, L.cz_probe_policy(h, d_obs.ptr, None, d_act.ptr))
            env.step_device(d_act, d_obs, d_rew, d_term, d_trunc)
        ck(hip.hipStreamEndCapture(stream, C.byref(graph)), "hipStreamEndCapture")
        ck(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
        for _ in range(5):
            ck(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")
        hip.hipEventCreate(C.byref(ev0)); hip.hipEventCreate(C.byref(ev1))
        s0 = env.stats()["env_steps"]
        hip.hipEventRecord(ev0, stream)
        for _ in range(reps):
            ck(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")
        hip.hipEventRecord(ev1, stream)
        ck(hip.hipEventSynchronize(ev1), "hipEventSynchronize")
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), ev0, ev1)
        us = ms.value * 1e3 / (K * reps)
        stepped = (env.stats()["env_steps"] - s0) / float(N * K * reps)
        return {"env_steps_per_s": stepped * N / (us * 1e-6), "us_per_step": us, "envs": N,
                "what": f"[cz_probe_policy -> cz_step_device] x {K} captured by hipStreamBeginCapture (global mode) on the caller's stream "
                        f"(cz_set_stream), the caller's graph replayed {reps} times, HIP events on that stream; the gap between two graph launches "
                        f"(about a microsecond per step at {K} steps per graph) is part of the figure"}
    finally:
        env.set_stream(None)
        for o, f in ((gexec, hip.hipGraphExecDestroy), (graph, hip.hipGraphDestroy), (ev0, hip.hipEventDestroy), (ev1, hip.hipEventDestroy),
                     (stream, hip.hipStreamDestroy)):
            if o.value:
                f(o)
        d_act.free()


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
                                         "wall clock around the cz_rollout_actions launches", "k_step<1,1,2,3,2>")
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
        # (7.1 of the 18.2 MB of rows per step reach HBM, profiles/r04/ring_fused_pmc.txt: the roof that binds is instruction issue)
        # algorithmic bytes of an in-place run: every step's outputs land on the same rows, only the last step's have to exist
        # afterwards - per env-step the state traffic plus 1/K of one set of outputs
        b_out = A * (8 * env.F + 10)
        out["roofline"] = roofline_block(algorithmic_bytes_per_env_step(env) - b_out + b_out / 2000.0, N, out["K=2000"]["us_per_step"],
                                         "cz::k_step<1,1,2,3,2> (fused over the ring's rows, outputs in place)",
                                         "wall clock of 2000-step regions between synchronisations", "k_step<1,1,2,3,2>/in_place")
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
           "roofline": roofline_block(b_alg, N, us, "cz::k_step<1,1,2,3,0>", "HIP events around one graph-replayed run of 512 launches", "k_step<1,1,2,3,0>/cooking")}
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


def leg_config4_on_one_gpu(device_id):
    """BASELINE config 4's WHOLE batch - 262 144 envs as eight shards of 32 768 with global env ids - on this one GPU (288 GB hold it):
    ShardedVecEnv(device_ids=[d] * 8), i.e. eight handles / streams / host threads, statistics reduced on the host (RCCL refuses
    duplicate devices).  Per env step: eight launches of the one-step kernel, one per shard.  What eight GPUs do in parallel this does
    in turn - the number is a lower bound of the 8-GPU value / 8, and evidence that the configuration runs in its full shape
    (parity of the same shape: tests/test_gpu_large.py::test_config4_full_shape_eight_shards_on_one_device)."""
    from cooking_zoo_amd.sharded import ShardedVecEnv
    G, K, R = 8, 50, 5
    n = CFG4_ENVS_PER_GPU * G
    senv = ShardedVecEnv(n, *WORKLOAD, device_ids=[device_id] * G, comm="host", **WORKLOAD_KW)
    try:
        senv.reset(return_obs=False)
        period = 16
        ring = senv.alloc((2,), np.int32, leading=(period,))
        ring.from_host(np.random.default_rng(7).integers(0, 5, size=(period, n, 2), dtype=np.int32))
        outs = (senv.alloc((2, senv.F), np.float64), senv.alloc((2,), np.float64), senv.alloc((2,), np.uint8), senv.alloc((2,), np.uint8))
        senv.ring_prepare(K, ring, period, 0, *outs)
        senv.step_device_ring(K, ring, period, 0, *outs)
        senv.sync()
        ts = []
        for _ in range(R):
            senv.sync()
            t0 = time.perf_counter()
            senv.step_device_ring(K, ring, period, 0, *outs)
            senv.sync()
            ts.append((time.perf_counter() - t0) * 1e6 / K)
        us = float(np.median(ts))
        b_alg = algorithmic_bytes_per_env_step(senv.shards[0])
        st = senv.stats()
        return {"workload": f"{n} envs = {G} shards of {CFG4_ENVS_PER_GPU} (global env ids) on ONE device, coop_test, 2 agents; {K}-step ring runs, median of {R}",
                "envs": n, "shards": [list(r) for r in senv.ranges], "us_per_step_of_the_whole_batch": us,
                "env_steps_per_s": 400.0 / 401.0 * n / (us * 1e-6), "stats_env_steps": st["env_steps"],
                "roofline": roofline_block(b_alg, n, us, "cz::k_step<1,1,2,3,0> x 8 shards per env step",
                                           "wall clock of the ring runs of all eight shards (eight streams of one device)")}
    finally:
        senv.close()


def guarded(fn, *a, **kw):
    """an extra leg of the line: its result, or {"error": ...} - never an exception that would lose the headline"""
    try:
        return fn(*a, **kw)
    except BaseException as exc:                     # noqa: BLE001  (incl. MemoryError; KeyboardInterrupt is re-raised)
        if isinstance(exc, KeyboardInterrupt):
            raise
        import traceback
        return {"error": f"{type(exc).__name__}: {exc}", "where": traceback.format_exc(limit=3).strip().splitlines()[-3:]}


SUSTAINED_S = 6.0                              # the `sustained` leg: longer than the 5 s between two samples of a utilisation monitor
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


WORKLOAD = ("coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"])      # BASELINE configs 2 and 4
WORKLOAD_KW = dict(action_scheme="scheme3", num_layouts=256, layout_seed=0, auto_reset=True)


def worker(args):
    _redirect_stdout()
    from cooking_zoo_amd import distributed as czd
    rdzv = czd.FileRendezvous.from_env(timeout=600.0)
    try:
        return worker_body(args, rdzv)
    finally:
        try:
            rdzv.close()
        except Exception:                                         # noqa: BLE001
            pass


def _redirect_stdout():
    # Native libraries print to stdout through C stdio (RCCL's version banner, for one), and whether that reaches fd 1
    # before or after this script's own line depends on who flushes when.  So fd 1 is handed to stderr for the whole
    # run, and the JSON line goes to a private duplicate of the real stdout: nothing else can appear there.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def timed_regions(senv, K, R, ring, outs, first_slot_of):
    """R regions of K steps of the whole batch: stream sync + all-rank barrier, K launches per shard (cz_step_device_ring: graph
    replay), stream sync; -> (elapsed seconds, env-steps executed by this process's shards) per region"""
    elapsed, steps_done = [], []
    period = ring.leading[0]
    run = senv.ring_runner(K, ring, period, *outs)               # addresses and strides resolved once: the region holds launches + sync
    for r in range(R):
        s0 = sum(st["env_steps"] for st in senv._each(lambda i, env: env.stats()))
        slot = first_slot_of(r)
        senv.barrier()                                           # stream sync + all-rank barrier
        t0 = time.perf_counter()
        run(slot)
        elapsed.append(time.perf_counter() - t0)
        senv.barrier()
        steps_done.append(sum(st["env_steps"] for st in senv._each(lambda i, env: env.stats())) - s0)   # (auto-reset passes are not counted)
    return elapsed, steps_done


def worker_body(args, rdzv):
    from cooking_zoo_amd import distributed as czd
    from cooking_zoo_amd.sharded import ShardedVecEnv
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if os.environ.get("CZ_BENCH_DEVICE") is not None:       # test hook: every rank on this device (a one-GPU box running N > 1 ranks; the
        local_rank = int(os.environ["CZ_BENCH_DEVICE"])     # statistics then go over the host path and the run exits with EXIT_COMM_FAILED)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr)
    N, K, Wm, R = args.envs, args.steps, args.warmup, max(1, args.repeats)
    sharded = dict(device_ids=[local_rank], world_size=world, rank=rank, rendezvous=rdzv if world > 1 else None)

    if args.dry_run:
        plan = ShardedVecEnv(N * world, *WORKLOAD, dry_run=True, **sharded, **WORKLOAD_KW)
        plan4 = ShardedVecEnv(CFG4_ENVS_PER_GPU * world, *WORKLOAD, dry_run=True, tables=None, **sharded, **dict(WORKLOAD_KW, num_layouts=4))
        (begin, count), (b4, n4) = plan.ranges[0], plan4.ranges[0]
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
                    "shards": [list(x) for x in plan.plan],
                    "stats_total_env_steps": sum(e["stats"]["env_steps"] for e in every),
                    "device_ids": [e["device_id"] for e in every], "env_id_bases": [e["env_id_base"] for e in every]}
            line.update(aggregate(every, K))
            if world > 1 and not args.no_extras:
                line["config4"] = config4_block(world, [e["cfg4"] for e in every], None)
            emit(line)
        return 0

    from cooking_zoo_amd import _native
    # the batch: N envs per GPU x world GPUs, this process's shard on device LOCAL_RANK; RCCL communicator under a deadline
    senv = ShardedVecEnv(N * world, *WORKLOAD, comm_deadline=COMM_DEADLINE_S, **sharded, **WORKLOAD_KW)
    env = senv.shards[0]                                   # (the single-handle legs below talk to this rank's handle)
    L, h = _native.lib(), env._h
    comm_ok, comm_msg = senv.comm_kind == "rccl", senv.comm_note
    senv.reset(return_obs=False)

    # inputs resident in HBM: a ring of int32 [N, A] action tensors, one slot per step (uniform over the 5 scheme3
    # actions); outputs: obs f64 [N, A, F], rewards f64, terminations / truncations u8.  The ring holds a whole number of
    # K-step runs so that every timed region replays one of a few graphs of exactly K launches.
    runs_in_ring = max(1, 256 // K) if K <= 256 else 1
    period = K * runs_in_ring if K <= 256 else 256
    rng = np.random.default_rng(1234 + rank)
    ring = senv.alloc((2,), np.int32, leading=(period,))
    ring.from_host(rng.integers(0, 5, size=(period, senv.local_envs, 2), dtype=np.int32))
    s_obs, s_rew = senv.alloc((2, env.F), np.float64), senv.alloc((2,), np.float64)
    s_term, s_trunc = senv.alloc((2,), np.uint8), senv.alloc((2,), np.uint8)
    outs = (None if args.no_obs else s_obs, s_rew, s_term, s_trunc)
    d_actions, d_obs, d_rew, d_term, d_trunc = ring.parts[0], s_obs.parts[0], s_rew.parts[0], s_term.parts[0], s_trunc.parts[0]

    def first_slot_of(r):
        return (r % runs_in_ring) * K if K <= 256 else 0

    # one-off graph captures outside the measurement (nothing is stepped)
    if Wm > 0:
        senv.ring_prepare(Wm, ring, period, 0, *outs)
    for r in range(min(R, runs_in_ring)):
        senv.ring_prepare(K, ring, period, first_slot_of(r), *outs)
    if Wm > 0:
        senv.step_device_ring(Wm, ring, period, 0, *outs)
    L.cz_launch_counts(h, None, None, 1)
    elapsed, steps_done = timed_regions(senv, K, R, ring, outs, first_slot_of)
    g_k, d_k = C.c_int64(), C.c_int64()
    L.cz_launch_counts(h, C.byref(g_k), C.byref(d_k), 0)

    # dominant-kernel duration: HIP events on the kernels' own stream around max(R*K, 4000) of the SAME launches issued back to back
    # (the host <-> GPU round trip of a synchronised region, ~20 us on this platform whatever K is, is not kernel time)
    ev_ms = C.c_float()
    n_ev = max(R * K, 4000)
    senv.ring_prepare(n_ev, ring, period, 0, *outs)
    senv.barrier()
    _native.check(h, L.cz_timer_start(h))
    env.step_device_ring(n_ev, d_actions, env.num_envs * 2, period, 0, None if args.no_obs else d_obs, d_rew, d_term, d_trunc)
    _native.check(h, L.cz_timer_stop(h, C.byref(ev_ms)))       # event after the last launch, synchronised
    kernel_us = ev_ms.value * 1e3 / n_ev
    mine = {"elapsed_s": elapsed, "env_steps": steps_done, "kernel_us": [kernel_us]}

    # ---- a sustained stretch of the headline's launches (a sampler of GPU activity sees the device busy)
    sustained = None
    if not args.no_extras:
        n_s = max(2000, K)
        senv.barrier()
        t0 = time.perf_counter()
        reps_s = 0
        while time.perf_counter() - t0 < SUSTAINED_S:
            senv.step_device_ring(n_s, ring, period, 0, *outs)
            reps_s += 1
            if reps_s % 8 == 0:
                senv.sync()
        senv.sync()
        dt_s = time.perf_counter() - t0
        sustained = {"seconds": dt_s, "launches": reps_s * n_s, "us_per_step": dt_s * 1e6 / (reps_s * n_s),
                     "env_steps_per_s_per_gpu": 400.0 / 401.0 * senv.local_envs * reps_s * n_s / dt_s,
                     "what": f"runs of {n_s} graph-replayed launches of the same kernel, back to back for {SUSTAINED_S:.0f} seconds, synchronised every 8 runs"}

    # ---- BASELINE config 4 when there are several ranks: 262144 envs over 8 GPUs = 32768 envs per rank, global env ids (the
    # statistics exchange below is that config's only collective); every rank times the same K-step regions between barriers
    cfg4_every = None
    if world > 1 and not args.no_extras:
        K4, R4 = CFG4_K, 5
        senv4 = ShardedVecEnv(CFG4_ENVS_PER_GPU * world, *WORKLOAD, comm="host", **sharded, **WORKLOAD_KW)
        senv4.reset(return_obs=False)
        ring4 = senv4.alloc((2,), np.int32, leading=(16,))
        ring4.from_host(np.random.default_rng(99 + rank).integers(0, 5, size=(16, senv4.local_envs, 2), dtype=np.int32))
        o4 = (senv4.alloc((2, senv4.F), np.float64), senv4.alloc((2,), np.float64), senv4.alloc((2,), np.uint8), senv4.alloc((2,), np.uint8))
        senv4.ring_prepare(K4, ring4, 16, 0, *o4)
        senv4.step_device_ring(16, ring4, 16, 0, *o4)
        el4, st4 = timed_regions(senv4, K4, R4, ring4, o4, lambda r: 0)
        cfg4_every = [json.loads(b) for b in rdzv.all_gather(json.dumps({"elapsed_s": el4, "env_steps": st4, "kernel_us": [0.0] * R4,
                                                                         "env_id_base": senv4.local_begin, "envs": senv4.local_envs}).encode())]
        b_alg4 = algorithmic_bytes_per_env_step(senv4.shards[0])
        senv4.close()

    # ---- episode statistics: RCCL all-gather over xGMI of one cz_stats per shard (the path's only collective), checked
    # against the same structs exchanged over the control plane.  The control-plane copy is taken HERE, after every leg that
    # steps this batch (the sustained stretch above included) and with nothing stepping between it and the RCCL gather -
    # a copy taken earlier describes an older batch and can never compare equal (BENCH_r05 left with rc 3).
    senv.sync()
    senv.barrier()
    mine["stats"] = env.stats()
    every = [json.loads(b) for b in rdzv.all_gather(json.dumps(mine).encode())]
    stats_all = {}
    rc = 0
    if comm_ok:
        done, res = with_deadline(senv.stats_per_shard, COMM_DEADLINE_S)
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
        # the floor of a launch that has to emit this much output: same grid shape, nothing but the stores
        out_only_us = None
        if not args.no_obs:
            us = C.c_float()
            if L.cz_probe_output_only(h, d_obs.ptr, N * 2 * env.F * 8, 500, C.byref(us)) == 0:
                out_only_us = float(us.value)
        rccl_path, hip_path = C.create_string_buffer(512), C.create_string_buffer(512)
        L.cz_runtime_paths(rccl_path, hip_path, 512)
        b_alg = algorithmic_bytes_per_env_step(env)
        api = (f"cooking_zoo_amd.ShardedVecEnv.step_device_ring -> cz_step_device_ring: one launch of cz::k_step<1,1,2,3,0> per env step, ordered "
               f"by launch boundaries; of the {g_k.value + d_k.value} timed launches 0 went out as overlapped launches (that mode was removed in "
               f"round 5), {g_k.value} were replayed from HIP graphs of {K if K <= 256 else 'up to 256'} launches and {d_k.value} launched "
               f"directly; actions/obs/rewards/flags resident in HBM")
        roof = roofline_block(b_alg, N, kernel_us, "cz::k_step<1,1,2,3,0> (one wavefront per env, 8 envs per workgroup)",
                              f"HIP events on the kernels' stream around {n_ev} launches issued back to back (graph replay, ordered by launch "
                              f"boundaries) - the launches `value` times -, divided by the number of launches; rocprofv3 --kernel-trace of the same "
                              f"command: {latest_profile('kernel_stats.csv')}", "k_step<1,1,2,3,0>")
        roof["traffic"] = pmc_traffic(f"k_step_{N}")
        roof["traffic_from"] = f"rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE separately) of the same kernel: {latest_profile('traffic.json')}"
        # measured on this box, same run: a kernel of the same grid shape that ONLY writes the observation bytes (write-through
        # 16-byte stores); write-only traffic does not reach the 8 TB/s read+write peak
        roof["output_only_launch_us"] = out_only_us
        roof["frac_of_output_only_launch"] = (out_only_us / kernel_us) if out_only_us else None
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
                       "envs_per_gpu": N, "parallelism": f"env-sharded x{world} (ShardedVecEnv: {senv.comm_note}), one wavefront per env, one process per GPU, no torch",
                       "api": api},
            "roofline": roof,
            "achieved_hbm_gbs_end_to_end": b_alg * agg["value"] / 1e9 / world,
            "runtime": {"hip": hip_path.value.decode(), "rccl": rccl_path.value.decode(), "torch_imported": "torch" in sys.modules},
            "episode_stats_allgather": stats_all,
        }
        if sustained is not None:
            line["sustained"] = sustained
        if not args.no_obs:
            fused = guarded(leg_fused, env, K)
            if "error" not in fused:
                fused["roofline"] = roofline_block(b_alg, N, fused["ms_per_step"] * 1e3, "cz::k_step<1,1,2,3,1> (32 steps per launch)",
                                                   "wall clock around the cz_rollout launches", "k_step<1,1,2,3,1>")
            line["fused_rollout"] = fused
        if cfg4_every is not None:
            line["config4"] = config4_block(world, cfg4_every, b_alg4)
        if world == 1 and not args.no_extras and not args.no_obs:
            # what users get beside the open-loop headline; each leg a fraction of a second of GPU time.  An extra leg must never be
            # able to lose the headline measured above: whatever it raises is recorded under its key.
            line["fused_actions"] = guarded(leg_fused_actions, env, K)
            line["fused_compact"] = guarded(leg_fused_compact, env, K)
            line["ring_fused"] = guarded(leg_ring_fused, env, K, d_obs, d_rew, d_term, d_trunc)
            line["closed_loop"] = guarded(leg_closed_loop, env, d_obs, d_rew, d_term, d_trunc)
            line["closed_loop_compact"] = guarded(leg_closed_loop_compact, env, d_rew, d_term, d_trunc)
            line["closed_loop_caller_graph"] = guarded(leg_closed_loop_captured, env, d_obs, d_rew, d_term, d_trunc)
            line["cooking_policy"] = guarded(leg_cooking_policy, local_rank)
            line["configs"] = guarded(leg_configs, local_rank)
            if isinstance(line["configs"], dict) and "error" not in line["configs"]:
                line["configs"]["config4_on_one_gpu"] = guarded(leg_config4_on_one_gpu, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = guarded(cpu_baseline, env)
        emit(line)
    sys.stdout.flush()
    try:
        rdzv.barrier()                           # nobody leaves before rank 0 has printed the line (the launcher ends the job with the first failing rank)
    except Exception:                            # noqa: BLE001
        pass
    if rc != 0:
        os._exit(rc)                             # a communicator stuck in bring-up cannot be torn down: leave, visibly failed
    senv.close()
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
