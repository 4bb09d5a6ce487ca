#!/usr/bin/env python3
"""bench.py -- env-steps/s of the CookingZoo step() hot path on MI355X (BASELINE.json metric).

A "step" is one batched env step (one `cz_step_device` launch) over every env of the rank: world dynamics, recipe
checks, rewards and the float64 feature-vector encode of all agents, with actions and outputs resident in HBM.
Workload at N GPUs = BASELINE config 2 per GPU (weak scaling): 4096 envs, level coop_test, 2 agents, recipes
[TomatoLettuceSalad, CarrotBanana], scheme3, max_steps 400, a pool of 256 layouts, next-step auto-reset, uniform
random actions.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes_per_env_step(env):
    """SURVEY.md 8(d):  A(8F + 8 + 1 + 1) written + 4A actions read + 2*S_dyn + W*H static cells read,
    S_dyn = 4A + 4D + action objects + linked + 4 (t) + ceil(nodes/8).  4655 B for config 2 (D = 12: 10 objects + 2
    Bread clones, 4 action objects: 3 Cutboards + 1 Blender, 8 recipe nodes)."""
    A, F, C = env.num_agents, env.F, env.dims.C
    D = max(l.slots_used for l in env.layouts)
    n_action = max(len(l.static_lists.get("Cutboard", [])) + len(l.static_lists.get("Blender", [])) for l in env.layouts)
    n_linked = max(len(l.static_lists.get("Switch", [])) + len(l.static_lists.get("Block", [])) for l in env.layouts)
    nodes = sum(int(env.recipe_table[r][0]) for r in env.recipe_ids[0][:env.num_recipes])
    s_dyn = 4 * A + 4 * D + n_action + n_linked + 4 + (nodes + 7) // 8
    return A * (8 * F + 10) + 4 * A + 2 * s_dyn + C


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
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/cz_oracle.c, {cores} threads x {envs_per_thread} envs x {T * reps} steps of the bench "
                      f"workload incl. obs encode ({total} env-steps in {dt:.1f} s); one thread alone: {single:.0f} env-steps/s"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="env instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-obs", action="store_true", help="ablation only: skip the observation encode (INVALID as a result)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    if world > 1 or os.environ.get("CZ_BENCH_FORCE_DIST"):      # (the variable exercises the multi-rank code path on one GPU)
        # torch first, and torch owns the GPU runtime of this process: its wheel bundles libamdhip64.so.7 / librccl.so.1
        # under the system libraries' SONAMEs, whichever copy is loaded first serves everybody, and only this order is
        # clean (library first: RCCL bring-up fails on the mixed stack and the process aborts in a destructor at exit;
        # measured).  Cost: 1-4 % per step against the system runtime of the single-GPU run.
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from cooking_zoo_amd import _native, distributed as czd
    from cooking_zoo_amd.vec_env import CookingVecEnv

    N = args.envs
    K, Wm = args.steps, args.warmup
    begin, count = czd.shard_range(N * world, world, rank)
    env = CookingVecEnv(count, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"],
                        action_scheme="scheme3", num_layouts=256, layout_seed=0, auto_reset=True,
                        device_id=local_rank, env_id_base=begin)
    L, h = _native.lib(), env._h
    env.reset(return_obs=False)

    # inputs resident in HBM: one int32 [N, A] action tensor per step (uniform over the 5 scheme3 actions);
    # outputs: obs f64 [N, A, F], rewards f64, terminations / truncations u8
    chunk = 256
    rng = np.random.default_rng(1234 + rank)
    d_actions = env.alloc((chunk, N, 2), np.int32)
    d_actions.from_host(rng.integers(0, 5, size=(chunk, N, 2), dtype=np.int32))
    d_obs = env.alloc((N, 2, env.F), np.float64)
    d_rew = env.alloc((N, 2), np.float64)
    d_term = env.alloc((N, 2), np.uint8)
    d_trunc = env.alloc((N, 2), np.uint8)
    obs_ptr = None if args.no_obs else d_obs.ptr

    def run_steps(k, first):
        # k launches of the step kernel, one per env step, issued from C; step j reads slot (first + j) % chunk of the
        # action ring (aligned runs of 32 launches are replayed from HIP graphs: cz_step_device_ring)
        rc = L.cz_step_device_ring(h, k, d_actions.ptr, N * 2, chunk, first % chunk, obs_ptr, d_rew.ptr, d_term.ptr, d_trunc.ptr)
        if rc:
            _native.check(h, rc)

    # one-off graph capture outside the measurement (it steps nothing)
    _native.check(h, L.cz_ring_prepare(h, d_actions.ptr, N * 2, chunk, obs_ptr, d_rew.ptr, d_term.ptr, d_trunc.ptr))

    def barrier():
        env.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    run_steps(Wm, 0)
    barrier()
    s0 = env.stats()["env_steps"]
    L.cz_timer_start(h)                                        # HIP event on the stream the kernels run on
    t0 = time.perf_counter()
    run_steps(K, Wm)
    ev_ms = C.c_float()
    L.cz_timer_stop(h, C.byref(ev_ms))                         # HIP event after the K-th launch, synchronised
    barrier()
    elapsed = time.perf_counter() - t0
    local_env_steps = env.stats()["env_steps"] - s0            # world steps executed (auto-reset passes are not counted)

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([local_env_steps], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_env_steps = int(cnt.item())
    else:
        total_env_steps = local_env_steps

    # the floor of a launch that has to emit this much output: same grid shape, nothing but the stores
    out_only_us = None
    if not args.no_obs:
        us = C.c_float()
        out_bytes = N * 2 * env.F * 8
        if L.cz_probe_output_only(h, d_obs.ptr, out_bytes, 500, C.byref(us)) == 0:
            out_only_us = float(us.value)

    # dominant-kernel duration: HIP events bracket the timed region on the kernels' own stream; the K launches run
    # back to back (rocprofv3 --kernel-trace shows no gaps, profiles/), so duration = event time / K
    kernel_us = ev_ms.value * 1e3 / K

    # secondary figure: the same work fused, T steps per launch with the on-device action stream and a trajectory buffer
    fused = None
    if not args.no_obs:
        T = 32
        d_traj = env.alloc((T, N, 2, env.F), np.float64)
        d_r = env.alloc((T, N, 2), np.float64)
        d_te = env.alloc((T, N, 2), np.uint8)
        d_tr = env.alloc((T, N, 2), np.uint8)
        reps = max(2, K // T)
        env.rollout(T, 1, 0, d_traj, d_r, d_te, d_tr)
        barrier()
        f0 = env.stats()["env_steps"]
        t0 = time.perf_counter()
        for r in range(reps):
            env.rollout(T, 1, (r + 1) * T, d_traj, d_r, d_te, d_tr)
        barrier()
        dtf = time.perf_counter() - t0
        fused = {"env_steps_per_s_per_gpu": (env.stats()["env_steps"] - f0) / dtf, "steps_per_launch": T,
                 "ms_per_step": dtf * 1e3 / (reps * T), "api": "cz_rollout, obs trajectory [T][N][A][F] in HBM"}
        for b in (d_traj, d_r, d_te, d_tr):
            b.free()

    # episode statistics: RCCL all-gather over xGMI of one cz_stats per rank (the path's only collective).
    # Done twice: through torch.distributed (backend nccl = RCCL) and through the C-ABI's own communicator
    # (cz_comm_init / cz_stats_allgather); the second runs under a watchdog so that a stuck communicator bring-up on an
    # unfamiliar node can never cost the timing result.
    stats_all = None
    rccl_hung = False
    if dist is not None:
        try:
            per_rank_t = czd.gather_stats_torch(env.stats(), device=torch.device("cuda", local_rank))
            stats_all = {"via": "torch.distributed nccl", "total": czd.reduce_stats(per_rank_t)}
        except Exception as exc:                 # keep the timing result even if the stats exchange fails
            per_rank_t = None
            stats_all = {"error": str(exc)}
        box = {}

        def direct():
            try:
                torch.cuda.set_device(local_rank)            # the current device is per thread
                def bcast(payload):
                    b = [payload]
                    dist.broadcast_object_list(b, src=0)
                    return b[0]
                box["per_rank"] = czd.gather_stats_rccl(env, world, rank, bcast)
            except Exception as exc:
                box["error"] = str(exc)
        th = threading.Thread(target=direct, daemon=True)
        th.start()
        th.join(timeout=120.0)
        flag = torch.tensor([1 if th.is_alive() else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)          # every rank takes the same exit path
        rccl_hung = bool(flag.item())
        if th.is_alive():
            stats_all["cz_stats_allgather"] = "timed out after 120 s (result above is from torch.distributed)"
        elif "error" in box:
            stats_all["cz_stats_allgather"] = "failed: " + box["error"]
        else:
            same = per_rank_t is not None and box["per_rank"] == per_rank_t
            stats_all["cz_stats_allgather"] = "ok, identical to the torch.distributed result" if same else "ok"
            if not same:
                stats_all["total_direct"] = czd.reduce_stats(box["per_rank"])

    if rank == 0:
        b_alg = algorithmic_bytes_per_env_step(env)
        achieved = b_alg * N / (kernel_us * 1e-6) / 1e9
        value = total_env_steps / elapsed
        line = {
            "metric": "env-steps/sec at N parallel envs (1/2/4/8 GPU) + achieved HBM GB/s",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 state, f64 obs+reward", "data": "synthetic",
            "config": {"workload": f"{N} envs per GPU x {world} GPU(s), level=coop_test, 2 agents, "
                                   f"recipes=[TomatoLettuceSalad, CarrotBanana], scheme3, max_steps=400, "
                                   f"256-layout pool, on-device auto-reset, feature_vector obs F={env.F} f64, "
                                   f"uniform random actions",
                       "envs_per_gpu": N, "parallelism": f"env-sharded x{world}, one wavefront per env",
                       "api": "cz_step_device_ring: one kernel launch per env step (runs of 32 launches replayed from HIP graphs), actions/obs/rewards/flags resident in HBM"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(f"k_step_{N}"),
                         "kernel": "cz::k_step<1,1,2,3,false> (one wavefront per env, 8 envs per workgroup)", "kernel_us": kernel_us, "alg_bytes_per_env_step": b_alg,
                         "units_per_launch": N,
                         # measured on this box, same run: a kernel of the same grid shape that ONLY writes the observation
                         # bytes (write-through 16-byte stores); write-only traffic does not reach the 8 TB/s read+write peak
                         "output_only_launch_us": out_only_us,
                         "frac_of_output_only_launch": (out_only_us / kernel_us) if out_only_us else None},
            "achieved_hbm_gbs_end_to_end": b_alg * value / 1e9 / world,
        }
        if fused is not None:
            line["fused_rollout"] = fused
        if stats_all is not None:
            line["episode_stats_allgather"] = stats_all
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(env)
        print(json.dumps(line), flush=True)
    if rccl_hung:
        os._exit(0)                              # a communicator stuck in bring-up cannot be torn down cleanly
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
