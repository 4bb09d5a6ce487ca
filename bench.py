#!/usr/bin/env python3
"""bench.py -- env-steps/s of the CookingZoo step() hot path on MI355X (BASELINE.json metric).

A "step" is one batched env step (one cz_step_device launch) over every env of the rank: world dynamics,
recipe checks, rewards and the float64 feature-vector encode of all agents, with actions and outputs resident
in HBM.  Workload at N GPUs = BASELINE config 2 per GPU (weak scaling): 4096 envs, level coop_test, 2 agents,
recipes [TomatoLettuceSalad, CarrotBanana], scheme3, max_steps 400, a pool of 256 layouts, next-step
auto-reset.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_env_step(env):
    """SURVEY.md 8(d):  A(8F + 8 + 1 + 1) written + 4A actions read + 2*S_dyn + W*H static cells read,
    S_dyn = 4A + 4D + action objects + linked + 4 (t) + ceil(nodes/8).  For cfg 2 this is 4655 B."""
    A, F, D = env.num_agents, env.F, 12 if env.dims.D <= 12 else env.dims.D
    lay = env.layouts[0]
    n_action = len(lay.static_lists.get("Cutboard", [])) + len(lay.static_lists.get("Blender", []))
    n_action = max(n_action, 4) if env.dims.C == 49 else n_action
    n_linked = len(lay.static_lists.get("Switch", [])) + len(lay.static_lists.get("Block", []))
    nodes = sum(int(env.recipe_table[r][0]) for r in env.recipe_ids[0][:env.num_recipes])
    s_dyn = 4 * A + 4 * D + n_action + n_linked + 4 + (nodes + 7) // 8
    return A * (8 * F + 10) + 4 * A + 2 * s_dyn + env.dims.C


def cpu_baseline(env, seconds_target=12.0):
    """The oracle (a bit-exact C port of the reference step path, oracle/cz_oracle.c) timed on this box's host
    cores on a bounded sample of the same workload: per thread a disjoint slice of envs, 400-step episodes with
    the same counter-based action stream, observations encoded every step."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle_binding import VecOracle
    cores = len(os.sched_getaffinity(0))
    envs_per_thread, T = 8, 400
    # calibrate on one thread
    vo = VecOracle.from_vec_env(env, num_envs=envs_per_thread)
    vo.reset()
    t0 = time.perf_counter()
    vo.rollout(T, 0)
    one = envs_per_thread * T / (time.perf_counter() - t0)
    reps = max(1, int(seconds_target * one / (envs_per_thread * T)))
    reps = min(reps, 64)
    workers = [VecOracle.from_vec_env(env, num_envs=envs_per_thread, env_id_base=i * envs_per_thread) for i in range(cores)]
    for w in workers:
        w.reset()

    def run(w):
        for r in range(reps):
            w.rollout(T, 0, r * T)

    ths = [threading.Thread(target=run, args=(w,)) for w in workers]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    total = cores * envs_per_thread * T * reps
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {envs_per_thread} envs x {T * reps} steps of the bench workload "
                      f"({total} env-steps, {dt:.1f} s; single-thread rate {one:.0f} env-steps/s)"}


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
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from cooking_zoo_amd import _native
    from cooking_zoo_amd.vec_env import CookingVecEnv

    N = args.envs
    K, Wm = args.steps, args.warmup
    env = CookingVecEnv(N, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"],
                        action_scheme="scheme3", num_layouts=256, layout_seed=0, auto_reset=True,
                        device_id=local_rank, env_id_base=rank * N)
    L, h = _native.lib(), env._h
    env.reset(return_obs=False)

    # inputs resident in HBM: one int32 [N, A] action tensor per step (uniform over the 5 scheme3 actions,
    # counter-based stream keyed by the global env id), outputs: obs f64 [N, A, F], rewards, flags
    chunk = 256
    acts = np.empty((chunk, N, 2), dtype=np.int32)
    e_ids = (rank * N + np.arange(N)).astype(np.int64)
    rng = np.random.default_rng(1234 + rank)
    acts[:] = rng.integers(0, 5, size=acts.shape, dtype=np.int32)
    d_actions = env.alloc((chunk, N, 2), np.int32)
    d_actions.from_host(acts)
    d_obs = env.alloc((N, 2, env.F), np.float64)
    d_rew = env.alloc((N, 2), np.float64)
    d_term = env.alloc((N, 2), np.uint8)
    d_trunc = env.alloc((N, 2), np.uint8)
    step_bytes = N * 2 * 4

    def run_steps(k, first):
        for t in range(first, first + k):
            rc = L.cz_step_device(h, d_actions.ptr + (t % chunk) * step_bytes, None if args.no_obs else d_obs.ptr, d_rew.ptr, d_term.ptr, d_trunc.ptr)
            if rc:
                _native.check(h, rc)

    def barrier():
        env.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    run_steps(Wm, 0)
    barrier()
    s0 = env.stats()["env_steps"]
    t0 = time.perf_counter()
    run_steps(K, Wm)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    s1 = env.stats()["env_steps"]
    local_env_steps = s1 - s0            # world steps actually executed (reset passes are not counted)

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([local_env_steps], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_env_steps = int(cnt.item())
    else:
        total_env_steps = local_env_steps

    # dominant-kernel duration, measured live with HIP events on the stream the kernel runs on
    L.cz_kernel_time_reset(h, 1)
    kl = min(K, 500)
    run_steps(kl, Wm + K)
    import ctypes as C
    ms, nl = C.c_double(), C.c_int64()
    L.cz_kernel_time_read(h, C.byref(ms), C.byref(nl))
    L.cz_kernel_time_reset(h, 0)
    kernel_us = ms.value * 1e3 / max(1, nl.value)

    # episode statistics: RCCL all-gather over xGMI of one cz_stats per rank (the path's only collective)
    stats_all = None
    if dist is not None:
        try:
            import torch
            uid = (C.c_uint8 * 128)()
            if rank == 0:
                _native.check(None, L.cz_comm_unique_id(uid))
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
            _native.check(h, L.cz_comm_init(h, world, rank, uid))
            out = (_native.CzStats * world)()
            _native.check(h, L.cz_stats_allgather(h, out))
            stats_all = [out[i].as_dict() for i in range(world)]
        except Exception as exc:                      # keep the timing result even if the stats exchange fails
            stats_all = {"error": str(exc)}

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
                                   f"256-layout pool, on-device auto-reset, feature_vector obs F={env.F} f64",
                       "envs_per_gpu": N, "parallelism": f"env-sharded x{world}, one wavefront per env",
                       "api": "cz_step_device, one launch per step, actions/obs/rewards resident in HBM"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_step<1,1>", "kernel_us": kernel_us, "alg_bytes_per_env_step": b_alg,
                         "units_per_launch": N},
            "achieved_hbm_gbs_end_to_end": b_alg * value / 1e9 / world,
        }
        if stats_all is not None:
            line["episode_stats_allgather"] = stats_all
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(env)
        print(json.dumps(line))
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
