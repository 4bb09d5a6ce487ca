"""GPU: the direct RCCL path of the C-ABI with a one-rank communicator (the multi-rank case needs >1 GPU)."""
import pytest

pytestmark = pytest.mark.gpu


def test_stats_allgather_single_rank():
    from cooking_zoo_amd import distributed as czd
    from cooking_zoo_amd.vec_env import CookingVecEnv
    env = CookingVecEnv(256, "coop_test", "example", 2, 20, ["TomatoLettuceSalad", "CarrotBanana"],
                        action_scheme="scheme3", num_layouts=8, auto_reset=True)
    env.reset(return_obs=False)
    env.rollout(50, 3)
    env.sync()
    local = env.stats()
    gathered = czd.gather_stats_rccl(env, 1, 0, lambda payload: payload)
    assert gathered == [local]
    assert czd.reduce_stats(gathered)["env_steps"] == local["env_steps"] > 0
    assert local["episodes"] >= 256 * 2
    env.close()


def test_comm_barrier_and_runtime_paths_single_rank():
    import ctypes as C
    from cooking_zoo_amd import _native, distributed as czd
    from cooking_zoo_amd.vec_env import CookingVecEnv
    import tempfile
    env = CookingVecEnv(64, "coop_test", "example", 2, 20, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                        num_layouts=4, auto_reset=True)
    env.reset(return_obs=False)
    L = _native.lib()
    assert L.cz_comm_barrier(env._h) != 0 and b"not initialised" in L.cz_last_error(env._h)
    rv = czd.FileRendezvous(tempfile.mkdtemp(prefix="cz_t_"), 0, 1, timeout=30.0)
    ok, msg = czd.comm_init_with_deadline(env, 1, 0, rv, 120.0)
    assert ok, msg
    env.rollout(30, 5)
    _native.check(env._h, L.cz_comm_barrier(env._h))
    assert czd.allgather_stats_rccl(env, 1) == [env.stats()]
    rccl, hip = C.create_string_buffer(512), C.create_string_buffer(512)
    L.cz_runtime_paths(rccl, hip, 512)
    assert b"librccl" in rccl.value and b"libamdhip64" in hip.value
    rv.close()
    env.close()


def test_bench_single_gpu_goes_through_the_multi_rank_code():
    """`python bench.py --gpus 1` at the driver's K = 20: no torch, the batch is a ShardedVecEnv whose RCCL communicator (1 rank)
    serves the barrier and the statistics all-gather, and the line says how its launches went out: every timed launch the
    boundary-ordered one-step kernel, replayed from graphs - the kernel the roofline block is about."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "7",
                        "--envs", "1024", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "timed launches 0 went out as overlapped launches" in d["config"]["api"]
    assert "0 were replayed from HIP graphs of 20 launches and 140 launched directly" in d["config"]["api"]      # (runs under 48 launches: direct)
    assert "k_step<1,1,2,3,0>" in d["roofline"]["kernel"] and "RCCL all-gather" in d["config"]["parallelism"]
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["repeats"] == 7
    assert d["episode_stats_allgather"]["cz_stats_allgather"].startswith("ok, identical")
    assert d["runtime"]["torch_imported"] is False and "/opt/rocm" in d["runtime"]["hip"]
    assert d["value"] > 1e6 and d["value_min"] <= d["value"] <= d["value_max"]
    assert 0 < d["roofline"]["frac"] < 1
    # the kernel's own time cannot exceed the region's time per step (one kernel per headline)
    assert d["roofline"]["kernel_us"] <= d["ms_per_step"] * 1e3 * 1.05


def test_bench_driver_command_exits_zero_with_every_leg():
    """The driver's own command, extras included (`python bench.py --gpus 1 --steps 20 --warmup 5`, only the CPU baseline left out):
    exit code 0, and the RCCL all-gather of the statistics equals the control-plane copy although the sustained stretch steps the
    batch after the timed regions (BENCH_r05 left with rc 3: the copy was taken before that stretch)."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["episode_stats_allgather"]["cz_stats_allgather"] == "ok, identical to the control-plane gather"
    assert d["sustained"]["launches"] >= 2000 and d["sustained"]["seconds"] >= 2.0
    # the statistics describe everything the batch was stepped through, the sustained stretch included
    assert d["episode_stats_allgather"]["total"]["env_steps"] >= int(4096 * d["sustained"]["launches"] * 0.9)
    for leg in ("fused_rollout", "fused_actions", "fused_compact", "ring_fused", "closed_loop", "closed_loop_compact",
                "closed_loop_caller_graph", "cooking_policy", "configs"):
        assert leg in d and "error" not in d[leg], (leg, d.get(leg))
    c4 = d["configs"]["config4_on_one_gpu"]                       # config 4's whole batch as eight shards on this one device
    assert "error" not in c4 and c4["envs"] == 262144 and c4["shards"][3] == [98304, 32768] and c4["env_steps_per_s"] > 1e8, c4


def test_two_ranks_on_one_device_agree_on_the_outcome():
    """Two rank processes that both open device 0 (all a one-GPU box can offer): RCCL refuses duplicate devices
    (ncclInvalidUsage), and the bring-up must end the same way on every rank - a refusal all ranks report, nobody hanging in a
    collective - or, should an RCCL accept it, with an all-gather that equals the ranks' local statistics."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "rccl_two_ranks_one_gpu.py")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    if d["rccl_two_ranks_on_one_device"] == "refused":
        assert "rank 0:" in d["message"] and "rank 1:" in d["message"]
    else:
        assert d["allgather_equals_local_stats"] is True and d["total_env_steps"] > 0


def test_bench_two_ranks_on_the_one_device():
    """`python bench.py --gpus 2` end to end on a one-GPU box: both ranks are put on device 0 (CZ_BENCH_DEVICE), so everything of the multi-rank
    path runs for real - the launcher, the rendezvous, ShardedVecEnv in its multi-process form, the headline regions between all-rank
    barriers, BASELINE config 4's shape (2 x 32 768 envs) - except the RCCL exchange between distinct devices, which RCCL refuses here: the
    statistics come over the host path, the line says so, and the exit code is the communicator-failed one (3), on every rank."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CZ_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CZ_RDZV_DIR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--repeats", "5",
                        "--envs", "2048", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 3 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 1e6
    assert "communicator not available" in d["episode_stats_allgather"]["cz_stats_allgather"]
    assert d["episode_stats_allgather"]["total"]["env_steps"] > 2 * 2048 * 20 * 5
    c4 = d["config4"]
    assert c4["envs"] == 65536 and c4["shards"] == [[0, 32768], [32768, 32768]] and c4["value"] > 1e6
