"""GPU: ShardedVecEnv - one batch over several handles.  A one-GPU box can only offer `device_ids=[0, 0]` (two handles, two
streams, two ranges of global env ids on the same device), which exercises everything but the RCCL exchange between distinct
devices: shards plus id offsets must equal ONE handle over the whole batch bit for bit, on every stepping path, with despawn /
respawn on a mixed-level batch and with the layout pool rotated under the running batch."""
import numpy as np
import pytest

from cooking_zoo_amd import soa
from cooking_zoo_amd.sharded import ShardedVecEnv
from cooking_zoo_amd.vec_env import CookingVecEnv

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def strip(recs):
    r = recs.copy()
    r[:, soa.RET_WORD0:soa.RET_WORD0 + 8] = 0
    return r


def same_stats(a, b):
    for k in ("env_steps", "episodes", "length_sum", "truncations", "terminations", "recipes_completed"):
        assert a[k] == b[k], (k, a[k], b[k])
    # (float64 sums: one handle adds its envs in one fixed order, the shards in theirs - equal up to rounding, each reproducible)
    assert np.allclose(a["return_sum"], b["return_sum"], rtol=1e-12, atol=1e-9)


CASES = [
    dict(level="coop_test", meta_file="example", num_agents=2, max_steps=23, recipes=["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3"),
    dict(level=["coop_test", "coexistence_test", "switch_test"], meta_file="example", num_agents=2, max_steps=31,
         recipes=["MashedCarrotBanana", "TomatoSalad"], action_scheme="scheme1", agent_despawn_rate=0.1, agent_respawn_rate=0.3, grace_period=2,
         spawn_seed=77),
    dict(level="large_16x16", meta_file="large_16x16", num_agents=4, max_steps=19,
         recipes=["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], action_scheme="scheme3"),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_shards_on_one_device_equal_one_handle(case, devices):
    kw = dict(CASES[case], num_layouts=6, layout_seed=3)
    n = 203                                                            # (odd: the shards differ in size)
    pos = [kw.pop(k) for k in ("level", "meta_file", "num_agents", "max_steps", "recipes")]
    one = CookingVecEnv(n, *pos, **kw)
    many = ShardedVecEnv(n, *pos, device_ids=devices, **kw)
    assert many.comm_kind == "host" and "share a device" in many.comm_note
    assert [c for _, c in many.ranges] == [len(x) for x in np.array_split(np.arange(n), len(devices))]
    A, F = one.num_agents, one.F
    assert np.array_equal(bits(one.reset()), bits(many.reset()))
    rng = np.random.default_rng(case)
    for t in range(40):                                                # host arrays, one launch per step
        acts = rng.integers(0, one.n_actions, size=(n, A), dtype=np.int32)
        a, b = one.step(acts), many.step(acts)
        assert all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b)), t
    codes_a, *_ = one.step_compact(acts)
    codes_b, *_ = many.step_compact(acts)
    assert np.array_equal(codes_a, codes_b) and many.codes_pitch == one.codes_pitch
    # fused rollouts over the on-device action stream (keyed by the global env id)
    T = 48
    d1 = [one.alloc((T, n, A, F), np.float64), one.alloc((T, n, A), np.float64), one.alloc((T, n, A), np.uint8), one.alloc((T, n, A), np.uint8)]
    dm = [many.alloc((A, F), np.float64, leading=(T,)), many.alloc((A,), np.float64, leading=(T,)), many.alloc((A,), np.uint8, leading=(T,)),
          many.alloc((A,), np.uint8, leading=(T,))]
    one.rollout(T, 9, 100, *d1)
    many.rollout(T, 9, 100, *dm)
    one.sync(); many.sync()
    for x, y in zip(d1, dm):
        assert np.array_equal(x.to_host().view(np.uint8), y.to_host().view(np.uint8))
    # fused steps over the caller's actions, and a ring run with the outputs in place
    acts = rng.integers(0, one.n_actions, size=(T, n, A), dtype=np.int32)
    a1, am = one.alloc((T, n, A), np.int32), many.alloc((A,), np.int32, leading=(T,))
    a1.from_host(acts); am.from_host(acts)
    one.rollout_actions(a1, T, *d1)
    many.rollout_actions(am, T, *dm)
    one.sync(); many.sync()
    assert np.array_equal(bits(d1[0].to_host()), bits(dm[0].to_host()))
    o1 = [one.alloc((n, A, F), np.float64), one.alloc((n, A), np.float64), one.alloc((n, A), np.uint8), one.alloc((n, A), np.uint8)]
    om = [many.alloc((A, F), np.float64), many.alloc((A,), np.float64), many.alloc((A,), np.uint8), many.alloc((A,), np.uint8)]
    many.ring_prepare(30, am, T, 5, *om)
    one.step_device_ring(30, a1, n * A, T, 5, *o1)
    many.step_device_ring(30, am, T, 5, *om)
    run = many.ring_runner(7, am, T, *om)                              # (launches + sync with the addresses resolved once)
    for first in (35, 42):
        one.step_device_ring(7, a1, n * A, T, first, *o1)
        run(first)
    one.sync(); many.sync()
    for x, y in zip(o1, om):
        assert np.array_equal(x.to_host().view(np.uint8), y.to_host().view(np.uint8))
    assert np.array_equal(one.get_state(), many.get_state())
    assert np.array_equal(bits(one.observe()), bits(many.observe()))
    same_stats(one.stats(), many.stats())
    assert len(many.stats_per_shard()) == len(devices)
    if "agent_despawn_rate" in kw:
        assert ((one.get_state()[:, soa.W_STATUS] >> 8) & 0xF).any(), "somebody should be despawned at this point"
        assert one.spawn_exhausted() == many.spawn_exhausted()
        # new rates on the live batch, across the 31-step boundary of the countdown fields (records are re-packed on every shard)
        for env in (one, many):
            env.set_spawn_rates(0.2, 0.4, 40)
        one.rollout(30, 11, 500); many.rollout(30, 11, 500)
        one.sync(); many.sync()
        assert np.array_equal(one.get_state(), many.get_state())
    one.close(); many.close()


def test_layout_rotation_fans_out():
    """the pool cut in two, switched and refreshed under the running batch: the same calls on a ShardedVecEnv and on one handle"""
    import random
    from cooking_zoo_amd.cooking_world.engine import load_level as ll
    pos = ("coop_test", "example", 2, 12, ["TomatoLettuceSalad", "CarrotBanana"])
    kw = dict(action_scheme="scheme3", num_layouts=8, layout_seed=1)
    n = 150
    one, many = CookingVecEnv(n, *pos, **kw), ShardedVecEnv(n, *pos, device_ids=[0, 0], **kw)
    one.reset(return_obs=False); many.reset(return_obs=False)
    r = random.Random(5)
    seen = set()
    for phase in range(6):
        for e in (one, many):
            e.set_layout_group(2, phase % 2)
        for e in (one, many):
            e.rollout(20, 3, phase * 20)
        fresh = [ll.instantiate(one.level_objects[0], one.meta, 2, r) for _ in range(4)]
        first = 4 * ((phase + 1) % 2)                                   # the half nobody draws from or plays on any more
        one.update_layouts(first, fresh); many.update_layouts(first, fresh)
        one.sync(); many.sync()
        a, b = one.get_state(), many.get_state()
        assert np.array_equal(a, b), phase
        seen |= set(int(v) for v in a[:, soa.W_LAYOUT])
    assert len(seen) == 8 and len(many.rotation_events) == len(one.rotation_events)
    same_stats(one.stats(), many.stats())
    one.close(); many.close()


def test_one_device_of_its_own_uses_the_rccl_exchange():
    """every shard on a device of its own (here: one shard): the statistics come through cz_stats_allgather"""
    with ShardedVecEnv(512, "coop_test", "example", 2, 30, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", device_ids=[0],
                       num_layouts=8) as env:
        assert env.comm_kind == "rccl", env.comm_note
        env.reset(return_obs=False)
        env.rollout(100, 1)
        env.barrier()
        st = env.stats()
        assert st == env.shards[0].stats() and st["env_steps"] > 0 and st["episodes"] > 0
    host = ShardedVecEnv(64, "coop_test", "example", 2, 30, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", device_ids=[0],
                         num_layouts=8, comm="host")
    assert host.comm_kind == "host"
    host.close()
    with pytest.raises(ValueError, match="device of its own"):
        ShardedVecEnv(64, "coop_test", "example", 2, 30, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", device_ids=[0, 0],
                      num_layouts=8, comm="rccl")


def test_config4_in_five_lines():
    """INTEGRATION.md section 3, scaled to what one device holds: the user code for BASELINE config 4"""
    from cooking_zoo_amd import ShardedVecEnv as S
    env = S(8192, "coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", device_ids=[0, 0, 0, 0])
    env.reset(return_obs=False)
    env.rollout(64, seed=1)
    st = env.stats()
    env.close()
    assert st["env_steps"] == 8192 * 64 and len(env.plan) == 4


_RANK_SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {repo!r})
from cooking_zoo_amd.distributed import FileRendezvous
from cooking_zoo_amd.sharded import ShardedVecEnv
rank, world = int(sys.argv[1]), int(sys.argv[2])
rv = FileRendezvous({rdzv!r}, rank, world, timeout=120.0)
env = ShardedVecEnv(301, ["coop_test", "switch_test"], "example", 2, 27, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                    num_layouts=6, device_ids=[0], world_size=world, rank=rank, rendezvous=rv, agent_despawn_rate=0.05, agent_respawn_rate=0.3,
                    grace_period=2, spawn_seed=9)
env.reset(return_obs=False)
env.barrier()
env.rollout(120, 4, 0)
acts = np.random.default_rng(1).integers(0, 5, size=(301, 2), dtype=np.int32)[env.local_begin:env.local_begin + env.local_envs]
obs, rew, term, trunc = env.step(acts)
env.barrier()
np.save({out!r} + f".state{{rank}}.npy", env.get_state())
np.save({out!r} + f".obs{{rank}}.npy", obs)
json.dump(dict(stats=env.stats(), per_shard=len(env.stats_per_shard()), comm=env.comm_kind, note=env.comm_note, ranges=env.ranges, plan=env.plan),
          open({out!r} + f".{{rank}}.json", "w"))
env.close()
"""


def test_two_processes_one_shard_each_equal_one_handle(tmp_path):
    """The multi-process form (one process per shard, FileRendezvous between them) - what `bench.py --gpus N` and torchrun start - on the one
    GPU a test box has: both ranks drive device 0, so the statistics go over the host path (RCCL refuses duplicate devices); states,
    observations and the statistics of the job must equal one handle over the whole batch, and every rank must report the same totals."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(repo=repo, rdzv=str(tmp_path / "rdzv"), out=str(tmp_path / "out")))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    res = [json.load(open(str(tmp_path / "out") + f".{r}.json")) for r in range(2)]
    assert res[0]["stats"] == res[1]["stats"] and res[0]["per_shard"] == 2 and res[0]["comm"] == "host" and "share a device" in res[0]["note"]
    assert res[0]["plan"] == [[0, 151], [151, 150]] and res[1]["ranges"] == [[151, 150]]
    kw = dict(action_scheme="scheme3", num_layouts=6, agent_despawn_rate=0.05, agent_respawn_rate=0.3, grace_period=2, spawn_seed=9)
    one = CookingVecEnv(301, ["coop_test", "switch_test"], "example", 2, 27, ["TomatoLettuceSalad", "CarrotBanana"], **kw)
    one.reset(return_obs=False)
    one.rollout(120, 4, 0)
    obs, *_ = one.step(np.random.default_rng(1).integers(0, 5, size=(301, 2), dtype=np.int32))
    state = np.concatenate([np.load(str(tmp_path / "out") + f".state{r}.npy") for r in range(2)])
    assert np.array_equal(state, one.get_state())
    assert np.array_equal(bits(np.concatenate([np.load(str(tmp_path / "out") + f".obs{r}.npy") for r in range(2)])), bits(obs))
    same_stats(one.stats(), res[0]["stats"])
    one.close()


def test_rotate_layouts_fans_out_to_every_shard():
    """`rotate_layouts` on a ShardedVecEnv: every shard runs the single handle's rotation with the same seed (its own producer process, the
    same refills at the same step numbers), so the batch sees what ONE handle that rotates sees - fresh layouts included"""
    pos = ("coop_test", "example", 2, 9, ["TomatoLettuceSalad", "CarrotBanana"])
    kw = dict(action_scheme="scheme3", num_layouts=8, layout_seed=2)
    n = 120
    one, many = CookingVecEnv(n, *pos, **kw), ShardedVecEnv(n, *pos, device_ids=[0, 0], **kw)
    try:
        one.reset(return_obs=False); many.reset(return_obs=False)
        one.rotate_layouts(40, groups=2, seed=7, prefetch=2)
        many.rotate_layouts(40, groups=2, seed=7, prefetch=2)
        for k in range(30):                                            # 600 steps: a dozen switches, as many refills
            one.rollout(20, 5, 20 * k); many.rollout(20, 5, 20 * k)
        one.sync(); many.sync()
        assert np.array_equal(one.get_state(), many.get_state())
        ev1, evm = one.rotation_events, many.rotation_events
        assert len(ev1) == len(evm) > 10 and [e[:2] for e in ev1] == [e[:2] for e in evm]
        assert sum(1 for e in ev1 if e[1] == "layouts") >= 5                # the pool really was refreshed under the running batch
        same_stats(one.stats(), many.stats())
    finally:
        one.stop_rotation(); many.stop_rotation()
        one.close(); many.close()
