"""CPU: the plan a ShardedVecEnv makes (no device is touched with dry_run=True) and the device-free tables it hands its shards."""
import numpy as np
import pytest

from cooking_zoo_amd.sharded import ShardedVecEnv, plan_shards
from cooking_zoo_amd.vec_env import BatchTables

CFG2 = ("coop_test", "example", 2, 400, ["TomatoLettuceSalad", "CarrotBanana"])


def test_config4_plan_of_eight_devices():
    env = ShardedVecEnv(262144, *CFG2, action_scheme="scheme3", device_ids=range(8), dry_run=True)
    assert env.plan == [(32768 * g, 32768) for g in range(8)] and env.ranges == env.plan
    assert env.local_envs == 262144 and env.F == 278 and env.n_actions == 5
    with pytest.raises(RuntimeError, match="dry_run"):
        env.reset()
    env.close()


def test_multi_process_form_owns_its_part_of_the_plan():
    """one process per GPU: rank r of 8 owns shard r (what bench.py --gpus 8 starts); two devices per process: shards 2r, 2r + 1"""
    class NoRendezvous:
        pass
    for r in range(8):
        env = ShardedVecEnv(262144, *CFG2, action_scheme="scheme3", device_ids=[r], world_size=8, rank=r, rendezvous=NoRendezvous(), dry_run=True)
        assert env.ranges == [(32768 * r, 32768)] and env.shard_ids == [r] and env.local_begin == 32768 * r
    env = ShardedVecEnv(1003, *CFG2, action_scheme="scheme3", device_ids=[0, 1], world_size=3, rank=1, rendezvous=NoRendezvous(), dry_run=True)
    assert env.shard_ids == [2, 3] and len(env.plan) == 6
    assert sum(c for _, c in env.plan) == 1003 and max(c for _, c in env.plan) - min(c for _, c in env.plan) == 1
    assert all(env.plan[g][0] + env.plan[g][1] == env.plan[g + 1][0] for g in range(5))
    with pytest.raises(ValueError, match="rendezvous"):
        ShardedVecEnv(64, *CFG2, action_scheme="scheme3", world_size=2, rank=0, dry_run=True)
    with pytest.raises(ValueError):
        plan_shards(3, 2, 2)


def test_shard_tables_are_views_of_the_batch_keyed_by_global_id():
    rid = np.array([[e % 8, (e + 1) % 8] for e in range(30)])
    t = BatchTables(30, ["coop_test", "coexistence_test", "switch_test"], "example", 2, 50, rid, action_scheme="scheme3", num_layouts=4,
                    agent_despawn_rate=0.1, agent_respawn_rate=0.2, spawn_seed=3)
    s = t.shard(7, 11)
    assert s.num_envs == 11 and s.env_id_base == 7
    assert np.array_equal(s.env_level, (7 + np.arange(11)) % 3)          # the level follows the GLOBAL env id
    assert np.array_equal(s.recipe_ids, t.recipe_ids[7:18])
    assert s.layouts == t.layouts and s.layouts is not t.layouts and s.pool_slices == t.pool_slices
    assert s.spawn_cells is t.spawn_cells and s.dims.as_tuple() == t.dims.as_tuple()
    with pytest.raises(ValueError):
        t.shard(25, 6)


def test_oracle_twins_of_shards_equal_the_twin_of_the_batch():
    """what sharding relies on, checked on the CPU with the oracle: stepping shards [0, 9) and [9, 20) of a mixed-level batch with
    despawn / respawn on equals stepping the batch (layout draws, action stream and spawn draws are keyed by the global id)"""
    from oracle_binding import VecOracle
    kw = dict(action_scheme="scheme1", num_layouts=6, agent_despawn_rate=0.1, agent_respawn_rate=0.3, grace_period=2, spawn_seed=11)
    t = BatchTables(20, ["coop_test", "switch_test"], "example", 2, 25, ["TomatoLettuceSalad", "CarrotBanana"], **kw)
    whole = VecOracle.from_vec_env(t)
    parts = [VecOracle.from_vec_env(t.shard(0, 9)), VecOracle.from_vec_env(t.shard(9, 11))]
    assert np.array_equal(whole.reset().view(np.uint64), np.concatenate([p.reset() for p in parts]).view(np.uint64))
    whole.rollout(90, 5, 0, want_obs=False)
    for p in parts:
        p.rollout(90, 5, 0, want_obs=False)
    assert np.array_equal(whole.records, np.concatenate([p.records for p in parts]))
