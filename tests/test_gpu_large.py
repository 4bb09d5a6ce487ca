"""GPU: the BASELINE configurations at their FULL sizes against the oracle, and all three observation-store flavours.

* `launch_step` (cz_api.hip) picks plain `global_store` instead of write-through buffer stores for the observation of
  the one-launch-per-step kernel once N*A*F*8 exceeds 128 MiB -- above ~30 k envs for config 2, ~5 k for config 5.  The
  small-batch tests never reach that branch, so it is pinned here twice: by forcing either flavour on small batches of
  every level family (CZ_WT=0/1/2, read at cz_create) and by running configs 3, 4 (one rank's shard) and 5 at full size.
* The oracle runs threaded (tests/oracle_binding.ShardedOracle); every comparison is bit for bit."""
import os

import numpy as np
import pytest

from cooking_zoo_amd import soa

pytestmark = pytest.mark.gpu

BOOK = ["TomatoSalad", "TomatoLettuceSalad", "TomatoLettuceOnionSalad", "CarrotBanana", "MashedCarrotBanana",
        "CucumberOnion", "AppleWatermelon", "no_recipe"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def strip(recs):
    r = recs.copy()
    r[:, soa.RET_WORD0:soa.RET_WORD0 + 8] = 0
    return r


def make(n, level, meta, agents, recipes, scheme="scheme3", **kw):
    from cooking_zoo_amd.vec_env import CookingVecEnv
    args = dict(max_steps=30, num_layouts=16, auto_reset=True)
    args.update(kw)
    ms = args.pop("max_steps")
    return CookingVecEnv(n, level, meta, agents, ms, recipes, action_scheme=scheme, **args)


FAMILIES = [
    ("coop_test", "example", 1, ["TomatoLettuceSalad"], "scheme3"),
    ("coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3"),
    ("switch_test", "example", 2, ["MashedCarrotBanana", "TomatoSalad"], "scheme1"),
    ("coexistence_test", "example", 2, ["AppleWatermelon", "TomatoLettuceOnionSalad"], "scheme3"),
    ("crowded_6x5", "crowded_6x5", 3, ["TomatoSalad", "no_recipe", "MashedCarrotBanana"], "scheme3"),
    ("crowded_6x5", "crowded_6x5", 4, ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"], "scheme1"),
    ("edge_8x8", "edge", 3, ["TomatoSalad", "MashedCarrotBanana", "TomatoLettuceSalad"], "scheme3"),
    ("edge_9x8", "edge", 1, ["TomatoSalad"], "scheme1"),
    ("large_16x16", "large_16x16", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], "scheme3"),
    ("dense_16x16", "dense_16x16", 2, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3"),
    ("limit_32x8", "limits", 2, ["TomatoSalad", "CarrotBanana"], "scheme3"),
    ("limit_8x31", "limits", 3, ["TomatoSalad", "CarrotBanana", "AppleWatermelon"], "scheme1"),
    # the third kernel instance (more than 128 slots or 256 cells)
    ("huge_32x32", "huge_32x32", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], "scheme3"),
    ("huge_20x20", "huge_20x20", 3, ["TomatoLettuceSalad", "MashedCarrotBanana", "TomatoSalad"], "scheme1"),
    ("huge_objs_16x16", "huge_objs_16x16", 2, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3"),
]


@pytest.mark.parametrize("wt", [0, 1, 2])
@pytest.mark.parametrize("level,meta,agents,recipes,scheme", FAMILIES)
def test_all_store_flavours_match_oracle(wt, level, meta, agents, recipes, scheme):
    from oracle_binding import VecOracle
    n, T = 80, 45
    os.environ["CZ_WT"] = str(wt)
    try:
        env = make(n, level, meta, agents, recipes, scheme, max_steps=20, num_layouts=6)
    finally:
        del os.environ["CZ_WT"]
    orc = VecOracle.from_vec_env(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    rng = np.random.default_rng(11 + wt)
    d_act = env.alloc((n, agents), np.int32)
    d_obs = env.alloc((n, agents, env.F), np.float64)
    d_rew = env.alloc((n, agents), np.float64)
    d_t = env.alloc((n, agents), np.uint8)
    d_u = env.alloc((n, agents), np.uint8)
    for t in range(T):
        acts = rng.integers(0, env.n_actions, size=(n, agents), dtype=np.int32)
        d_act.from_host(acts)
        env.step_device(d_act, d_obs, d_rew, d_t, d_u)              # the device-pointer path the bench uses
        env.sync()
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(d_obs.to_host()), bits(oo)), f"obs @ step {t} (wt={wt})"
        assert np.array_equal(bits(d_rew.to_host()), bits(ro)), f"rewards @ step {t}"
        assert np.array_equal(d_t.to_host(), to) and np.array_equal(d_u.to_host(), uo), f"flags @ step {t}"
    assert np.array_equal(strip(env.get_state()), orc.records)
    env.close()


def _ring_run_vs_oracle(env, orc, K, period, rng, first_slot=0):
    """K steps through cz_step_device_ring (graph replay / direct launches) against the oracle: what the last
    step left in the output buffers, the records, the episode statistics."""
    import ctypes as C
    from cooking_zoo_amd import _native
    n, A = env.num_envs, env.num_agents
    L = _native.lib()
    ring_host = rng.integers(0, env.n_actions, size=(period, n, A), dtype=np.int32)
    d_ring = env.alloc((period, n, A), np.int32)
    d_ring.from_host(ring_host)
    d_obs = env.alloc((n, A, env.F), np.float64)
    d_rew = env.alloc((n, A), np.float64)
    d_t = env.alloc((n, A), np.uint8)
    d_u = env.alloc((n, A), np.uint8)
    _native.check(env._h, L.cz_step_device_ring(env._h, K, d_ring.ptr, n * A, period, first_slot, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr))
    env.sync()
    for k in range(K):
        oo, ro, to, uo = orc.step(ring_host[(first_slot + k) % period], k == K - 1)
    assert np.array_equal(strip(env.get_state()), orc.records), "records after the run"
    assert np.array_equal(bits(d_obs.to_host()), bits(oo)), "observation of the last step"
    assert np.array_equal(bits(d_rew.to_host()), bits(ro)), "rewards of the last step"
    assert np.array_equal(d_t.to_host(), to) and np.array_equal(d_u.to_host(), uo), "flags of the last step"
    for b in (d_ring, d_obs, d_rew, d_t, d_u):
        b.free()


@pytest.mark.parametrize("level,meta,agents,recipes,scheme", FAMILIES)
def test_ring_runs_match_oracle(level, meta, agents, recipes, scheme):
    """Runs of cz_step_device_ring (captured into HIP graphs on first use, replayed afterwards; short pieces launched directly):
    same results as stepping the oracle one step at a time, for every level family, with episodes ending and restarting
    inside the runs; and the library says how the launches went out."""
    import ctypes as C
    from cooking_zoo_amd import _native
    from oracle_binding import VecOracle
    env = make(200, level, meta, agents, recipes, scheme, max_steps=17, num_layouts=6)
    orc = VecOracle.from_vec_env(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    rng = np.random.default_rng(5)
    L = _native.lib()
    L.cz_launch_counts(env._h, None, None, 1)
    for K, period, first in ((2, 8, 0), (37, 16, 5), (3, 3, 2), (60, 64, 63)):
        _ring_run_vs_oracle(env, orc, K, period, rng, first)
    g, d = C.c_int64(), C.c_int64()
    L.cz_launch_counts(env._h, C.byref(g), C.byref(d), 0)
    assert g.value + d.value == 2 + 37 + 3 + 60 and g.value > 0
    st = env.stats()
    assert st["episodes"] == int(orc.records[:, soa.W_EPISODE].sum()) + int((orc.records[:, soa.W_STATUS] & 1).sum())
    env.close()


@pytest.mark.parametrize("n", [4096, 16384])
def test_long_ring_runs(n):
    """300-step runs of the config-2 workload through cz_step_device_ring"""
    from oracle_binding import ShardedOracle
    env = make(n, "coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3", max_steps=40, num_layouts=64)
    orc = ShardedOracle(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    _ring_run_vs_oracle(env, orc, 300, 32, np.random.default_rng(77))
    env.close()


def test_many_short_ring_runs():
    """80 runs of 2-40 steps from random slots, mostly without synchronising in between (graph cache churn: at most 64 graphs are
    kept per handle)"""
    from cooking_zoo_amd import _native
    from oracle_binding import ShardedOracle
    n, A, period = 4096, 2, 32
    env = make(n, "coop_test", "example", A, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3", max_steps=40, num_layouts=64)
    orc = ShardedOracle(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    L = _native.lib()
    rng = np.random.default_rng(9)
    ring_host = rng.integers(0, env.n_actions, size=(period, n, A), dtype=np.int32)
    d_ring = env.alloc((period, n, A), np.int32)
    d_ring.from_host(ring_host)
    d_obs, d_rew = env.alloc((n, A, env.F), np.float64), env.alloc((n, A), np.float64)
    d_t, d_u = env.alloc((n, A), np.uint8), env.alloc((n, A), np.uint8)
    for r in range(80):
        K, first = int(rng.integers(2, 41)), int(rng.integers(period))
        _native.check(env._h, L.cz_step_device_ring(env._h, K, d_ring.ptr, n * A, period, first, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr))
        if r % 7 == 0:
            env.sync()
        for k in range(K):
            oo, ro, to, uo = orc.step(ring_host[(first + k) % period], False)
    env.sync()
    assert np.array_equal(strip(env.get_state()), orc.records)
    assert np.array_equal(bits(d_rew.to_host()), bits(ro)) and np.array_equal(d_t.to_host(), to) and np.array_equal(d_u.to_host(), uo)
    env.close()


def test_step_device_many_matches_oracle():
    from cooking_zoo_amd import _native
    from oracle_binding import VecOracle
    a = make(64, "coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"])
    orc = VecOracle.from_vec_env(a)
    assert np.array_equal(bits(a.reset()), bits(orc.reset()))
    L = _native.lib()
    rng = np.random.default_rng(3)
    acts = rng.integers(0, a.n_actions, size=(6, 64, 2), dtype=np.int32)
    d_acts, d_obs = a.alloc(acts.shape, np.int32), a.alloc((64, 2, a.F), np.float64)
    d_rew, d_t, d_u = a.alloc((64, 2), np.float64), a.alloc((64, 2), np.uint8), a.alloc((64, 2), np.uint8)
    d_acts.from_host(acts)
    _native.check(a._h, L.cz_step_device_many(a._h, 6, d_acts.ptr, 64 * 2, 6, d_obs.ptr, d_rew.ptr, d_t.ptr, d_u.ptr))
    a.sync()
    for k in range(6):
        oo, ro, to, uo = orc.step(acts[k])
    assert np.array_equal(bits(d_obs.to_host()), bits(oo)) and np.array_equal(strip(a.get_state()), orc.records)
    a.close()


def _full_size_case(env, steps, T_fused, seed):
    """`steps` one-launch-per-step launches with external actions (every output compared at every step), then one fused
    rollout of T_fused steps (final records, last observation, last rewards / flags compared)."""
    from oracle_binding import ShardedOracle
    n, A = env.num_envs, env.num_agents
    orc = ShardedOracle(env)
    og = env.reset()
    assert np.array_equal(bits(og), bits(orc.reset())), "reset observation"
    del og
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = rng.integers(0, env.n_actions, size=(n, A), dtype=np.int32)
        og, rg, tg, ug = env.step(acts)
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(rg), bits(ro)), f"rewards @ step {t}"
        assert np.array_equal(tg, to) and np.array_equal(ug, uo), f"flags @ step {t}"
        assert np.array_equal(bits(og), bits(oo)), f"obs @ step {t}"
        del og, oo
    assert np.array_equal(strip(env.get_state()), orc.records), "records after the per-step launches"
    d_rew = env.alloc((T_fused, n, A), np.float64)
    d_t = env.alloc((T_fused, n, A), np.uint8)
    d_u = env.alloc((T_fused, n, A), np.uint8)
    env.rollout(T_fused, seed, 1000, None, d_rew, d_t, d_u)
    env.sync()
    oo, ro, to, uo = orc.rollout(T_fused, seed, 1000)
    assert np.array_equal(strip(env.get_state()), orc.records), "records after the fused rollout"
    assert np.array_equal(bits(d_rew.to_host()[-1]), bits(ro)), "last rewards of the fused rollout"
    assert np.array_equal(d_t.to_host()[-1], to) and np.array_equal(d_u.to_host()[-1], uo)
    assert np.array_equal(bits(env.observe()), bits(oo)), "observation after the fused rollout"
    st = env.stats()
    assert st["episodes"] == int(orc.records[:, soa.W_EPISODE].sum()) + int((orc.records[:, soa.W_STATUS] & 1).sum())


def test_config3_full_size_mixed_levels_whole_recipe_book():
    """BASELINE config 3: 65 536 envs, env e -> level e % 3 of (coop_test, coexistence_test, switch_test), recipes
    R[e % 8], R[(e + 1) % 8]; short horizon so that auto-resets happen inside the window."""
    n = 65536
    rid = np.array([[e % 8, (e + 1) % 8] for e in range(n)])
    from cooking_zoo_amd.vec_env import CookingVecEnv
    env = CookingVecEnv(n, ["coop_test", "coexistence_test", "switch_test"], "example", 2, 12, rid, action_scheme="scheme3",
                        num_layouts=256, auto_reset=True)
    assert n * 2 * env.F * 8 > (128 << 20), "this size must take the plain-store branch of launch_step"
    _full_size_case(env, steps=16, T_fused=40, seed=31)
    env.close()


def test_config3_full_size_with_despawn_respawn_and_compact_observations():
    """BASELINE config 3's shape - 65 536 envs on three levels - with despawn / respawn on (spawn areas per level,
    parsing.py:118-151; round 3 refused mixed levels) and the observation taken as codes: device-pointer steps writing both the
    float64 rows and the codes, then a fused rollout with a compact trajectory, against the oracle's restatement of the rule
    (pinned to the reference by tests/golden/spawn_keyed_*.npz) on host threads."""
    from cooking_zoo_amd.vec_env import CookingVecEnv
    from oracle_binding import ShardedOracle
    n, A = 65536, 2
    rid = np.array([[e % 8, (e + 1) % 8] for e in range(n)])
    env = CookingVecEnv(n, ["coop_test", "coexistence_test", "switch_test"], "example", A, 12, rid, action_scheme="scheme3", num_layouts=256,
                        auto_reset=True, agent_despawn_rate=0.1, agent_respawn_rate=0.3, grace_period=2, spawn_seed=17)
    orc = ShardedOracle(env)
    env.reset(return_obs=False)
    orc.reset()
    table, F, Fp = env.obs_table(), env.F, env.codes_pitch
    d_act = env.alloc((n, A), np.int32)
    d_obs, d_codes = env.alloc((n, A, F), np.float64), env.alloc((n, A, Fp), np.uint8)
    d_rew, d_t, d_u = env.alloc((n, A), np.float64), env.alloc((n, A), np.uint8), env.alloc((n, A), np.uint8)
    rng = np.random.default_rng(35)
    n_gone = 0
    for t in range(12):
        acts = rng.integers(0, 5, size=(n, A), dtype=np.int32)
        d_act.from_host(acts)
        env.step_device_compact(d_act, d_codes, d_rew, d_t, d_u, d_obs if t % 2 == 0 else None)
        env.sync()
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(table[d_codes.to_host()[:, :, :F]]), bits(oo)), f"decoded observation @ step {t}"
        if t % 2 == 0:
            assert np.array_equal(bits(d_obs.to_host()), bits(oo)), f"float64 observation @ step {t}"
        assert np.array_equal(bits(d_rew.to_host()), bits(ro)) and np.array_equal(d_t.to_host(), to) and np.array_equal(d_u.to_host(), uo), t
        n_gone += int((((orc.records[:, soa.W_STATUS] >> 8) & 0xF) != 0).sum())
        del oo
    assert np.array_equal(strip(env.get_state()), orc.records)
    assert n_gone > 10000
    T = 24
    d_traj = env.alloc((T, n, A, Fp), np.uint8)
    d_tu = env.alloc((T, n, A), np.uint8)
    env.rollout_compact(T, 36, 500, d_traj, None, None, None, d_tu)
    env.sync()
    oo, ro, to, uo = orc.rollout(T, 36, 500)
    assert np.array_equal(strip(env.get_state()), orc.records), "records after the fused rollout"
    assert np.array_equal(bits(table[d_traj.to_host()[-1][:, :, :F]]), bits(oo)), "last observation of the compact trajectory"
    assert np.array_equal(d_tu.to_host()[-1], uo)
    assert env.spawn_exhausted() == 0
    env.close()


def test_config5_full_size_four_agents_16x16():
    """BASELINE config 5: 65 536 envs x 4 competing agents on large_16x16 (F = 840, 26.9 KB of observation per env-step)."""
    n = 65536
    env = make(n, "large_16x16", "large_16x16", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"],
               max_steps=6, num_layouts=256)
    assert env.F == 840
    _full_size_case(env, steps=8, T_fused=16, seed=32)
    env.close()


def test_config4_one_ranks_shard():
    """BASELINE config 4, the shard of rank 3 of 8: 32 768 envs with global ids from 98 304 (layout draws and the
    on-device action stream are keyed by the global id)."""
    n, base = 32768, 98304
    env = make(n, "coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"], max_steps=14, num_layouts=256,
               env_id_base=base)
    assert n * 2 * env.F * 8 > (128 << 20)
    _full_size_case(env, steps=16, T_fused=48, seed=33)
    env.close()


def test_config4_full_shape_eight_shards_on_one_device():
    """BASELINE config 4 in its full shape - 262 144 envs cut into 8 shards of 32 768, global env ids - on the one device a box has:
    `ShardedVecEnv(device_ids=[0] * 8)` = eight handles, eight streams, eight host threads, the statistics of eight shards; only RCCL's
    transport between DISTINCT devices is replaced by the host reduction (RCCL refuses duplicate devices).  Against the oracle over the
    whole batch: host-array steps with every output compared, then a fused rollout over the on-device action stream (keyed by the global
    id, so the shards draw what one handle over 262 144 envs would draw)."""
    from cooking_zoo_amd import ShardedVecEnv
    from oracle_binding import ShardedOracle
    n, A, G = 262144, 2, 8
    senv = ShardedVecEnv(n, "coop_test", "example", A, 14, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=256,
                         auto_reset=True, device_ids=[0] * G, comm="host")
    assert senv.ranges == [(g * 32768, 32768) for g in range(G)] and senv.comm_kind == "host"
    orc = ShardedOracle(senv.tables)
    senv.reset(return_obs=False)
    orc.reset()
    rng = np.random.default_rng(43)
    for t in range(5):
        acts = rng.integers(0, 5, size=(n, A), dtype=np.int32)
        og, rg, tg, ug = senv.step(acts)
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(rg), bits(ro)) and np.array_equal(tg, to) and np.array_equal(ug, uo), f"rewards / flags @ step {t}"
        assert np.array_equal(bits(og), bits(oo)), f"obs @ step {t}"
        del og, oo
    T = 32
    senv.rollout(T, 44, 700)
    senv.sync()
    orc.rollout(T, 44, 700, want_obs=False)
    assert np.array_equal(strip(senv.get_state()), orc.records), "records after the fused rollout"
    per = senv.stats_per_shard()
    assert len(per) == G and all(p["env_steps"] > 32768 * (5 + T) * 0.9 for p in per)
    st = senv.stats()
    assert st["env_steps"] == sum(p["env_steps"] for p in per)
    assert st["episodes"] == int(orc.records[:, soa.W_EPISODE].sum()) + int((orc.records[:, soa.W_STATUS] & 1).sum())
    senv.close()


def test_stats_reduction_order_and_size():
    """The two-stage statistics reduction (64 workgroups of chains + one tree) against a numpy model of its fixed
    summation order, bit for bit, at 65 536 envs: chain c adds envs c, c+256, ... in order, then a binary tree."""
    n, T = 65536, 45
    env = make(n, "coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"], max_steps=20, num_layouts=64)
    env.reset(return_obs=False)
    d_rew = env.alloc((T, n, 2), np.float64)
    d_t = env.alloc((T, n, 2), np.uint8)
    d_u = env.alloc((T, n, 2), np.uint8)
    env.rollout(T, 8, 0, None, d_rew, d_t, d_u)
    env.sync()
    rew, done = d_rew.to_host(), (d_t.to_host()[:, :, 0] | d_u.to_host()[:, :, 0]).astype(bool)
    # per-env sums of finished-episode returns, accumulated the way the kernel does (running return += reward per step)
    cur = np.zeros((n, 2))
    fin = np.zeros((n, 2))
    was_done = np.zeros(n, bool)
    episodes = 0
    for t in range(T):
        stepped = ~was_done
        cur[stepped] += rew[t][stepped]
        ended = stepped & done[t]
        fin[ended] += cur[ended]
        cur[ended] = 0
        episodes += int(ended.sum())
        was_done = ended                                  # the next launch is the reset pass of an ended env
    chains = np.zeros((256, 2))
    for j in range(n // 256):
        chains += fin[j * 256:(j + 1) * 256]
    level = chains
    while level.shape[0] > 1:
        half = level.shape[0] // 2
        level = level[:half] + level[half:]
    st = env.stats()
    assert st["episodes"] == episodes
    assert np.array_equal(bits(np.array(st["return_sum"][:2])), bits(level[0]))
    env.close()
