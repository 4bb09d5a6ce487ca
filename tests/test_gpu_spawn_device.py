"""Agent despawn / respawn evaluated by the step kernels (cz_set_spawn, Ops::handle_agent_spawn) against the host model
(cooking_zoo_amd/spawn.py SpawnBook, and the scalar transliteration of the reference's rule in test_spawn_book.py) on top of
the oracle: every stepping path - cz_step, cz_step_device, ring runs, cz_rollout - and shard invariance; and
against the oracle's own restatement of the rule (pinned to the reference by tests/golden/spawn_keyed_*.npz) on mixed-level
batches (spawn areas per level) and on every kernel instance."""
import ctypes as C

import numpy as np
import pytest

from cooking_zoo_amd import soa
from cooking_zoo_amd.spawn import SpawnBook, grace_bits, status_bits

pytestmark = pytest.mark.gpu
KW = dict(action_scheme="scheme3", num_layouts=6, auto_reset=True, agent_despawn_rate=0.15, agent_respawn_rate=0.25, grace_period=2, spawn_seed=5)
RECIPES = ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def make(n, base=0, max_steps=40, **kw):
    from cooking_zoo_amd.vec_env import CookingVecEnv
    k = dict(KW)
    k.update(kw)
    return CookingVecEnv(n, "crowded_6x5", "crowded_6x5", 4, max_steps, RECIPES, env_id_base=base, **k)


class Model:
    """oracle world step + the host model of handle_agent_spawn (vectorised SpawnBook, or the scalar rule)"""

    def __init__(self, env, scalar=False):
        from oracle_binding import VecOracle
        self.env, self.orc, self.scalar = env, VecOracle.from_vec_env(env), scalar
        self.orc.oracle.set_spawn(0, 0, 0, 0, [])            # (the oracle's own rule off: this model does the bookkeeping itself)
        d, r, g, seed = env._spawn_cfg
        self.book = SpawnBook(env.num_envs, env.num_agents, env.spawn_cells, despawn_rate=d, respawn_rate=r, grace_period=g,
                              seed=seed, env_id_base=env.env_id_base, level_of_layout=env.level_of_layout)

    def reset(self):
        self.book.reset_all()
        return self.orc.reset()

    def records(self):
        r = self.orc.records.copy()
        r[:, soa.W_STATUS] |= status_bits(self.book.active, self.book.grace, grace_bits(self.book.grace_period, self.book.A))
        return r

    def step(self, acts):
        orc, book, dims = self.orc, self.book, self.env.dims
        done_before = (orc.records[:, soa.W_STATUS] & 1).astype(bool)
        obs, rew, term, trunc = orc.step(book.mask_actions(acts))
        book.reset_envs(done_before)                       # auto-reset pass: a fresh world, everybody present
        if self.scalar:
            moved = self._scalar_after_step(~done_before)
        else:
            moved = book.after_step(orc.records, dims, stepped=~done_before)
        for e in moved:
            obs[e] = orc.oracle.observe(orc.records[e])
        trunc = trunc | (book.changed & ~book.active).astype(np.uint8)
        return obs, rew, term, trunc

    def _scalar_after_step(self, stepped):
        from test_spawn_book import scalar_rule
        book, recs, dims = self.book, self.orc.records, self.env.dims
        book._key = (recs[:, soa.W_EPISODE].astype(np.uint64) << np.uint64(32)) | recs[:, soa.W_T].astype(np.uint64)
        moved = []
        for e in np.nonzero(stepped)[0]:
            st = dict(active=book.active[e].tolist(), changed=book.changed[e].tolist(), grace=book.grace[e].tolist(), seed=book.seed,
                      despawn=book.despawn_rate, respawn=book.respawn_rate, grace_period=book.grace_period)
            before_active = list(st["active"])
            scalar_rule(st, recs[e], dims, int(book.env_ids[e]), int(book._key[e]))
            for i in range(book.A):                            # respawn_agent: the new location (parsing.py:154-167)
                if st["active"][i] and not before_active[i]:
                    x, y = book._generate_location(recs[e], dims, i, int(e))
                    _, _, o, h = soa.unpack_agent(recs[e, soa.AGENT_WORD0 + i])
                    recs[e, soa.AGENT_WORD0 + i] = soa.pack_agent(x, y, o, h)
                    moved.append(int(e))
            book.active[e], book.changed[e], book.grace[e] = st["active"], st["changed"], st["grace"]
        return sorted(set(moved))


def strip(recs):
    r = recs.copy()
    r[:, soa.RET_WORD0:soa.RET_WORD0 + 8] = 0
    return r


def test_host_array_step_and_device_step_match_the_model():
    n, A = 48, 4
    env, twin = make(n), make(n)
    mdl = Model(env)
    og = env.reset()
    twin.reset(return_obs=False)
    assert np.array_equal(bits(og), bits(mdl.reset()))
    assert np.array_equal(strip(env.get_state()), mdl.records())
    d_act = twin.alloc((n, A), np.int32)
    d_obs, d_rew = twin.alloc((n, A, twin.F), np.float64), twin.alloc((n, A), np.float64)
    d_t, d_u = twin.alloc((n, A), np.uint8), twin.alloc((n, A), np.uint8)
    rng = np.random.default_rng(1)
    n_gone = n_back = 0
    for t in range(150):
        acts = rng.integers(0, 5, size=(n, A), dtype=np.int32)
        og, rg, tg, ug = env.step(acts)
        om, rm, tm, um = mdl.step(acts)
        assert np.array_equal(bits(og), bits(om)), f"observation at step {t}"
        assert np.array_equal(bits(rg), bits(rm)) and np.array_equal(tg, tm), f"rewards / terminations at step {t}"
        assert np.array_equal(ug, um), f"truncations at step {t}"
        assert np.array_equal(strip(env.get_state()), mdl.records()), f"records at step {t}"
        assert np.array_equal(env.spawn.active, mdl.book.active) and np.array_equal(env.spawn.grace, mdl.book.grace)
        n_gone += int((mdl.book.changed & ~mdl.book.active).sum())
        n_back += int((mdl.book.changed & mdl.book.active).sum())
        # the device-pointer path does the same
        d_act.from_host(acts)
        twin.step_device(d_act, d_obs, d_rew, d_t, d_u)
        twin.sync()
        assert np.array_equal(bits(d_obs.to_host()), bits(om)) and np.array_equal(d_u.to_host(), um)
    assert np.array_equal(strip(twin.get_state()), mdl.records())
    assert n_gone > 100 and n_back > 100
    env.close()
    twin.close()


def test_rollout_matches_the_scalar_rule_and_is_shard_invariant():
    """cz_rollout with on-device actions: 48 worlds against the scalar transliteration of the reference's rule fed with the same
    keyed draws; the same 48 worlds as six handles of 8 (global env ids) end in the same records."""
    from cooking_zoo_amd import _native
    n, A, T, seed = 48, 4, 40, 77
    env = make(n)
    parts = [make(8, base=8 * k) for k in range(6)]
    mdl = Model(env, scalar=True)
    env.reset(return_obs=False)
    [p.reset(return_obs=False) for p in parts]
    mdl.reset()
    L = _native.lib()
    d_obs, d_rew = env.alloc((T, n, A, env.F), np.float64), env.alloc((T, n, A), np.float64)
    d_t, d_u = env.alloc((T, n, A), np.uint8), env.alloc((T, n, A), np.uint8)
    for chunk in range(3):
        step0 = chunk * T
        env.rollout(T, seed, step0, d_obs, d_rew, d_t, d_u)
        [p.rollout(T, seed, step0) for p in parts]
        env.sync()
        obs, rew, term, trunc = d_obs.to_host(), d_rew.to_host(), d_t.to_host(), d_u.to_host()
        for t in range(T):
            acts = np.array([[L.cz_action(seed, e, a, step0 + t, env.n_actions) for a in range(A)] for e in range(n)], dtype=np.int32)
            om, rm, tm, um = mdl.step(acts)
            assert np.array_equal(bits(obs[t]), bits(om)), (chunk, t)
            assert np.array_equal(bits(rew[t]), bits(rm)) and np.array_equal(term[t], tm) and np.array_equal(trunc[t], um), (chunk, t)
        assert np.array_equal(strip(env.get_state()), mdl.records()), chunk
        whole = strip(env.get_state())
        split = np.concatenate([strip(p.get_state()) for p in parts])
        assert np.array_equal(whole, split), f"sharded records after chunk {chunk}"
    active = mdl.book.active
    assert (~active).sum() > 5 and active.any(axis=1).all()
    env.close()
    [p.close() for p in parts]


def test_ring_runs_do_the_same_bookkeeping():
    n, A, period, K = 256, 4, 16, 48
    env, ref = make(n, max_steps=25), make(n, max_steps=25)
    env.reset(return_obs=False)
    ref.reset(return_obs=False)
    rng = np.random.default_rng(3)
    ring_host = rng.integers(0, 5, size=(period, n, A), dtype=np.int32)
    d_ring = env.alloc((period, n, A), np.int32)
    d_ring.from_host(ring_host)
    outs = [env.alloc((n, A, env.F), np.float64), env.alloc((n, A), np.float64), env.alloc((n, A), np.uint8), env.alloc((n, A), np.uint8)]
    env.step_device_ring(K, d_ring, n * A, period, 0, *outs)
    env.sync()
    for k in range(K):
        o, r, t, u = ref.step(ring_host[k % period])
    assert np.array_equal(strip(env.get_state()), strip(ref.get_state()))
    assert np.array_equal(bits(outs[0].to_host()), bits(o)) and np.array_equal(outs[3].to_host(), u)
    env.close()
    ref.close()


@pytest.mark.parametrize("levels,meta,agents,recipes,scheme", [
    (["coop_test", "coexistence_test", "switch_test"], "example", 2, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3"),      # config 3's shape
    (["switch_test", "coop_test"], "example", 2, ["MashedCarrotBanana", "TomatoSalad"], "scheme1"),
    (["large_16x16"], "large_16x16", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], "scheme3"),
    (["large_16x16"], "large_16x16", 3, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon"], "scheme1"),
    (["huge_objs_16x16"], "huge_objs_16x16", 3, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon"], "scheme3"),
    (["huge_20x20"], "huge_20x20", 3, ["TomatoLettuceSalad", "MashedCarrotBanana", "TomatoSalad"], "scheme1"),            # one-cell areas: exhausted respawns
    (["crowded_6x5"], "crowded_6x5", 1, ["TomatoSalad"], "scheme3"),                                                       # a lone agent never leaves
])
def test_spawning_batches_match_the_oracle_rule(levels, meta, agents, recipes, scheme):
    """host-array steps, then a fused rollout, then a ring run, against the oracle (which evaluates the same keyed rule);
    mixed levels take their spawn areas from each world's own level file (parsing.py:118-151)"""
    from cooking_zoo_amd.vec_env import CookingVecEnv
    from oracle_binding import VecOracle
    n, A = 96, agents
    kw = dict(action_scheme=scheme, num_layouts=6, auto_reset=True, agent_despawn_rate=0.12, agent_respawn_rate=0.3, grace_period=3, spawn_seed=21)
    env = CookingVecEnv(n, levels if len(levels) > 1 else levels[0], meta, A, 35, recipes, env_id_base=500, **kw)
    # (env e of a batch plays level e % len(levels): halves of 48 envs keep that assignment for 1, 2 and 3 levels)
    parts = [CookingVecEnv(n // 2, levels if len(levels) > 1 else levels[0], meta, A, 35, recipes, env_id_base=500 + k * (n // 2), **kw) for k in range(2)]
    orc = VecOracle.from_vec_env(env)
    og, oo = env.reset(), orc.reset()
    [p.reset(return_obs=False) for p in parts]
    assert np.array_equal(bits(og), bits(oo)) and np.array_equal(strip(env.get_state()), orc.records)
    rng = np.random.default_rng(8)
    n_gone_seen = 0
    for t in range(60):
        acts = rng.integers(0, env.n_actions, size=(n, A), dtype=np.int32)
        og, rg, tg, ug = env.step(acts)
        oo, ro, to, uo = orc.step(acts)
        [p.step(acts[k * (n // 2):(k + 1) * (n // 2)], return_obs=False) for k, p in enumerate(parts)]
        assert np.array_equal(bits(og), bits(oo)), f"observation at step {t}"
        assert np.array_equal(bits(rg), bits(ro)) and np.array_equal(tg, to) and np.array_equal(ug, uo), f"rewards / flags at step {t}"
        assert np.array_equal(strip(env.get_state()), orc.records), f"records at step {t}"
        gone = (orc.records[:, soa.W_STATUS] >> 8) & 0xF
        n_gone_seen += int((gone != 0).sum())
        assert (np.array([bin(int(g)).count("1") for g in gone]) < max(A, 2)).all()          # never everybody
    T = 50
    d_obs, d_u = env.alloc((T, n, A, env.F), np.float64), env.alloc((T, n, A), np.uint8)
    env.rollout(T, 9, 1000, d_obs, None, None, d_u)
    [p.rollout(T, 9, 1000) for p in parts]
    env.sync()
    oo, ro, to, uo = orc.rollout(T, 9, 1000)
    assert np.array_equal(strip(env.get_state()), orc.records) and np.array_equal(bits(d_obs.to_host()[-1]), bits(oo))
    assert np.array_equal(d_u.to_host()[-1], uo)
    assert np.array_equal(strip(env.get_state()), np.concatenate([strip(p.get_state()) for p in parts])), "shard invariance"
    assert (n_gone_seen > 100) if A > 1 else (n_gone_seen == 0)       # (a lone agent never leaves: cooking_world.py:273)
    if levels == ["huge_20x20"]:
        assert env.spawn_exhausted() > 0
    env.close()
    [p.close() for p in parts]


@pytest.mark.parametrize("agents,grace,level,meta,recipes", [
    (2, 200, "coop_test", "example", ["TomatoLettuceSalad", "CarrotBanana"]),
    (3, 63, "crowded_6x5", "crowded_6x5", ["TomatoSalad", "TomatoLettuceSalad", "MashedCarrotBanana"]),
    (1, 5000, "coop_test", "example", ["TomatoSalad"]),
])
def test_grace_periods_beyond_31(agents, grace, level, meta, recipes):
    """the reference takes any grace_period (cooking_env.py:28, parsing.py:142).  The status word has 20 bits for the countdowns of all
    agents: 5 bits each while the period is at most 31 (the layout of every fixture), else 20 / A bits each - 1023 steps for two agents,
    63 for three, any practical value for one.  Device vs the oracle's rule, and the view the env gives of the countdowns."""
    from cooking_zoo_amd.spawn import decode_status, grace_bits, max_grace
    from cooking_zoo_amd.vec_env import CookingVecEnv
    from oracle_binding import VecOracle
    assert grace > 31 and grace <= max_grace(agents)
    n = 128
    kw = dict(action_scheme="scheme3", num_layouts=4, agent_despawn_rate=0.3, agent_respawn_rate=0.5, grace_period=grace, spawn_seed=3)
    env = CookingVecEnv(n, level, meta, agents, 400, recipes, **kw)
    orc = VecOracle.from_vec_env(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    b = grace_bits(grace, agents)
    _, g0 = decode_status(env.get_state()[:, soa.W_STATUS], agents, b)
    assert (g0 == grace).all() and (env.spawn.grace == grace).all()
    rng = np.random.default_rng(2)
    for t in range(12):
        acts = rng.integers(0, 5, size=(n, agents), dtype=np.int32)
        og, *_ = env.step(acts)
        oo, *_ = orc.step(acts)
        assert np.array_equal(bits(og), bits(oo)) and np.array_equal(strip(env.get_state()), orc.records), t
    assert (env.spawn.grace == grace - 12).all()
    T = min(grace + 40, 380)
    env.rollout(T, 4, 100); env.sync()
    orc.rollout(T, 4, 100, want_obs=False)
    assert np.array_equal(strip(env.get_state()), orc.records)
    if agents > 1:
        # somebody has left and come back by now: its countdown was restarted, a value beyond what five bits hold lives in the record
        _, g = decode_status(orc.records[:, soa.W_STATUS], agents, b)
        assert (g > 31).any() and (g <= grace).all()
    env.close()


def test_grace_width_change_repacks_running_episodes():
    """cz_set_spawn across the 31 boundary while episodes run: the countdown fields change their width and the library
    re-packs the resident records - every agent keeps its countdown (agent a > 0's field moves), going back narrows with a clamp to 31 -
    and the batch then steps exactly like an oracle whose records were re-packed the same way."""
    from cooking_zoo_amd.spawn import decode_status, grace_bits, status_bits
    from cooking_zoo_amd.vec_env import CookingVecEnv
    from oracle_binding import VecOracle
    n, A = 96, 2
    kw = dict(action_scheme="scheme3", num_layouts=4, agent_despawn_rate=0.3, agent_respawn_rate=0.5, grace_period=20, spawn_seed=9)
    env = CookingVecEnv(n, "coop_test", "example", A, 400, ["TomatoLettuceSalad", "CarrotBanana"], **kw)
    env.reset(return_obs=False)
    env.rollout(30, 4, 100); env.sync()
    st5 = env.get_state()
    act5, g5 = decode_status(st5[:, soa.W_STATUS], A, 5)
    assert (g5 > 0).any() and (~act5).any()
    env.set_spawn_rates(0.3, 0.5, 200)                                  # 5 -> 10 bits per countdown
    st10 = env.get_state()
    act10, g10 = decode_status(st10[:, soa.W_STATUS], A, grace_bits(200, A))
    assert np.array_equal(act10, act5) and np.array_equal(g10, g5)
    other = np.ones(st5.shape[1], dtype=bool); other[soa.W_STATUS] = False
    assert np.array_equal(st10[:, other], st5[:, other]) and np.array_equal(st10[:, soa.W_STATUS] & 0xFFF, st5[:, soa.W_STATUS] & 0xFFF)
    assert np.array_equal(env.spawn.refresh().grace, g5)               # (the view decodes with the width of the current period)
    # from here on: an oracle of the new configuration, started from the re-packed records
    orc = VecOracle.from_vec_env(env)
    orc.reset()
    orc.records[:] = strip(st10)
    rng = np.random.default_rng(4)
    for t in range(60):
        acts = rng.integers(0, 5, size=(n, A), dtype=np.int32)
        og, rg, tg, ug = env.step(acts)
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(og), bits(oo)) and np.array_equal(tg, to) and np.array_equal(ug, uo), t
        assert np.array_equal(strip(env.get_state()), orc.records), t
    _, g = decode_status(env.get_state()[:, soa.W_STATUS], A, 10)
    assert (g > 31).any()                                              # somebody came back under the long period
    env.set_spawn_rates(0.3, 0.5, 8)                                    # 10 -> 5 bits: clamp to 31
    act_b, g_b = decode_status(env.get_state()[:, soa.W_STATUS], A, 5)
    act_a, _ = decode_status(orc.records[:, soa.W_STATUS], A, 10)
    assert np.array_equal(act_b, act_a) and np.array_equal(g_b, np.minimum(g, 31))
    env.close()


def test_grace_period_beyond_the_field_is_refused():
    from cooking_zoo_amd import _native
    from cooking_zoo_amd.vec_env import CookingVecEnv
    with pytest.raises(_native.NativeError, match="grace_period <= 31 for 4 agent"):
        CookingVecEnv(8, "crowded_6x5", "crowded_6x5", 4, 40, RECIPES, action_scheme="scheme3", num_layouts=2, agent_despawn_rate=0.1,
                      agent_respawn_rate=0.1, grace_period=32)
    with pytest.raises(_native.NativeError, match="grace_period <= 1023 for 2 agent"):
        CookingVecEnv(8, "coop_test", "example", 2, 40, ["TomatoSalad", "TomatoSalad"], action_scheme="scheme3", num_layouts=2,
                      agent_despawn_rate=0.1, agent_respawn_rate=0.1, grace_period=1024)
