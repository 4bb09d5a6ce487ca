"""GPU: the compact observation (cz_step_device_compact): one byte per feature, the index of the feature's value in a table of
256 float64 (cz_obs_table) - losslessly the float64 observation of cooking_env.py:352-373.  `table[codes] == oracle obs` as
uint64 on every level family / kernel instance / scheme / agent count, codes-only launches and launches that write both forms,
padding bytes, the closed loop over codes against the closed loop over float64 rows."""
import ctypes as C

import numpy as np
import pytest

from cooking_zoo_amd import soa

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def strip(recs):
    r = recs.copy()
    r[:, soa.RET_WORD0:soa.RET_WORD0 + 8] = 0
    return r


def make(n, level, meta, agents, recipes, scheme, max_steps=30, **kw):
    from cooking_zoo_amd.vec_env import CookingVecEnv
    args = dict(action_scheme=scheme, num_layouts=8, auto_reset=True)
    args.update(kw)
    return CookingVecEnv(n, level, meta, agents, max_steps, recipes, **args)


CASES = [
    ("scheme3", "coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], "example"),
    ("scheme3", "coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], "example_odd"),          # F = 283: not a multiple of 4
    ("scheme1", "switch_test", 2, ["MashedCarrotBanana", "TomatoSalad"], "example"),
    ("scheme3", "coexistence_test", 1, ["TomatoLettuceOnionSalad"], "example"),
    ("scheme3", "crowded_6x5", 4, ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"], "crowded_6x5"),
    ("scheme1", "crowded_6x5", 3, ["TomatoSalad", "TomatoLettuceSalad", "MashedCarrotBanana"], "crowded_6x5"),
    ("scheme3", "large_16x16", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], "large_16x16"),
    ("scheme1", "dense_16x16", 2, ["TomatoLettuceOnionSalad", "MashedCarrotBanana"], "dense_16x16"),
    ("scheme3", "edge_8x8", 3, ["TomatoSalad", "MashedCarrotBanana", "TomatoLettuceSalad"], "edge"),
    ("scheme3", "limit_32x8", 3, ["TomatoSalad", "MashedCarrotBanana", "TomatoLettuceSalad"], "limits"),
    ("scheme3", "limit_8x31", 2, ["TomatoSalad", "CarrotBanana"], "limits"),
    ("scheme3", "huge_20x20", 3, ["TomatoLettuceSalad", "MashedCarrotBanana", "TomatoSalad"], "huge_20x20"),
    ("scheme1", "huge_32x32", 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], "huge_32x32"),
]


@pytest.mark.parametrize("scheme,level,agents,recipes,meta", CASES)
def test_codes_decode_to_the_oracle_observation(scheme, level, agents, recipes, meta):
    from oracle_binding import VecOracle
    n, T = 48, 70
    env = make(n, level, meta, agents, recipes, scheme)
    orc = VecOracle.from_vec_env(env)
    env.reset(return_obs=False)
    orc.reset()
    F, Fp, A = env.F, env.codes_pitch, agents
    assert Fp % 16 == 0 and F <= Fp < F + 16
    table = env.obs_table()
    assert table.shape == (256,) and table[255] == 0.0
    d_act = env.alloc((n, A), np.int32)
    d_codes, d_obs = env.alloc((n, A, Fp), np.uint8), env.alloc((n, A, F), np.float64)
    d_rew, d_t, d_u = env.alloc((n, A), np.float64), env.alloc((n, A), np.uint8), env.alloc((n, A), np.uint8)
    rng = np.random.default_rng(5)
    for t in range(T):
        acts = rng.integers(0, env.n_actions, size=(n, A), dtype=np.int32)
        d_act.from_host(acts)
        both = t % 3 == 0                                    # every third step writes the float64 rows as well
        d_codes.from_host(np.full((n, A, Fp), 7, dtype=np.uint8))
        env.step_device_compact(d_act, d_codes, d_rew, d_t, d_u, d_obs if both else None)
        env.sync()
        oo, ro, to, uo = orc.step(acts)
        codes = d_codes.to_host()
        assert np.array_equal(bits(table[codes[:, :, :F]]), bits(oo)), f"decoded observation at step {t}"
        assert (codes[:, :, F:] == 255).all(), "padding bytes"
        if both:
            assert np.array_equal(bits(d_obs.to_host()), bits(oo)), f"float64 observation at step {t}"
        assert np.array_equal(bits(d_rew.to_host()), bits(ro)) and np.array_equal(d_t.to_host(), to) and np.array_equal(d_u.to_host(), uo)
    assert np.array_equal(strip(env.get_state()), orc.records)
    assert int(env.get_state()[:, soa.W_EPISODE].min()) >= 1          # reset passes were encoded too
    env.close()


def test_closed_loop_over_codes_takes_the_same_actions_as_over_float64():
    """the two measurement loops of bench.py (cz_probe_closed_loop / _compact): their policies read the same four features -
    as float64 or as table indices - so both loops must walk the same trajectory"""
    from cooking_zoo_amd import _native
    n, A, K = 512, 2, 60
    a, b = (make(n, "coop_test", "example", A, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3", max_steps=25) for _ in range(2))
    L = _native.lib()
    us = C.c_float()
    acts0 = np.random.default_rng(1).integers(0, 5, size=(n, A), dtype=np.int32)
    for env in (a, b):
        env.reset(return_obs=False)
    da, db = a.alloc((n, A), np.int32), b.alloc((n, A), np.int32)
    da.from_host(acts0)
    db.from_host(acts0)
    d_obs = a.alloc((n, A, a.F), np.float64)
    d_codes = b.alloc((n, A, b.codes_pitch), np.uint8)
    ra, rb = a.alloc((n, A), np.float64), b.alloc((n, A), np.float64)
    fa, fb = [a.alloc((n, A), np.uint8) for _ in range(2)], [b.alloc((n, A), np.uint8) for _ in range(2)]
    _native.check(a._h, L.cz_probe_closed_loop(a._h, K, 1, da.ptr, d_obs.ptr, ra.ptr, fa[0].ptr, fa[1].ptr, C.byref(us)))
    _native.check(b._h, L.cz_probe_closed_loop_compact(b._h, K, 1, db.ptr, d_codes.ptr, rb.ptr, fb[0].ptr, fb[1].ptr, C.byref(us)))
    a.sync()
    b.sync()
    assert np.array_equal(da.to_host(), db.to_host())
    assert np.array_equal(strip(a.get_state()), strip(b.get_state()))
    assert np.array_equal(bits(b.obs_table()[d_codes.to_host()[:, :, :b.F]]), bits(d_obs.to_host()))
    assert a.stats() == b.stats()
    a.close()
    b.close()


def test_compact_step_argument_errors():
    from cooking_zoo_amd import _native
    env = make(8, "coop_test", "example", 2, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3")
    env.reset(return_obs=False)
    L = _native.lib()
    d = env.alloc((8, 2), np.int32)
    assert L.cz_step_device_compact(env._h, d.ptr, None, None, None, None, None) != 0
    assert b"codes" in L.cz_last_error(env._h)
    env.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_host_array_step_with_codes(pinned):
    """CookingVecEnv.step_compact (cz_step_compact): pageable arrays through the staging block, pinned ones written by the kernel"""
    from oracle_binding import VecOracle
    n, A = 300, 2
    env = make(n, "coop_test", "example", A, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3", pinned_outputs=pinned)
    orc = VecOracle.from_vec_env(env)
    env.reset(return_obs=False)
    orc.reset()
    table = env.obs_table()
    rng = np.random.default_rng(3)
    for t in range(40):
        acts = rng.integers(0, 5, size=(n, A), dtype=np.int32)
        codes, rg, tg, ug = env.step_compact(acts)
        oo, ro, to, uo = orc.step(acts)
        assert codes.shape == (n, A, env.codes_pitch)
        assert np.array_equal(bits(table[codes[:, :, :env.F]]), bits(oo)), t
        assert np.array_equal(bits(rg), bits(ro)) and np.array_equal(tg, to) and np.array_equal(ug, uo)
    # the ordinary step afterwards is untouched by the compact one
    og, rg, tg, ug = env.step(acts)
    oo, ro, to, uo = orc.step(acts)
    assert np.array_equal(bits(og), bits(oo))
    env.close()


@pytest.mark.parametrize("scheme,level,agents,recipes,meta", [CASES[0], CASES[1], CASES[4], CASES[6], CASES[11]])
def test_fused_rollout_with_a_compact_trajectory(scheme, level, agents, recipes, meta):
    """cz_rollout_compact: T fused steps, codes of every step (and, second launch, the float64 trajectory beside them)"""
    from oracle_binding import VecOracle
    n, T, seed = 40, 45, 321
    env = make(n, level, meta, agents, recipes, scheme, max_steps=20)
    orc = VecOracle.from_vec_env(env)
    env.reset(return_obs=False)
    orc.reset()
    A, F, Fp = agents, env.F, env.codes_pitch
    table = env.obs_table()
    d_codes, d_obs = env.alloc((T, n, A, Fp), np.uint8), env.alloc((T, n, A, F), np.float64)
    d_rew, d_u = env.alloc((T, n, A), np.float64), env.alloc((T, n, A), np.uint8)
    lib = orc.oracle.lib
    for launch, with_obs in enumerate((False, True)):
        env.rollout_compact(T, seed, launch * T, d_codes, d_obs if with_obs else None, d_rew, None, d_u)
        env.sync()
        codes, rew, trunc = d_codes.to_host(), d_rew.to_host(), d_u.to_host()
        obs = d_obs.to_host() if with_obs else None
        for t in range(T):
            acts = np.array([[lib.czo_action(seed, e, a, launch * T + t, env.n_actions) for a in range(A)] for e in range(n)], dtype=np.int32)
            oo, ro, to, uo = orc.step(acts)
            assert np.array_equal(bits(table[codes[t][:, :, :F]]), bits(oo)), (launch, t)
            assert (codes[t][:, :, F:] == 255).all()
            if with_obs:
                assert np.array_equal(bits(obs[t]), bits(oo)), (launch, t)
            assert np.array_equal(bits(rew[t]), bits(ro)) and np.array_equal(trunc[t], uo)
        assert np.array_equal(strip(env.get_state()), orc.records)
    env.close()


def test_ring_runs_with_compact_output_set_on_the_handle():
    """cz_set_compact_output: ring runs (graph replay) write the codes too; switching it off restores the plain
    kernels, and the graphs captured before the switch are not replayed with a stale pointer"""
    from oracle_binding import VecOracle
    n, A, period, K = 192, 2, 8, 20
    env = make(n, "coop_test", "example", A, ["TomatoLettuceSalad", "CarrotBanana"], "scheme3", max_steps=25)
    orc = VecOracle.from_vec_env(env)
    env.reset(return_obs=False)
    orc.reset()
    table = env.obs_table()
    rng = np.random.default_rng(9)
    ring_host = rng.integers(0, 5, size=(period, n, A), dtype=np.int32)
    d_ring = env.alloc((period, n, A), np.int32)
    d_ring.from_host(ring_host)
    d_obs, d_codes = env.alloc((n, A, env.F), np.float64), env.alloc((n, A, env.codes_pitch), np.uint8)
    outs = [env.alloc((n, A), np.float64), env.alloc((n, A), np.uint8), env.alloc((n, A), np.uint8)]
    step = 0
    for phase, (codes, obs) in enumerate([(None, d_obs), (d_codes, None), (d_codes, d_obs), (None, d_obs)]):
        env.set_compact_output(codes)
        if codes is not None:
            d_codes.from_host(np.zeros((n, A, env.codes_pitch), dtype=np.uint8))
        env.step_device_ring(K, d_ring, n * A, period, step % period, obs, *outs)
        env.sync()
        for k in range(K):
            oo, ro, to, uo = orc.step(ring_host[(step + k) % period], k == K - 1)
        step += K
        if obs is not None:
            assert np.array_equal(bits(d_obs.to_host()), bits(oo)), phase
        if codes is not None:
            assert np.array_equal(bits(table[d_codes.to_host()[:, :, :env.F]]), bits(oo)), phase
        assert np.array_equal(strip(env.get_state()), orc.records), phase
    env.close()


@pytest.mark.parametrize("scheme,level,agents,recipes,meta", [CASES[0], CASES[1], CASES[4], CASES[6], CASES[-1]])
def test_first_observation_as_codes_after_reset_and_set_state(scheme, level, agents, recipes, meta):
    """cz_observe_compact / cz_observe_device: the compact form of observe() before any step has run - after reset and after
    set_state - decodes to the float64 observation bit for bit; the padding bytes of a row are 255 and entry 255 of the table
    is 0.0 (what include/cookingzoo.h promises a consumer that reads whole rows)."""
    from oracle_binding import VecOracle
    n = 96
    env = make(n, level, meta, agents, recipes, scheme)
    orc = VecOracle.from_vec_env(env)
    table = env.obs_table()
    assert table[soa.LUT_ABSENT] == 0.0 and soa.LUT_ABSENT == 255
    F, Fp = env.F, env.codes_pitch
    codes = env.reset(return_codes=True)
    assert codes.shape == (n, agents, Fp) and (codes[:, :, F:] == 255).all()
    assert np.array_equal(bits(table[codes[:, :, :F]]), bits(orc.reset()))
    for _ in range(12):
        acts = np.random.default_rng(1).integers(0, env.n_actions, size=(n, agents), dtype=np.int32)
        env.step(acts, return_obs=False)
        oo, *_ = orc.step(acts)
    assert np.array_equal(bits(table[env.observe_compact()[:, :, :F]]), bits(oo))
    # into device buffers, both forms at once, for a part of the batch
    d_obs, d_codes = env.alloc((40, agents, F), np.float64), env.alloc((40, agents, Fp), np.uint8)
    d_codes.from_host(np.zeros((40, agents, Fp), np.uint8))
    env.observe_device(d_obs, d_codes, env_begin=17, env_count=40)
    env.sync()
    assert np.array_equal(bits(d_obs.to_host()), bits(oo[17:57]))
    got = d_codes.to_host()
    assert np.array_equal(bits(table[got[:, :, :F]]), bits(oo[17:57])) and (got[:, :, F:] == 255).all()
    from cooking_zoo_amd import _native
    with pytest.raises(_native.NativeError, match="null"):
        env.observe_device(None, None)
    env.close()


def test_set_layouts_on_a_multi_level_spawning_batch_is_refused():
    """replacing the whole pool of a multi-level batch with despawn / respawn on would lose the layout -> level map"""
    env = make(32, ["coop_test", "switch_test"], "example", 2, ["TomatoSalad", "TomatoSalad"], "scheme3", agent_despawn_rate=0.1,
               agent_respawn_rate=0.2, spawn_seed=1)
    with pytest.raises(ValueError, match="multi-level"):
        env.set_layouts(env.layouts[:4])
    # ... and on the C side a reloaded pool invalidates the spawn tables until cz_set_spawn is called again
    from cooking_zoo_amd import _native
    env._upload_layouts()
    with pytest.raises(_native.NativeError, match="cz_set_spawn"):
        env.step(np.zeros((32, 2), np.int32))
    env._set_spawn()
    env.reset(return_obs=False)
    env.step(np.zeros((32, 2), np.int32))
    env.close()
