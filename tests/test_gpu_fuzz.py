"""GPU: randomised differential test of the HIP path against the oracle -- random level / scheme / agent count / recipe
assignment / horizon / reward scheme / end condition per case, and (round 4) despawn / respawn rates, a user recipe book with
wide tables, a layout pool that is switched and refreshed mid-run; a fused rollout of T steps over the counter-based action
stream, then final records, last observation and statistics must agree bit for bit.

Round 5: every case is also run under a BIASED action source (tests/fuzz_policy.py: walk to a cell that holds an object or is
an action object and use it; scheme1 draws EXECUTE / PICK_UP_SPECIAL with raised probability) computed from the oracle's
records, fed to the device through cz_rollout_actions and the one-step paths - uniform random actions hardly ever plate a
dish, absorb from a counter, deliver or complete a recipe.  Which transitions a run exercised is counted from the oracle's
record diffs; the last test of the module prints the table and asserts that nothing stayed at zero (all events in a soak,
the common ones in the default handful).

Default: a handful of cases (seconds).  CZ_FUZZ_CASES=N widens it for one-off soak runs (200 cases are about
37 M env-steps); CZ_FUZZ_FIRST=K starts at case K (cases are a function of their index: earlier soaks covered 0 ... 1499)."""
import os

import numpy as np
import pytest

from cooking_zoo_amd import soa
from test_gpu_rollout import bits, make, oracle_for, strip

pytestmark = pytest.mark.gpu

LEVELS = [("coop_test", "example", 2), ("coexistence_test", "example", 2), ("switch_test", "example", 2),
          ("large_16x16", "large_16x16", 4), ("crowded_6x5", "crowded_6x5", 4), ("edge_8x8", "edge", 3),
          ("edge_9x8", "edge", 3), ("edge_empty", "edge", 3), ("limit_32x8", "limits", 3), ("limit_8x31", "limits", 3), ("dense_16x16", "dense_16x16", 4),
          ("huge_32x32", "huge_32x32", 4), ("huge_20x20", "huge_20x20", 3), ("huge_objs_16x16", "huge_objs_16x16", 3)]
BOOK = ["TomatoSalad", "TomatoLettuceSalad", "CarrotBanana", "MashedCarrotBanana", "CucumberOnion", "AppleWatermelon",
        "TomatoLettuceOnionSalad", "no_recipe"]
N_CASES = int(os.environ.get("CZ_FUZZ_CASES", "6"))
FIRST_CASE = int(os.environ.get("CZ_FUZZ_FIRST", "0"))          # soak runs on cases no earlier run has drawn: CZ_FUZZ_FIRST=2000


def draw_case(i):
    rng = np.random.default_rng(1000 + i)
    level, meta, max_agents = LEVELS[int(rng.integers(len(LEVELS)))]
    agents = int(rng.integers(1, max_agents + 1))
    n_rec = int(rng.integers(agents, 5))
    recipes = [BOOK[int(k)] for k in rng.integers(len(BOOK), size=n_rec)]
    reward = None
    if rng.random() < 0.5:
        reward = {"recipe_reward": float(rng.integers(1, 30)), "max_time_penalty": float(-rng.integers(0, 9)),
                  "recipe_penalty": float(-rng.integers(0, 50)), "recipe_node_reward": float(rng.integers(0, 4))}
    kw = dict(level=level, meta_file=meta, num_agents=agents, recipes=recipes,
              action_scheme="scheme3" if rng.random() < 0.6 else "scheme1", max_steps=int(rng.integers(5, 260)),
              end_condition_all_dishes=bool(rng.random() < 0.3), reward_scheme=reward,
              num_layouts=int(rng.integers(1, 40)), layout_seed=int(rng.integers(1 << 20)))
    seed = int(rng.integers(1 << 30))
    # ---- the device paths of rounds 3 / 4 (drawn from a second stream, so that the cases above stay what they were)
    rng2 = np.random.default_rng(77000 + i)
    extra = {"wide": False, "rotate": False}
    spawn_ok = not (level == "edge_empty" and agents > 2)     # (that level file places two agents only: no spawn area for a third)
    if rng2.random() < 0.4 and spawn_ok:              # despawn / respawn inside the kernels (cz_set_spawn)
        kw.update(agent_despawn_rate=float(rng2.choice([0.02, 0.1, 0.3])), agent_respawn_rate=float(rng2.choice([0.05, 0.25, 0.6])),
                  grace_period=int(rng2.integers(0, 6)), spawn_seed=int(rng2.integers(1 << 40)))
    if rng2.random() < 0.2:                           # a user recipe book with a 10-node graph: wide recipe tables
        extra["wide"] = True
        kw["recipes"] = [WIDE_BOOK[int(k)] for k in rng2.integers(len(WIDE_BOOK), size=len(recipes))]
    if rng2.random() < 0.35:                          # the layout pool cut in two, switched and refreshed under the running batch
        extra["rotate"] = True
        kw["num_layouts"] = 2 * int(rng2.integers(1, 12))
    extra["rng"] = rng2
    return kw, seed, extra


WIDE_BOOK = ["FruitFeast", "PickyBanana", "BreadSnack"]


def fresh_layouts(env, count, rng):
    """`count` newly instantiated layouts per level (the reference's draw order, engine/load_level.py)"""
    import random
    from cooking_zoo_amd.cooking_world.engine import load_level as ll
    r = random.Random(int(rng.integers(1 << 30)))
    return [[ll.instantiate(lv, env.meta, env.num_agents, r) for _ in range(count)] for lv in env.level_objects]


@pytest.mark.parametrize("i", range(FIRST_CASE, FIRST_CASE + N_CASES))
def test_random_configuration_matches_oracle(i):
    kw, seed, extra = draw_case(i)
    if extra["wide"]:
        from cooking_zoo_amd.cooking_book import recipe_drawer as rd
        from test_custom_recipes import register_fixture_recipes
        assert not rd.RECIPE_STORE
        register_fixture_recipes()
    try:
        run_case(i, kw, seed, extra)
    finally:
        if extra["wide"]:
            rd.RECIPE_STORE.clear()


def run_case(i, kw, seed, extra):
    n, T = 384, 480
    env = make(n, **kw)
    orc = oracle_for(env)
    env.reset(return_obs=False)
    orc.reset()
    if extra["wide"]:
        assert env.recipe_nodes == 16
    A = kw["num_agents"]
    chunk = T // 3
    d_rew = env.alloc((chunk, n, A), np.float64)
    d_term = env.alloc((chunk, n, A), np.uint8)
    d_trunc = env.alloc((chunk, n, A), np.uint8)
    ctx = f"case {i}: {kw}"
    for t0 in range(0, T, chunk):                   # three launches: the state round-trips through HBM in between
        if extra["rotate"] and t0 == chunk:         # from here the envs draw their next layouts from the second half of the pool
            env.set_layout_group(2, 1)
            orc.set_layout_group(2, 1)
        if extra["rotate"] and t0 == 2 * chunk and kw["max_steps"] + 2 <= chunk:
            # every episode that started on the first half is over: refresh it (cz_update_layouts) and go back to it
            half = kw["num_layouts"] // 2
            for (base, count), lays in zip(env.pool_slices, fresh_layouts(env, half, extra["rng"])):
                env.update_layouts(base, lays)
                orc.update_layouts(base, lays)
            env.set_layout_group(2, 0)
            orc.set_layout_group(2, 0)
        done = 0
        if t0 == chunk:                             # the second launch starts with 16 steps of the codes-only fused kernel
            done = 16
            d_codes = env.alloc((done, n, A, env.codes_pitch), np.uint8)
            env.rollout_compact(done, seed, t0, d_codes, None, d_rew, d_term, d_trunc)
            env.sync()
            oo, ro, to, uo = orc.rollout(done, seed, t0)
            assert np.array_equal(strip(env.get_state()), orc.records), ctx
            assert np.array_equal(bits(env.obs_table()[d_codes.to_host()[-1][:, :, :env.F]]), bits(oo)), ctx
            assert np.array_equal(bits(d_rew.to_host()[done - 1]), bits(ro)), ctx
            d_codes.free()
        env.rollout(chunk - done, seed, t0 + done, None, d_rew, d_term, d_trunc)
        env.sync()
        oo, ro, to, uo = orc.rollout(chunk - done, seed, t0 + done)
        assert np.array_equal(strip(env.get_state()), orc.records), ctx
        assert np.array_equal(bits(d_rew.to_host()[chunk - done - 1]), bits(ro)), ctx
        assert np.array_equal(d_term.to_host()[chunk - done - 1], to) and np.array_equal(d_trunc.to_host()[chunk - done - 1], uo), ctx
        assert np.array_equal(bits(env.observe()), bits(oo)), ctx
    # ... and on from there with one launch per step and actions from the host (the other kernel variant)
    rng = np.random.default_rng(seed)
    n_act = 5 if kw["action_scheme"] == "scheme3" else 8
    table = env.obs_table()
    for t in range(25):
        acts = rng.integers(0, n_act, size=(n, A), dtype=np.int32)
        if t % 4 == 3:                               # the compact observation (its own kernel instance): decoded, it is the same
            codes, r, te, tr = env.step_compact(acts)
            o = table[codes[:, :, :env.F]]
        else:
            o, r, te, tr = env.step(acts)
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(o), bits(oo)) and np.array_equal(bits(r), bits(ro)), (ctx, t)
        assert np.array_equal(te, to) and np.array_equal(tr, uo), (ctx, t)
    assert np.array_equal(strip(env.get_state()), orc.records), ctx
    # ... and a ring run replayed from graphs (cz_step_device_ring)
    from cooking_zoo_amd import _native
    K, period = int(rng.integers(2, 40)), int(rng.integers(2, 12))
    first = int(rng.integers(period))
    ring_host = rng.integers(0, n_act, size=(period, n, A), dtype=np.int32)
    d_ring, d_obs = env.alloc((period, n, A), np.int32), env.alloc((n, A, env.F), np.float64)
    d_ring.from_host(ring_host)
    _native.check(env._h, _native.lib().cz_step_device_ring(env._h, K, d_ring.ptr, n * A, period, first, d_obs.ptr, d_rew.ptr,
                                                            d_term.ptr, d_trunc.ptr))
    env.sync()
    for k in range(K):
        oo, ro, to, uo = orc.step(ring_host[(first + k) % period], k == K - 1)
    assert np.array_equal(strip(env.get_state()), orc.records), ctx
    assert np.array_equal(bits(d_obs.to_host()), bits(oo)) and np.array_equal(bits(d_rew.to_host()[0]), bits(ro)), ctx
    assert np.array_equal(d_term.to_host()[0], to) and np.array_equal(d_trunc.to_host()[0], uo), ctx
    # ... and the same kind of run as fused launches over the ring's rows (cz_set_ring_fused: outputs written in place)
    env.set_ring_fused(True)
    K2, first2 = int(rng.integers(2, 40)), int(rng.integers(period))
    _native.check(env._h, _native.lib().cz_step_device_ring(env._h, K2, d_ring.ptr, n * A, period, first2, d_obs.ptr, d_rew.ptr,
                                                            d_term.ptr, d_trunc.ptr))
    env.sync()
    assert env.ring_fused_steps() == K2, ctx
    for k in range(K2):
        oo, ro, to, uo = orc.step(ring_host[(first2 + k) % period], k == K2 - 1)
    assert np.array_equal(strip(env.get_state()), orc.records), ctx
    assert np.array_equal(bits(d_obs.to_host()), bits(oo)) and np.array_equal(bits(d_rew.to_host()[0]), bits(ro)), ctx
    assert np.array_equal(d_term.to_host()[0], to) and np.array_equal(d_trunc.to_host()[0], uo), ctx
    st = env.stats()
    assert st["episodes"] == int(orc.records[:, soa.W_EPISODE].sum()) + int((orc.records[:, soa.W_STATUS] & 1).sum()), ctx
    env.close()


# ---------------------------------------------------------------------------------------------------------------------------
# the same configurations under the biased action source, with event accounting
# ---------------------------------------------------------------------------------------------------------------------------
from fuzz_policy import EVENTS, BumperActions, EventCounter   # noqa: E402

EVENT_TOTALS = {k: 0 for k in EVENTS}
EVENT_STEPS = [0, 0]                      # live env-steps, cases
# ... and the same per kernel instance (1 slot / 1 cell per lane, 2 / 4, 4 / 16: cz_kernels.h), so that a soak says what each one reached
INSTANCES = ("small", "large", "huge")
EVENT_BY_INSTANCE = {n: {k: 0 for k in EVENTS} for n in INSTANCES}
CASES_BY_INSTANCE = {n: 0 for n in INSTANCES}


def instance_of(dims):
    return "small" if dims.D <= 64 and dims.C <= 64 else "large" if dims.D <= 128 and dims.C <= 256 else "huge"


def event_table():
    lines = [f"event coverage of {EVENT_STEPS[1]} biased cases, {EVENT_STEPS[0]} live env-steps (device == oracle on all of them); "
             f"cases per kernel instance: " + ", ".join(f"{n} {CASES_BY_INSTANCE[n]}" for n in INSTANCES),
             f"  {'event':20s} {'all':>10s} " + " ".join(f"{n:>10s}" for n in INSTANCES)]
    for k in EVENTS:
        lines.append(f"  {k:20s} {EVENT_TOTALS[k]:10d} " + " ".join(f"{EVENT_BY_INSTANCE[n][k]:10d}" for n in INSTANCES))
    return "\n".join(lines)
# the default handful must reach these; a soak (>= 200 cases) must reach every event of fuzz_policy.EVENTS
CORE_EVENTS = ["pick_up", "put_down", "chop", "plate_add", "static_accepts", "delivery", "marks_changed", "truncation"]


@pytest.mark.parametrize("i", range(FIRST_CASE, FIRST_CASE + N_CASES))
def test_biased_actions_match_oracle(i):
    kw, seed, extra = draw_case(i)
    if extra["wide"]:
        from cooking_zoo_amd.cooking_book import recipe_drawer as rd
        from test_custom_recipes import register_fixture_recipes
        assert not rd.RECIPE_STORE
        register_fixture_recipes()
    try:
        run_biased_case(i, kw, seed)
    finally:
        if extra["wide"]:
            rd.RECIPE_STORE.clear()


def run_biased_case(i, kw, seed):
    n, chunk, n_chunks, singles = 256, 96, 3, 48
    if kw["level"].startswith(("huge", "limit", "dense")):  # (the policy's distance maps over 1024 cells are the slow part, on the host)
        n = 96
    kw = dict(kw)
    kw["max_steps"] = max(kw["max_steps"], 60)              # deep states need a few dozen steps (the uniform half keeps short horizons)
    env = make(n, **kw)
    orc = oracle_for(env)
    env.reset(return_obs=False)
    orc.reset()
    A, F = kw["num_agents"], env.F
    ctx = f"biased case {i}: {kw}"
    rng = np.random.default_rng(seed ^ 0x5EED)
    pol = BumperActions(env.dims, env.scheme_class.CODE, rng)
    ev = EventCounter(env.dims)

    def oracle_steps(T, want_last_obs=True):
        """T oracle steps under the policy: the actions, every step's rewards / flags, the last observation"""
        acts = np.empty((T, n, A), np.int32)
        rew, term, trunc = np.empty((T, n, A)), np.empty((T, n, A), np.uint8), np.empty((T, n, A), np.uint8)
        obs = None
        for t in range(T):
            before = orc.records.copy()
            acts[t] = pol.act(before)
            obs, rew[t], term[t], trunc[t] = orc.step(acts[t], want_obs=(t == T - 1) and want_last_obs)
            pol.observe_result(orc.records)
            ev.update(before, orc.records, term[t], trunc[t])
        return acts, obs, rew, term, trunc

    d_act = env.alloc((chunk, n, A), np.int32)
    d_obs, d_rew = env.alloc((chunk, n, A, F), np.float64), env.alloc((chunk, n, A), np.float64)
    d_t, d_u = env.alloc((chunk, n, A), np.uint8), env.alloc((chunk, n, A), np.uint8)
    for c in range(n_chunks):                               # fused steps over the policy's actions (kernel mode 2)
        acts, oo, ro, to, uo = oracle_steps(chunk)
        d_act.from_host(acts)
        env.rollout_actions(d_act, chunk, d_obs, d_rew, d_t, d_u)
        env.sync()
        assert np.array_equal(strip(env.get_state()), orc.records), (ctx, "records after chunk", c)
        assert np.array_equal(bits(d_rew.to_host()), bits(ro)), (ctx, "rewards of chunk", c)
        assert np.array_equal(d_t.to_host(), to) and np.array_equal(d_u.to_host(), uo), (ctx, "flags of chunk", c)
        assert np.array_equal(bits(d_obs.to_host()[-1]), bits(oo)), (ctx, "last observation of chunk", c)
    table = env.obs_table()
    for t in range(singles):                                # one launch per step (modes 0 and 3), host arrays
        acts, oo, ro, to, uo = oracle_steps(1)
        if t % 4 == 3:
            codes, r, te, tr = env.step_compact(acts[0])
            o = table[codes[:, :, :F]]
        else:
            o, r, te, tr = env.step(acts[0])
        assert np.array_equal(bits(o), bits(oo)) and np.array_equal(bits(r), bits(ro[0])), (ctx, "single step", t)
        assert np.array_equal(te, to[0]) and np.array_equal(tr, uo[0]), (ctx, "single step flags", t)
    assert np.array_equal(strip(env.get_state()), orc.records), ctx
    inst = instance_of(env.dims)
    for k in EVENTS:
        EVENT_TOTALS[k] += ev.counts[k]
        EVENT_BY_INSTANCE[inst][k] += ev.counts[k]
    CASES_BY_INSTANCE[inst] += 1
    EVENT_STEPS[0] += ev.steps
    EVENT_STEPS[1] += 1
    env.close()
    if os.environ.get("CZ_FUZZ_EVENT_LOG"):          # soak runs: the running table after every case (a killed run still leaves it)
        with open(os.environ["CZ_FUZZ_EVENT_LOG"], "w") as f:
            f.write(f"(running table; last case: {i})\n" + event_table() + "\n")


def test_zz_biased_runs_reached_the_deep_transitions():
    """(runs last in the module) the event table of the biased runs; nothing that must be reached stayed at zero"""
    if EVENT_STEPS[1] == 0:
        pytest.skip("no biased case ran in this session")
    print("\n" + event_table())
    must = EVENTS if EVENT_STEPS[1] >= 200 else (CORE_EVENTS if EVENT_STEPS[1] >= 6 else [])
    missing = [k for k in must if EVENT_TOTALS[k] == 0]
    assert not missing, f"never exercised: {missing}"
    if EVENT_STEPS[1] >= 200:
        # a soak: every kernel instance on its own reaches the interaction repertoire (its levels have the objects for it);
        # Switches exist only in levels of the small and the huge instance
        per = ["pick_up", "put_down", "chop", "plate_add", "static_accepts", "delivery", "marks_changed", "truncation", "despawn", "respawn"]
        for n in INSTANCES:
            assert CASES_BY_INSTANCE[n] > 0, f"no case ran on the {n} instance"
            lacking = [k for k in per if EVENT_BY_INSTANCE[n][k] == 0]
            assert not lacking, f"the {n} instance never exercised: {lacking}"
