"""Fresh layouts under a stepping batch (cz_update_layouts / cz_set_layout_group, CookingVecEnv.rotate_layouts): the batched
counterpart of the reference instantiating a new level at every reset (cooking_env.py:191-195, parsing.py:21-151)."""
import time

import numpy as np
import pytest

from cooking_zoo_amd import soa

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def strip(recs):
    r = recs.copy()
    r[:, soa.RET_WORD0:soa.RET_WORD0 + 8] = 0          # running returns: device-side statistics only
    return r


def make(n, max_steps, num_layouts):
    from cooking_zoo_amd.vec_env import CookingVecEnv
    return CookingVecEnv(n, "coop_test", "example", 2, max_steps, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3",
                         num_layouts=num_layouts, layout_seed=11, auto_reset=True)


def test_rotating_pool_is_bit_exact_and_uses_new_layouts():
    """512 envs, 2000 steps, max_steps 20, a 64-slot pool in two parts, switched every 50 steps and refilled 22 steps after
    each switch: every output of every step equals the oracle's, to which the same switches / refills are applied at the
    same step indices; far more than 64 distinct layouts are played on."""
    from oracle_binding import VecOracle
    n, T = 512, 2000
    env = make(n, 20, 64)
    orc = VecOracle.from_vec_env(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    keys = {l.key() for l in env.layouts}
    assert 32 < len(keys) <= 64             # (coop_test has ~1800 distinct instantiations: a few of 64 draws coincide)
    env.rotate_layouts(50, groups=2, seed=3, prefetch=2)
    seen = 0
    played = set()
    rng = np.random.default_rng(2)
    for t in range(T):
        for ev in env.rotation_events[seen:]:               # what the env switched / replaced after its previous step
            orc.apply_rotation_event(ev)
            if ev[1] == "layouts":
                keys |= {l.key() for l in ev[3]}
        seen = len(env.rotation_events)
        acts = rng.integers(0, env.n_actions, size=(n, 2), dtype=np.int32)
        og, rg, tg, ug = env.step(acts)
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(og), bits(oo)), f"observation at step {t}"
        assert np.array_equal(bits(rg), bits(ro)) and np.array_equal(tg, to) and np.array_equal(ug, uo), f"rewards / flags at step {t}"
        if t % 97 == 0:
            recs = env.get_state()
            assert np.array_equal(strip(recs), orc.records), f"records at step {t}"
            played |= {env.layouts[i].key() for i in np.unique(recs[:, soa.W_LAYOUT])}
    assert np.array_equal(strip(env.get_state()), orc.records)
    refills = [ev for ev in env.rotation_events if ev[1] == "layouts"]
    switches = [ev for ev in env.rotation_events if ev[1] == "group"]
    assert len(refills) >= 30 and len(switches) >= 35
    assert len(keys) > 64 + 200, f"the refills brought {len(keys)} distinct layouts into the pool"
    assert len(played) > 64, f"envs were seen playing on {len(played)} distinct layouts"
    from cooking_zoo_amd import _native
    assert _native.lib().cz_layout_updates(env._h) == 32 * len(refills)
    env.close()


def test_non_blocking_rotation_replays_from_its_events():
    """rotate_layouts(blocking=False): a refill whose layouts the background process has not finished yet is retried at the next
    call instead of waited for, so the schedule depends on timing - but `rotation_events` still says what was switched /
    replaced before which step, and the oracle fed with them stays bit-exact."""
    from oracle_binding import VecOracle
    n, T = 256, 700
    env = make(n, 20, 32)
    orc = VecOracle.from_vec_env(env)
    assert np.array_equal(bits(env.reset()), bits(orc.reset()))
    env.rotate_layouts(23, groups=2, seed=9, prefetch=1, blocking=False)
    seen = 0
    rng = np.random.default_rng(6)
    for t in range(T):
        for ev in env.rotation_events[seen:]:
            orc.apply_rotation_event(ev)
        seen = len(env.rotation_events)
        acts = rng.integers(0, env.n_actions, size=(n, 2), dtype=np.int32)
        og, rg, tg, ug = env.step(acts)
        oo, ro, to, uo = orc.step(acts)
        assert np.array_equal(bits(og), bits(oo)), f"observation at step {t}"
        assert np.array_equal(bits(rg), bits(ro)) and np.array_equal(tg, to) and np.array_equal(ug, uo), f"rewards / flags at step {t}"
    assert np.array_equal(strip(env.get_state()), orc.records)
    refills = [ev for ev in env.rotation_events if ev[1] == "layouts"]
    switches = [ev for ev in env.rotation_events if ev[1] == "group"]
    assert len(switches) >= 2 and len(refills) >= 1, (len(switches), len(refills))
    env.close()


def test_rotation_keeps_the_step_rate():
    """Device-resident stepping (2000 steps as 40 ring calls of 50) with the pool rotating (switch every 500 steps, refills
    prepared ahead by the background thread) against the same run on a static pool: within 3 %."""
    from oracle_binding import VecOracle
    n, K, calls, period = 512, 50, 40, 64

    def run(rotate):
        env = make(n, 20, 64)
        env.reset(return_obs=False)
        rng = np.random.default_rng(4)
        ring_host = rng.integers(0, env.n_actions, size=(period, n, 2), dtype=np.int32)
        d_ring = env.alloc((period, n, 2), np.int32)
        d_ring.from_host(ring_host)
        d_obs, d_rew = env.alloc((n, 2, env.F), np.float64), env.alloc((n, 2), np.float64)
        d_t, d_u = env.alloc((n, 2), np.uint8), env.alloc((n, 2), np.uint8)
        orc = VecOracle.from_vec_env(env)
        orc.reset()
        if rotate:
            env.rotate_layouts(500, groups=2, seed=5, prefetch=4)
            deadline = time.monotonic() + 60
            while env.rotation_ready() < 4 and time.monotonic() < deadline:
                time.sleep(0.01)
        env.step_device_ring(K, d_ring, n * 2, period, 0, d_obs, d_rew, d_t, d_u)      # graphs captured, caches warm
        env.sync()
        t0 = time.perf_counter()
        for c in range(1, calls):
            env.step_device_ring(K, d_ring, n * 2, period, (c * K) % period, d_obs, d_rew, d_t, d_u)
        env.sync()
        dt = time.perf_counter() - t0
        # the same run on the oracle: events applied at the call boundaries they were issued at
        evs = list(env.rotation_events)
        step = 0
        for c in range(calls):
            for ev in [e for e in evs if e[0] == step]:
                orc.apply_rotation_event(ev)
            for k in range(K):
                orc.step(ring_host[(c * K + k) % period], False)
            step += K
        assert np.array_equal(strip(env.get_state()), orc.records)
        n_refills = sum(1 for e in evs if e[1] == "layouts")
        env.close()
        return dt, n_refills

    # (wall-clock of 9 ms runs on a shared box: the best of up to six runs each, taken alternately)
    best, log = {}, []
    for attempt in range(6):
        for rotate in (False, True):
            dt, n_refills = run(rotate)
            best[rotate] = min(best.get(rotate, 1e9), dt)
            log.append((rotate, round(dt * 1e3, 2)))
            if rotate:
                assert n_refills >= 2
        if attempt >= 1 and best[True] <= best[False] * 1.03:
            break
    print("rotation rate runs (rotating?, ms):", log)
    assert best[True] <= best[False] * 1.03, f"rotating {best[True] * 1e3:.2f} ms against static {best[False] * 1e3:.2f} ms for {(calls - 1) * K} steps: {log}"
