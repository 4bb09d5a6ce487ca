"""CPU: the biased action source and the event accounting of the device fuzz (tests/fuzz_policy.py) against the oracle alone -
so that what test_gpu_fuzz.py relies on (deep transitions are reached, counters see them) is checked without a GPU."""
import numpy as np
import pytest

from cooking_zoo_amd.vec_env import BatchTables
from fuzz_policy import EVENTS, BumperActions, EventCounter
from oracle_binding import VecOracle


def run(level, meta, agents, scheme, recipes, steps=300, n=192, max_steps=150, **kw):
    t = BatchTables(n, level, meta, agents, max_steps, recipes, action_scheme=scheme, num_layouts=8, **kw)
    orc = VecOracle.from_vec_env(t)
    orc.reset()
    pol = BumperActions(t.dims, t.scheme_class.CODE, np.random.default_rng(3))
    ev = EventCounter(t.dims)
    for _ in range(steps):
        before = orc.records.copy()
        acts = pol.act(before)
        assert acts.dtype == np.int32 and acts.min() >= 0 and acts.max() < t.n_actions
        _, _, te, tr = orc.step(acts, want_obs=False)
        pol.observe_result(orc.records)
        ev.update(before, orc.records, te, tr)
    return ev


def test_scheme3_reaches_plating_and_delivery():
    ev = run("coop_test", "example", 2, "scheme3", ["TomatoLettuceSalad", "CarrotBanana"])
    for k in ("pick_up", "put_down", "chop", "bread_clone", "blend", "plate_add", "plate_absorb", "static_accepts", "delivery",
              "marks_changed", "plate_with_2plus", "truncation"):
        assert ev.counts[k] > 0, (k, ev.table())
    assert ev.counts["pick_up_special"] == 0                      # scheme3 has no such action (action_scheme3.py)


def test_scheme1_reaches_special_pick_up_switch_and_completion():
    ev = run("crowded_6x5", "crowded_6x5", 4, "scheme1", ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"],
             steps=420, max_steps=200)
    for k in ("chop", "blend", "blender_toggle", "plate_add", "plate_absorb", "pick_up_special", "switch_press", "delivery",
              "termination", "chopped_and_mashed"):
        assert ev.counts[k] > 0, (k, ev.table())


def test_spawn_events_are_counted():
    ev = run("coop_test", "example", 2, "scheme3", ["TomatoSalad", "TomatoSalad"], steps=120, agent_despawn_rate=0.1,
             agent_respawn_rate=0.3, grace_period=2, spawn_seed=5)
    assert ev.counts["despawn"] > 0 and ev.counts["respawn"] > 0, ev.table()


def test_uniform_actions_reach_far_less():
    """the reason the policy exists: the same budget of uniform random actions plates and delivers several times less often"""
    t = BatchTables(192, "coop_test", "example", 2, 150, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3", num_layouts=8)
    orc = VecOracle.from_vec_env(t)
    orc.reset()
    ev, rng = EventCounter(t.dims), np.random.default_rng(3)
    for _ in range(300):
        before = orc.records.copy()
        _, _, te, tr = orc.step(rng.integers(0, 5, size=(192, 2), dtype=np.int32), want_obs=False)
        ev.update(before, orc.records, te, tr)
    biased = run("coop_test", "example", 2, "scheme3", ["TomatoLettuceSalad", "CarrotBanana"])
    assert set(ev.counts) == set(EVENTS)
    for k in ("plate_add", "plate_absorb", "plate_with_2plus", "chop", "termination"):
        assert biased.counts[k] > 3 * ev.counts[k], (k, biased.counts[k], ev.counts[k])
