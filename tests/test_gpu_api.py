"""GPU: the drop-in surface (parallel_env / gym-style wrappers) against wrapper-level traces captured from the reference."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = json.load(open(os.path.join(HERE, "golden", "api_traces.json")))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("case", TRACES, ids=lambda c: f"{c['kwargs']['level']}-A{c['kwargs']['num_agents']}-{c['kwargs']['action_scheme']}")
def test_parallel_env_matches_reference_trace(case):
    from cooking_zoo_amd.environment.cooking_env import parallel_env
    random.seed(case["seed"])
    np.random.seed(case["seed"])
    env = parallel_env(**case["kwargs"])
    assert env.possible_agents == case["possible_agents"]
    assert list(env.observation_space("player_0").shape) == case["obs_shape"]
    assert env.action_space("player_0").n == case["n_actions"]
    obs, infos = env.reset()
    assert infos == {a: {} for a in env.possible_agents}
    for a, o in case["reset_obs"].items():
        assert obs[a].dtype == np.float64 and np.array_equal(bits(obs[a]), bits(np.array(o)))
    for st in case["steps"]:
        # (with despawn / respawn rates > 0 the action dict follows env.agents: it was drawn on the fly and recorded)
        o, r, te, tr, inf = env.step(st.get("action_dict") or {f"player_{i}": a for i, a in enumerate(st["actions"])})
        assert set(o) == set(st["obs"]) and set(r) == set(st["rewards"]) and set(inf) == set(st["infos"])
        for a in st["obs"]:
            assert np.array_equal(bits(o[a]), bits(np.array(st["obs"][a]))), a
            assert isinstance(r[a], np.float64) and np.array_equal(bits(r[a]), bits(np.array(st["rewards"][a])))
            assert te[a] == st["terminations"][a] and tr[a] == st["truncations"][a]
            ref_info = st["infos"][a]
            assert set(inf[a]) == set(ref_info)
            assert inf[a]["t"] == ref_info["t"] and inf[a]["termination_info"] == ref_info["termination_info"]
            assert inf[a]["recipe_done"] == ref_info["recipe_done"] and inf[a]["action"] == ref_info["action"]
            assert inf[a]["task"] == ref_info["task"] and list(inf[a]["goal_vector"]) == ref_info["goal_vector"]
        assert env.agents == st["agents_after"]
    if not env.agents:
        with pytest.raises(RuntimeError):
            env.step({a: 0 for a in env.possible_agents})
    env.close()


def test_full_reset_false_reuses_layout_and_gym_wrappers():
    from cooking_zoo_amd.environment import GymCookingEnvironment, GymCookingEnvironmentMA
    random.seed(3)
    g = GymCookingEnvironment("coop_test", "example", 20, ["TomatoLettuceSalad"], action_scheme="scheme3")
    o1, i1 = g.reset()
    assert o1.shape == (278,) and i1 == {}
    first = g.zoo_env.world.cells.copy()
    g.zoo_env.reset(options={"full_reset": False})
    assert np.array_equal(g.zoo_env.world.cells, first)
    o, r, te, tr, info = g.step(0)
    assert o.shape == (278,) and r == -5 / 20 and te is False and tr is False and info["t"] == 1
    g.close()
    m = GymCookingEnvironmentMA("coop_test", "example", 2, 3, ["TomatoLettuceSalad", "CarrotBanana"], action_scheme="scheme3")
    obs, infos = m.reset()
    assert len(obs) == 2 and len(infos) == 2
    for t in range(3):
        obs, rew, te, tr, inf = m.step([1, 2])
    assert tr == [True, True] and m.zoo_env.agents == []
    m.close()


def test_unsupported_modes_fail_loudly():
    from cooking_zoo_amd.environment.cooking_env import parallel_env
    kw = dict(level="coop_test", meta_file="example", num_agents=1, max_steps=10, recipes=["TomatoSalad"])
    with pytest.raises(AssertionError):
        parallel_env(obs_spaces=["pixels"], action_scheme="scheme3", **kw)         # cooking_env.py:85-86
    with pytest.raises(AttributeError):
        parallel_env(action_scheme="scheme2", **kw)


def test_last_marks_matches_state_on_both_host_paths():
    """cz_last_marks (what the facade reports as recipe_done) == record word 1, on the zero-copy small-batch path of
    cz_step and on the staged-copy path."""
    from cooking_zoo_amd import soa
    from cooking_zoo_amd.vec_env import CookingVecEnv
    for n in (3, 700):                      # 3 envs: pinned device-mapped block; 700 envs x 2 x 278 x 8 B > 256 KiB: staged copies
        env = CookingVecEnv(n, "coop_test", "example", 2, 30, ["TomatoLettuceSalad", "CarrotBanana"],
                            action_scheme="scheme3", num_layouts=4, auto_reset=True)
        env.reset()
        with pytest.raises(Exception):
            env.last_marks()                # no step yet
        rng = np.random.default_rng(n)
        for t in range(40):
            obs, rew, term, trunc = env.step(rng.integers(0, 5, size=(n, 2)))
            assert np.array_equal(env.last_marks(), env.get_state()[:, soa.W_MARKS].astype(np.uint64)), (n, t)
        assert np.array_equal(bits(obs), bits(env.observe()))
        env.close()


import gzip  # noqa: E402
AEC_TRACES = json.loads(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "aec_traces.json.gz")).read())


@pytest.mark.parametrize("case", AEC_TRACES, ids=lambda c: f"{c['kwargs']['level']}-A{c['kwargs']['num_agents']}-{c['kwargs']['action_scheme']}"
                                                          + ("-spawn" if c['kwargs'].get('agent_despawn_rate') else ""))
def test_aec_env_matches_reference_trace(case):
    """The agent-iterator API (cooking_env.py:215-241) call by call: selected agent, last() tuple, and the bookkeeping
    dicts after every step(action | None), through a truncation (agent list empties) and a termination (no progress)."""
    from cooking_zoo_amd.environment.cooking_env import env as aec_env
    random.seed(case["seed"])
    np.random.seed(case["seed"])
    e = aec_env(**case["kwargs"])
    e.reset()
    assert e.agent_selection == case["after_reset"]["agent_selection"] and e.agents == case["after_reset"]["agents"]
    for k, call in enumerate(case["calls"]):
        assert e.agents, k
        assert e.agent_selection == call["agent"], k
        obs, cum, term, trunc, info = e.last()
        ref = call["last"]
        assert np.array_equal(bits(obs), bits(np.array(ref["obs"]))), k
        assert np.array_equal(bits(np.float64(cum)), bits(np.float64(ref["reward"]))), (k, cum, ref["reward"])
        assert (term, trunc) == (ref["termination"], ref["truncation"]), k
        assert set(info) == set(ref["info"]) and all(
            (list(info[q]) if q == "goal_vector" else info[q]) == ref["info"][q] for q in info), k
        e.step(call["action"])
        after = call["after"]
        assert e.agents == after["agents"], k
        if after["agents"]:
            assert e.agent_selection == after["agent_selection"], k
        for name, want in (("rewards", after["rewards"]), ("_cumulative_rewards", after["cumulative"])):
            got = getattr(e, name)
            assert set(got) == set(want), (k, name)
            for a in want:
                assert np.array_equal(bits(np.float64(got[a])), bits(np.float64(want[a]))), (k, name, a, got[a], want[a])
        assert {a: bool(v) for a, v in e.terminations.items()} == after["terminations"], k
        assert {a: bool(v) for a, v in e.truncations.items()} == after["truncations"], k
        assert e.t == after["t"], k
    if case["kwargs"].get("agent_despawn_rate"):
        e.close()
        return
    # agent_iter drives the same loop
    e.reset()
    seen = []
    for agent in e.agent_iter(max_iter=2 * len(e.possible_agents)):
        seen.append(agent)
        e.step(0)
    assert seen == (e.possible_agents * 2)[:len(seen)] and len(seen) == 2 * len(e.possible_agents)
    with pytest.raises(ValueError):
        dead = aec_env(**{**case["kwargs"], "max_steps": 1})
        dead.reset()
        for _ in dead.possible_agents:
            dead.step(0)
        dead.step(0)                                  # truncated: only None is valid now
    e.close()



def test_pinned_output_buffers_give_the_same_results():
    """pinned_outputs=True: step / reset / observe fill page-locked buffers owned by the env (views, overwritten by the
    next call); results equal the fresh-array path bit for bit, on the staged-copy path and the small zero-copy path."""
    from cooking_zoo_amd.vec_env import CookingVecEnv
    for n in (3, 700):
        kw = dict(action_scheme="scheme3", num_layouts=4, auto_reset=True)
        a = CookingVecEnv(n, "coop_test", "example", 2, 30, ["TomatoLettuceSalad", "CarrotBanana"], pinned_outputs=True, **kw)
        b = CookingVecEnv(n, "coop_test", "example", 2, 30, ["TomatoLettuceSalad", "CarrotBanana"], **kw)
        oa, ob = a.reset(), b.reset()
        assert np.array_equal(oa.view(np.uint64), ob.view(np.uint64))
        rng = np.random.default_rng(n)
        first = None
        for t in range(40):
            acts = rng.integers(0, 5, size=(n, 2), dtype=np.int32)
            ra, rb = a.step(acts), b.step(acts)
            for x, y in zip(ra, rb):
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), t
            if first is None:
                first = ra[0]
        again_a, again_b = a.step(acts), b.step(acts)
        assert first is not None and np.shares_memory(first, again_a[0])          # a view of the same pinned buffer
        assert np.array_equal(again_a[0].view(np.uint64), again_b[0].view(np.uint64))
        assert np.array_equal(a.observe().view(np.uint64), b.observe().view(np.uint64))
        assert np.array_equal(a.get_state(), b.get_state())
        a.close(); b.close()
