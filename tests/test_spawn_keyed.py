"""Pins the batched despawn / respawn rule to the REFERENCE: tests/golden/spawn_keyed_*.npz are trajectories of the unmodified
`CookingWorld.handle_agent_spawn` / `parsing.generate_location` (cooking_world.py:267-290, parsing.py:154-167) whose random
sources were answered with the keyed stream the batched env uses (tools/gen_golden.py capture_spawn_keyed_episode).  Here the
oracle (oracle/cz_oracle.c handle_agent_spawn) replays them on the CPU: records incl. the status word, observations of every
agent, rewards, flags incl. the truncated-once report; tests/test_gpu_spawn_keyed.py does the same through the kernels."""
import ctypes as C

import numpy as np
import pytest

from cooking_zoo_amd import soa, spawn
from golden_io import GoldenSet, spawn_keyed_sets
from oracle_binding import Oracle, load_lib


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def keyed_oracle(gs, ep):
    off, cells = ep.static_table()
    orc = Oracle(ep.dims, gs.meta, gs.recipe_table, [(ep.states[0], off, cells)], scheme=gs.scheme, max_steps=gs.cfg["max_steps"],
                 end_condition_all=gs.cfg["end_condition_all_dishes"], num_recipes=len(gs.cfg["recipes"]),
                 reward_scheme=gs.cfg.get("reward_scheme"), env_id_base=ep.env_id)
    d, r, g = gs.cfg["rates"]
    orc.set_spawn(d, r, g, gs.cfg["spawn_seed"], [[(xs, ys) for xs, ys in ep.spawn_areas]])
    return orc


def test_there_are_keyed_sets_for_every_kernel_instance():
    names = spawn_keyed_sets()
    dims = {n: GoldenSet(n).episodes[0].dims for n in names}
    assert any(d.D <= 64 and d.C <= 64 for d in dims.values())                                  # 1 slot / 1 cell per lane
    assert any((d.D > 64 or d.C > 64) and d.D <= 128 and d.C <= 256 for d in dims.values())     # 2 / 4
    assert any(d.D > 128 or d.C > 256 for d in dims.values())                                   # 4 / 16
    assert {GoldenSet(n).scheme for n in names} == {1, 3}
    assert {d.A for d in dims.values()} >= {2, 3, 4}


@pytest.mark.parametrize("name", spawn_keyed_sets())
def test_oracle_replays_keyed_despawn_respawn(name):
    gs = GoldenSet(name)
    n_gone = n_back = n_moved = 0
    for ei, ep in enumerate(gs.episodes):
        orc = keyed_oracle(gs, ep)
        rec = ep.states[0].copy()
        # reset path: a fresh world starts with everybody present and the grace periods running (parsing.py:142)
        rec2 = ep.states[0].copy()
        rec2[soa.W_STATUS] = 0
        assert orc.lib.czo_reset_env(C.byref(orc.ctx), C.c_int64(0), C.c_uint32(0), rec2.ctypes.data_as(C.c_void_p), None) == 0
        assert np.array_equal(rec2, ep.states[0]), f"{name} ep{ei}: reset record"
        for t in range(len(ep.actions)):
            before = rec.copy()
            err, obs, rew, term, trunc = orc.step_env(rec, ep.actions[t])
            ctx = f"{name} ep{ei} (seed {ep.seed}, {ep.policy}, env {ep.env_id}, episode {ep.episode_no}) step {t}"
            assert err == 0, ctx
            if not np.array_equal(rec, ep.states[t + 1]):
                pytest.fail(f"{ctx}: state differs (status {rec[soa.W_STATUS]:#x} vs {ep.states[t + 1][soa.W_STATUS]:#x})\n-- oracle\n"
                            f"{soa.describe_record(ep.dims, rec)}\n-- reference\n{soa.describe_record(ep.dims, ep.states[t + 1])}")
            assert np.array_equal(bits(obs), bits(ep.obs[t + 1])), f"{ctx}: observation"
            assert np.array_equal(bits(rew), bits(ep.rewards[t])), f"{ctx}: reward {rew} vs {ep.rewards[t]}"
            assert np.array_equal(term, ep.terms[t]) and np.array_equal(trunc, ep.truncs[t]), f"{ctx}: flags"
            a0, _ = spawn.decode_status(before[soa.W_STATUS:soa.W_STATUS + 1], ep.dims.A, 5)     # (fixtures: grace periods <= 31)
            a1, _ = spawn.decode_status(rec[soa.W_STATUS:soa.W_STATUS + 1], ep.dims.A, 5)
            n_gone += int((a0 & ~a1).sum())
            n_back += int((~a0 & a1).sum())
            n_moved += sum(int(before[soa.AGENT_WORD0 + a] & 0xFFFF != rec[soa.AGENT_WORD0 + a] & 0xFFFF) for a in range(ep.dims.A) if not a0[0, a] and a1[0, a])
    assert n_gone >= 4 and n_back >= 4 and n_moved >= 3, (n_gone, n_back, n_moved)       # every set exercises both directions


def test_generator_action_stream_is_the_devices():
    """the "stream" episodes take their actions from tools/gen_golden.py's restatement of the counter-based action stream: it
    must be the oracle's (which test_host_logic pins to the library's)"""
    lib = load_lib()
    for name in spawn_keyed_sets():
        gs = GoldenSet(name)
        n_act = 5 if gs.scheme == 3 else 8
        for ep in gs.episodes:
            if ep.policy != "stream":
                continue
            want = np.array([[lib.czo_action(gs.cfg["spawn_seed"], ep.env_id, a, t, n_act) for a in range(ep.dims.A)] for t in range(len(ep.actions))])
            assert np.array_equal(want, ep.actions), name


def test_keyed_stream_mirrors_agree():
    """spawn.uniform (numpy), the oracle's czo_spawn_uniform: the same function"""
    lib = load_lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed, env, ep, t = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 2 ** 20))
        a, d = int(rng.integers(0, 4)), int(rng.integers(0, 2004))
        assert lib.czo_spawn_uniform(seed, env, (ep << 32) | t, a, d) == float(spawn.uniform(seed, env, (ep << 32) | t, a, d))
