"""Batched agent despawn / respawn bookkeeping (cooking_zoo_amd/spawn.py): the vectorised pass over all worlds against a
scalar transliteration of the reference's per-world rule (cooking_world.py:267-290) fed with the same keyed draws."""
import numpy as np
import pytest

from cooking_zoo_amd import soa
from cooking_zoo_amd.spawn import SpawnBook, uniform


def scalar_rule(book_like, rec, dims, e, step):
    """handle_agent_spawn for ONE world, written the way the reference writes it (draw only when the `and` chain gets
    there); state = (active[A], changed[A], grace[A]) lists; returns whether somebody was moved"""
    active, grace, changed = book_like["active"], book_like["grace"], book_like["changed"]
    A = len(active)
    moved = False
    for i in range(A):
        changed[i] = False
    for i in range(A):
        if grace[i] > 0:
            grace[i] -= 1
            continue
        if active.count(True) > 1 and active[i] and uniform(book_like["seed"], e, step, i, 0) < book_like["despawn"]:
            if soa.unpack_agent(rec[soa.AGENT_WORD0 + i])[3] >= 0:
                continue
            active[i] = False
            changed[i] = True
        elif not active[i] and uniform(book_like["seed"], e, step, i, 1) < book_like["respawn"]:
            active[i] = True
            changed[i] = True
            grace[i] = book_like["grace_period"]
            moved = True
    return moved


def test_vectorised_bookkeeping_equals_the_scalar_rule():
    N, A, T = 300, 4, 120
    dims = soa.Dims(6, 5, 8, A, 10)
    rng = np.random.default_rng(0)
    recs = np.zeros((N, dims.RW), dtype=np.uint32)
    cells = np.zeros(30, dtype=np.uint8)                                  # all Floor: every spawn cell is valid
    for e in range(N):
        soa.record_cells(dims, recs[e])[:] = cells
        for a in range(A):
            recs[e, soa.AGENT_WORD0 + a] = soa.pack_agent(a, 0, 1, -1)
    spawn_cells = [[([1, 2, 3, 4], [1, 2, 3])] * A]
    book = SpawnBook(N, A, spawn_cells, despawn_rate=0.3, respawn_rate=0.4, grace_period=2, seed=7, env_id_base=1000)
    twins = [dict(active=[True] * A, changed=[False] * A, grace=[2] * A, seed=7, despawn=0.3, respawn=0.4, grace_period=2) for _ in range(N)]
    saw_despawn = saw_respawn = saw_holding_stay = 0
    for t in range(1, T + 1):
        for e in range(N):                                                 # random "holding" flags: such agents must stay
            for a in range(A):
                x, y, o, _ = soa.unpack_agent(recs[e, soa.AGENT_WORD0 + a])
                recs[e, soa.AGENT_WORD0 + a] = soa.pack_agent(x, y, o, 0 if rng.random() < 0.2 else -1)
        recs[:, soa.W_T] = t                                                # (the draws are keyed by each world's episode and step counter)
        recs[:, soa.W_EPISODE] = np.arange(N) % 3
        before = recs.copy()
        moved = set(book.after_step(recs, dims).tolist())
        for e in range(N):
            m = scalar_rule(twins[e], before[e], dims, 1000 + e, ((e % 3) << 32) | t)
            assert book.active[e].tolist() == twins[e]["active"], (t, e)
            assert book.changed[e].tolist() == twins[e]["changed"], (t, e)
            assert book.grace[e].tolist() == twins[e]["grace"], (t, e)
            assert (e in moved) == m, (t, e)
            assert book.active[e].any()                                     # the last agent never leaves
            for a in range(A):
                if book.changed[e, a] and not book.active[e, a]:
                    saw_despawn += 1
                    assert soa.unpack_agent(before[e, soa.AGENT_WORD0 + a])[3] < 0
                if book.changed[e, a] and book.active[e, a]:
                    saw_respawn += 1
                    x, y, _, _ = soa.unpack_agent(recs[e, soa.AGENT_WORD0 + a])
                    assert x in spawn_cells[0][a][0] and y in spawn_cells[0][a][1]
                    others = [soa.unpack_agent(recs[e, soa.AGENT_WORD0 + b])[:2] for b in range(A) if b != a]
                    assert (x, y) not in others
    assert saw_despawn > 500 and saw_respawn > 500


def test_keyed_draws_do_not_depend_on_the_batch():
    whole = uniform(3, np.arange(0, 64), 9, 2, 1)
    parts = np.concatenate([uniform(3, np.arange(0, 20), 9, 2, 1), uniform(3, np.arange(20, 64), 9, 2, 1)])
    assert np.array_equal(whole, parts) and whole.min() >= 0.0 and whole.max() < 1.0
    assert not np.array_equal(whole, uniform(3, np.arange(0, 64), 10, 2, 1))


@pytest.mark.gpu
def test_batched_env_with_spawning_is_shard_invariant_and_consistent():
    """CookingVecEnv with despawn / respawn rates: one batch of 48 worlds == six batches of 8 with the matching global ids
    (keyed draws), despawned agents stand still and are reported truncated once, nobody leaves with something in hand."""
    from cooking_zoo_amd.vec_env import CookingVecEnv
    kw = dict(action_scheme="scheme3", num_layouts=6, auto_reset=True, agent_despawn_rate=0.15, agent_respawn_rate=0.25,
              grace_period=2, spawn_seed=5)
    mk = lambda n, base: CookingVecEnv(n, "crowded_6x5", "crowded_6x5", 4, 40,
                                       ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"], env_id_base=base, **kw)
    whole, parts = mk(48, 0), [mk(8, 8 * k) for k in range(6)]
    ow = whole.reset()
    op = np.concatenate([p.reset() for p in parts])
    assert np.array_equal(ow.view(np.uint64), op.view(np.uint64))
    rng = np.random.default_rng(1)
    prev_active = whole.spawn.active.copy()
    prev_xy = whole.get_state()[:, soa.AGENT_WORD0:soa.AGENT_WORD0 + 4] & 0xFFFF
    n_desp = 0
    for t in range(90):
        acts = rng.integers(0, 5, size=(48, 4), dtype=np.int32)
        rw = whole.step(acts)
        rp = [p.step(acts[8 * k:8 * k + 8]) for k, p in enumerate(parts)]
        for j in range(4):
            assert np.array_equal(rw[j].view(np.uint8), np.concatenate([r[j] for r in rp]).view(np.uint8)), (t, j)
        assert np.array_equal(whole.spawn.active, np.concatenate([p.spawn.active for p in parts]))
        st = whole.get_state()
        xy = st[:, soa.AGENT_WORD0:soa.AGENT_WORD0 + 4] & 0xFFFF
        stayed_out = ~prev_active & ~whole.spawn.active                       # inactive before and after: must not have moved
        fresh = st[:, soa.W_T] == 0
        assert np.array_equal(xy[stayed_out & ~fresh[:, None]], prev_xy[stayed_out & ~fresh[:, None]])
        gone = whole.spawn.changed & ~whole.spawn.active
        n_desp += int(gone.sum())
        assert np.all(rw[3][gone] == 1)                                       # reported truncated in the step they leave
        assert np.all((st[:, soa.AGENT_WORD0:soa.AGENT_WORD0 + 4][gone] >> 24) == 0)
        assert whole.spawn.active.any(axis=1).all()
        prev_active, prev_xy = whole.spawn.active.copy(), xy
    assert n_desp > 50
    whole.close()
    [p.close() for p in parts]


def test_grace_field_width_follows_the_period_and_the_agent_count():
    """status-word layout of the countdowns (csrc/cz_device.h spawn_grace_bits): 5 bits each while the period is at most 31 - what every
    fixture holds -, else 20 // A bits each; encode / decode round-trip at both widths"""
    from cooking_zoo_amd.spawn import decode_status, grace_bits, max_grace, status_bits
    assert [grace_bits(g, a) for g, a in ((0, 4), (31, 2), (32, 2), (200, 2), (40, 3), (5000, 1))] == [5, 5, 10, 10, 6, 20]
    assert [max_grace(a) for a in (1, 2, 3, 4)] == [(1 << 20) - 1, 1023, 63, 31]
    rng = np.random.default_rng(0)
    for A, bits in ((2, 5), (2, 10), (3, 6), (4, 5), (1, 20)):
        active = rng.random((50, A)) < 0.5
        grace = rng.integers(0, 1 << bits, size=(50, A))
        a2, g2 = decode_status(status_bits(active, grace, bits), A, bits)
        assert np.array_equal(a2, active) and np.array_equal(g2, grace)
