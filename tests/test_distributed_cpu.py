"""The N > 1 host path on CPU: two processes over gloo.  Each rank steps its shard of the envs with the ORACLE (no GPU
here), using the same global-id keyed layout draws and action stream the device uses; the per-rank statistics are
all-gathered with torch.distributed and reduced in rank order; the result must equal a single-process run over all envs."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _oracle_shard_stats(begin, count, T, seed):
    """rollout of envs [begin, begin+count) with the oracle; returns the statistics dict and the final records"""
    sys.path.insert(0, HERE)
    sys.path.insert(0, REPO)
    import random
    from cooking_zoo_amd import soa
    from cooking_zoo_amd.cooking_world.engine import load_level as ll
    from cooking_zoo_amd.cooking_world.layout import feature_length
    from golden_io import RECIPE_NAMES, recipe_table
    from oracle_binding import VecOracle
    meta = ll.load_meta_file("example")
    lv = ll.load_level_file("coop_test")
    rng = random.Random(0)
    lays = [ll.instantiate(lv, meta, 2, rng) for _ in range(8)]
    dims = soa.Dims(7, 7, ll.level_max_dyn(lv), 2, feature_length(meta))
    rid = np.full((count, 4), 0xFF, np.uint8)
    rid[:, 0], rid[:, 1] = RECIPE_NAMES.index("TomatoLettuceSalad"), RECIPE_NAMES.index("CarrotBanana")
    vo = VecOracle(layouts=lays, meta=meta, recipe_table=recipe_table(), recipe_ids=rid, dims=dims, scheme=3, max_steps=20,
                   end_condition_all=False, num_recipes=2, reward_scheme=None, pool_slices=[(0, 8)],
                   env_level=np.zeros(count, int), num_envs=count, env_id_base=begin)
    vo.reset()
    st = dict(env_steps=0, episodes=0, length_sum=0, truncations=0, terminations=0, recipes_completed=[0] * 4,
              return_sum=[0.0] * 4)
    lib = vo.oracle.lib
    cur = np.zeros((count, 2))
    for t in range(T):
        acts = np.array([[lib.czo_action(seed, begin + e, a, t, 5) for a in range(2)] for e in range(count)], dtype=np.int32)
        was_done = (vo.records[:, soa.W_STATUS] & 1).astype(bool)
        _, rew, term, trunc = vo.step(acts, want_obs=False)
        for e in range(count):
            if was_done[e]:
                continue
            st["env_steps"] += 1
            cur[e] += rew[e]
            if vo.records[e, soa.W_STATUS] & 1:
                st["episodes"] += 1
                st["length_sum"] += int(vo.records[e, soa.W_T])
                st["truncations"] += int(trunc[e, 0])
                st["terminations"] += int(term[e, 0])
                for a in range(2):
                    st["return_sum"][a] += float(cur[e, a])
                    st["recipes_completed"][a] += (int(vo.records[e, soa.W_MARKS]) >> (8 * a)) & 1
                cur[e] = 0
    return st, vo.records.copy()


def _worker(rank, world, port, total, T, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from cooking_zoo_amd import distributed as czd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    begin, count = czd.shard_range(total, world, rank)
    st, recs = _oracle_shard_stats(begin, count, T, seed)
    gathered = czd.gather_stats_torch(st)
    dist.barrier()
    q.put((rank, begin, count, czd.reduce_stats(gathered), gathered, recs))
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process():
    import multiprocessing as mp          # (stdlib: the parent never imports torch; the spawned ranks do)
    from cooking_zoo_amd import distributed as czd
    total, T, seed, world = 22, 45, 77, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole_stats, whole_recs = _oracle_shard_stats(0, total, T, seed)
    assert results[0][3] == results[1][3], "every rank reduces to the same totals"
    reduced = results[0][3]
    for k in czd.STAT_KEYS:
        assert reduced[k] == whole_stats[k], k
    assert reduced["recipes_completed"] == whole_stats["recipes_completed"]
    assert np.allclose(reduced["return_sum"], whole_stats["return_sum"], rtol=0, atol=1e-9)
    # shard invariance of the state itself: concatenated shard records == single-process records
    cat = np.concatenate([r[5] for r in results])
    assert np.array_equal(cat, whole_recs)
    assert [(r[1], r[2]) for r in results] == [(0, 11), (11, 11)]


def test_shard_range_partitions_everything():
    from cooking_zoo_amd.distributed import shard_range
    for total in (1, 7, 4096, 262144, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_stats_struct_roundtrip():
    from cooking_zoo_amd.distributed import stats_from_bytes, stats_to_bytes
    st = dict(env_steps=123456789012, episodes=77, length_sum=3080, truncations=70, terminations=7,
              recipes_completed=[3, 4, 0, 0], return_sum=[-12.5, 19.9875, 0.0, 0.0])
    assert stats_from_bytes(stats_to_bytes(st)) == st


def test_rendezvous_poison_wakes_waiting_ranks_and_a_fresh_subdir_starts_over(tmp_path):
    """FileRendezvous.poison: a rank that waits in a collective (or enters one later) raises RendezvousPoisoned with the
    reason; `subdir` gives every rank a fresh rendezvous whose call numbers start at 0 again (bench.py's second attempt)."""
    import threading
    from cooking_zoo_amd.distributed import FileRendezvous, RendezvousPoisoned
    base = [FileRendezvous(str(tmp_path / "r"), r, 2, timeout=20.0) for r in range(2)]
    first = [b.subdir("attempt0") for b in base]
    got = {}
    t0 = threading.Thread(target=lambda: got.__setitem__("ag0", first[0].all_gather(b"a")))
    t0.start()

    def rank0():
        try:
            first[0].barrier()                                     # rank 1 never comes: it poisons instead
        except RendezvousPoisoned as e:
            got["r0"] = str(e)
    # complete the all_gather above for rank 1, then let rank 0 wait in a barrier
    assert first[1].all_gather(b"b") == [b"a", b"b"]
    t0.join(10)
    assert got["ag0"] == [b"a", b"b"]
    t = threading.Thread(target=rank0)
    t.start()
    first[1].poison("hand-off abandoned on rank 1")
    t.join(10)
    assert got.get("r0") == "hand-off abandoned on rank 1"
    with pytest.raises(RendezvousPoisoned):
        first[1].barrier()                                         # entering it later fails as well
    second = [b.subdir("attempt1") for b in base]
    out = {}
    ts = [threading.Thread(target=lambda r=r: out.__setitem__(r, second[r].all_gather(bytes([r])))) for r in range(2)]
    for x in ts:
        x.start()
    for x in ts:
        x.join(10)
    assert out[0] == out[1] == [b"\x00", b"\x01"]
