"""`ShardedVecEnv`: one batch of CookingZoo envs over several GPUs - multi-GPU as an API, not a benchmark script.

The step path shards trivially (BASELINE north_star: "the batch shards trivially across the 8 GPUs of one node with RCCL
all-gather of episode stats over xGMI only"): the envs of the batch are cut into contiguous ranges of GLOBAL env ids, one
`CookingVecEnv` (one C-ABI handle, one HIP stream) per device, and everything that is drawn on the device - layout draws at
reset, the on-device action stream, the despawn / respawn draws - is keyed by the global id, so the results do not depend on
how many devices (or processes) the batch is spread over.  There is no data-path collective.  The only exchange is the
episode statistics: an RCCL all-gather of one `cz_stats` per shard through the C-ABI (`cz_comm_init` / `cz_stats_allgather`)
when every shard has a device of its own, a host-side reduction otherwise (several shards on one device, e.g. in tests).

Two ways to use it, one code path:

* one process, several devices:   `ShardedVecEnv(262144, ..., device_ids=range(8))`
  One host thread per handle (the C-ABI's contract: one host thread per handle): every call fans out to the shards'
  threads and returns when all of them have issued their part.
* one process per GPU (what `python bench.py --gpus N`, torchrun or any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE
  starts):   `ShardedVecEnv(262144, ..., device_ids=[local_rank], world_size=W, rank=r, rendezvous=rdzv)`
  `num_envs` is always the size of the WHOLE batch; this process then owns shards `rank * len(device_ids) ...`.

Reference semantics: every shard is the batched form of `cooking_env.py:243-288` (`accumulated_step`, `observe`) for its envs;
see `vec_env.CookingVecEnv`.  The kwargs keep the reference's names (`cooking_env.py:26-28`) and add `num_envs`, `device_ids`
("Config / flags").
"""
from __future__ import annotations

import ctypes as C
import json
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from cooking_zoo_amd import _native
from cooking_zoo_amd import distributed as czd
from cooking_zoo_amd.vec_env import BatchTables, CookingVecEnv


class ShardedBuffer:
    """One device buffer per local shard for an array whose env axis is cut like the batch: shard i holds
    `leading + (count_i,) + per_env` elements.  `to_host` / `from_host` join / cut along the env axis."""

    def __init__(self, owner: "ShardedVecEnv", leading: Tuple[int, ...], per_env: Tuple[int, ...], dtype):
        self.owner, self.leading, self.per_env, self.dtype = owner, tuple(leading), tuple(per_env), np.dtype(dtype)
        self.axis = len(self.leading)
        self.parts = owner._each(lambda i, env: env.alloc(self.leading + (env.num_envs,) + self.per_env, self.dtype))

    def to_host(self) -> np.ndarray:
        return np.concatenate(self.owner._each(lambda i, env: self.parts[i].to_host()), axis=self.axis)

    def from_host(self, arr) -> None:
        arr = np.asarray(arr, dtype=self.dtype)
        cuts = self.owner._local_cuts
        self.owner._each(lambda i, env: self.parts[i].from_host(np.ascontiguousarray(np.take(arr, range(cuts[i], cuts[i + 1]), axis=self.axis))))

    def free(self) -> None:
        self.owner._each(lambda i, env: self.parts[i].free())


def plan_shards(num_envs: int, world_size: int, devices_per_process: int) -> List[Tuple[int, int]]:
    """(first global env id, count) of every shard of the job, in shard order: shard g = rank * devices_per_process + i"""
    G = world_size * devices_per_process
    if num_envs < G:
        raise ValueError(f"{num_envs} envs cannot be cut into {G} shards")
    return [czd.shard_range(num_envs, G, g) for g in range(G)]


class ShardedVecEnv:
    def __init__(self, num_envs, level, meta_file, num_agents, max_steps, recipes, end_condition_all_dishes=False,
                 action_scheme="scheme1", reward_scheme=None, *, device_ids: Sequence[int] = (0,), world_size: int = 1, rank: int = 0,
                 rendezvous=None, comm: str = "auto", comm_deadline: float = 120.0, dry_run: bool = False, num_layouts=256,
                 layout_seed=0, layouts=None, auto_reset=True, max_dyn=None, pinned_outputs=False, agent_respawn_rate=0.0,
                 grace_period=20, agent_despawn_rate=0.0, spawn_seed=0, tables: Optional[BatchTables] = None):
        """`num_envs`: the whole batch (all processes, all devices).  `device_ids`: the HIP ordinals this process drives, one
        shard each (an ordinal may appear more than once: several shards on one device).  `world_size` / `rank` /
        `rendezvous` (a `distributed.FileRendezvous`): the multi-process form.  `comm`: "auto" = RCCL statistics exchange when
        every shard of the job sits on a device of its own, host reduction otherwise; "host" = never RCCL; "rccl" = required.
        `dry_run`: plan only - no device is touched (launcher tests on machines without GPUs).  Everything else as for
        `CookingVecEnv` (the reference's kwargs, cooking_env.py:26-28,62-64)."""
        self.device_ids = [int(d) for d in device_ids]
        if not self.device_ids:
            raise ValueError("device_ids must name at least one device")
        self.world_size, self.rank, self.rendezvous = int(world_size), int(rank), rendezvous
        if self.world_size > 1 and rendezvous is None:
            raise ValueError("world_size > 1 needs a rendezvous (cooking_zoo_amd.distributed.FileRendezvous)")
        self.num_envs = int(num_envs)
        k = len(self.device_ids)
        self.plan = plan_shards(self.num_envs, self.world_size, k)
        self.shard_ids = [self.rank * k + i for i in range(k)]
        self.ranges = [self.plan[g] for g in self.shard_ids]
        self.local_envs = sum(c for _, c in self.ranges)
        self.local_begin = self.ranges[0][0]
        self._local_cuts = np.concatenate([[0], np.cumsum([c for _, c in self.ranges])]).astype(int)
        self.dry_run = bool(dry_run)
        self.comm_kind, self.comm_note = "host", ""
        self._pools: List[ThreadPoolExecutor] = []
        self.shards: List[CookingVecEnv] = []
        self._closed = False
        if tables is None:
            tables = BatchTables(self.num_envs, level, meta_file, num_agents, max_steps, recipes, end_condition_all_dishes,
                                 action_scheme, reward_scheme, num_layouts=num_layouts, layout_seed=layout_seed, layouts=layouts,
                                 max_dyn=max_dyn, agent_respawn_rate=agent_respawn_rate, grace_period=grace_period,
                                 agent_despawn_rate=agent_despawn_rate, spawn_seed=spawn_seed)
        elif tables.num_envs != self.num_envs:
            raise ValueError("`tables` describes another batch size")
        self.tables = tables
        for name in ("num_agents", "max_steps", "F", "n_actions", "dims", "action_scheme", "scheme_class", "meta", "levels",
                     "recipe_nodes", "num_recipes", "reward_scheme", "end_condition_all_dishes"):
            setattr(self, name, getattr(tables, name))
        if self.dry_run:
            return
        # one host thread per handle; a single local shard is driven by the calling thread itself (a thread hop costs tens of
        # microseconds per call - more than a 20-step region's synchronisation)
        self._pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"cz-shard{g}") for g in self.shard_ids] if k > 1 else [None]
        try:
            self.shards = [None] * k
            made = self._each(lambda i, _: CookingVecEnv(self.ranges[i][1], tables=tables.shard(*self.ranges[i]), auto_reset=auto_reset,
                                                         device_id=self.device_ids[i], pinned_outputs=pinned_outputs), need_env=False)
            self.shards = list(made)
            self._bring_up_comm(comm, comm_deadline)
        except BaseException:
            self.close()
            raise

    # ------------------------------------------------------------------ plumbing
    def _each(self, fn, need_env=True):
        """fn(i, shard_i) on every local shard's own thread, all at once; the results in shard order.  The first exception
        (in shard order) is re-raised after all of them have finished."""
        if self.dry_run:
            raise RuntimeError("ShardedVecEnv(dry_run=True) holds a plan only: no device work")
        if len(self._pools) == 1 and self._pools[0] is None:
            return [fn(0, self.shards[0] if need_env else None)]
        futs = [pool.submit(fn, i, self.shards[i] if need_env else None) for i, pool in enumerate(self._pools)]
        out, err = [], None
        for f in futs:
            try:
                out.append(f.result())
            except BaseException as exc:                              # noqa: BLE001
                out.append(None)
                err = err or exc
        if err is not None:
            raise err
        return out

    def _cut(self, arr, i, axis=0):
        return np.ascontiguousarray(np.take(np.asarray(arr), range(self._local_cuts[i], self._local_cuts[i + 1]), axis=axis))

    @staticmethod
    def _part(buf, i):
        return None if buf is None else (buf.parts[i] if isinstance(buf, ShardedBuffer) else buf[i])

    # ------------------------------------------------------------------ statistics exchange
    def _bring_up_comm(self, comm, deadline):
        G = len(self.plan)
        if comm not in ("auto", "host", "rccl"):
            raise ValueError("comm must be 'auto', 'host' or 'rccl'")
        distinct_here = len(set(self.device_ids)) == len(self.device_ids)
        if self.world_size > 1:
            every = [json.loads(b) for b in self.rendezvous.all_gather(json.dumps(self.device_ids).encode())]
            # (ranks of one node: a device may be driven by one shard only for RCCL to accept the communicator)
            flat = [d for ids in every for d in ids]
            distinct = len(set(flat)) == len(flat)
        else:
            distinct = distinct_here
        if comm == "host" or (comm == "auto" and not distinct):
            self.comm_note = "host reduction" + ("" if distinct else ": several shards share a device (RCCL refuses duplicate devices)")
            if comm == "host":
                self.comm_note = "host reduction (requested)"
            return
        if comm == "rccl" and not distinct:
            raise ValueError("comm='rccl' needs every shard on a device of its own")
        L = _native.lib()
        uid = (C.c_uint8 * 128)()
        payload = None
        if self.rank == 0:
            payload = bytes(uid) if L.cz_comm_unique_id(uid) == 0 else b""
        if self.world_size > 1:
            payload = self.rendezvous.broadcast(payload, src=0)
        box: Dict[int, str] = {}
        if payload:
            uid = (C.c_uint8 * 128).from_buffer_copy(payload)

            def bring_up(i, env):
                rc = L.cz_comm_init(env._h, G, self.shard_ids[i], uid)
                box[i] = "" if rc == 0 else (L.cz_last_error(env._h) or b"cz_comm_init failed").decode()

            # ncclCommInitRank blocks until every rank has joined: all shards at once, each on its own thread, under a deadline
            pools = self._pools if self._pools[0] is not None else [ThreadPoolExecutor(max_workers=1, thread_name_prefix="cz-comm")]
            futs = [pool.submit(bring_up, i, self.shards[i]) for i, pool in enumerate(pools)]
            done = threading.Event()

            def wait_all():
                for f in futs:
                    try:
                        f.result()
                    except BaseException as exc:                      # noqa: BLE001
                        box.setdefault(-1, str(exc))
                done.set()
            threading.Thread(target=wait_all, daemon=True).start()
            mine = "" if done.wait(timeout=deadline) else "timed out"
            if pools is not self._pools:
                pools[0].shutdown(wait=False)                 # (the helper thread of a single local shard)
            mine = mine or "; ".join(f"shard {self.shard_ids[i]}: {m}" for i, m in sorted(box.items()) if m)
        else:
            mine = "rank 0 could not create an RCCL unique id (librccl missing?)"
        if self.world_size > 1:
            every = [b.decode() for b in self.rendezvous.all_gather(mine.encode())]
            bad = "; ".join(f"rank {r}: {m}" for r, m in enumerate(every) if m)
        else:
            bad = mine
        if bad:
            self.comm_stuck = "timed out" in bad          # (a bring-up that hangs cannot be torn down: the caller should leave with os._exit)
            if comm == "rccl":
                raise _native.NativeError("RCCL communicator: " + bad)
            self.comm_note = "host reduction: RCCL communicator not available (" + bad + ")"
            return
        self.comm_kind, self.comm_note = "rccl", f"RCCL all-gather of one cz_stats per shard over {G} shard(s)"

    def stats_per_shard(self) -> List[Dict]:
        """one statistics dict per shard OF THE WHOLE JOB, in shard order (on every process).  COLLECTIVE in the multi-process form: every
        process has to call it at the same point of its program (it is an all-gather - RCCL's, or the rendezvous' on the host path)."""
        G = len(self.plan)
        if self.comm_kind == "rccl":
            def gather(i, env):
                out = (_native.CzStats * G)()
                _native.check(env._h, _native.lib().cz_stats_allgather(env._h, out))
                return [out[g].as_dict() for g in range(G)]
            return self._each(gather)[0]
        local = self._each(lambda i, env: env.stats())
        if self.world_size == 1:
            return local
        every = [json.loads(b) for b in self.rendezvous.all_gather(json.dumps(local).encode())]
        return [st for part in every for st in part]

    def stats(self) -> Dict:
        """episode statistics of the whole batch: the shards' cz_stats summed in shard order (bitwise reproducible float64).  Collective
        like `stats_per_shard`."""
        return czd.reduce_stats(self.stats_per_shard())

    def reset_stats(self):
        self._each(lambda i, env: env.reset_stats())

    def barrier(self):
        """every local stream drained; with several processes also a barrier over all of them (and, with RCCL, on the GPUs)"""
        self.sync()
        if self.world_size > 1:
            self.rendezvous.barrier()
        if self.comm_kind == "rccl":
            self._each(lambda i, env: _native.check(env._h, _native.lib().cz_comm_barrier(env._h)))

    # ------------------------------------------------------------------ host-array API (arrays cover this process's envs)
    def reset(self, return_obs=True):
        parts = self._each(lambda i, env: env.reset(return_obs=return_obs))
        return np.concatenate(parts) if return_obs else None

    def step(self, actions, return_obs=True):
        parts = self._each(lambda i, env: env.step(self._cut(actions, i), return_obs=return_obs))
        return tuple(None if p[0] is None else np.concatenate(p) for p in zip(*parts))

    def step_compact(self, actions):
        parts = self._each(lambda i, env: env.step_compact(self._cut(actions, i)))
        return tuple(np.concatenate(p) for p in zip(*parts))

    def observe(self):
        return np.concatenate(self._each(lambda i, env: env.observe()))

    def get_state(self):
        return np.concatenate(self._each(lambda i, env: env.get_state()))

    def set_state(self, records):
        self._each(lambda i, env: env.set_state(self._cut(records, i)))

    def obs_table(self):
        return self.shards[0].obs_table()

    @property
    def codes_pitch(self):
        return self.shards[0].codes_pitch

    def spawn_exhausted(self):
        return sum(self._each(lambda i, env: env.spawn_exhausted()))

    def set_spawn_rates(self, agent_despawn_rate, agent_respawn_rate, grace_period, spawn_seed=None):
        """`CookingVecEnv.set_spawn_rates` on every shard (the draws stay keyed by the global env id: the batch behaves like one handle)"""
        self._each(lambda i, env: env.set_spawn_rates(agent_despawn_rate, agent_respawn_rate, grace_period, spawn_seed))

    # ------------------------------------------------------------------ device-resident API
    def alloc(self, per_env_shape, dtype, leading=()):
        """a buffer with `leading + (envs of the shard,) + per_env_shape` elements on every local shard's device"""
        return ShardedBuffer(self, tuple(leading), tuple(per_env_shape), dtype)

    def step_device(self, d_actions, d_obs, d_rewards, d_term, d_trunc):
        P = self._part
        self._each(lambda i, env: env.step_device(P(d_actions, i), P(d_obs, i), P(d_rewards, i), P(d_term, i), P(d_trunc, i)))

    def step_device_compact(self, d_actions, d_codes, d_rewards, d_term, d_trunc, d_obs=None):
        P = self._part
        self._each(lambda i, env: env.step_device_compact(P(d_actions, i), P(d_codes, i), P(d_rewards, i), P(d_term, i), P(d_trunc, i), P(d_obs, i)))

    def step_device_ring(self, K, d_ring, action_period, first_slot, d_obs, d_rewards, d_term, d_trunc):
        """K steps; step k reads slot (first_slot + k) % action_period of `d_ring` (ShardedBuffer with leading = (action_period,))"""
        P = self._part
        self._each(lambda i, env: env.step_device_ring(K, P(d_ring, i), env.num_envs * env.num_agents, action_period, first_slot,
                                                        P(d_obs, i), P(d_rewards, i), P(d_term, i), P(d_trunc, i)))

    def step_device_ring_sync(self, K, d_ring, action_period, first_slot, d_obs, d_rewards, d_term, d_trunc):
        """`step_device_ring` + `sync` in one fan-out (one thread hop per shard instead of two)"""
        P = self._part

        def run(i, env):
            env.step_device_ring(K, P(d_ring, i), env.num_envs * env.num_agents, action_period, first_slot, P(d_obs, i), P(d_rewards, i),
                                 P(d_term, i), P(d_trunc, i))
            env.sync()
        self._each(run)

    def ring_runner(self, K, d_ring, action_period, d_obs, d_rewards, d_term, d_trunc):
        """-> `run(first_slot)`: `step_device_ring_sync` with everything that does not change between calls resolved once (buffer
        addresses, strides, the bound C functions) - for loops that issue short runs back to back, where the Python work per call
        (a few microseconds) is a visible part of a 100 us region"""
        from cooking_zoo_amd.vec_env import _dev_ptr as p
        P, L = self._part, _native.lib()
        fixed = [(env, env._h, (int(K), p(P(d_ring, i)), env.num_envs * env.num_agents, int(action_period)),
                  (p(P(d_obs, i)), p(P(d_rewards, i)), p(P(d_term, i)), p(P(d_trunc, i)))) for i, env in enumerate(self.shards)]
        ring, sync, check = L.cz_step_device_ring, L.cz_sync, _native.check

        def one(i, env, first_slot):
            _, h, a, o = fixed[i]
            check(h, ring(h, *a, first_slot, *o))
            check(h, sync(h))
            env._advance(a[0])

        if len(fixed) == 1:
            env0 = fixed[0][0]
            return lambda first_slot: one(0, env0, int(first_slot))
        return lambda first_slot: self._each(lambda i, env: one(i, env, int(first_slot)))

    def ring_prepare(self, K, d_ring, action_period, first_slot, d_obs, d_rewards, d_term, d_trunc):
        """build the HIP graphs of such a run up front (nothing is stepped)"""
        P = self._part
        from cooking_zoo_amd.vec_env import _dev_ptr as p

        def prep(i, env):
            _native.check(env._h, _native.lib().cz_ring_prepare(env._h, int(K), p(P(d_ring, i)), env.num_envs * env.num_agents, int(action_period),
                                                                int(first_slot), p(P(d_obs, i)), p(P(d_rewards, i)), p(P(d_term, i)), p(P(d_trunc, i))))
        self._each(prep)

    def set_ring_fused(self, enabled=True):
        return self._each(lambda i, env: env.set_ring_fused(enabled))[0]

    def rollout(self, T, seed, step0=0, d_obs=None, d_rewards=None, d_term=None, d_trunc=None):
        P = self._part
        self._each(lambda i, env: env.rollout(T, seed, step0, P(d_obs, i), P(d_rewards, i), P(d_term, i), P(d_trunc, i)))

    def rollout_compact(self, T, seed, step0, d_codes, d_obs=None, d_rewards=None, d_term=None, d_trunc=None):
        P = self._part
        self._each(lambda i, env: env.rollout_compact(T, seed, step0, P(d_codes, i), P(d_obs, i), P(d_rewards, i), P(d_term, i), P(d_trunc, i)))

    def rollout_actions(self, d_actions, T, d_obs=None, d_rewards=None, d_term=None, d_trunc=None):
        P = self._part
        self._each(lambda i, env: env.rollout_actions(P(d_actions, i), T, P(d_obs, i), P(d_rewards, i), P(d_term, i), P(d_trunc, i)))

    def sync(self):
        self._each(lambda i, env: env.sync())

    # ------------------------------------------------------------------ layout pool (the same tables on every shard)
    def update_layouts(self, first, layouts):
        self._each(lambda i, env: env.update_layouts(first, layouts))

    def set_layout_group(self, groups, active):
        self._each(lambda i, env: env.set_layout_group(groups, active))

    def rotate_layouts(self, every, **kw):
        """`CookingVecEnv.rotate_layouts` on every shard with the same seed: every shard's producer draws the same layouts, and
        the switches fall on the same step numbers - the batch behaves like one handle that rotates"""
        self._each(lambda i, env: env.rotate_layouts(every, **kw))

    def stop_rotation(self):
        self._each(lambda i, env: env.stop_rotation())

    def advance(self, k):
        """`CookingVecEnv.advance` on every shard: k env steps ran as replays of a graph of the caller that holds captured launches"""
        self._each(lambda i, env: env.advance(k))

    @property
    def rotation_events(self):
        return self.shards[0].rotation_events

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if self._closed:
            return
        self._closed = True
        stuck = getattr(self, "comm_stuck", False)
        for i, pool in enumerate(self._pools):
            env = self.shards[i] if i < len(self.shards) else None
            if env is not None and not stuck:
                try:
                    env.close() if pool is None else pool.submit(env.close).result(timeout=60)
                except BaseException:                                 # noqa: BLE001
                    pass
            if pool is not None:
                pool.shutdown(wait=not stuck)
        self._pools, self.shards = [], []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:                                             # noqa: BLE001
            pass
