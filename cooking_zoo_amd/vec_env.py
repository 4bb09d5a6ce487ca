"""CookingVecEnv: N independent CookingZoo worlds advanced by one HIP grid launch per step.

This is the batched form of the reference's `CookingEnvironment` (environment/cooking_env.py:49-385):
same kwargs (level, meta_file, num_agents, max_steps, recipes, end_condition_all_dishes, action_scheme,
reward_scheme), plus the batch geometry.  All world state lives in HBM behind the C-ABI of
include/cookingzoo.h; this class only prepares tables (layout pool, recipe tables) and moves arrays.
"""
from __future__ import annotations

import ctypes as C
import queue as _queue
import random as _random

import numpy as np

from cooking_zoo_amd import _native, soa
from cooking_zoo_amd.cooking_book import recipe_drawer
from cooking_zoo_amd.cooking_world.actions import ACTION_SCHEMES
from cooking_zoo_amd.cooking_world.engine import load_level as _ll
from cooking_zoo_amd.cooking_world.layout import feature_length

DEFAULT_REWARD_SCHEME = {"recipe_reward": 20, "max_time_penalty": -5, "recipe_penalty": -40, "recipe_node_reward": 0}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _produce_layouts(q, stop, level_objects, meta, num_agents, dims_tuple, pool_slices, groups, seed, start=0):
    """Body of the layout-rotation process (CookingVecEnv.rotate_layouts): batch b refills part (start + b) % groups of every
    level's pool slice (`start` = the part the envs drew from when the rotation began: the first one to be retired); its
    layouts come from random.Random streams keyed by (seed, b, level), so a run can be replayed."""
    dims = soa.Dims(*dims_tuple)
    b = 0
    while not stop.is_set():
        g = (start + b) % groups
        batch = []
        for li, lv in enumerate(level_objects):
            base, count = pool_slices[li]
            sub = count // groups
            rng = _random.Random((int(seed) * 1000003 + b) * 101 + li)
            lays = [_ll.instantiate(lv, meta, num_agents, rng) for _ in range(sub)]
            first = base + g * sub
            recs = np.stack([l.init_record(dims, first + k) for k, l in enumerate(lays)])
            desc = np.stack([l.obs_descriptor(meta, dims) for l in lays])
            batch.append((first, lays, recs, desc))
        while not stop.is_set():
            try:
                q.put(batch, timeout=0.1)
                break
            except _queue.Full:
                pass
        b += 1


class DeviceBuffer:
    """A typed HBM allocation owned by a CookingVecEnv (no torch involved)."""

    def __init__(self, env, shape, dtype):
        self.env, self.shape, self.dtype = env, tuple(int(s) for s in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = _native.lib().cz_dev_alloc(env._h, self.nbytes)
        if not self.ptr:
            _native.check(env._h, 1)

    def to_host(self, out=None):
        out = np.empty(self.shape, self.dtype) if out is None else out
        _native.check(self.env._h, _native.lib().cz_memcpy_d2h(self.env._h, _ptr(out), self.ptr, self.nbytes))
        return out

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        _native.check(self.env._h, _native.lib().cz_memcpy_h2d(self.env._h, self.ptr, _ptr(arr), self.nbytes))

    def free(self):
        if self.ptr:
            _native.lib().cz_dev_free(self.env._h, self.ptr)
            self.ptr = None


def _dev_ptr(b):
    """device address of a DeviceBuffer / raw int / object with data_ptr() (e.g. a torch tensor); None stays None"""
    if b is None:
        return None
    if hasattr(b, "ptr"):
        return b.ptr
    if hasattr(b, "data_ptr"):
        if hasattr(b, "is_contiguous") and not b.is_contiguous():
            raise ValueError("device buffers must be contiguous")
        return C.c_void_p(b.data_ptr())
    return C.c_void_p(int(b))


def resolve_recipe_tables(recipes):
    """-> (registry dict, names list).  Same rule as cooking_env.py:100-105: a non-empty user
    RECIPE_STORE replaces the default book."""
    book = recipe_drawer.RECIPE_STORE if recipe_drawer.RECIPE_STORE else recipe_drawer.RECIPES
    return book, list(book.keys())


class SpawnView:
    """Read-only view of the device's despawn / respawn bookkeeping (record word W_STATUS, cz_set_spawn): `active` [N, A],
    `grace` [N, A] as of the last `refresh()` (every host-array `step` / `reset` refreshes), `changed` [N, A] = whose
    presence flipped between the last two refreshes of a running episode (cooking_world.py status_changed)."""

    def __init__(self, env):
        self._env = env
        self.active = np.ones((env.num_envs, env.num_agents), dtype=bool)
        self.grace = np.full((env.num_envs, env.num_agents), env._spawn_cfg[2], dtype=np.int64)
        self.changed = np.zeros((env.num_envs, env.num_agents), dtype=bool)
        self._episode = None

    def refresh(self, reset=False):
        from cooking_zoo_amd.spawn import decode_status, grace_bits
        recs = self._env.get_state()
        active, grace = decode_status(recs[:, soa.W_STATUS], self._env.num_agents, grace_bits(self._env._spawn_cfg[2], self._env.num_agents))
        episode = recs[:, soa.W_EPISODE].copy()
        same = (not reset) and self._episode is not None
        self.changed = (active != self.active) & (episode == self._episode)[:, None] if same else np.zeros_like(active)
        self.active, self.grace, self._episode = active, grace, episode
        return self

    def relevant(self):
        """[N, A] bool: the agents a step reports on (active, or despawned in this very step): cooking_world.py:292-293"""
        return self.active | self.changed


class BatchTables:
    """Everything of a batch that is prepared on the host and never touches the device: meta file, levels, recipe tables and
    per-env recipe ids, the layout pool (one contiguous slice per level), dims, per-env level, spawn areas.  A `CookingVecEnv`
    is one of these plus a device handle; `ShardedVecEnv` makes one for the whole batch and hands every device a `shard` of
    it (same pool and tables, a contiguous range of global env ids), so that nothing is drawn or parsed twice."""

    def __init__(self, num_envs, level, meta_file, num_agents, max_steps, recipes, end_condition_all_dishes=False,
                 action_scheme="scheme1", reward_scheme=None, *, num_layouts=256, layout_seed=0, layouts=None,
                 env_id_base=0, max_dyn=None, agent_respawn_rate=0.0, grace_period=20, agent_despawn_rate=0.0, spawn_seed=0):
        self._spawn_cfg = (float(agent_despawn_rate), float(agent_respawn_rate), int(grace_period), int(spawn_seed))
        if action_scheme not in ACTION_SCHEMES:
            raise ValueError("action_scheme must be 'scheme1' or 'scheme3' (scheme2 raises AttributeError in the "
                             "reference: action_scheme2.py:15)")
        self.num_envs, self.num_agents, self.max_steps = int(num_envs), int(num_agents), int(max_steps)
        self.env_id_base = int(env_id_base)
        self.action_scheme = action_scheme
        self.scheme_class = ACTION_SCHEMES[action_scheme]
        self.n_actions = len(self.scheme_class.ACTIONS)
        self.meta = _ll.load_meta_file(meta_file)
        assert self.num_agents <= self.meta["Agent"], "Too many agents for this level"      # cooking_env.py:93
        self.levels = [level] if isinstance(level, str) else list(level)
        self.level_objects = [_ll.load_level_file(l) for l in self.levels]
        self.reward_scheme = dict(reward_scheme or DEFAULT_REWARD_SCHEME)
        self.end_condition_all_dishes = bool(end_condition_all_dishes)
        self.F = feature_length(self.meta)

        # ---- recipe tables
        self.book, self.book_names = resolve_recipe_tables(recipes)
        graphs = [self.book[n]() for n in self.book_names]
        # compact tables while every graph of the book has at most 8 nodes, wide ones (16) otherwise
        self.recipe_nodes = soa.NARROW_NODES if max(len(g.node_list) for g in graphs) <= soa.NARROW_NODES else soa.MAX_NODES
        self.recipe_table = np.stack([g.flatten(self.recipe_nodes) for g in graphs])
        if isinstance(recipes, np.ndarray):
            rid = np.asarray(recipes, dtype=np.int64)
            assert rid.shape[0] == self.num_envs
            self.recipe_names = None
        else:
            self.recipe_names = list(recipes)
            rid = np.tile(np.array([self.book_names.index(r) for r in recipes], dtype=np.int64), (self.num_envs, 1))
        self.num_recipes = rid.shape[1]
        if not (self.num_agents <= self.num_recipes <= soa.MAX_RECIPES_PER_ENV):
            raise ValueError("need one recipe per agent and at most 4 recipes (cooking_env.py:329)")
        self.recipe_ids = np.full((self.num_envs, 4), 0xFF, dtype=np.uint8)
        self.recipe_ids[:, :self.num_recipes] = rid

        # ---- layout pool (per level a contiguous slice)
        rng = _random.Random(layout_seed)
        self.layouts = []
        self.pool_slices = []
        for li, lv in enumerate(self.level_objects):
            base = len(self.layouts)
            if layouts is not None:
                mine = layouts[li] if isinstance(layouts[0], (list, tuple)) else layouts
            else:
                mine = [_ll.instantiate(lv, self.meta, self.num_agents, rng) for _ in range(int(num_layouts))]
            self.layouts += list(mine)
            self.pool_slices.append((base, len(mine)))
        W, H = self.layouts[0].width, self.layouts[0].height
        if any((l.width, l.height) != (W, H) for l in self.layouts):
            raise ValueError("all levels of one batch must share the grid size")
        D = max_dyn if max_dyn is not None else max(max(_ll.level_max_dyn(lv) for lv in self.level_objects),
                                                    max(l.slots_used for l in self.layouts))
        self.dims = soa.Dims(W, H, max(1, D), self.num_agents, self.F)
        # env e of the whole (unsharded) batch plays level e % len(levels): keyed by the GLOBAL id, so a shard sees its own part
        self.env_level = (self.env_id_base + np.arange(self.num_envs)) % len(self.levels)
        self.spawn_cells = self.level_of_layout = None
        if self._spawn_cfg[0] or self._spawn_cfg[1]:
            # spawn areas per level (parsing.py:118-151: the AGENTS entries of the level file, one per agent up to MAX_COUNT each)
            self.spawn_cells = []                                        # [level][agent] -> (x candidates, y candidates)
            for lv in self.level_objects:
                cells = [(spec["X_POSITION"], spec["Y_POSITION"]) for spec in lv["AGENTS"] for _ in range(spec["MAX_COUNT"])][:self.num_agents]   # parsing.py:145
                self.spawn_cells.append([(list(xs), list(ys)) for xs, ys in cells])
            self.level_of_layout = np.zeros(len(self.layouts), dtype=np.uint8)
            for li, (base, count) in enumerate(self.pool_slices):
                self.level_of_layout[base:base + count] = li

    def shard(self, begin, count):
        """The tables of envs [begin, begin + count) of this batch: the same pool, recipes and spawn areas (shared, not
        copied), global ids env_id_base + begin ..."""
        import copy
        if not (0 <= begin and begin + count <= self.num_envs and count > 0):
            raise ValueError(f"shard [{begin}, {begin + count}) outside the batch of {self.num_envs} envs")
        t = copy.copy(self)
        t.num_envs, t.env_id_base = int(count), self.env_id_base + int(begin)
        t.recipe_ids = np.ascontiguousarray(self.recipe_ids[begin:begin + count])
        t.env_level = self.env_level[begin:begin + count]
        t.layouts = list(self.layouts)            # (every handle keeps its own record of what its pool holds after updates)
        return t


class CookingVecEnv:
    def __init__(self, num_envs, level=None, meta_file=None, num_agents=None, max_steps=None, recipes=None,
                 end_condition_all_dishes=False, action_scheme="scheme1", reward_scheme=None, *, num_layouts=256,
                 layout_seed=0, layouts=None, auto_reset=True, device_id=0, env_id_base=0, max_dyn=None, pinned_outputs=False,
                 agent_respawn_rate=0.0, grace_period=20, agent_despawn_rate=0.0, spawn_seed=0, tables=None):
        """`level` may be one level name/path or a list (env e uses levels[e % len], e the global id); `recipes` is a list of
        names (every env the same) or an int array [num_envs, R] of indices into the recipe book.
        `layouts` (optional) supplies pre-instantiated Layout objects per level instead of drawing
        `num_layouts` of them with random.Random(layout_seed).
        `pinned_outputs=True`: `step` / `reset` / `observe` return views of page-locked host buffers owned by the env
        (overwritten by the next call) instead of fresh arrays -- the copy engines then write them directly, which is what
        a host-array step of thousands of envs spends its time on (18 MB of observations per step at 4096 envs).
        `agent_despawn_rate` / `agent_respawn_rate` / `grace_period`: agent despawn / respawn (cooking_world.py:267-290)
        for every world of the batch, evaluated by the step kernels themselves with keyed random streams (cz_set_spawn) -
        on every stepping path (`step`, `step_device`, `step_device_ring`, `rollout`).  `self.spawn` is a read-only view of
        that bookkeeping (`active`, `grace`, and after a host-array `step` also `changed`).
        `tables`: a prepared `BatchTables` (then the configuration arguments before `auto_reset` are ignored)."""
        self._pinned = bool(pinned_outputs)
        self._pin_bufs = {}
        self.spawn = None
        if tables is None:
            tables = BatchTables(num_envs, level, meta_file, num_agents, max_steps, recipes, end_condition_all_dishes,
                                 action_scheme, reward_scheme, num_layouts=num_layouts, layout_seed=layout_seed,
                                 layouts=layouts, env_id_base=env_id_base, max_dyn=max_dyn,
                                 agent_respawn_rate=agent_respawn_rate, grace_period=grace_period,
                                 agent_despawn_rate=agent_despawn_rate, spawn_seed=spawn_seed)
        self.tables = tables
        for k, v in vars(tables).items():           # the env reads (and, for the layout pool, edits) its own copies of the names
            setattr(self, k, v)
        W, H = self.dims.W, self.dims.H

        # ---- device handle
        L = _native.lib()
        rs = self.reward_scheme
        self._cfg = _native.CzConfig(self.num_envs, self.num_agents, W, H, self.dims.D, self.F, self.scheme_class.CODE,
                                     self.max_steps, int(self.end_condition_all_dishes), self.num_recipes,
                                     int(bool(auto_reset)), int(device_id), int(self.env_id_base),
                                     float(rs["recipe_reward"]), float(rs["max_time_penalty"]),
                                     float(rs["recipe_penalty"]), float(rs["recipe_node_reward"]))
        h = C.c_void_p()
        rc = L.cz_create(C.byref(self._cfg), C.byref(h))
        if rc != 0:
            raise _native.NativeError((L.cz_last_error(None) or b"cz_create failed").decode())
        self._h = h
        self.device_id = int(device_id)
        assert L.cz_record_words(self._h) == self.dims.RW
        _native.check(self._h, L.cz_load_recipes(self._h, _ptr(self.recipe_table), len(self.book_names), self.recipe_nodes))
        self._upload_layouts()
        self._buffers = []
        self._steps = 0                       # env steps issued so far (every stepping method counts; `advance` adds replayed ones)
        self.captured_steps = 0               # steps captured into graphs of the caller (they run at replay, see `advance`)
        self._caller_stream, self._warned_capture = False, False
        self._lay_groups, self._lay_active = 1, 0
        self._rot = None                      # rotate_layouts state
        self.rotation_events = []             # (step index, "group", groups, active) / (step index, "layouts", first slot, [Layout])
        if self.spawn_cells is not None:
            self._set_spawn()
            self.spawn = SpawnView(self)

    def _set_spawn(self):
        L, nl, A = _native.lib(), len(self.spawn_cells), self.num_agents
        stride = max(max(len(xs), len(ys)) for lv in self.spawn_cells for xs, ys in lv)
        if not 1 <= stride <= 1024:
            raise ValueError("a spawn area lists 1..1024 x and y candidates")
        sx, sy = np.zeros((nl, A, stride), dtype=np.uint8), np.zeros((nl, A, stride), dtype=np.uint8)
        nx, ny = np.zeros((nl, A), dtype=np.int32), np.zeros((nl, A), dtype=np.int32)
        for li, lv in enumerate(self.spawn_cells):
            if len(lv) < A:
                raise ValueError(f"level {self.levels[li]} has spawn areas for {len(lv)} agent(s) only")
            for a, (xs, ys) in enumerate(lv):
                if not xs or not ys:
                    raise ValueError("a spawn area lists 1..1024 x and y candidates")
                nx[li, a], ny[li, a] = len(xs), len(ys)
                # (a candidate beyond the grid can never be taken: 255 stands for any of them)
                sx[li, a, :len(xs)] = [v if 0 <= v < 255 else 255 for v in xs]
                sy[li, a, :len(ys)] = [v if 0 <= v < 255 else 255 for v in ys]
        d, r, g, seed = self._spawn_cfg
        _native.check(self._h, L.cz_set_spawn(self._h, d, r, g, seed, nl, _ptr(self.level_of_layout), stride, _ptr(sx), _ptr(nx), _ptr(sy), _ptr(ny)))

    def set_spawn_rates(self, agent_despawn_rate, agent_respawn_rate, grace_period, spawn_seed=None):
        """Change the despawn / respawn parameters of a live batch (cz_set_spawn: takes effect with the next step; running episodes keep
        their countdowns, re-packed by the library if the period crosses 31 and the fields change their width)."""
        if self.spawn_cells is None:
            raise ValueError("the env was made without despawn / respawn (no spawn areas were collected)")
        seed = self._spawn_cfg[3] if spawn_seed is None else int(spawn_seed)
        self._spawn_cfg = (float(agent_despawn_rate), float(agent_respawn_rate), int(grace_period), seed)
        self._set_spawn()

    def spawn_exhausted(self):
        """respawns that found no free cell of their spawn area in 1001 tries (the reference raises there) since the env was made"""
        return int(_native.lib().cz_spawn_exhausted(self._h))

    # ------------------------------------------------------------------ tables
    def _upload_layouts(self):
        recs = np.stack([l.init_record(self.dims, i) for i, l in enumerate(self.layouts)])
        desc = np.stack([l.obs_descriptor(self.meta, self.dims) for l in self.layouts])
        self._lay_records, self._lay_desc = recs, desc
        _native.check(self._h, _native.lib().cz_load_layouts(self._h, _ptr(recs), _ptr(desc), len(self.layouts)))

    def set_layouts(self, layouts):
        """Replace the whole pool (single-level batches; used by the single-env facade at reset)."""
        if self.spawn is not None and len(self.levels) > 1:
            raise ValueError("set_layouts replaces the pool of a single-level batch; a multi-level batch with despawn / respawn "
                             "would lose its layout -> level map (use update_layouts, which keeps every slot's level)")
        self.layouts = list(layouts)
        self.pool_slices = [(0, len(self.layouts))]
        self._upload_layouts()
        if self.spawn is not None:                                       # (the spawn tables map every layout of the pool to its level)
            self.level_of_layout = np.zeros(len(self.layouts), dtype=np.uint8)
            self._set_spawn()

    # ------------------------------------------------------------------ fresh layouts under a stepping batch
    def update_layouts(self, first, layouts):
        """Replace pool slots [first, first + len(layouts)) while the batch keeps stepping (cz_update_layouts: a copy stream
        of the library's own; the handle's stream is not touched).  Only slots that no env can draw or is playing on."""
        recs = np.stack([l.init_record(self.dims, first + k) for k, l in enumerate(layouts)])
        desc = np.stack([l.obs_descriptor(self.meta, self.dims) for l in layouts])
        self._update_layout_arrays(first, layouts, recs, desc)

    def _update_layout_arrays(self, first, layouts, recs, desc):
        _native.check(self._h, _native.lib().cz_update_layouts(self._h, int(first), len(layouts), _ptr(recs), _ptr(desc)))
        self.layouts[first:first + len(layouts)] = list(layouts)
        self._lay_records[first:first + len(layouts)] = recs
        self._lay_desc[first:first + len(layouts)] = desc
        self.rotation_events.append((self._steps, "layouts", int(first), list(layouts)))

    def set_layout_group(self, groups, active):
        """From the next step on the envs draw their next episodes from part `active` of their pool slices cut into `groups`
        equal parts (cz_set_layout_group; waits on the device for the updates issued so far)."""
        if any(count % groups for _, count in self.pool_slices):
            raise ValueError("every level's pool slice must be a multiple of `groups` layouts long")
        _native.check(self._h, _native.lib().cz_set_layout_group(self._h, int(groups), int(active)))
        self._lay_groups, self._lay_active = int(groups), int(active)
        self.rotation_events.append((self._steps, "group", int(groups), int(active)))

    def rotate_layouts(self, every, *, groups=2, seed=0, prefetch=2, blocking=True):
        """Keep the layout pool fresh while the batch steps - the batched counterpart of the reference instantiating a new
        level at every reset (cooking_env.py:191-195, parsing.py:21-151).  The pool slices are cut into `groups` parts; the
        envs draw from one; every `every` steps (at the next call boundary) the next part becomes the one drawn from, and
        once the episodes that started on the old part are over (max_steps + 2 steps later) it is refilled with layouts
        a background process has instantiated meanwhile (engine/load_level.py, its own seeded stream per refill).  Which
        layouts an env sees is a function of the sequence of stepping calls only (the refill waits for that process if it
        has to), so a run can be replayed: `rotation_events` lists what was switched / replaced at which step.
        `blocking=False` never waits: a refill whose layouts are not ready yet is tried again at the next call boundary (and
        the next switch with it), which keeps the stepping thread free of stalls at the price of a timing-dependent schedule.
        Steps that run as replays of a graph of the CALLER (captured `step_device*` launches) are invisible here: report them with
        `advance(k)` after every replay - the switches and refills then happen in that call, between replays."""
        if self._rot is not None:
            raise RuntimeError("rotate_layouts is already running")
        if groups < 2 or any(count % groups for _, count in self.pool_slices):
            raise ValueError("need groups >= 2 and pool slices that are multiples of `groups` long")
        if every < self.max_steps + 3:
            raise ValueError("`every` must be at least max_steps + 3 steps: a part is refilled max_steps + 2 steps after the "
                             "envs stopped drawing from it, and before they draw from it again")
        start = self._lay_active if self._lay_groups == groups else 0        # the part in use now: the first one to be refilled
        rot = {"groups": int(groups), "every": int(every), "flip_due": self._steps + int(every), "refill_due": None, "n_refills": 0,
               "blocking": bool(blocking), "start": int(start)}

        # The instantiation is plain Python (engine/load_level.py, the reference's draw order) and takes milliseconds per batch:
        # in a thread it would hold the interpreter lock against the thread that issues the steps, so it runs in a process of
        # its own (spawned: a fresh interpreter that never loads the HIP library) and hands finished arrays over a queue.
        import multiprocessing as _mp
        ctx = _mp.get_context("spawn")
        rot["queue"] = ctx.Queue(maxsize=max(1, int(prefetch)))
        rot["stop"] = ctx.Event()
        rot["process"] = ctx.Process(target=_produce_layouts, name="cz-layout-rotation", daemon=True,
                                     args=(rot["queue"], rot["stop"], self.level_objects, self.meta, self.num_agents, self.dims.as_tuple(),
                                           list(self.pool_slices), int(groups), int(seed), int(start)))
        # ... and a small thread takes the batches off the process queue (unpickling costs a millisecond) into a local one
        rot["ready"] = _queue.Queue(maxsize=max(1, int(prefetch)))        # (bounded: when it is full everything upstream sleeps)

        def drain():
            batch = None
            while not rot["stop"].is_set():
                try:
                    if batch is None:
                        batch = rot["queue"].get(timeout=0.1)
                    rot["ready"].put(batch, timeout=0.1)
                    batch = None
                except (_queue.Empty, _queue.Full):
                    pass
                except (EOFError, OSError):
                    return
        import threading as _threading
        rot["drain"] = _threading.Thread(target=drain, name="cz-layout-rotation-drain", daemon=True)
        self._rot = rot
        _native.check(self._h, _native.lib().cz_update_layouts(self._h, 0, 0, None, None))     # copy stream + staging, ahead of time
        self.set_layout_group(groups, start)
        rot["process"].start()
        rot["drain"].start()

    def stop_rotation(self):
        if self._rot is not None:
            rot, self._rot = self._rot, None
            rot["stop"].set()
            try:
                while True:                                   # (a producer blocked on a full queue sees the stop flag within 0.1 s)
                    rot["queue"].get_nowait()
            except _queue.Empty:
                pass
            rot["drain"].join(timeout=5)
            rot["process"].join(timeout=10)
            if rot["process"].is_alive():
                rot["process"].terminate()

    @staticmethod
    def _rotation_pending(rot):
        """a batch still on its way from the producer's queue to the local one"""
        try:
            return not rot["queue"].empty()
        except (OSError, ValueError):
            return False

    def rotation_ready(self):
        """refills the background process has ready right now (a measurement may want to start with a full queue)"""
        return 0 if self._rot is None else self._rot["ready"].qsize()

    def _issued(self, k):
        """a device-pointer call of k steps has returned: count them - unless the call was CAPTURED into a graph of the caller
        (nothing ran; the steps run whenever that graph is replayed, which Python does not see: the caller reports them with
        `advance`)"""
        if self._caller_stream and _native.lib().cz_stream_capturing(self._h):
            self.captured_steps += int(k)
            if self._rot is not None and not self._warned_capture:
                self._warned_capture = True
                import warnings
                warnings.warn("steps captured into a graph of the caller while rotate_layouts is active: call env.advance(k) after "
                              "every replay of k captured steps, or the layout pool is never switched / refilled", RuntimeWarning, stacklevel=3)
            return
        self._advance(k)

    def advance(self, k):
        """Tell the env that k env steps ran which it did not issue itself: replays of a graph of the caller that holds captured
        `step_device*` / `rollout*` launches (hipGraphLaunch, torch.cuda.CUDAGraph.replay()).  Keeps `rotate_layouts` on schedule:
        the switch to the next part of the pool and the refill of the retired one happen here, between replays."""
        if self._caller_stream and _native.lib().cz_stream_capturing(self._h):
            raise RuntimeError("advance() inside a stream capture: call it after the replay, outside the capture")
        self._advance(k)

    def _advance(self, k):
        """k more env steps have been issued: switch / refill when due"""
        self._steps += int(k)
        rot = self._rot
        if rot is None:
            return
        if rot["refill_due"] is not None and self._steps >= rot["refill_due"]:
            batch = None
            while batch is None:
                try:
                    # (blocking: waits for the producer if it is behind - in slices, so that a producer that died is noticed)
                    batch = rot["ready"].get(timeout=0.25) if rot["blocking"] else rot["ready"].get_nowait()
                except _queue.Empty:
                    dead = not rot["process"].is_alive() and rot["ready"].empty() and not self._rotation_pending(rot)
                    rot["dead_polls"] = rot.get("dead_polls", 0) + 1 if dead else 0   # (twice in a row: a batch may be in the drain thread's hands)
                    if rot["dead_polls"] >= 2:
                        code = rot["process"].exitcode
                        self.stop_rotation()
                        raise RuntimeError(f"rotate_layouts: the layout producer process died (exit code {code}); rotation stopped, "
                                           f"the envs keep drawing from part {self._lay_active} of {self._lay_groups}")
                    if not rot["blocking"]:
                        return                                              # not ready: next call
            for first, lays, recs, desc in batch:
                self._update_layout_arrays(first, lays, recs, desc)
            rot["refill_due"] = None
            rot["n_refills"] += 1
        if rot["refill_due"] is None and self._steps >= rot["flip_due"]:
            old = self._lay_active
            if old != (rot["start"] + rot["n_refills"]) % rot["groups"]:      # the part the producer's next batch is made for
                self.stop_rotation()
                raise RuntimeError("rotate_layouts: the active layout group was changed behind the rotation's back "
                                   "(set_layout_group while rotate_layouts runs); rotation stopped")
            self.set_layout_group(rot["groups"], (old + 1) % rot["groups"])
            rot["refill_due"] = self._steps + self.max_steps + 2
            rot["flip_due"] = self._steps + rot["every"]

    # ------------------------------------------------------------------ reset / step
    def initial_layout_ids(self):
        """Episode-0 layout of every env: the same keyed draw auto-reset uses (shard invariant)."""
        ids = np.empty(self.num_envs, dtype=np.int32)
        pools = np.empty(self.num_envs, dtype=np.uint32)
        L = _native.lib()
        for e in range(self.num_envs):
            base, count = self.pool_slices[self.env_level[e]]
            pools[e] = base | (count << 16)
            ids[e] = L.cz_next_layout_group(self.env_id_base + e, 0, int(pools[e]), len(self.layouts), self._lay_groups, self._lay_active)
        return ids, pools

    def reset(self, layout_ids=None, return_obs=True, env_begin=0, env_count=None, return_codes=False):
        """-> the float64 observation [n, A, F] (return_obs), the compact one uint8 [n, A, codes_pitch] (return_codes; then
        return_obs is ignored), or None"""
        if return_codes:
            self.reset(layout_ids, False, env_begin, env_count)
            return self.observe_compact(env_begin, env_count)
        n = self.num_envs if env_count is None else int(env_count)
        ids, pools = self.initial_layout_ids()
        if layout_ids is not None:
            ids = np.ascontiguousarray(layout_ids, dtype=np.int32)
        else:
            ids = ids[env_begin:env_begin + n].copy()
        pools = np.ascontiguousarray(pools[env_begin:env_begin + n])
        obs = self._host_array("obs", (n, self.num_agents, self.F), np.float64) if return_obs else None
        rid = np.ascontiguousarray(self.recipe_ids[env_begin:env_begin + n])
        _native.check(self._h, _native.lib().cz_reset(self._h, env_begin, n, _ptr(ids), _ptr(rid), _ptr(pools), _ptr(obs)))
        if self.spawn is not None:
            self.spawn.refresh(reset=True)
        return obs

    def _host_array(self, key, shape, dtype):
        """a fresh array, or (pinned_outputs) the env's page-locked buffer of that role"""
        if not self._pinned:
            return np.empty(shape, dtype=dtype)
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        buf = self._pin_bufs.get(key)
        if buf is None or buf[1] < nbytes:
            if buf is not None:
                _native.lib().cz_host_free(self._h, buf[0])
            p = _native.lib().cz_host_alloc(self._h, max(nbytes, 1))
            if not p:
                _native.check(self._h, 1)
            buf = self._pin_bufs[key] = (p, max(nbytes, 1))
        raw = (C.c_uint8 * nbytes).from_address(buf[0])
        return np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def step(self, actions, return_obs=True):
        """actions int [N, A] -> (obs f64 [N, A, F] | None, rewards f64 [N, A], terminations u8, truncations u8).
        An action of -1 means "this agent is despawned": it is left out of the step (cooking_world.py:105-108)."""
        N, A = self.num_envs, self.num_agents
        acts = self._host_array("act", (N, A), np.int32)
        acts[...] = np.asarray(actions).reshape(N, A)
        if acts.size and int(acts.max()) >= self.n_actions:
            raise ValueError(f"actions must be in [0, {self.n_actions}) for {self.action_scheme} (negative = despawned agent)")
        obs = self._host_array("obs", (N, A, self.F), np.float64) if return_obs else None
        rew = self._host_array("rew", (N, A), np.float64)
        term = self._host_array("term", (N, A), np.uint8)
        trunc = self._host_array("trunc", (N, A), np.uint8)
        _native.check(self._h, _native.lib().cz_step(self._h, _ptr(acts), _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc)))
        if self.spawn is not None:
            self.spawn.refresh()                                          # (the kernel did the bookkeeping: this only reads it back)
        self._advance(1)
        return obs, rew, term, trunc

    def step_compact(self, actions):
        """`step` with the observation as one byte per feature: -> (codes uint8 [N, A, codes_pitch], rewards, terminations,
        truncations); `obs_table()[codes[..., :F]]` is the float64 observation bit for bit.  An eighth of the bytes over PCIe."""
        N, A = self.num_envs, self.num_agents
        acts = self._host_array("act", (N, A), np.int32)
        acts[...] = np.asarray(actions).reshape(N, A)
        if acts.size and int(acts.max()) >= self.n_actions:
            raise ValueError(f"actions must be in [0, {self.n_actions}) for {self.action_scheme} (negative = despawned agent)")
        codes = self._host_array("codes", (N, A, self.codes_pitch), np.uint8)
        rew = self._host_array("rew", (N, A), np.float64)
        term = self._host_array("term", (N, A), np.uint8)
        trunc = self._host_array("trunc", (N, A), np.uint8)
        _native.check(self._h, _native.lib().cz_step_compact(self._h, _ptr(acts), _ptr(codes), _ptr(rew), _ptr(term), _ptr(trunc)))
        if self.spawn is not None:
            self.spawn.refresh()
        self._advance(1)
        return codes, rew, term, trunc

    def last_marks(self):
        """Recipe-node marks after the most recent host-array `step`: uint64 [N], bit 8r + j = node j of the env's r-th
        recipe (compact tables) or bit 16r + j (wide tables, `recipe_nodes == 16`)."""
        words = np.empty((self.num_envs, 2), dtype=np.uint32)
        _native.check(self._h, _native.lib().cz_last_marks(self._h, _ptr(words)))
        return words[:, 0].astype(np.uint64) | (words[:, 1].astype(np.uint64) << np.uint64(32))

    def marks_of(self, marks, r):
        """the node marks of recipe r inside a marks value (an int): bit j = node j of node_list"""
        bits = 8 if self.recipe_nodes == soa.NARROW_NODES else 16
        return (int(marks) >> (bits * r)) & ((1 << bits) - 1)

    def observe(self, env_begin=0, env_count=None):
        n = self.num_envs if env_count is None else int(env_count)
        obs = self._host_array("obs", (n, self.num_agents, self.F), np.float64)
        _native.check(self._h, _native.lib().cz_observe(self._h, env_begin, n, _ptr(obs)))
        return obs

    def observe_compact(self, env_begin=0, env_count=None):
        """`observe` as one byte per feature: uint8 [n, A, codes_pitch]; `obs_table()[codes[..., :F]]` is the float64 observation"""
        n = self.num_envs if env_count is None else int(env_count)
        codes = self._host_array("codes", (n, self.num_agents, self.codes_pitch), np.uint8)
        _native.check(self._h, _native.lib().cz_observe_compact(self._h, env_begin, n, _ptr(codes)))
        return codes

    def observe_device(self, d_obs=None, d_codes=None, env_begin=0, env_count=None):
        """the current observation into device buffers (float64 [n, A, F] and / or codes uint8 [n, A, codes_pitch]), stream-ordered:
        what a device-resident consumer reads before its first step"""
        n = self.num_envs if env_count is None else int(env_count)
        _native.check(self._h, _native.lib().cz_observe_device(self._h, env_begin, n, _dev_ptr(d_obs), _dev_ptr(d_codes)))

    # ------------------------------------------------------------------ device-resident API
    def alloc(self, shape, dtype):
        b = DeviceBuffer(self, shape, dtype)
        self._buffers.append(b)
        return b

    def set_ring_fused(self, enabled=True):
        """Runs of two or more steps of `step_device_ring` whose action slots are densely packed (`action_stride == num_envs *
        num_agents`) go out as fused launches - one per stretch of consecutive slots, the state in registers across the steps,
        every step's outputs written in place: the same final state, outputs and statistics as one launch per step, at the cost
        per step of a fused rollout (4.0-4.3 instead of 5.1-5.8 us at 4096 envs).  Open loop only.  Returns the previous setting."""
        rc = _native.lib().cz_set_ring_fused(self._h, 1 if enabled else 0)
        if rc < 0:
            raise _native.NativeError((_native.lib().cz_last_error(self._h) or b"cz_set_ring_fused failed").decode())
        return bool(rc)

    def ring_fused_steps(self, reset=False):
        return int(_native.lib().cz_ring_fused_steps(self._h, 1 if reset else 0))

    def set_stream(self, stream=None):
        """Order this env's device work on the caller's HIP stream: an int / ctypes pointer, or an object with a
        `cuda_stream` attribute such as torch.cuda.current_stream().  None = the env's own stream."""
        raw = getattr(stream, "cuda_stream", stream)
        raw = getattr(raw, "value", raw)                               # (ctypes.c_void_p)
        _native.check(self._h, _native.lib().cz_set_stream(self._h, C.c_void_p(int(raw)) if raw else None))
        self._caller_stream = bool(raw)

    def step_device(self, d_actions, d_obs, d_rewards, d_term, d_trunc):
        """Device-resident step.  Buffers: DeviceBuffer, a raw device address (int), or anything with `data_ptr()`
        (a torch tensor on this GPU: int32 [N, A] actions, float64 [N, A, F] observations, float64 [N, A] rewards,
        uint8 [N, A] flags, all contiguous)."""
        p = _dev_ptr
        _native.check(self._h, _native.lib().cz_step_device(self._h, p(d_actions), p(d_obs), p(d_rewards), p(d_term),
                                                            p(d_trunc)))
        self._issued(1)

    def step_device_compact(self, d_actions, d_codes, d_rewards, d_term, d_trunc, d_obs=None):
        """Device-resident step whose observation is one byte per feature: d_codes uint8 [N, A, codes_pitch] receives the index
        of every feature's value in `obs_table()` (256 float64; `obs_table()[codes]` is the float64 observation bit for bit).
        d_obs (optional): the float64 [N, A, F] observation as well."""
        p = _dev_ptr
        _native.check(self._h, _native.lib().cz_step_device_compact(self._h, p(d_actions), p(d_codes), p(d_obs), p(d_rewards), p(d_term),
                                                                    p(d_trunc)))
        self._issued(1)

    def set_compact_output(self, d_codes=None):
        """every one-step launch from now on (step_device, step_device_ring, step) also writes the compact observation to d_codes
        (uint8 [N, A, codes_pitch]); None switches it off.  Pass d_obs=None to those calls for codes only."""
        _native.check(self._h, _native.lib().cz_set_compact_output(self._h, _dev_ptr(d_codes)))

    @property
    def codes_pitch(self):
        """row length of the compact observation in bytes: F rounded up to a multiple of 16 (padding bytes are 255)"""
        return int(_native.lib().cz_codes_pitch(self._h))

    def obs_table(self):
        """the 256 float64 values a compact-observation code stands for"""
        t = np.empty(256, dtype=np.float64)
        _native.check(self._h, _native.lib().cz_obs_table(self._h, _ptr(t)))
        return t

    def step_device_ring(self, K, d_ring, action_stride, action_period, first_slot, d_obs, d_rewards, d_term, d_trunc):
        """K consecutive device-resident steps in one call (cz_step_device_ring: graph replay, or fused launches after set_ring_fused); step k
        reads its actions from ring slot (first_slot + k) % action_period."""
        p = _dev_ptr
        _native.check(self._h, _native.lib().cz_step_device_ring(self._h, int(K), p(d_ring), int(action_stride), int(action_period),
                                                                 int(first_slot), p(d_obs), p(d_rewards), p(d_term), p(d_trunc)))
        self._issued(K)

    def rollout(self, T, seed, step0=0, d_obs=None, d_rewards=None, d_term=None, d_trunc=None):
        p = _dev_ptr
        _native.check(self._h, _native.lib().cz_rollout(self._h, int(T), int(seed), int(step0), p(d_obs), p(d_rewards),
                                                        p(d_term), p(d_trunc)))
        self._issued(T)

    def rollout_compact(self, T, seed, step0, d_codes, d_obs=None, d_rewards=None, d_term=None, d_trunc=None):
        """`rollout` with a compact trajectory: d_codes uint8 [T, N, A, codes_pitch] (and, optionally, the float64 one)"""
        p = _dev_ptr
        _native.check(self._h, _native.lib().cz_rollout_compact(self._h, int(T), int(seed), int(step0), p(d_codes), p(d_obs), p(d_rewards),
                                                                p(d_term), p(d_trunc)))
        self._issued(T)

    def rollout_actions(self, d_actions, T, d_obs=None, d_rewards=None, d_term=None, d_trunc=None):
        """T fused steps over the caller's actions (device int32 [T, N, A]); every step's outputs go to the trajectory
        buffers ([T, N, A, F] / [T, N, A]).  Same results as T `step_device` calls over those rows."""
        p = _dev_ptr
        _native.check(self._h, _native.lib().cz_rollout_actions(self._h, int(T), p(d_actions), p(d_obs), p(d_rewards),
                                                                p(d_term), p(d_trunc)))
        self._issued(T)

    def sync(self):
        _native.check(self._h, _native.lib().cz_sync(self._h))

    # ------------------------------------------------------------------ state / stats
    def get_state(self, env_begin=0, env_count=None):
        n = self.num_envs if env_count is None else int(env_count)
        recs = np.empty((n, self.dims.RW), dtype=np.uint32)
        _native.check(self._h, _native.lib().cz_get_state(self._h, env_begin, n, _ptr(recs)))
        return recs

    def set_state(self, records, env_begin=0):
        recs = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1, self.dims.RW)
        _native.check(self._h, _native.lib().cz_set_state(self._h, env_begin, recs.shape[0], _ptr(recs)))

    def stats(self):
        st = _native.CzStats()
        _native.check(self._h, _native.lib().cz_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def reset_stats(self):
        _native.check(self._h, _native.lib().cz_reset_stats(self._h))

    def close(self):
        if getattr(self, "_rot", None) is not None:
            self.stop_rotation()
        if getattr(self, "_h", None):
            for b in self._buffers:
                b.free()
            for p, _ in self._pin_bufs.values():
                _native.lib().cz_host_free(self._h, p)
            self._pin_bufs = {}
            _native.lib().cz_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
