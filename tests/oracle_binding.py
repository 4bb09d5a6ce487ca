"""ctypes binding of the TEST ORACLE (oracle/libcz_oracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by cooking_zoo_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from cooking_zoo_amd import soa

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(REPO, "oracle", "libcz_oracle.so")


class CzoConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_dyn", C.c_int32), ("num_agents", C.c_int32),
                ("feat_len", C.c_int32), ("action_scheme", C.c_int32), ("max_steps", C.c_int32),
                ("end_condition_all", C.c_int32), ("num_recipes", C.c_int32), ("auto_reset", C.c_int32),
                ("num_layouts", C.c_int32), ("record_words", C.c_int32), ("recipe_nodes", C.c_int32),
                ("recipe_reward", C.c_double), ("max_time_penalty", C.c_double), ("recipe_penalty", C.c_double),
                ("recipe_node_reward", C.c_double)]


class CzoLayout(C.Structure):
    _fields_ = [("init_record", C.c_void_p), ("static_off", C.c_void_p), ("static_cells", C.c_void_p)]


class CzoMeta(C.Structure):
    _fields_ = [("cls", C.c_int), ("num", C.c_int)]


class CzoSpawn(C.Structure):
    _fields_ = [("despawn_rate", C.c_double), ("respawn_rate", C.c_double), ("seed", C.c_uint64), ("grace_period", C.c_int32),
                ("n_levels", C.c_int32), ("stride", C.c_int32), ("level_of_layout", C.c_void_p), ("n_x", C.c_void_p),
                ("n_y", C.c_void_p), ("xs", C.c_void_p), ("ys", C.c_void_p)]


class CzoCtx(C.Structure):
    _fields_ = [("cfg", C.POINTER(CzoConfig)), ("recipe_table", C.c_void_p), ("layouts", C.POINTER(CzoLayout)),
                ("meta", C.POINTER(CzoMeta)), ("n_meta", C.c_int), ("env_id_base", C.c_int64),
                ("pool_groups", C.c_int32), ("pool_active", C.c_int32), ("spawn", C.POINTER(CzoSpawn))]


def load_lib():
    if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(REPO, "oracle", "cz_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(_LIB_PATH)
    assert lib.czo_sizeof_config() == C.sizeof(CzoConfig)
    lib.czo_action.restype = C.c_uint32
    lib.czo_action.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_uint32, C.c_uint32]
    lib.czo_next_layout.restype = C.c_uint32
    lib.czo_next_layout.argtypes = [C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.czo_spawn_uniform.restype = C.c_double
    lib.czo_spawn_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    return lib


def meta_class_id(name):
    if name == "Agent":
        return 32
    return soa.class_node_id(name)


DEFAULT_REWARD = {"recipe_reward": 20, "max_time_penalty": -5, "recipe_penalty": -40, "recipe_node_reward": 0}


class Oracle:
    """Batch of envs stepped by the C oracle.  `layouts` = list of (init_record, static_off, static_cells)."""

    def __init__(self, dims: soa.Dims, meta: dict, recipe_table: np.ndarray, layouts, *, scheme=3, max_steps=400,
                 end_condition_all=False, num_recipes=None, auto_reset=0, reward_scheme=None, env_id_base=0):
        self.lib = load_lib()
        self.dims = dims
        rs = dict(DEFAULT_REWARD)
        rs.update(reward_scheme or {})
        self.cfg = CzoConfig(dims.W, dims.H, dims.D, dims.A, dims.F, int(scheme), int(max_steps),
                             int(bool(end_condition_all)), int(num_recipes or dims.A), int(auto_reset), len(layouts),
                             dims.RW, 16 if np.asarray(recipe_table).shape[1] > 9 else 8, float(rs["recipe_reward"]), float(rs["max_time_penalty"]),
                             float(rs["recipe_penalty"]), float(rs["recipe_node_reward"]))
        self.recipe_table = np.ascontiguousarray(recipe_table, dtype=np.uint32)
        self._keep = []
        self.layout_arr = (CzoLayout * len(layouts))()
        for i, (rec, off, cells) in enumerate(layouts):
            rec = np.ascontiguousarray(rec, dtype=np.uint32)
            off = np.ascontiguousarray(off, dtype=np.int32)
            cells = np.ascontiguousarray(cells, dtype=np.int16)
            assert rec.size == dims.RW
            self._keep += [rec, off, cells]
            self.layout_arr[i] = CzoLayout(rec.ctypes.data, off.ctypes.data, cells.ctypes.data)
        self.meta_arr = (CzoMeta * len(meta))(*[CzoMeta(meta_class_id(k), int(v)) for k, v in meta.items()])
        self.ctx = CzoCtx(C.pointer(self.cfg), self.recipe_table.ctypes.data, self.layout_arr, self.meta_arr,
                          len(meta), int(env_id_base), 1, 0, None)

    def set_spawn(self, despawn_rate, respawn_rate, grace_period, seed, spawn_cells, level_of_layout=None):
        """agent despawn / respawn with keyed draws (mirror of cz_set_spawn): spawn_cells[level][agent] = (x candidates, y
        candidates) as the level file lists them; level_of_layout[layout id] = level (None: one level).  Rates 0, 0: off."""
        if not (despawn_rate or respawn_rate):
            self.ctx.spawn = None
            return
        A, nl = self.dims.A, len(spawn_cells)
        stride = max(max(len(xs), len(ys)) for lv in spawn_cells for xs, ys in lv)
        xs_a, ys_a = np.zeros((nl, A, stride), dtype=np.int32), np.zeros((nl, A, stride), dtype=np.int32)
        nx, ny = np.zeros((nl, A), dtype=np.int32), np.zeros((nl, A), dtype=np.int32)
        for li, lv in enumerate(spawn_cells):
            for a, (xs, ys) in enumerate(lv[:A]):
                nx[li, a], ny[li, a] = len(xs), len(ys)
                xs_a[li, a, :len(xs)], ys_a[li, a, :len(ys)] = xs, ys
        lol = None if level_of_layout is None else np.ascontiguousarray(level_of_layout, dtype=np.uint8)
        self._spawn_keep = (xs_a, ys_a, nx, ny, lol)
        self._spawn = CzoSpawn(float(despawn_rate), float(respawn_rate), int(seed), int(grace_period), nl, stride,
                               None if lol is None else lol.ctypes.data, nx.ctypes.data, ny.ctypes.data, xs_a.ctypes.data, ys_a.ctypes.data)
        self.ctx.spawn = C.pointer(self._spawn)

    def set_layout(self, i, rec, off, cells):
        """replace layout i of the pool (mirror of cz_update_layouts)"""
        rec = np.ascontiguousarray(rec, dtype=np.uint32)
        off = np.ascontiguousarray(off, dtype=np.int32)
        cells = np.ascontiguousarray(cells, dtype=np.int16)
        assert rec.size == self.dims.RW
        self._keep += [rec, off, cells]                  # (the replaced arrays stay alive too: cheap, and nothing can dangle)
        self.layout_arr[i] = CzoLayout(rec.ctypes.data, off.ctypes.data, cells.ctypes.data)

    def set_layout_group(self, groups, active):
        """mirror of cz_set_layout_group"""
        self.ctx.pool_groups, self.ctx.pool_active = int(groups), int(active)

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def reset_env(self, rec, layout_id, obs=None):
        return self.lib.czo_reset_env(C.byref(self.ctx), C.c_int64(0), C.c_uint32(layout_id), self._p(rec), self._p(obs))

    def observe(self, rec):
        obs = np.empty((self.dims.A, self.dims.F), dtype=np.float64)
        err = self.lib.czo_observe_env(C.byref(self.ctx), self._p(rec), self._p(obs))
        assert err == 0, err
        return obs

    def step_env(self, rec, actions, env_local=0):
        A, F = self.dims.A, self.dims.F
        obs = np.empty((A, F), dtype=np.float64)
        rew = np.empty(A, dtype=np.float64)
        term = np.empty(A, dtype=np.uint8)
        trunc = np.empty(A, dtype=np.uint8)
        acts = np.ascontiguousarray(actions, dtype=np.int32)
        err = self.lib.czo_step_env(C.byref(self.ctx), C.c_int64(env_local), self._p(rec), self._p(acts), self._p(obs),
                                    self._p(rew), self._p(term), self._p(trunc))
        return err, obs, rew, term, trunc

    def step_batch(self, records, actions, want_obs=True):
        n = records.shape[0]
        A, F = self.dims.A, self.dims.F
        obs = np.empty((n, A, F), dtype=np.float64) if want_obs else None
        rew = np.empty((n, A), dtype=np.float64)
        term = np.empty((n, A), dtype=np.uint8)
        trunc = np.empty((n, A), dtype=np.uint8)
        acts = np.ascontiguousarray(actions, dtype=np.int32)
        err = self.lib.czo_step_batch(C.byref(self.ctx), C.c_int64(n), self._p(records), self._p(acts), self._p(obs),
                                      self._p(rew), self._p(term), self._p(trunc))
        return err, obs, rew, term, trunc

    def rollout(self, records, T, seed, step0=0, want_obs=True, want_actions=False):
        n = records.shape[0]
        A, F = self.dims.A, self.dims.F
        obs = np.empty((n, A, F), dtype=np.float64) if want_obs else None
        rew = np.empty((n, A), dtype=np.float64)
        term = np.empty((n, A), dtype=np.uint8)
        trunc = np.empty((n, A), dtype=np.uint8)
        acts = np.empty((T, n, A), dtype=np.int32) if want_actions else None
        err = self.lib.czo_rollout(C.byref(self.ctx), C.c_int64(n), self._p(records), C.c_int32(T), C.c_uint64(seed),
                                   C.c_uint32(step0), self._p(obs), self._p(rew), self._p(term), self._p(trunc),
                                   self._p(acts))
        return err, obs, rew, term, trunc, acts


class VecOracle:
    """Oracle twin of a cooking_zoo_amd CookingVecEnv configuration (same layout pool, recipes, rewards):
    the checker for smoke()/bench parity and the `cpu_baseline` leg of bench.py.  Built only from host-side
    tables (layouts, meta, recipe table) -- it never touches the GPU library."""

    def __init__(self, *, layouts, meta, recipe_table, recipe_ids, dims, scheme, max_steps, end_condition_all,
                 num_recipes, reward_scheme, pool_slices, env_level, num_envs, env_id_base=0, auto_reset=1):
        lay = []
        for i, l in enumerate(layouts):
            off, cells = l.static_table()
            lay.append((l.init_record(dims, i), off, cells))
        self.oracle = Oracle(dims, meta, recipe_table, lay, scheme=scheme, max_steps=max_steps,
                             end_condition_all=end_condition_all, num_recipes=num_recipes, auto_reset=auto_reset,
                             reward_scheme=reward_scheme, env_id_base=env_id_base)
        self.dims, self.num_envs, self.env_id_base = dims, num_envs, env_id_base
        self.records = np.zeros((num_envs, dims.RW), dtype=np.uint32)
        self.recipe_ids = np.asarray(recipe_ids, dtype=np.uint8)
        self.pool_slices, self.env_level = pool_slices, env_level
        self.n_layouts = len(layouts)

    @classmethod
    def from_vec_env(cls, env, num_envs=None, env_id_base=None, auto_reset=1):
        """env: a CookingVecEnv or its device-free half, a cooking_zoo_amd.vec_env.BatchTables"""
        n = env.num_envs if num_envs is None else num_envs
        me = cls._from_vec_env(env, n, env_id_base, auto_reset)
        if getattr(env, "spawn_cells", None) is not None:        # the batch evaluates despawn / respawn on the device: so does its twin
            d, r, g, seed = env._spawn_cfg
            me.oracle.set_spawn(d, r, g, seed, env.spawn_cells, env.level_of_layout)
        return me

    @classmethod
    def _from_vec_env(cls, env, n, env_id_base, auto_reset):
        return cls(layouts=env.layouts, meta=env.meta, recipe_table=env.recipe_table, recipe_ids=env.recipe_ids[:n],
                   dims=env.dims, scheme=env.scheme_class.CODE, max_steps=env.max_steps,
                   end_condition_all=env.end_condition_all_dishes, num_recipes=env.num_recipes,
                   reward_scheme=env.reward_scheme, pool_slices=env.pool_slices, env_level=env.env_level[:n],
                   num_envs=n, env_id_base=env.env_id_base if env_id_base is None else env_id_base,
                   auto_reset=auto_reset)

    def reset(self):
        lib = self.oracle.lib
        obs = np.empty((self.num_envs, self.dims.A, self.dims.F))
        for e in range(self.num_envs):
            base, count = self.pool_slices[self.env_level[e]]
            pool = base | (count << 16)
            lay = lib.czo_next_layout(self.env_id_base + e, 0, pool, self.n_layouts)
            rec = self.records[e]
            rec[:] = 0
            rid = self.recipe_ids[e]
            rec[soa.W_RECIPES] = int(rid[0]) | (int(rid[1]) << 8) | (int(rid[2]) << 16) | (int(rid[3]) << 24)
            rec[soa.W_POOL] = pool
            # czo_reset_env keys nothing on env index; env-local offset only matters for auto-reset draws
            err = lib.czo_reset_env(C.byref(self.oracle.ctx), C.c_int64(e), C.c_uint32(lay),
                                    rec.ctypes.data_as(C.c_void_p), obs[e].ctypes.data_as(C.c_void_p))
            assert err == 0
        return obs

    def step(self, actions, want_obs=True):
        err, obs, rew, term, trunc = self.oracle.step_batch(self.records, actions, want_obs)
        assert err == 0, err
        return obs, rew, term, trunc

    def rollout(self, T, seed, step0=0, want_obs=True):
        err, obs, rew, term, trunc, _ = self.oracle.rollout(self.records, T, seed, step0, want_obs)
        assert err == 0, err
        return obs, rew, term, trunc

    # ---- the layout rotation of CookingVecEnv.rotate_layouts, mirrored (tests apply the env's `rotation_events` here)
    def update_layouts(self, first, layouts):
        for k, l in enumerate(layouts):
            off, cells = l.static_table()
            self.oracle.set_layout(first + k, l.init_record(self.dims, first + k), off, cells)

    def set_layout_group(self, groups, active):
        self.oracle.set_layout_group(groups, active)

    def apply_rotation_event(self, ev):
        if ev[1] == "group":
            self.set_layout_group(ev[2], ev[3])
        else:
            self.update_layouts(ev[2], ev[3])


class ShardedOracle:
    """The same twin for BASELINE-sized batches: the envs are cut into contiguous slices, one VecOracle (and one host
    thread, the C calls release the GIL) per slice, each keyed with the global id of its first env -- so the result is
    what one VecOracle over all envs would produce, in a fraction of the time."""

    def __init__(self, env, threads=None):
        import os
        n = env.num_envs
        threads = threads or max(1, min(len(os.sched_getaffinity(0)), 64, n // 256 or 1))
        cuts = [n * i // threads for i in range(threads + 1)]
        self.spans = [(lo, hi) for lo, hi in zip(cuts, cuts[1:]) if hi > lo]
        self.parts = [VecOracle(layouts=env.layouts, meta=env.meta, recipe_table=env.recipe_table,
                                recipe_ids=env.recipe_ids[lo:hi], dims=env.dims, scheme=env.scheme_class.CODE,
                                max_steps=env.max_steps, end_condition_all=env.end_condition_all_dishes,
                                num_recipes=env.num_recipes, reward_scheme=env.reward_scheme, pool_slices=env.pool_slices,
                                env_level=env.env_level[lo:hi], num_envs=hi - lo, env_id_base=env.env_id_base + lo)
                      for lo, hi in self.spans]
        if getattr(env, "spawn", None) is not None:
            d, r, g, seed = env._spawn_cfg
            for p in self.parts:
                p.oracle.set_spawn(d, r, g, seed, env.spawn_cells, env.level_of_layout)
        self.num_envs, self.dims = n, env.dims

    def _each(self, fn):
        import threading
        out = [None] * len(self.parts)

        def run(i):
            out[i] = fn(i, self.parts[i])
        ths = [threading.Thread(target=run, args=(i,)) for i in range(len(self.parts))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return out

    @property
    def records(self):
        return np.concatenate([p.records for p in self.parts])

    def reset(self):
        return np.concatenate(self._each(lambda i, p: p.reset()))

    def step(self, actions, want_obs=True):
        res = self._each(lambda i, p: p.step(actions[self.spans[i][0]:self.spans[i][1]], want_obs))
        return tuple(None if res[0][k] is None else np.concatenate([r[k] for r in res]) for k in range(4))

    def rollout(self, T, seed, step0=0, want_obs=True):
        res = self._each(lambda i, p: p.rollout(T, seed, step0, want_obs))
        return tuple(None if res[0][k] is None else np.concatenate([r[k] for r in res]) for k in range(4))
