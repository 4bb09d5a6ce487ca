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
                ("num_layouts", C.c_int32), ("record_words", C.c_int32),
                ("recipe_reward", C.c_double), ("max_time_penalty", C.c_double), ("recipe_penalty", C.c_double),
                ("recipe_node_reward", C.c_double)]


class CzoLayout(C.Structure):
    _fields_ = [("init_record", C.c_void_p), ("static_off", C.c_void_p), ("static_cells", C.c_void_p)]


class CzoMeta(C.Structure):
    _fields_ = [("cls", C.c_int), ("num", C.c_int)]


class CzoCtx(C.Structure):
    _fields_ = [("cfg", C.POINTER(CzoConfig)), ("recipe_table", C.c_void_p), ("layouts", C.POINTER(CzoLayout)),
                ("meta", C.POINTER(CzoMeta)), ("n_meta", C.c_int), ("env_id_base", C.c_int64)]


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
                             dims.RW, float(rs["recipe_reward"]), float(rs["max_time_penalty"]),
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
                          len(meta), int(env_id_base))

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
