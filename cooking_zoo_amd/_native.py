"""ctypes binding of include/cookingzoo.h (libcookingzoo_hip.so).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible, creating an
environment raises.  (Pure-data modules such as soa.py / layout.py import without it.)
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CZ_LIB", os.path.join(_HERE, "csrc", "libcookingzoo_hip.so"))   # CZ_LIB: diagnostic builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cookingzoo.h")


def header_abi_version() -> int:
    """CZ_ABI_VERSION of include/cookingzoo.h - the only place the number is written down."""
    import re
    m = re.search(r"^#define\s+CZ_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read(), flags=re.M)
    if not m:
        raise RuntimeError(f"no CZ_ABI_VERSION in {HEADER_PATH}")
    return int(m.group(1))


class CzConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_agents", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("max_dyn", C.c_int32), ("feat_len", C.c_int32), ("action_scheme", C.c_int32),
                ("max_steps", C.c_int32), ("end_condition_all", C.c_int32), ("num_recipes", C.c_int32),
                ("auto_reset", C.c_int32), ("device_id", C.c_int32), ("env_id_base", C.c_int64),
                ("recipe_reward", C.c_double), ("max_time_penalty", C.c_double), ("recipe_penalty", C.c_double),
                ("recipe_node_reward", C.c_double)]


class CzStats(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("episodes", C.c_uint64), ("length_sum", C.c_uint64),
                ("truncations", C.c_uint64), ("terminations", C.c_uint64), ("recipes_completed", C.c_uint64 * 4),
                ("return_sum", C.c_double * 4)]

    def as_dict(self):
        return {"env_steps": int(self.env_steps), "episodes": int(self.episodes), "length_sum": int(self.length_sum),
                "truncations": int(self.truncations), "terminations": int(self.terminations),
                "recipes_completed": [int(v) for v in self.recipes_completed],
                "return_sum": [float(v) for v in self.return_sum]}


# every symbol include/cookingzoo.h declares: (name, restype, argtypes)
_VP, _I32, _I64, _U32, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
SYMBOLS = [
    ("cz_create", C.c_int, [C.POINTER(CzConfig), C.POINTER(_VP)]),
    ("cz_destroy", C.c_int, [_VP]),
    ("cz_last_error", C.c_char_p, [_VP]),
    ("cz_record_words", _I32, [_VP]),
    ("cz_abi_version", _I32, []),
    ("cz_sizeof_config", _I32, []),
    ("cz_sizeof_stats", _I32, []),
    ("cz_debug_set_stamps", C.c_int, [_VP, _VP]),
    ("cz_debug_set_timeline", C.c_int, [_VP, _VP, _I32]),
    ("cz_sync", C.c_int, [_VP]),
    ("cz_load_recipes", C.c_int, [_VP, _VP, _I32, _I32]),
    ("cz_load_layouts", C.c_int, [_VP, _VP, _VP, _I32]),
    ("cz_update_layouts", C.c_int, [_VP, _I32, _I32, _VP, _VP]),
    ("cz_set_layout_group", C.c_int, [_VP, _I32, _I32]),
    ("cz_layout_updates", _I64, [_VP]),
    ("cz_set_spawn", C.c_int, [_VP, C.c_double, C.c_double, _I32, _U64, _I32, _VP, _I32, _VP, _VP, _VP, _VP]),
    ("cz_spawn_exhausted", _I64, [_VP]),
    ("cz_spawn_uniform", C.c_double, [_U64, _I64, _U32, _U32, _I32, _U32]),
    ("cz_set_state", C.c_int, [_VP, _I64, _I64, _VP]),
    ("cz_get_state", C.c_int, [_VP, _I64, _I64, _VP]),
    ("cz_reset", C.c_int, [_VP, _I64, _I64, _VP, _VP, _VP, _VP]),
    ("cz_observe", C.c_int, [_VP, _I64, _I64, _VP]),
    ("cz_observe_compact", C.c_int, [_VP, _I64, _I64, _VP]),
    ("cz_observe_device", C.c_int, [_VP, _I64, _I64, _VP, _VP]),
    ("cz_step", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    ("cz_step_device", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    ("cz_step_device_compact", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("cz_set_compact_output", C.c_int, [_VP, _VP]),
    ("cz_step_compact", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    ("cz_codes_pitch", _I32, [_VP]),
    ("cz_obs_table", C.c_int, [_VP, _VP]),
    ("cz_obs_table_device", _VP, [_VP]),
    ("cz_probe_closed_loop_compact", C.c_int, [_VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, C.POINTER(C.c_float)]),
    ("cz_step_device_many", C.c_int, [_VP, _I32, _VP, _I64, _I32, _VP, _VP, _VP, _VP]),
    ("cz_step_device_ring", C.c_int, [_VP, _I32, _VP, _I64, _I32, _I32, _VP, _VP, _VP, _VP]),
    ("cz_ring_prepare", C.c_int, [_VP, _I32, _VP, _I64, _I32, _I32, _VP, _VP, _VP, _VP]),
    ("cz_launch_counts", C.c_int, [_VP, C.POINTER(_I64), C.POINTER(_I64), _I32]),
    ("cz_set_ring_fused", C.c_int, [_VP, _I32]),
    ("cz_ring_fused_steps", C.c_int64, [_VP, _I32]),
    ("cz_last_marks", C.c_int, [_VP, _VP]),
    ("cz_set_stream", C.c_int, [_VP, _VP]),
    ("cz_probe_output_only", C.c_int, [_VP, _VP, C.c_size_t, _I32, _VP]),
    ("cz_probe_policy", C.c_int, [_VP, _VP, _VP, _VP]),
    ("cz_probe_closed_loop", C.c_int, [_VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, C.POINTER(C.c_float)]),
    ("cz_rollout", C.c_int, [_VP, _I32, _U64, _U32, _VP, _VP, _VP, _VP]),
    ("cz_rollout_compact", C.c_int, [_VP, _I32, _U64, _U32, _VP, _VP, _VP, _VP, _VP]),
    ("cz_rollout_actions", C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP]),
    ("cz_action", _U32, [_U64, _I64, _I32, _U32, _U32]),
    ("cz_next_layout", _U32, [_I64, _U32, _U32, _U32]),
    ("cz_next_layout_group", _U32, [_I64, _U32, _U32, _U32, _U32, _U32]),
    ("cz_dev_alloc", _VP, [_VP, C.c_size_t]),
    ("cz_dev_free", C.c_int, [_VP, _VP]),
    ("cz_host_alloc", _VP, [_VP, C.c_size_t]),
    ("cz_host_free", C.c_int, [_VP, _VP]),
    ("cz_memcpy_h2d", C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    ("cz_memcpy_d2h", C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    ("cz_timer_start", C.c_int, [_VP]),
    ("cz_timer_stop", C.c_int, [_VP, C.POINTER(C.c_float)]),
    ("cz_kernel_time_reset", C.c_int, [_VP, _I32]),
    ("cz_kernel_time_read", C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(_I64)]),
    ("cz_get_stats", C.c_int, [_VP, C.POINTER(CzStats)]),
    ("cz_reset_stats", C.c_int, [_VP]),
    ("cz_comm_unique_id", C.c_int, [_VP]),
    ("cz_comm_init", C.c_int, [_VP, _I32, _I32, _VP]),
    ("cz_stats_allgather", C.c_int, [_VP, _VP]),
    ("cz_comm_barrier", C.c_int, [_VP]),
    ("cz_runtime_paths", C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
]

_lib = None


class NativeError(RuntimeError):
    pass


class _LateTorchGuard:
    """PyTorch's ROCm wheels bundle their own libamdhip64.so.7 / librccl.so.1 under the SONAMEs of the system libraries
    this library links.  Loaded after this library, they sit next to the system copies already in use: everything seems
    to work until the process aborts in a destructor at exit ("double free or corruption", rc 134).  The working order is
    torch first (both then share torch's runtime), so a first `import torch` that comes too late is refused with an
    explanation - at the moment torch's module is EXECUTED, not when somebody merely asks whether torch is installed
    (`importlib.util.find_spec("torch")` keeps returning a spec), and only for a torch that really bundles a HIP runtime of
    its own (a CPU-only build, or one built against the system ROCm, imports as usual).  Installed as a side effect of
    loading the library; CZ_ALLOW_LATE_TORCH=1 removes the guard."""

    MESSAGE = ("`import torch` after cooking_zoo_amd has loaded libcookingzoo_hip.so: torch's bundled ROCm runtime would be loaded next "
               "to the system one already in use and the process would abort at exit. Import torch BEFORE creating the first "
               "cooking_zoo_amd environment (see INTEGRATION.md), or set CZ_ALLOW_LATE_TORCH=1 to take the risk.")

    class _Refuse:
        def __init__(self, inner):
            self._inner = inner

        def create_module(self, spec):
            return None

        def exec_module(self, module):
            raise ImportError(_LateTorchGuard.MESSAGE)

    def find_spec(self, name, path=None, target=None):
        if name != "torch":
            return None
        import glob
        import importlib.machinery
        spec = importlib.machinery.PathFinder.find_spec(name, path)
        if spec is None or not spec.submodule_search_locations:
            return None                                   # not installed: the ordinary "No module named torch"
        bundled = any(glob.glob(os.path.join(loc, "lib", pat)) for loc in spec.submodule_search_locations
                      for pat in ("libamdhip64.so*", "librccl.so*"))
        if not bundled:
            return None                                   # no runtime of its own: nothing to collide with
        spec.loader = self._Refuse(spec.loader)
        return spec


def lib():
    """Load libcookingzoo_hip.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"or `make -C cooking_zoo_amd/csrc` (hipcc --offload-arch=gfx950). "
                              f"cooking_zoo_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.cz_abi_version.restype = _I32
        if L.cz_abi_version() != header_abi_version():
            raise NativeError(f"{LIB_PATH} reports ABI {L.cz_abi_version()}, include/cookingzoo.h declares "
                              f"{header_abi_version()}: rebuild it (make -C cooking_zoo_amd/csrc)")
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.cz_sizeof_config() != C.sizeof(CzConfig) or L.cz_sizeof_stats() != C.sizeof(CzStats):
            raise NativeError("libcookingzoo_hip.so and cooking_zoo_amd/_native.py disagree on struct layouts")
        _lib = L
        if "torch" not in sys.modules and not os.environ.get("CZ_ALLOW_LATE_TORCH"):
            sys.meta_path.insert(0, _LateTorchGuard())
    return _lib


def check(handle, rc):
    if rc != 0:
        msg = lib().cz_last_error(handle)
        raise NativeError(msg.decode() if msg else f"libcookingzoo_hip error {rc}")
