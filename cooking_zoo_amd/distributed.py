"""Multi-GPU host logic: env sharding and the episode-statistics exchange.

Env instances are independent, so the step path has no data-path collective: rank r of G owns the contiguous global
env ids [r*N/G, (r+1)*N/G) (the layout draws and the on-device action stream are keyed by the GLOBAL id, so results do
not depend on G).  The only exchange is an all-gather of one `cz_stats` struct per rank, reduced locally in rank order
(deterministic float64 sums):

* on GPUs: RCCL over xGMI through the C-ABI (`cz_comm_init` / `cz_stats_allgather`), the unique id travelling over the
  torch.distributed store that launched the job;
* `gather_stats_torch` does the same exchange with `torch.distributed.all_gather` (gloo on CPU, nccl on GPU); the CPU
  tests use it, and bench.py falls back to it if the direct RCCL path is unavailable.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np

from cooking_zoo_amd import _native

STAT_KEYS = ("env_steps", "episodes", "length_sum", "truncations", "terminations")


def shard_range(total_envs: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first global env id, count) of `rank`; contiguous, sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(total_envs, world_size)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def stats_to_bytes(stats: Dict) -> np.ndarray:
    st = _native.CzStats()
    for k in STAT_KEYS:
        setattr(st, k, int(stats[k]))
    for a in range(4):
        st.recipes_completed[a] = int(stats["recipes_completed"][a])
        st.return_sum[a] = float(stats["return_sum"][a])
    return np.frombuffer(bytes(st), dtype=np.uint8).copy()


def stats_from_bytes(buf) -> Dict:
    return _native.CzStats.from_buffer_copy(bytes(bytearray(buf))).as_dict()


def reduce_stats(per_rank: List[Dict]) -> Dict:
    """Sum in rank order (fixed order -> bitwise reproducible float64 totals)."""
    out = {k: 0 for k in STAT_KEYS}
    out["recipes_completed"] = [0, 0, 0, 0]
    out["return_sum"] = [0.0, 0.0, 0.0, 0.0]
    for st in per_rank:
        for k in STAT_KEYS:
            out[k] += st[k]
        for a in range(4):
            out["recipes_completed"][a] += st["recipes_completed"][a]
            out["return_sum"][a] += st["return_sum"][a]
    return out


def gather_stats_torch(local: Dict, device=None) -> List[Dict]:
    """All-gather of the stats struct with torch.distributed (backend chosen by the caller: gloo or nccl)."""
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(stats_to_bytes(local))
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [stats_from_bytes(t.cpu().numpy()) for t in out]


def gather_stats_rccl(env, world_size: int, rank: int, broadcast_bytes) -> List[Dict]:
    """RCCL all-gather through the C-ABI.  `broadcast_bytes(payload_or_None) -> bytes` must return rank 0's payload
    on every rank (e.g. via torch.distributed.broadcast_object_list)."""
    L = _native.lib()
    uid = (C.c_uint8 * 128)()
    payload = None
    if rank == 0:
        # a failure here must reach every rank (they are all waiting in the broadcast): ship b"" instead of raising
        payload = bytes(uid) if L.cz_comm_unique_id(uid) == 0 else b""
    payload = broadcast_bytes(payload)
    if not payload:
        raise _native.NativeError("rank 0 could not create an RCCL unique id (librccl missing?)")
    uid = (C.c_uint8 * 128).from_buffer_copy(payload)
    _native.check(env._h, L.cz_comm_init(env._h, world_size, rank, uid))
    out = (_native.CzStats * world_size)()
    _native.check(env._h, L.cz_stats_allgather(env._h, out))
    return [out[i].as_dict() for i in range(world_size)]
