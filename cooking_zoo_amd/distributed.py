"""Multi-GPU host logic: env sharding and the episode-statistics exchange.

Env instances are independent, so the step path has no data-path collective: rank r of G owns the contiguous global
env ids [r*N/G, (r+1)*N/G) (the layout draws and the on-device action stream are keyed by the GLOBAL id, so results do
not depend on G).  The only exchange is an all-gather of one `cz_stats` struct per rank, reduced locally in rank order
(deterministic float64 sums):

* on GPUs: RCCL over xGMI through the C-ABI (`cz_comm_init` / `cz_stats_allgather`); the 128-byte unique id travels
  over any control plane -- `FileRendezvous` below (a directory in /dev/shm: enough for the ranks of one node, no
  torch, no sockets) or the store of whatever launched the job;
* `gather_stats_torch` does the same exchange with `torch.distributed.all_gather` (gloo on CPU, nccl on GPU) for
  callers that already live in a torch.distributed job; the gloo CPU test uses it.

`launch_local` starts one fresh process per GPU of this node (the parent never touches the GPU) -- what
`python bench.py --gpus N` does when it was not started by a launcher.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple

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


# ---------------------------------------------------------------------------------------------------------
# control plane for the ranks of one node (no torch): a rendezvous directory + a local launcher
# ---------------------------------------------------------------------------------------------------------

class RendezvousTimeout(RuntimeError):
    pass


class RendezvousPoisoned(RuntimeError):
    """another rank has declared this rendezvous void (`FileRendezvous.poison`): whoever waits in it, or enters it, stops"""


class FileRendezvous:
    """Small-message collectives between the ranks of ONE node through a shared directory (tmpfs).

    Every rank must issue the same sequence of calls; call number i of rank r lives in the file `<i>.<r>` (written to a
    temporary name and renamed, so a reader never sees a partial file).  `timeout` bounds every wait: a rank that died
    turns into RendezvousTimeout on the others instead of a hang."""

    def __init__(self, path: str, rank: int, world_size: int, timeout: float = 300.0):
        if not 0 <= rank < world_size:
            raise ValueError("rank out of range")
        self.path, self.rank, self.world, self.timeout = path, int(rank), int(world_size), float(timeout)
        self._seq = 0
        os.makedirs(path, exist_ok=True)

    @classmethod
    def from_env(cls, env=None, timeout: float = 300.0) -> "FileRendezvous":
        """RANK / WORLD_SIZE as torchrun and `launch_local` set them; the directory is CZ_RDZV_DIR when given, otherwise
        derived from what the ranks of one launcher share: MASTER_PORT and the launcher's pid + start time."""
        env = os.environ if env is None else env
        rank, world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
        path = env.get("CZ_RDZV_DIR")
        if not path and world == 1 and "WORLD_SIZE" not in env:
            # a lone process nobody launched: a directory of its own (two `python bench.py` runs of one shell share parent
            # pid and start time, and a crashed earlier run may have left files behind under the derived name)
            base = "/dev/shm" if os.path.isdir("/dev/shm") else None
            path = tempfile.mkdtemp(prefix=f"cz_rdzv_solo_{os.getpid()}_", dir=base)
        if not path:
            ppid = os.getppid()
            try:
                started = open(f"/proc/{ppid}/stat").read().rsplit(")", 1)[1].split()[19]      # field 22: starttime
            except (OSError, IndexError):
                started = "0"
            base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
            path = os.path.join(base, f"cz_rdzv_{env.get('MASTER_PORT', '0')}_{ppid}_{started}")
        return cls(path, rank, world, timeout)

    def _file(self, seq: int, rank: int) -> str:
        return os.path.join(self.path, f"{seq}.{rank}")

    def subdir(self, name: str) -> "FileRendezvous":
        """A fresh rendezvous (call numbers start at 0 again) in a sub-directory, for a phase all ranks (re)start together."""
        return FileRendezvous(os.path.join(self.path, name), self.rank, self.world, self.timeout)

    def poison(self, reason: str) -> None:
        """Declare this rendezvous void: every rank that waits in it or enters it from now on raises RendezvousPoisoned
        (used when one rank has to abandon a phase that the others are still in)."""
        tmp = os.path.join(self.path, f"poison.tmp.{self.rank}")
        with open(tmp, "w") as f:
            f.write(reason)
        os.rename(tmp, os.path.join(self.path, "poison"))

    def _check_poison(self) -> None:
        try:
            with open(os.path.join(self.path, "poison")) as f:
                raise RendezvousPoisoned(f.read())
        except FileNotFoundError:
            pass

    def all_gather(self, payload: bytes) -> List[bytes]:
        self._check_poison()
        seq, self._seq = self._seq, self._seq + 1
        tmp = self._file(seq, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.rename(tmp, self._file(seq, self.rank))
        out: List[Optional[bytes]] = [None] * self.world
        deadline = time.monotonic() + self.timeout
        missing = set(range(self.world))
        pause = 20e-6
        while missing:
            for r in list(missing):
                try:
                    with open(self._file(seq, r), "rb") as f:
                        out[r] = f.read()
                    missing.discard(r)
                except FileNotFoundError:
                    pass
            if missing:
                self._check_poison()
                if time.monotonic() > deadline:
                    raise RendezvousTimeout(f"rank {self.rank}: ranks {sorted(missing)} did not reach collective #{seq} "
                                            f"within {self.timeout:.0f} s")
                time.sleep(pause)
                pause = min(pause * 1.5, 2e-3)
        return out          # type: ignore[return-value]

    def broadcast(self, payload: Optional[bytes], src: int = 0) -> bytes:
        return self.all_gather(payload if self.rank == src and payload is not None else b"")[src]

    def barrier(self) -> None:
        self.all_gather(b"")

    def close(self) -> None:
        """Last call of every rank.  Rank 0 removes the directory, but only after every other rank has said that it will
        not read from it again (a `bye.<rank>` file written after that rank has left the final barrier)."""
        try:
            self.barrier()
        except (RendezvousTimeout, RendezvousPoisoned):
            pass
        if self.rank != 0:
            try:
                open(os.path.join(self.path, f"bye.{self.rank}"), "wb").close()
            except OSError:
                pass
            return
        deadline = time.monotonic() + min(self.timeout, 30.0)
        while time.monotonic() < deadline and not all(os.path.exists(os.path.join(self.path, f"bye.{r}")) for r in range(1, self.world)):
            time.sleep(1e-3)
        shutil.rmtree(self.path, ignore_errors=True)


def launch_local(n_ranks: int, argv: Sequence[str], *, extra_env: Optional[Dict[str, str]] = None, timeout: float = 1800.0,
                 python: Optional[str] = None) -> int:
    """Start `n_ranks` fresh processes `python argv...` on this node, one per GPU, and wait for them.

    The caller must not have touched the GPU (nothing here does): every child is a new interpreter that gets RANK,
    LOCAL_RANK, WORLD_SIZE and CZ_RDZV_DIR (a fresh tmpfs directory for `FileRendezvous`) in its environment and pins
    its own device.  Rank 0 inherits stdout (the one JSON line of bench.py); the other ranks' stdout goes to stderr.
    Returns 0 if every rank exited with 0.  If a rank fails or the deadline passes, the remaining ranks are killed (each
    child is its own process group, killed by that exact group id) and the first non-zero code (or 124) is returned."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    rdzv = tempfile.mkdtemp(prefix="cz_rdzv_", dir=base)
    procs: List[subprocess.Popen] = []

    def kill_live() -> None:
        """every child that is still running: killed by its own process-group id (start_new_session: pgid == pid of exactly
        this child), then reaped"""
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    # The children live in sessions of their own, so a signal aimed at this process (Ctrl-C, `timeout N python bench.py
    # --gpus 8`, a driver's kill) does not reach them by itself: it is turned into an exception here, and the `finally`
    # below kills them before the rendezvous directory goes away.  (SIGKILL of this process cannot be caught: the children
    # then notice the missing parent through the rendezvous timeout.)
    class _Signalled(BaseException):
        pass

    def on_signal(signum, frame):
        raise _Signalled(signum)

    previous = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            try:
                previous[sig] = signal.signal(sig, on_signal)
            except (ValueError, OSError):
                pass
    try:
        for r in range(n_ranks):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                        "CZ_RDZV_DIR": rdzv, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            env.update(extra_env or {})
            procs.append(subprocess.Popen([python or sys.executable, *argv], env=env, start_new_session=True,
                                          stdout=None if r == 0 else sys.stderr))
        deadline = time.monotonic() + timeout
        rc = 0
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    if code != 0 and rc == 0:
                        rc = code
            if rc != 0 or time.monotonic() > deadline:
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.02)
        return rc
    except _Signalled as sig:
        return 128 + int(sig.args[0])
    finally:
        kill_live()                                      # a no-op when every rank has exited by itself
        for sig, old in previous.items():
            try:
                signal.signal(sig, old)
            except (ValueError, OSError):
                pass
        shutil.rmtree(rdzv, ignore_errors=True)          # only after the children are gone


def comm_init_with_deadline(env, world_size: int, rank: int, rdzv: FileRendezvous, seconds: float = 120.0) -> Tuple[bool, str]:
    """cz_comm_init for every rank, under a deadline.  The unique id is created by rank 0 and broadcast over `rdzv` on the
    calling thread; only the RCCL bring-up itself runs on a helper thread, so a communicator that hangs on an unfamiliar
    node costs `seconds`, not the run.  All ranks agree on the outcome (it is all-gathered): returns (ok, message).  After
    a False the helper thread may still be stuck inside RCCL: the caller must finish with os._exit(non-zero)."""
    L = _native.lib()
    uid = (C.c_uint8 * 128)()
    payload = None
    if rank == 0:
        payload = bytes(uid) if L.cz_comm_unique_id(uid) == 0 else b""
    payload = rdzv.broadcast(payload, src=0)
    box = {}
    if payload:
        uid = (C.c_uint8 * 128).from_buffer_copy(payload)

        def bring_up():
            rc = L.cz_comm_init(env._h, world_size, rank, uid)
            box["msg"] = "" if rc == 0 else (L.cz_last_error(env._h) or b"cz_comm_init failed").decode()

        th = threading.Thread(target=bring_up, daemon=True)
        th.start()
        th.join(timeout=seconds)
        mine = "timed out" if th.is_alive() else box.get("msg", "")
    else:
        mine = "rank 0 could not create an RCCL unique id (librccl missing?)"
    every = [b.decode() for b in rdzv.all_gather(mine.encode())]
    bad = [f"rank {r}: {m}" for r, m in enumerate(every) if m]
    return (not bad), "; ".join(bad)


def allgather_stats_rccl(env, world_size: int) -> List[Dict]:
    """cz_stats_allgather on an initialised communicator: one cz_stats per rank, in rank order."""
    out = (_native.CzStats * world_size)()
    _native.check(env._h, _native.lib().cz_stats_allgather(env._h, out))
    return [out[i].as_dict() for i in range(world_size)]
