"""The torch-free multi-rank path of bench.py on a machine without GPUs: `python bench.py --gpus 2 --dry-run` makes the
parent start two fresh rank processes (cooking_zoo_amd.distributed.launch_local), the ranks meet through the tmpfs
rendezvous directory, exchange their (made-up) timings, and rank 0 prints the aggregated line: whole-job throughput =
sum of the ranks' env-steps / the slowest rank's time, per timed region, median over the regions."""
import json
import os
import subprocess
import sys
import textwrap
import threading

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _run_bench(*extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CZ_RDZV_DIR"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *extra], capture_output=True, text=True, env=e, timeout=180)
    return p


def test_gpus_flag_spawns_ranks_and_aggregates():
    p = _run_bench("--gpus", "2", "--dry-run", "--steps", "20", "--warmup", "5", "--repeats", "5", "--envs", "64")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["repeats"] == 5
    assert d["shards"] == [[0, 64], [64, 64]]
    # dry-run timings: rank r, region i takes 1e-3 * (r + 1 + 0.01 i) s and "does" K * envs env-steps
    per_region = [2 * 20 * 64 / (1e-3 * (2 + 0.01 * i)) for i in range(5)]       # slowest rank = rank 1
    assert d["value"] == pytest.approx(sorted(per_region)[2])
    assert d["value_max"] == pytest.approx(max(per_region)) and d["value_min"] == pytest.approx(min(per_region))
    assert d["ms_per_step"] == pytest.approx(sorted(1e-3 * (2 + 0.01 * i) * 1e3 / 20 for i in range(5))[2])
    assert d["stats_total_env_steps"] == 2 * 20 * 64 * 5
    assert "DRY RUN" in d["data"]


def test_single_rank_goes_through_the_same_path():
    p = _run_bench("--gpus", "1", "--dry-run", "--steps", "8", "--warmup", "0", "--repeats", "4", "--envs", "16")
    assert p.returncode == 0, p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["shards"] == [[0, 16]] and d["repeats"] == 4


def test_launcher_env_is_honoured_like_torchrun():
    """Started by an external launcher (RANK / WORLD_SIZE set, no CZ_RDZV_DIR): the ranks derive the same directory."""
    import tempfile
    d = tempfile.mkdtemp(prefix="cz_t_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", CZ_RDZV_DIR=d)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "10",
                                       "--repeats", "3", "--envs", "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert outs[0][0].count("{") >= 1 and outs[1][0].strip() == ""          # only rank 0 prints
    assert json.loads(outs[0][0].splitlines()[-1])["n_gpus"] == 2


def test_launch_local_propagates_failure_and_kills_the_rest(tmp_path):
    from cooking_zoo_amd.distributed import launch_local
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)
    """))
    import time
    t0 = time.monotonic()
    assert launch_local(3, [str(script)], timeout=60.0) == 7
    assert time.monotonic() - t0 < 30


def test_launch_local_deadline(tmp_path):
    from cooking_zoo_amd.distributed import launch_local
    script = tmp_path / "child.py"
    script.write_text("import time; time.sleep(600)\n")
    assert launch_local(2, [str(script)], timeout=1.0) == 124


def test_file_rendezvous_collectives_and_timeout(tmp_path):
    from cooking_zoo_amd.distributed import FileRendezvous, RendezvousTimeout
    world = 3
    got = {}

    def rank_main(r):
        rv = FileRendezvous(str(tmp_path / "rv"), r, world, timeout=20.0)
        got[r] = (rv.all_gather(f"hello {r}".encode()), rv.broadcast(b"id-bytes" if r == 0 else None), rv.broadcast(b"x" if r == 2 else None, src=2))
        rv.barrier()
    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for r in range(world):
        assert got[r] == ([b"hello 0", b"hello 1", b"hello 2"], b"id-bytes", b"x")
    lonely = FileRendezvous(str(tmp_path / "rv2"), 0, 2, timeout=0.3)
    with pytest.raises(RendezvousTimeout):
        lonely.barrier()


def test_launcher_interrupted_kills_its_ranks(tmp_path):
    """SIGTERM to the launching process (a driver's kill, `timeout N python bench.py --gpus 8`): the ranks, which live in
    sessions of their own, are killed by the launcher before it leaves; nothing keeps running (and holding a GPU)."""
    import signal
    import time
    pidfile = tmp_path / "pids"
    child = textwrap.dedent(f"""
        import os, time
        open({str(pidfile)!r} + "." + os.environ["RANK"], "w").write(str(os.getpid()))
        time.sleep(120)
    """)
    script = tmp_path / "child.py"
    script.write_text(child)
    launcher = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {REPO!r})
        from cooking_zoo_amd.distributed import launch_local
        sys.exit(launch_local(2, [{str(script)!r}], timeout=100.0))
    """)
    p = subprocess.Popen([sys.executable, "-c", launcher])
    deadline = time.monotonic() + 60
    while time.monotonic() < deadline and not all(os.path.exists(f"{pidfile}.{r}") and open(f"{pidfile}.{r}").read() for r in range(2)):
        time.sleep(0.05)
    pids = [int(open(f"{pidfile}.{r}").read()) for r in range(2)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.2)
    for pid in pids:
        gone = False
        try:
            os.kill(pid, 0)
            gone = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] == "Z"     # a zombie nobody reaps is gone too
        except (ProcessLookupError, FileNotFoundError):
            gone = True
        assert gone, f"rank process {pid} survived its launcher"


def test_eight_ranks_dry_run_has_config4_and_one_device_per_rank():
    """First-contact insurance for the round-end 8-GPU run (no hardware needed): `bench.py --gpus 8 --dry-run` through
    launch_local - eight fresh rank processes, rendezvous, aggregation - gives one line with n_gpus 8, BASELINE config 4
    (262 144 envs) cut into eight contiguous shards of 32 768 global env ids, and every rank on the device LOCAL_RANK names."""
    p = _run_bench("--gpus", "8", "--dry-run", "--steps", "20", "--warmup", "5", "--repeats", "3", "--envs", "4096")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["device_ids"] == list(range(8))                       # rank r drives device LOCAL_RANK = r, nobody shares one
    assert d["shards"] == [[4096 * r, 4096] for r in range(8)] and d["env_id_bases"] == [4096 * r for r in range(8)]
    c4 = d["config4"]
    assert c4["envs"] == 262144 and "262144 envs" in c4["workload"]
    assert c4["shards"] == [[32768 * r, 32768] for r in range(8)]
    covered = sorted((b, b + n) for b, n in c4["shards"])
    assert covered[0][0] == 0 and covered[-1][1] == 262144 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    # whole-job value: all ranks' env-steps over the slowest rank's time (dry-run: rank r's region takes 1e-2 (r + 1) s)
    assert c4["value"] == pytest.approx(8 * 100 * 32768 / (1e-2 * 8))


def test_gpus_1_equals_the_plain_run():
    """`--gpus 1` (launcher-less single rank) and a plain run are the same code path: same line but for the timings"""
    a = _run_bench("--gpus", "1", "--dry-run", "--steps", "8", "--warmup", "0", "--repeats", "4", "--envs", "16")
    b = _run_bench("--dry-run", "--steps", "8", "--warmup", "0", "--repeats", "4", "--envs", "16")
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    da, db = (json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0]) for p in (a, b))
    assert da == db and da["value"] == pytest.approx(db["value"]) and da["device_ids"] == [0] and "config4" not in da


def test_the_drivers_torchrun_command_dry_run():
    """The driver's own launch form for N > 1 - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` - with `--dry-run`: torchrun's ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT from its
    agent) find each other through the rendezvous directory derived from what they share, rank 0 prints the one line, exit code 0."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, PYTHONPATH=REPO)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CZ_RDZV_DIR", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"],
                       capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["device_ids"] == [0, 1] and d["env_id_bases"] == [0, 4096] and d["steps"] == 20 and d["warmup"] == 5
    assert "config4" in d and d["config4"]["envs"] == 2 * 32768
