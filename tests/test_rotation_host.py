"""Host logic of CookingVecEnv.rotate_layouts that needs no GPU: a producer process that died is reported instead of waited
for, the producer refills the part that was in use when the rotation began, and a group switched behind the rotation's back
is an error (not an assert that -O removes)."""
import queue
import types

import pytest

from cooking_zoo_amd import vec_env


class _DeadProcess:
    exitcode = -9

    def is_alive(self):
        return False


class _EmptyQueue:
    def empty(self):
        return True


def _fake_env(rot):
    stopped = []
    env = types.SimpleNamespace(_steps=0, _rot=rot, _lay_active=0, _lay_groups=2, max_steps=10)
    env.stop_rotation = lambda: (stopped.append(True), setattr(env, "_rot", None))
    env._rotation_pending = vec_env.CookingVecEnv._rotation_pending
    env._stopped = stopped
    return env


@pytest.mark.parametrize("blocking", [True, False])
def test_a_dead_producer_is_reported_not_waited_for(blocking):
    rot = {"groups": 2, "every": 20, "flip_due": 1000, "refill_due": 0, "n_refills": 0, "blocking": blocking, "start": 0,
           "ready": queue.Queue(), "queue": _EmptyQueue(), "process": _DeadProcess()}
    env = _fake_env(rot)
    with pytest.raises(RuntimeError, match="producer process died.*-9"):
        for _ in range(3):                                   # (non-blocking: reported at the second call boundary)
            vec_env.CookingVecEnv._advance(env, 1)
    assert env._stopped


def test_group_switched_behind_the_rotation_is_an_error():
    rot = {"groups": 2, "every": 20, "flip_due": 0, "refill_due": None, "n_refills": 0, "blocking": True, "start": 0,
           "ready": queue.Queue(), "queue": _EmptyQueue(), "process": _DeadProcess()}
    env = _fake_env(rot)
    env._lay_active = 1                                      # somebody called set_layout_group(2, 1) meanwhile
    with pytest.raises(RuntimeError, match="behind the rotation"):
        vec_env.CookingVecEnv._advance(env, 1)
    assert env._stopped


def test_producer_starts_with_the_part_in_use():
    """batch b of the producer refills part (start + b) % groups: with start = 1 the first batch is for part 1"""
    import threading
    from cooking_zoo_amd import soa
    from cooking_zoo_amd.cooking_world.engine import load_level as ll
    meta = ll.load_meta_file("example")
    lv = ll.load_level_file("coop_test")
    dims = soa.Dims(7, 7, 12, 2, 278)
    q, stop = queue.Queue(maxsize=1), threading.Event()
    th = threading.Thread(target=vec_env._produce_layouts, args=(q, stop, [lv], meta, 2, dims.as_tuple(), [(0, 8)], 2, 5, 1), daemon=True)
    th.start()
    firsts = [q.get(timeout=30)[0][0] for _ in range(3)]
    stop.set()
    th.join(timeout=10)
    assert firsts == [4, 0, 4]                               # part 1 (slots 4..7), then part 0, then part 1 again
