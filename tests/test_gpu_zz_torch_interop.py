"""GPU + PyTorch in one process: device-pointer steps on torch tensors, ordered on torch's stream.

PyTorch's ROCm wheels bundle their own libamdhip64.so.7; the step library links the system one under the same SONAME.
Whichever is loaded first serves both, and only the order "torch first" works.  Inside a full `pytest tests -m gpu` run
the step library is already loaded when this file's turn comes (importing torch then would abort the process at exit),
so the test body always runs in a FRESH child interpreter that imports torch first; the parent asserts on its exit code.
The only skip left is "torch is not installed"."""
import importlib.machinery
import os
import subprocess
import sys

import numpy as np
import pytest

CHILD_FLAG = "CZ_TORCH_INTEROP_CHILD"

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_torch_tensors_on_the_callers_stream():
    """Device-pointer steps driven from PyTorch: tensors as buffers, the env ordered on torch's current stream
    (cz_set_stream), no host synchronisation between the producer of the actions, the step and the consumer."""
    # (PathFinder: sys.meta_path may hold cooking_zoo_amd's guard against importing torch AFTER the step library)
    if "torch" not in sys.modules and importlib.machinery.PathFinder.find_spec("torch") is None:
        pytest.skip("torch is not installed")
    if not os.environ.get(CHILD_FLAG):
        env = dict(os.environ)
        env[CHILD_FLAG] = "1"
        p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, f"child pytest failed (rc {p.returncode}):\n{p.stdout[-4000:]}\n{p.stderr[-2000:]}"
        assert "1 passed" in p.stdout, p.stdout[-2000:]
        return
    import torch                                  # first: its bundled HIP runtime then serves the step library too
    torch.cuda.init()
    from cooking_zoo_amd.vec_env import CookingVecEnv
    n, A, T = 512, 2, 40
    kw = dict(action_scheme="scheme3", num_layouts=8, auto_reset=True)
    env = CookingVecEnv(n, "coop_test", "example", A, 25, ["TomatoLettuceSalad", "CarrotBanana"], **kw)
    ref = CookingVecEnv(n, "coop_test", "example", A, 25, ["TomatoLettuceSalad", "CarrotBanana"], **kw)
    env.reset(return_obs=False)
    ref.reset(return_obs=False)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    obs = torch.empty((n, A, env.F), dtype=torch.float64, device=dev)
    rew = torch.empty((n, A), dtype=torch.float64, device=dev)
    term = torch.empty((n, A), dtype=torch.uint8, device=dev)
    trunc = torch.empty((n, A), dtype=torch.uint8, device=dev)
    ret = torch.zeros((n, A), dtype=torch.float64, device=dev)
    acts_log, obs_sum = [], []
    with torch.cuda.stream(side):
        env.set_stream(torch.cuda.current_stream())
        for t in range(T):
            acts = torch.randint(0, 5, (n, A), dtype=torch.int32, device=dev, generator=gen)      # "policy" on the GPU
            env.step_device(acts, obs, rew, term, trunc)
            ret += rew                                                                           # consumer on the GPU
            obs_sum.append(obs.sum())
            acts_log.append(acts)
        side.synchronize()
    env.set_stream(None)
    want_ret = np.zeros((n, A))
    for t in range(T):
        o, r, te, tr = ref.step(acts_log[t].cpu().numpy())
        want_ret += r
        assert float(obs_sum[t].cpu()) == float(torch.from_numpy(o).sum()) or np.isclose(float(obs_sum[t].cpu()), o.sum(), rtol=1e-12)
    assert np.array_equal(bits(obs.cpu().numpy()), bits(o)) and np.array_equal(term.cpu().numpy(), te)
    assert np.array_equal(bits(ret.cpu().numpy()), bits(want_ret))
    assert np.array_equal(env.get_state(), ref.get_state())
    # the compact observation into a torch tensor, decoded with the table on the GPU: the float64 observation bit for bit
    codes = torch.empty((n, A, env.codes_pitch), dtype=torch.uint8, device=dev)
    table = torch.from_numpy(env.obs_table()).to(dev)
    env.step_device_compact(acts_log[0], codes, rew, term, trunc, obs)
    torch.cuda.synchronize()
    decoded = table[codes[..., :env.F].long()]
    assert torch.equal(decoded.view(torch.int64), obs.view(torch.int64))
    o, r, te, tr = ref.step(acts_log[0].cpu().numpy())
    assert np.array_equal(bits(decoded.cpu().numpy()), bits(o))
    with pytest.raises(ValueError):
        env.step_device(acts_log[0].t(), obs, rew, term, trunc)                                  # not contiguous
    # ---- torch.cuda.graph: [a torch policy -> cz_step_device] captured once (torch's default capture mode is "global", the
    # strictest), replayed 200 times; the closed loop must do what the same loop does eagerly on the twin env
    def policy(o):
        return torch.remainder(o.view(torch.int64)[:, :, :8].sum(-1), 5).to(torch.int32)
    cap = torch.cuda.Stream(device=dev)
    act = torch.zeros((n, A), dtype=torch.int32, device=dev)
    env.sync()
    env.observe_device(obs)
    torch.cuda.synchronize()
    env.set_stream(cap)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        for _ in range(4):
            act.copy_(policy(obs))
            env.step_device(act, obs, rew, term, trunc)
    R = 200
    for _ in range(R):
        graph.replay()
    torch.cuda.synchronize()
    o = ref.observe()
    for _ in range(4 * R):
        a = np.remainder(o.view(np.int64)[:, :, :8].sum(-1), 5).astype(np.int32)
        o, r, te, tr = ref.step(a)
    assert np.array_equal(act.cpu().numpy(), a)
    assert np.array_equal(bits(obs.cpu().numpy()), bits(o)) and np.array_equal(bits(rew.cpu().numpy()), bits(r))
    assert np.array_equal(env.get_state(), ref.get_state())
    env.set_stream(None)
    env.close()
    ref.close()
