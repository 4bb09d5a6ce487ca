#!/usr/bin/env python3
"""Times the UNMODIFIED reference (parallel_env through tools/refshim) in the build container: env-steps/s for
BASELINE config 1 and the config-2 shape, one process and one process per core.  The reference cannot
travel to the GPU box, so the figures are committed as data (profiles/rNN/reference_python_timing.json) and bench.py
carries them in `cpu_baseline.reference_python` next to the C port it times on the GPU box's own cores.

    python tools/time_reference.py [out.json]
"""
import json, multiprocessing as mp, os, platform, random, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools", "refshim")); sys.path.insert(1, "/root/reference")


def run(job):
    num_agents, recipes, steps, seed = job
    from cooking_zoo.environment.cooking_env import parallel_env
    random.seed(seed); np.random.seed(seed)
    env = parallel_env(level="coop_test", meta_file="example", num_agents=num_agents, max_steps=400, recipes=recipes,
                       obs_spaces=["feature_vector"] * num_agents, action_scheme="scheme3")
    env.reset()
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    n = 0
    while n < steps:
        if not env.agents:
            env.reset()
        env.step({a: int(rng.integers(5)) for a in env.agents})
        n += 1
    return n, time.perf_counter() - t0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


if __name__ == "__main__":
    cores = len(os.sched_getaffinity(0))
    steps = 4000
    out = {"cpu": cpu_model(), "cores_available": cores, "python": platform.python_version(), "numpy": np.__version__,
           "method": "unmodified reference parallel_env (tools/refshim stand-ins for pettingzoo/gymnasium/pygame), scheme3, "
                     "coop_test, meta example, max_steps=400, uniform random actions, reset when the agent list empties, "
                     f"{steps} steps per process; aggregate = sum of steps / slowest worker", "cases": {}}
    for name, na, rec in (("config1_1agent", 1, ["TomatoLettuceSalad"]), ("config2_shape_2agents", 2, ["TomatoLettuceSalad", "CarrotBanana"])):
        n, dt = run((na, rec, steps, 0))
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(run, [(na, rec, steps, s) for s in range(cores)])
        out["cases"][name] = {"env_steps_per_s_1_process": round(n / dt, 1),
                              f"env_steps_per_s_{cores}_processes": round(sum(r[0] for r in res) / max(r[1] for r in res), 1)}
        print(name, out["cases"][name])
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
