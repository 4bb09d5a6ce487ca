#!/usr/bin/env python3
"""Times the UNMODIFIED reference (parallel_env through tools/refshim) in the build container: env-steps/s for
BASELINE config 1 and the config-2 shape, one process.  The reference cannot travel to the GPU box."""
import os, random, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools", "refshim")); sys.path.insert(1, "/root/reference")
from cooking_zoo.environment.cooking_env import parallel_env

def run(num_agents, recipes, steps=4000):
    random.seed(0); np.random.seed(0)
    env = parallel_env(level="coop_test", meta_file="example", num_agents=num_agents, max_steps=400, recipes=recipes,
                       obs_spaces=["feature_vector"] * num_agents, action_scheme="scheme3")
    env.reset()
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    n = 0
    while n < steps:
        if not env.agents:
            env.reset()
        env.step({a: int(rng.integers(5)) for a in env.agents})
        n += 1
    return n / (time.perf_counter() - t0)

print("config 1 (1 env, 1 agent, TomatoLettuceSalad):", round(run(1, ["TomatoLettuceSalad"])), "env-steps/s on 1 core")
print("config-2 shape (1 env, 2 agents):", round(run(2, ["TomatoLettuceSalad", "CarrotBanana"])), "env-steps/s on 1 core")
