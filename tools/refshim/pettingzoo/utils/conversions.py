from collections import defaultdict


class aec_to_parallel_wrapper:
    """Paraphrase of pettingzoo's AEC->parallel adapter (plumbing only)."""

    def __init__(self, aec_env):
        self.aec_env = aec_env
        self.possible_agents = aec_env.possible_agents
        self.metadata = getattr(aec_env, "metadata", {})
        self.agents = []

    @property
    def unwrapped(self):
        return self.aec_env

    def observation_space(self, agent):
        return self.aec_env.observation_space(agent)

    def action_space(self, agent):
        return self.aec_env.action_space(agent)

    def reset(self, seed=None, options=None):
        self.aec_env.reset(seed=seed, options=options)
        self.agents = self.aec_env.agents[:]
        observations = {a: self.aec_env.observe(a) for a in self.aec_env.agents
                        if not (self.aec_env.terminations[a] or self.aec_env.truncations[a])}
        infos = dict(**self.aec_env.infos)
        return observations, infos

    def step(self, actions):
        rewards = defaultdict(int)
        for agent in list(self.aec_env.agents):
            assert agent == self.aec_env.agent_selection
            self.aec_env.step(actions[agent])
            for a in self.aec_env.agents:
                rewards[a] += self.aec_env.rewards[a]
        terminations = dict(**self.aec_env.terminations)
        truncations = dict(**self.aec_env.truncations)
        infos = dict(**self.aec_env.infos)
        observations = {a: self.aec_env.observe(a) for a in self.aec_env.agents}
        guard = 0
        while self.aec_env.agents and (self.aec_env.terminations[self.aec_env.agent_selection]
                                       or self.aec_env.truncations[self.aec_env.agent_selection]):
            self.aec_env.step(None)
            guard += 1
            if guard > 16:      # termination cannot make progress in the reference
                self.aec_env.agents = []
                break
        self.agents = self.aec_env.agents
        return observations, dict(rewards), terminations, truncations, infos

    def render(self):
        return self.aec_env.render()

    def close(self):
        return self.aec_env.close()


def parallel_wrapper_fn(env_fn):
    def par_fn(**kwargs):
        return aec_to_parallel_wrapper(env_fn(**kwargs))
    return par_fn
