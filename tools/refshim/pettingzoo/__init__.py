"""Import-time stand-in for `pettingzoo` (NOT installed in the build container).

Turn-taking / dict plumbing only, paraphrased from the public API; no step arithmetic.
"""
from . import utils  # noqa: F401


class AECEnv:
    def __init__(self):
        pass

    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    def last(self, observe=True):
        agent = self.agent_selection
        obs = self.observe(agent) if observe else None
        return (obs, self._cumulative_rewards[agent], self.terminations[agent],
                self.truncations[agent], self.infos[agent])

    def _was_dead_step(self, action):
        raise RuntimeError("dead step not modelled by the shim")
