"""`cookingEnvMA-v1`: the list-in / list-out gym-style environment for several agents (reference:
environment/multi_agent_gym.py:5-35).  `step([a0, a1, ...])` returns five lists in agent order, `reset()` two; an agent
that has left the episode (despawned) is simply absent from the lists, as in the reference."""
from cooking_zoo_amd.environment.gym_adapter import ZooAdapter


class GymCookingEnvironment(ZooAdapter):
    metadata = {"render.modes": ["human"], "name": "multi_agent_cooking_zoo"}

    def __init__(self, level, meta_file, num_agents, max_steps, recipes, *positional, **options):
        super().__init__(num_agents, level, meta_file, max_steps, recipes, *positional, **options)
