"""`cookingEnv-v1`: the single-agent gym-style environment (reference: environment/environment.py:5-32; its constructor
has no `num_agents`).  The work is in gym_adapter.ZooAdapter; this class fixes the agent count at one and the scalar
call shape: `step(action) -> (obs, reward, terminated, truncated, info)`, `reset() -> (obs, info)`."""
from cooking_zoo_amd.environment.gym_adapter import ZooAdapter


class GymCookingEnvironment(ZooAdapter):
    scalar = True
    metadata = {"render.modes": ["human"], "name": "cooking_zoo"}

    def __init__(self, level, meta_file, max_steps, recipes, *positional, **options):
        super().__init__(1, level, meta_file, max_steps, recipes, *positional, **options)
