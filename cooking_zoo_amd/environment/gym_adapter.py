"""One adapter from the dict-keyed parallel environment to the two flat gym-style call shapes of the reference.

The reference ships two 30-line classes for this (`environment/environment.py:5-32` for one agent -- scalar in, scalar
out -- and `environment/multi_agent_gym.py:5-35` for several -- list in, list out).  Both are the same operation: put
the caller's positional actions under the `player_i` keys, step the parallel environment, and take the five result
dicts apart again in agent order.  Here that operation exists once (`ZooAdapter._exchange`); the two public classes
only say whether one value or a list crosses the boundary.  Contract kept from the reference: the class name
`GymCookingEnvironment` in both modules, the constructor keywords, the attributes `zoo_env`, `observation_space`,
`action_space`, `metadata`, and `step` / `reset` / `render` / `close`.
"""
from cooking_zoo_amd.environment import cooking_env

_FORWARDED = ("agent_visualization", "obs_spaces", "end_condition_all_dishes", "action_scheme", "render", "reward_scheme")


class ZooAdapter:
    #: one value per call (single-agent form) instead of one list entry per agent
    scalar = False
    metadata = {"render.modes": ["human"], "name": "cooking_zoo_adapter"}

    def __init__(self, num_agents, level, meta_file, max_steps, recipes, *positional, **options):
        # the reference's constructors also take these six positionally, in this order
        if len(positional) > len(_FORWARDED):
            raise TypeError(f"at most {len(_FORWARDED)} optional positional arguments: {_FORWARDED}")
        for name, value in zip(_FORWARDED, positional):
            if name in options:
                raise TypeError(f"got multiple values for argument '{name}'")
            options[name] = value
        unknown = set(options) - set(_FORWARDED)
        if unknown:
            raise TypeError(f"unexpected keyword argument(s): {sorted(unknown)}")
        options.setdefault("action_scheme", "scheme1")
        self.zoo_env = cooking_env.parallel_env(level=level, meta_file=meta_file, num_agents=num_agents, max_steps=max_steps,
                                                recipes=recipes, **options)
        self._names = tuple(self.zoo_env.possible_agents)
        lead = self._names[0]
        self.observation_space = self.zoo_env.observation_space(lead)
        self.action_space = self.zoo_env.action_space(lead)

    # -- the one operation
    def _take_apart(self, *dicts):
        """-> one tuple per dict, holding the values of the agents that are present, in agent order"""
        present = [a for a in self._names if a in dicts[0]]
        columns = tuple([d[a] for a in present] for d in dicts)
        return tuple(c[0] for c in columns) if self.scalar else columns

    def _exchange(self, actions):
        acts = (actions,) if self.scalar else tuple(actions)
        return self._take_apart(*self.zoo_env.step(dict(zip(self._names, acts))))

    # -- gym surface
    def step(self, action):
        return self._exchange(action)

    def reset(self, **kwargs):
        return self._take_apart(*self.zoo_env.reset())

    def render(self, mode="human"):
        self.zoo_env.render()

    def close(self):
        self.zoo_env.close()
