"""Drop-in `parallel_env` for the step() path, backed by the HIP kernels (one env instance on one wavefront).

Same constructor kwargs, method names, dict keys and return shapes as the reference's
`cooking_zoo.environment.cooking_env.parallel_env` (cooking_env.py:26-46, 62-64, 178-210, 243-269, 271-288):

    env = parallel_env(level=..., meta_file=..., num_agents=..., max_steps=..., recipes=[...],
                       obs_spaces=["feature_vector", ...], action_scheme="scheme3", reward_scheme=None, ...)
    obs, infos = env.reset()
    obs, rewards, terminations, truncations, infos = env.step({"player_0": a0, "player_1": a1})

`env(...)` returns the agent-iterator (AEC) view of the same environment (AECCookingEnvironment below).

Agent despawn / respawn (rates > 0) is supported: the random bookkeeping stays on the host (same global RNG streams as
the reference), the device step is told who acts.

`obs_spaces` entries other than "feature_vector" are served on the host from the env record: "symbolic" rebuilds the
reference's object view (cooking_world/symbolic.py), "full" is the reference's numeric dict (a constant zero tensor
upstream).  Deliberately not offered: pygame rendering and scheme2 (which raises AttributeError in the reference itself).
For many envs at once use cooking_zoo_amd.vec_env.CookingVecEnv -- this facade is the num_envs = 1 case of it.
"""
from __future__ import annotations

import random

import numpy as np

from cooking_zoo_amd import soa, spaces
from cooking_zoo_amd.cooking_book import recipe_drawer
from cooking_zoo_amd.cooking_world import symbolic
from cooking_zoo_amd.cooking_world.actions import ACTION_SCHEMES
from cooking_zoo_amd.cooking_world.engine import load_level as _ll
from cooking_zoo_amd.vec_env import CookingVecEnv

FPS = 20
COLORS = ['blue', 'magenta', 'yellow', 'green']


class WorldView:
    """Read-only snapshot of the world for callers that peek at `env.unwrapped.world` (demos do)."""

    def __init__(self, dims, record):
        self.width, self.height = dims.W, dims.H
        self.agents = []
        for a in range(dims.A):
            x, y, o, h = soa.unpack_agent(record[soa.AGENT_WORD0 + a])
            self.agents.append({"location": (x, y), "orientation": o, "holding_slot": h, "name": f"agent-{a + 1}",
                                "color": COLORS[a]})
        self.cells = soa.record_cells(dims, record).reshape(dims.H, dims.W).copy()
        self.dynamic_objects = []
        for s in range(dims.D):
            x, y, c, f = soa.unpack_dyn0(record[dims.dyn0_word0 + s])
            cont, seq = soa.unpack_dyn1(record[dims.dyn1_word0 + s])
            if f & soa.DYN_ALIVE:
                self.dynamic_objects.append({"class": soa.DYNAMIC_CLASSES[c], "location": (x, y), "slot": s,
                                             "chopped": bool(f & soa.DYN_CHOPPED), "mashed": bool(f & soa.DYN_MASHED),
                                             "free": bool(f & soa.DYN_FREE), "inside_plate_slot": cont})


class CookingEnvironment:
    """Parallel (dict-in / dict-out) CookingZoo environment; what `parallel_env(...)` returns."""

    metadata = {"render_modes": ["human", "rgb_array"], "name": "cookingzoo_v1", "is_parallelizable": True,
                "render_fps": FPS}

    def __init__(self, level, meta_file, num_agents, max_steps, recipes, agent_visualization=None, obs_spaces=None,
                 end_condition_all_dishes=False, allowed_objects=None, action_scheme="scheme1", render=False,
                 reward_scheme=None, agent_respawn_rate=0.0, grace_period=20, agent_despawn_rate=0.0, device_id=0):
        obs_spaces = obs_spaces or ["feature_vector"]
        allowed = ["symbolic", "full", "feature_vector"]
        assert len(set(obs_spaces + allowed)) == 3, f"Selected invalid obs spaces. Allowed {allowed}"
        # one entry per agent (cooking_env.py:272); agents past the end of the list get the feature vector
        self._obs_kind = [obs_spaces[i] if i < len(obs_spaces) else "feature_vector" for i in range(num_agents)]
        if action_scheme == "scheme2":
            raise AttributeError("'CookingWorld' object has no attribute 'perform_agent_action' (scheme2 is unusable in "
                                 "the reference: action_scheme2.py:15)")
        # agent despawn / respawn (cooking_world.py:267-290): the bookkeeping -- two draws from numpy's global stream per
        # agent and step, respawn cells from Python's global `random` -- stays on the host, exactly where the reference
        # does it; the device step only learns which agents act (action -1 = not in the active list)
        self.agent_respawn_rate, self.agent_despawn_rate, self.grace_period = agent_respawn_rate, agent_despawn_rate, grace_period
        self._spawning = bool(agent_respawn_rate or agent_despawn_rate)
        self.level, self.meta_file, self.max_steps = level, meta_file, max_steps
        self.action_scheme = action_scheme
        self.action_scheme_class = ACTION_SCHEMES[action_scheme]
        self.obs_spaces = obs_spaces
        self.possible_agents = ["player_" + str(r) for r in range(num_agents)]
        self.agents = self.possible_agents[:]
        self.agent_visualization = agent_visualization or ["human"] * num_agents
        self.reward_scheme = reward_scheme or {"recipe_reward": 20, "max_time_penalty": -5, "recipe_penalty": -40,
                                               "recipe_node_reward": 0}
        self.recipe_names = list(recipes)
        self.end_condition_all_dishes = end_condition_all_dishes
        self.render_flag = render
        self.termination_info = ""
        self.t = 0

        meta = _ll.load_meta_file(meta_file)
        assert num_agents <= meta["Agent"], "Too many agents for this level"
        level_object = _ll.load_level_file(level)
        # the reference's constructor loads the level once (cooking_env.py:109): same draws from `random`
        first = _ll.instantiate(level_object, meta, num_agents, random)
        self._vec = CookingVecEnv(1, level, meta_file, num_agents, max_steps, recipes,
                                  end_condition_all_dishes=end_condition_all_dishes, action_scheme=action_scheme,
                                  reward_scheme=self.reward_scheme, layouts=[first], auto_reset=False,
                                  device_id=device_id, max_dyn=max(_ll.level_max_dyn(level_object), 1))
        self._level_object, self._meta = level_object, meta
        self._cached_layout = first
        # parse_agents appends one (x candidates, y candidates) pair per created agent (parsing.py:145)
        self._spawn_cells = [(spec["X_POSITION"], spec["Y_POSITION"]) for spec in level_object["AGENTS"]
                             for _ in range(spec["MAX_COUNT"])][:num_agents]
        self._active = [True] * num_agents
        self._status_changed = [False] * num_agents
        self._grace = [grace_period] * num_agents
        self.num_goals = recipe_drawer.NUM_GOALS if recipe_drawer.RECIPE_STORE else recipe_drawer.DEFAULT_NUM_GOALS
        self.recipe_graphs = [self._vec.book[r]() for r in recipes]
        self.loaded_recipes = list(self._vec.book.keys())
        # goal vectors are indexed by agent position, not by recipe (reference defect kept: cooking_env.py:159-161)
        eye = np.eye(len(self.loaded_recipes))
        self.goal_vectors = {a: eye[i] for i, a in zip(range(len(self.loaded_recipes)), self.possible_agents)}
        F = self._vec.F
        self.feature_vector_representation_length = F
        self.feature_obs_space = spaces.Box(low=-1, high=1, shape=(F,))
        dims = self._vec.dims
        self.graph_representation_length = symbolic.GRAPH_REPRESENTATION_LENGTH                  # cooking_env.py:110
        numeric_obs_space = {"feature_vector": spaces.Box(low=0, high=10, shape=(dims.W, dims.H, self.graph_representation_length),
                                                          dtype=np.int32),
                             "agent_location": spaces.Box(low=0, high=max(dims.W, dims.H), shape=(2,)),
                             "goal_vector": spaces.MultiBinary(self.num_goals)}
        by_kind = {"full": numeric_obs_space, "feature_vector": self.feature_obs_space, "symbolic": {}}   # cooking_env.py:126-128
        self.observation_spaces = {a: by_kind[k] for a, k in zip(self.possible_agents, self._obs_kind)}
        self.current_tensor_observation = np.zeros((dims.W, dims.H, self.graph_representation_length))   # never written upstream
        self.action_spaces = {a: spaces.Discrete(len(self.action_scheme_class.ACTIONS)) for a in self.possible_agents}
        self.rewards = {a: 0 for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}
        self.infos = {a: {} for a in self.agents}
        self._needs_reset = True
        self.render_mode = "human"
        self.np_random = None

    # ------------------------------------------------------------------ PettingZoo parallel API
    @property
    def unwrapped(self):
        return self

    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    def observation_space(self, agent):
        return self.observation_spaces[agent]

    def action_space(self, agent):
        return self.action_spaces[agent]

    def state(self):
        return None

    def seed(self, seed=None):
        self.np_random = np.random.default_rng(seed)

    def reset(self, seed=None, return_info=False, options=None):
        """cooking_env.py:178-210.  `seed` is ignored, as in the reference (level draws come from Python's global
        `random`); options={"full_reset": False} re-uses the most recently drawn layout."""
        options = options or {"full_reset": True}
        self.t = 0
        self.termination_info = ""
        self.agents = self.possible_agents[:]
        if options["full_reset"]:
            self._cached_layout = _ll.instantiate(self._level_object, self._meta, len(self.possible_agents), random)
            self._vec.set_layouts([self._cached_layout])
        obs = self._vec.reset(layout_ids=[0])
        self._refresh_marks()
        n = len(self.possible_agents)
        self._active, self._status_changed = [True] * n, [False] * n            # load_level.py:67-68
        if options["full_reset"]:
            self._grace = [self.grace_period] * n                                # a new CookingWorld: parsing.py:142
        # (a world that is re-used keeps the countdowns of the finished episode: parsing.py appends behind them)
        self.rewards = {a: 0 for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}
        self.infos = {a: {} for a in self.agents}
        self._needs_reset = False
        return self._observations(obs, range(n)), dict(self.infos)

    def step(self, actions):
        """One accumulated_step (cooking_env.py:243-269) + observe for every agent (cooking_env.py:271-288).
        `actions` has one entry per agent in `self.agents` (the active ones)."""
        if self._needs_reset or not self.agents:
            raise RuntimeError("the episode is over (or reset() was never called): call reset() before step()")
        A = len(self.possible_agents)
        active_start = self._active[:]
        acts = [int(actions[a]) if active_start[i] else -1 for i, a in enumerate(self.possible_agents)]
        n = self.action_spaces[self.possible_agents[0]].n
        if any(active_start[i] and not 0 <= acts[i] < n for i in range(A)):
            raise ValueError(f"actions must be in [0, {n}) for {self.action_scheme}")
        obs, rew, term, trunc = self._vec.step(np.asarray([acts], dtype=np.int32))
        self.t += 1
        self._set_marks(int(self._vec.last_marks()[0]))
        self._status_changed = [False] * A                                       # cooking_world.py:106
        if self._spawning and self._handle_agent_spawn():
            obs = self._vec.observe()                                             # somebody was put on a new cell
        relevant = [i for i in range(A) if self._active[i] or self._status_changed[i]]   # cooking_world.py:292-293
        # compute_truncated (cooking_env.py:333-350)
        out_of_time = bool(trunc[0, 0])
        if out_of_time:
            self.termination_info = f"Terminating because {self.max_steps} timesteps passed"
            # (the reference sizes this list with the CURRENT agent count and then indexes past its end when somebody
            # is despawned at this moment -- IndexError at cooking_env.py:257; the build ends the episode instead)
            self._active = [False] * A
            self._status_changed = [i in relevant for i in range(A)]
        truncated = {i: out_of_time or (self._status_changed[i] and not self._active[i]) for i in relevant}
        done = bool(term[0, 0])
        info = {"t": self.t, "termination_info": self.termination_info}
        self.rewards, self.terminations, self.truncations, self.infos = {}, {}, {}, {}
        observations = self._observations(obs, relevant)
        for k, i in enumerate(relevant):
            a = self.possible_agents[i]
            # the reference scatters the per-RECIPE rewards by position in the relevant list (cooking_env.py:255-261):
            # with everybody present that is the agent's own recipe, otherwise the k-th one
            self.rewards[a] = np.float64(rew[0, k])
            self.terminations[a] = done
            self.truncations[a] = bool(truncated[i])
            self.infos[a] = {"goal_vector": self.goal_vectors[a], **info,
                             "recipe_done": self.recipe_graphs[i].completed(),
                             "action": acts[i] if active_start[i] else 0, "task": self.recipe_names[i]}
        if out_of_time or done:
            # the reference empties the agent list after the final step (truncation: cooking_env.py:336-339,267-268;
            # termination: its wrapper loop cannot make progress, SURVEY A.12(5) -- the build ends the episode cleanly)
            self.agents = []
            self._needs_reset = True
        else:
            # whoever was despawned in this step is reported once (truncated) and then leaves the agent list
            # (cooking_env.py:216-224); whoever was respawned joins it
            self.agents = [a for i, a in enumerate(self.possible_agents) if self._active[i]]
        return observations, dict(self.rewards), dict(self.terminations), dict(self.truncations), dict(self.infos)

    def _handle_agent_spawn(self):
        """cooking_world.py:267-290 on the host: same draws from the same global streams as the reference.  Returns
        True if an agent was moved (the caller re-encodes the observation)."""
        A = len(self.possible_agents)
        rec = None
        moved = False
        for i in range(A):
            if self._grace[i] > 0:
                self._grace[i] -= 1
                continue
            if self._active.count(True) > 1 and self._active[i] and np.random.random() < self.agent_despawn_rate:
                rec = self._vec.get_state()[0] if rec is None else rec
                if soa.unpack_agent(rec[soa.AGENT_WORD0 + i])[3] >= 0:
                    continue                                                     # holding something: stays (:280-281)
                self._active[i] = False
                self._status_changed[i] = True
            elif not self._active[i] and np.random.random() < self.agent_respawn_rate:
                rec = self._vec.get_state()[0] if rec is None else rec
                self._active[i] = True
                self._status_changed[i] = True
                self._grace[i] = self.grace_period
                x, y = self._generate_location(rec, *self._spawn_cells[i])
                _, _, o, h = soa.unpack_agent(rec[soa.AGENT_WORD0 + i])
                rec[soa.AGENT_WORD0 + i] = soa.pack_agent(x, y, o, h)         # location only (cooking_world.py:290)
                self._vec.set_state(rec[None, :])
                moved = True
        return moved

    def _generate_location(self, rec, x_positions, y_positions):
        """parsing.py:154-167: a Floor cell nobody (active or not) stands on, drawn with `random.sample`."""
        dims = self._vec.dims
        cells = soa.record_cells(dims, rec)
        agents = [soa.unpack_agent(rec[soa.AGENT_WORD0 + a])[:2] for a in range(dims.A)]
        time_out = 0
        while True:
            x = random.sample(x_positions, 1)[0]
            y = random.sample(y_positions, 1)[0]
            if x < 0 or y < 0 or x > dims.W or y > dims.H:
                raise ValueError(f"Position {x} {y} is out of bounds set by the level layout!")
            on_grid = x < dims.W and y < dims.H
            if on_grid and (x, y) not in agents and (cells[y * dims.W + x] & soa.CELL_TYPE_MASK) == soa.FLOOR:
                return int(x), int(y)
            time_out += 1
            if time_out > 1000:
                raise ValueError(f"Can't find valid position in {time_out} steps")

    def observe(self, agent):
        """cooking_env.py:271-288: the observation of one agent in the mode its `obs_spaces` entry names."""
        i = self.possible_agents.index(agent)
        obs = self._vec.observe() if self._obs_kind[i] == "feature_vector" else None
        return self._observations(obs, [i])[agent]

    def _observations(self, obs, agent_indices):
        """{player_i: observation} for the given agents.  "feature_vector": the device-encoded float64 vector;
        "symbolic": a fresh object view of the world rebuilt from the env record (cooking_world/symbolic.py; the
        reference deep-copies its object graph, so every agent gets objects of its own); "full": the reference's numeric
        dict, whose tensor is constant zeros upstream (cooking_env.py:149,273-278)."""
        out = {}
        record = None
        for i in agent_indices:
            kind, a = self._obs_kind[i], self.possible_agents[i]
            if kind == "feature_vector":
                out[a] = obs[0, i].copy()
                continue
            if record is None:
                record = self._vec.get_state()[0]
            if kind == "symbolic":
                out[a] = symbolic.materialize(self._vec.dims, self._cached_layout, record)
            else:
                x, y, _, _ = soa.unpack_agent(record[soa.AGENT_WORD0 + i])
                out[a] = {"feature_vector": self.current_tensor_observation,
                          "agent_location": np.asarray((x, y), np.int32),
                          "goal_vector": self.recipe_graphs[i].goals_completed(self.num_goals)}
        return out

    def _refresh_marks(self):
        rec = self._vec.get_state()[0]
        self._set_marks(int(rec[soa.W_MARKS]) | (int(rec[soa.W_MARKS_HI]) << 32))

    def _set_marks(self, marks):
        for r, g in enumerate(self.recipe_graphs):
            g.set_marks(self._vec.marks_of(marks, r))

    @property
    def world(self):
        return WorldView(self._vec.dims, self._vec.get_state()[0])

    def render(self, **kwargs):
        raise NotImplementedError("rendering (pygame GraphicPipeline) is outside the accelerated step path")

    def close(self):
        self._vec.close()


class AECCookingEnvironment:
    """Agent-iterator (AEC) view of the same environment: what the reference's raw `CookingEnvironment` offers through
    `step(action)` / `last()` / `agent_iter()` (cooking_env.py:215-241; gym id cookingZooEnv-v0).

    Agents act in turn; the world advances once the last agent of the round has chosen (one batched device step).
    Bookkeeping follows the reference call by call (tests/golden/aec_traces.json.gz), including two things a PettingZoo
    user would not expect: rewards are zeroed for everybody on every sub-step, and after each call the cumulative
    reward that is cleared is the LAST agent's (the loop variable of cooking_env.py:229 shadows the acting agent), so
    every other agent's `last()` reward keeps growing over the episode.
    """

    metadata = CookingEnvironment.metadata

    def __init__(self, *args, **kwargs):
        self._core = CookingEnvironment(*args, **kwargs)
        self.possible_agents = self._core.possible_agents[:]
        self.agents = self.possible_agents[:]
        self.observation_spaces, self.action_spaces = self._core.observation_spaces, self._core.action_spaces
        self.agent_selection = None
        self._pending = []
        self._turn = 0
        self._status_changed = False
        self.rewards, self._cumulative_rewards, self.terminations, self.truncations, self.infos = {}, {}, {}, {}, {}
        self._obs = {}

    # -- plumbing shared with the parallel facade
    unwrapped = property(lambda self: self)
    num_agents = property(lambda self: len(self.agents))
    max_num_agents = property(lambda self: len(self.possible_agents))
    world = property(lambda self: self._core.world)
    t = property(lambda self: self._core.t)
    recipe_graphs = property(lambda self: self._core.recipe_graphs)

    def observation_space(self, agent):
        return self.observation_spaces[agent]

    def action_space(self, agent):
        return self.action_spaces[agent]

    def reset(self, seed=None, return_info=False, options=None):
        """cooking_env.py:178-210"""
        self._obs, _ = self._core.reset(seed=seed, options=options)
        self.agents = self.possible_agents[:]
        self._turn = 0
        self.agent_selection = self.agents[0]
        self._pending = []
        self._status_changed = False
        self.rewards = {a: 0 for a in self.agents}
        self._cumulative_rewards = {a: 0 for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}
        self.infos = {a: {} for a in self.agents}

    def observe(self, agent):
        """The feature vector of the current world (it only changes when a round completes)."""
        return self._obs[agent].copy()

    def last(self, observe=True):
        a = self.agent_selection
        return (self.observe(a) if observe else None, self._cumulative_rewards[a], self.terminations[a],
                self.truncations[a], self.infos[a])

    def agent_iter(self, max_iter=2 ** 63):
        n = 0
        while self.agents and n < max_iter:
            yield self.agent_selection
            n += 1

    def step(self, action):
        """cooking_env.py:215-241"""
        if self.agent_selection is None:
            raise RuntimeError("reset() must be called before step()")
        if action is None:
            # a finished agent is acknowledged with None (cooking_env.py:216-224): if somebody's spawn status changed in
            # the last world step -- a despawn, a respawn, or the time limit (which deactivates everybody) -- the agent
            # list becomes the active agents and the turn goes to the first of them; after a recipe-completion
            # termination the reference changes nothing and neither does this.
            if self._status_changed:
                self.agents = [a for a in self.possible_agents if a in self._core.agents]
                self._turn = 0
                if self.agents:
                    self.agent_selection = self.agents[0]
            return
        me = self.agent_selection
        if self.terminations[me] or self.truncations[me]:
            raise ValueError("when an agent is terminated or truncated, the only valid action is None")
        self._pending.append(int(action))
        for a in self.agents:
            self.rewards[a] = 0
        tail = self.agents[-1]                       # whose cumulative reward the reference clears (see class docstring)
        if me == self.agents[-1]:
            actions, self._pending = self._pending, []
            obs, rew, term, trunc, infos = self._core.step(dict(zip(self.agents, actions)))
            self._obs.update(obs)
            self.rewards, self.terminations, self.truncations, self.infos = rew, term, trunc, infos
            for a in rew:
                self._cumulative_rewards[a] += rew[a]
            # the agents this step reports on: the active ones and whoever was despawned by it (cooking_env.py:267-268)
            self.agents = [a for a in self.possible_agents if a in rew]
            self._status_changed = any(self._core._status_changed)
            self._turn = 0                            # a fresh selector: the next round starts with the first agent
            for a in self.agents:
                if term[a] or trunc[a]:
                    self.agent_selection = a
                    self._cumulative_rewards[tail] = 0
                    return
            self.agent_selection = self.agents[0]
        else:
            self.agent_selection = self.agents[self._turn + 1] if self._turn + 1 < len(self.agents) else self.agents[0]
            self._turn += 1
        self._cumulative_rewards[tail] = 0

    def render(self, **kwargs):
        return self._core.render(**kwargs)

    def close(self):
        self._core.close()


def env(level, meta_file, num_agents, max_steps, recipes, agent_visualization=None, obs_spaces=None,
        end_condition_all_dishes=False, action_scheme="scheme1", render=False, reward_scheme=None,
        agent_respawn_rate=0.0, grace_period=20, agent_despawn_rate=0.0, device_id=0):
    """The AEC environment (cooking_env.py:26-43, without PettingZoo's stdout-capture / order-enforcing wrappers)."""
    return AECCookingEnvironment(level, meta_file, num_agents, max_steps, recipes, agent_visualization, obs_spaces,
                                 end_condition_all_dishes=end_condition_all_dishes, action_scheme=action_scheme,
                                 render=render, reward_scheme=reward_scheme, agent_respawn_rate=agent_respawn_rate,
                                 grace_period=grace_period, agent_despawn_rate=agent_despawn_rate, device_id=device_id)


def parallel_env(level, meta_file, num_agents, max_steps, recipes, agent_visualization=None, obs_spaces=None,
                 end_condition_all_dishes=False, action_scheme="scheme1", render=False, reward_scheme=None,
                 agent_respawn_rate=0.0, grace_period=20, agent_despawn_rate=0.0, device_id=0):
    return CookingEnvironment(level, meta_file, num_agents, max_steps, recipes, agent_visualization, obs_spaces,
                              end_condition_all_dishes=end_condition_all_dishes, action_scheme=action_scheme,
                              render=render, reward_scheme=reward_scheme, agent_respawn_rate=agent_respawn_rate,
                              grace_period=grace_period, agent_despawn_rate=agent_despawn_rate, device_id=device_id)
