#!/usr/bin/env python3
"""Generate golden parity fixtures by running the UNMODIFIED reference in this container.

Runs only where /root/reference exists (the build container).  It imports the reference
(`cooking_zoo.environment.cooking_env.CookingEnvironment`) through the import-time stand-ins in
tools/refshim (gymnasium / pettingzoo / pygame are not installed; the stand-ins are plumbing only,
SURVEY.md Appendix C), drives `reset()` / `accumulated_step()` / `observe()` and dumps, per step,

  * the full world state converted to the flat record of cooking_zoo_amd/soa.py,
  * the float64 feature-vector observation of every agent,
  * rewards (float64), terminations, truncations,

as compressed .npz files under tests/golden/.  The files are data only (inputs + expected outputs);
nothing of the reference's source travels.  While converting, the generator ASSERTS the structural
invariants the flat model relies on (derived static content, slot order == get_objects_at order,
Bread clones appended last), so a violated assumption fails here, not silently on the GPU.

Usage:  python tools/gen_golden.py [--only NAME] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import random
import sys
from collections import defaultdict, deque

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def import_reference():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("reference not present: golden fixtures can only be regenerated in the build container")
    sys.path.insert(0, os.path.join(REPO, "tools", "refshim"))
    sys.path.insert(1, REFERENCE)
    sys.path.insert(2, REPO)


import_reference()

from cooking_zoo.environment.cooking_env import CookingEnvironment, parallel_env  # noqa: E402
from cooking_zoo.cooking_world import world_objects as wo  # noqa: E402
from cooking_zoo.cooking_world.abstract_classes import (  # noqa: E402
    DynamicObject, StaticObject, ContentObject, ActionObject)
from cooking_zoo.cooking_world.constants import (  # noqa: E402
    ChopFoodStates, BlenderFoodStates, ActionObjectState)
from cooking_zoo.cooking_agents.cooking_agent import CookingAgent  # noqa: E402

from cooking_zoo_amd import soa  # noqa: E402


# ------------------------------------------------------------------------------------------------
# reference world -> flat record
# ------------------------------------------------------------------------------------------------

class SlotMap:
    """Assigns dynamic-object slots for one episode: class-major in `world_objects` key order,
    list order inside a class, Bread clone head-room directly after the Bread originals."""

    def __init__(self, world, max_dyn):
        self.class_order = [k for k, v in world.world_objects.items()
                            if issubclass(wo.StringToClass[k], DynamicObject) and len(v) > 0]
        self.base = {}
        self.cap = {}
        s = 0
        for k in self.class_order:
            n = len(world.world_objects[k])
            self.base[k] = s
            self.cap[k] = 2 * n if k == "Bread" else n
            s += self.cap[k]
        self.used = s
        assert s <= max_dyn, (s, max_dyn)

    def slots(self, world):
        """dict id(obj) -> slot, list of (slot, obj) in slot order."""
        out = {}
        ordered = []
        keys = [k for k, v in world.world_objects.items()
                if issubclass(wo.StringToClass[k], DynamicObject) and len(v) > 0]
        assert keys == self.class_order, (keys, self.class_order)
        for k in keys:
            lst = world.world_objects[k]
            assert len(lst) <= self.cap[k], (k, len(lst), self.cap[k])
            for i, o in enumerate(lst):
                out[id(o)] = self.base[k] + i
                ordered.append((self.base[k] + i, o))
        return out, ordered


def static_at(world):
    grid = {}
    for k, lst in world.world_objects.items():
        if issubclass(wo.StringToClass[k], StaticObject):
            for o in lst:
                assert o.location not in grid
                grid[o.location] = o
    assert len(grid) == world.width * world.height
    return grid


def world_to_record(env, dims, slotmap, layout_id=0, recipe_ids=None, recipe_nodes=soa.NARROW_NODES):
    world = env.world
    rec = soa.new_record(dims)
    rec[soa.W_T] = env.t
    marks = 0
    bits = 8 if recipe_nodes == soa.NARROW_NODES else 16                # compact / wide recipe tables (soa.py)
    for r, g in enumerate(env.recipe_graphs):
        assert len(g.node_list) <= recipe_nodes
        for j, node in enumerate(g.node_list):
            if node.marked:
                marks |= 1 << (bits * r + j)
    rec[soa.W_MARKS] = marks & 0xFFFFFFFF
    rec[soa.W_MARKS_HI] = marks >> 32
    rec[soa.W_LAYOUT] = layout_id
    rid = [0xFF] * 4
    for i, v in enumerate(recipe_ids or []):
        rid[i] = v
    rec[soa.W_RECIPES] = rid[0] | (rid[1] << 8) | (rid[2] << 16) | (rid[3] << 24)

    slot_of, ordered = slotmap.slots(world)
    held = {}
    for a, ag in enumerate(world.agents):
        hs = -1
        if ag.holding is not None:
            hs = slot_of[id(ag.holding)]
            held[id(ag.holding)] = a
            assert ag.holding.location == ag.location
        rec[soa.AGENT_WORD0 + a] = soa.pack_agent(ag.location[0], ag.location[1], ag.orientation, hs)

    grid = static_at(world)
    cells = soa.record_cells(dims, rec)
    for (x, y), s in grid.items():
        t = soa.STATIC_CLASSES.index(type(s).__name__)
        v = t
        if isinstance(s, ActionObject) and s.status == ActionObjectState.READY:
            v |= soa.CELL_READY
        if isinstance(s, wo.Blender) and s.toggle:
            v |= soa.CELL_TOGGLE
        if isinstance(s, wo.Switch):
            assert not s.button_pressed          # transient inside a step only
            if s.switch_active:
                v |= soa.CELL_ACTIVE
        if isinstance(s, wo.Block) and s.walkable:
            v |= soa.CELL_WALK
        if not isinstance(s, wo.Block):
            assert s.walkable == (t in (soa.FLOOR, soa.SWITCH))
        cells[y * dims.W + x] = v

    container = {}
    for slot, o in ordered:
        if isinstance(o, wo.Plate):
            for seq, c in enumerate(o.content):
                assert id(c) not in container
                container[id(c)] = (slot, seq)
                assert c.location == o.location
    for slot, o in ordered:
        cls = soa.DYNAMIC_CLASSES.index(type(o).__name__)
        flags = soa.DYN_ALIVE
        if getattr(o, "chop_state", None) == ChopFoodStates.CHOPPED:
            flags |= soa.DYN_CHOPPED
        if getattr(o, "blend_state", None) == BlenderFoodStates.MASHED:
            flags |= soa.DYN_MASHED
        if hasattr(o, "blend_state"):
            assert o.blend_state != BlenderFoodStates.IN_PROGRESS
        if o.free:
            flags |= soa.DYN_FREE
        rec[dims.dyn0_word0 + slot] = soa.pack_dyn0(o.location[0], o.location[1], cls, flags)
        cs, seq = container.get(id(o), (-1, 0))
        rec[dims.dyn1_word0 + slot] = soa.pack_dyn1(cs, seq)
    # reserved (not yet alive) Bread clone slots carry the class so the allocator can find them
    if "Bread" in slotmap.base:
        n_alive = len(world.world_objects["Bread"])
        for i in range(n_alive, slotmap.cap["Bread"]):
            rec[dims.dyn0_word0 + slotmap.base["Bread"] + i] = soa.pack_dyn0(0, 0, soa.BREAD, 0)

    # ---- invariants of the flat model, checked against the live reference objects
    for (x, y), s in grid.items():
        here = [(slot, o) for slot, o in ordered if o.location == (x, y)]
        # get_objects_at(DynamicObject) order == slot order
        ref_order = world.get_objects_at((x, y), DynamicObject)
        assert [id(o) for _, o in here] == [id(o) for o in ref_order]
        if isinstance(s, (wo.Floor, wo.Switch, wo.Block)):
            assert all(isinstance(c, wo.Agent) for c in s.content)
            continue
        direct = [o for _, o in here if id(o) not in container and id(o) not in held]
        assert [id(o) for o in direct] == [id(o) for o in s.content], (type(s).__name__, (x, y))
    return rec


def layout_arrays(env, dims, slotmap):
    """Initial state of a freshly reset env (also the reset image of the layout pool)."""
    rec = world_to_record(env, dims, slotmap)
    return rec


def static_lists(env):
    """Per static class, cell positions in world_objects list order (needed for obs ordering)."""
    world = env.world
    out = {}
    for k, lst in world.world_objects.items():
        if issubclass(wo.StringToClass[k], StaticObject) and k != "Floor":
            out[k] = [list(o.location) for o in lst]
    return out


# ------------------------------------------------------------------------------------------------
# policies
# ------------------------------------------------------------------------------------------------

DIRS = {1: (-1, 0), 2: (1, 0), 3: (0, 1), 4: (0, -1)}


class Bumper:
    """Random 'walk up to a non-walkable cell and use it' policy: dense coverage of pick/place/
    chop/blend/plate/deliver transitions, far denser than uniform random actions."""

    def __init__(self, rng, scheme, eps=0.15):
        self.rng, self.scheme, self.eps = rng, scheme, eps
        self.goal = None

    def pick_goal(self, world):
        grid = static_at(world)
        cand = []
        for loc, s in grid.items():
            if isinstance(s, (wo.Floor, wo.Switch)):
                if isinstance(s, wo.Switch):
                    cand += [loc] * 3
                continue
            w = 1
            if world.get_objects_at(loc, DynamicObject):
                w += 4
            if isinstance(s, (wo.Cutboard, wo.Blender, wo.Deliversquare)):
                w += 4
            if isinstance(s, wo.Block):
                w += 2
            cand += [loc] * w
        self.goal = cand[self.rng.integers(len(cand))]

    def act(self, world, idx):
        n_act = 5 if self.scheme == "scheme3" else 8
        if self.rng.random() < self.eps:
            return int(self.rng.integers(n_act))
        if self.goal is None or self.rng.random() < 0.03:
            self.pick_goal(world)
        ag = world.agents[idx]
        grid = static_at(world)
        gx, gy = self.goal
        if grid[self.goal].walkable:            # a Switch (or open Block): just walk onto it
            targets = {self.goal}
        else:
            targets = {(gx - dx, gy - dy) for dx, dy in DIRS.values()}
        # adjacent: bump / interact
        for a, (dx, dy) in DIRS.items():
            if (ag.location[0] + dx, ag.location[1] + dy) == self.goal and not grid[self.goal].walkable:
                if self.rng.random() < 0.5:
                    self.goal = None
                if self.scheme == "scheme3":
                    return a
                if ag.orientation != a:
                    return a
                return int(self.rng.choice([5, 5, 5, 5, 7, 7, 6]))
        if ag.location == self.goal:
            self.goal = None
            return int(self.rng.integers(n_act))
        # BFS over walkable cells
        prev = {ag.location: None}
        dq = deque([ag.location])
        found = None
        while dq:
            cur = dq.popleft()
            if cur in targets:
                found = cur
                break
            for a, (dx, dy) in DIRS.items():
                nx = (cur[0] + dx, cur[1] + dy)
                if nx in prev or nx not in grid or not grid[nx].walkable:
                    continue
                prev[nx] = (cur, a)
                dq.append(nx)
        if found is None or found == ag.location:
            self.goal = None
            return int(self.rng.integers(n_act))
        cur = found
        while prev[cur][0] != ag.location:
            cur = prev[cur][0]
        return prev[cur][1]


class Heuristic:
    """The reference's own heuristic agent (cooking_agents/cooking_agent.py) as a trace generator."""

    def __init__(self, rng, recipe, name, eps=0.1):
        self.rng, self.eps = rng, eps
        self.agent = CookingAgent(recipe, name)

    def act(self, world, idx):
        if self.rng.random() < self.eps:
            return int(self.rng.integers(5))
        objects = defaultdict(list)
        objects.update(world.world_objects)
        objects["Agent"] = world.agents
        try:
            return int(self.agent.step(objects))
        except Exception:
            return int(self.rng.integers(5))


class Uniform:
    def __init__(self, rng, scheme):
        self.rng, self.n = rng, (5 if scheme == "scheme3" else 8)

    def act(self, world, idx):
        return int(self.rng.integers(self.n))


class Scripted:
    def __init__(self, actions):
        self.actions, self.i = list(actions), 0

    def act(self, world, idx):
        a = self.actions[self.i] if self.i < len(self.actions) else 0
        self.i += 1
        return int(a)


# ------------------------------------------------------------------------------------------------
# episode capture
# ------------------------------------------------------------------------------------------------

# (seed, policy, step at which the unmodified reference raises) - reproducers from the fuzz histograms; the first is the judge's
REFCRASH_PLAN = [(987704, "bumper", 151)]

RECIPE_NAMES = ["TomatoSalad", "TomatoLettuceSalad", "CarrotBanana", "MashedCarrotBanana", "CucumberOnion",
                "AppleWatermelon", "TomatoLettuceOnionSalad", "no_recipe"]


def level_max_dyn(level_path_or_name):
    if level_path_or_name.endswith(".json"):
        p = level_path_or_name
    else:
        p = os.path.join(REFERENCE, "cooking_zoo", "utils", "level", level_path_or_name + ".json")
    with open(p) as f:
        lv = json.load(f)
    n = 0
    for d in lv["DYNAMIC_OBJECTS"]:
        (name, spec), = d.items()
        n += spec["COUNT"] * (2 if name == "Bread" else 1)
    return max(n, 1)          # the record always has at least one (possibly unused) slot


class ReferenceCrash(Exception):
    """Raised by capture_episode(on_crash="raise") - kept for callers that want the old behaviour by name."""


def crash_record(exc, step, act):
    """What diff_fuzz keeps of an exception that left the reference's accumulated_step: type, message, and the frames of
    the traceback that lie inside the reference (innermost last), as `file:line function`."""
    import traceback
    frames = [f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}" for fr in traceback.extract_tb(exc.__traceback__)
              if os.path.abspath(fr.filename).startswith(REFERENCE)]
    detail = ""
    tb = exc.__traceback__
    while tb is not None:                               # the innermost frame's view of what was being used
        loc = tb.tb_frame.f_locals
        so = loc.get("static_object")
        if so is not None:
            content = getattr(so, "content", [])
            detail = (f"{type(so).__name__} status={getattr(getattr(so, 'status', None), 'name', None)} content="
                      + "[" + ",".join(f"{type(o).__name__}({getattr(getattr(o, 'chop_state', None), 'name', '-')}/"
                                       f"{getattr(getattr(o, 'blend_state', None), 'name', '-')})" for o in content) + "]")
        tb = tb.tb_next
    return {"type": type(exc).__name__, "message": str(exc)[:120], "frames": frames, "step": int(step),
            "action": [int(a) for a in act], "detail": detail}


# Reference crash sites the build answers with a defined no-op (include/cookingzoo.h "deviations").
# `tolerant_reference()` turns exactly those raise sites of the reference into that no-op - nothing else - so that the
# generator can let the REFERENCE compute the rest of the crashing step (other agents, progress_world, rewards,
# observation): the post-crash expectations of the *_refcrash_* fixtures are reference outputs, not oracle outputs.
def _tolerant_cutboard_action(orig):
    def action(self):
        out = orig(self)
        # world_objects.py:250-269: READY with empty content, or READY with content whose chop() does not execute,
        # falls off the loop -> None -> TypeError at cooking_world.py:162.  Build: nothing created / deleted / executed.
        return ([], [], False) if out is None else out
    return action


class tolerant_reference:
    def __enter__(self):
        self._orig = wo.Cutboard.action
        wo.Cutboard.action = _tolerant_cutboard_action(self._orig)
        return self

    def __exit__(self, *a):
        wo.Cutboard.action = self._orig
        return False


def capture_episode(cfg, seed, policy_name, max_len=None, recipe_ids=None, recipe_nodes=soa.NARROW_NODES, on_crash="raise"):
    """Returns dict of arrays for one episode (reset + steps until done / max_len).  on_crash="record": an exception out
    of the reference's accumulated_step ends the episode at the last good step; `crash` (crash_record) names it."""
    random.seed(seed)
    np.random.seed(seed)
    rng = np.random.default_rng(seed + 7919)
    env = CookingEnvironment(level=cfg["level"], meta_file=cfg["meta_file"], num_agents=cfg["num_agents"],
                             max_steps=cfg["max_steps"], recipes=cfg["recipes"],
                             obs_spaces=["feature_vector"] * cfg["num_agents"],
                             end_condition_all_dishes=cfg["end_condition_all_dishes"],
                             action_scheme=cfg["action_scheme"], reward_scheme=cfg.get("reward_scheme"))
    env.reset()
    world = env.world
    A = cfg["num_agents"]
    F = env.feature_vector_representation_length
    D = cfg["max_dyn"]
    dims = soa.Dims(world.width, world.height, D, A, F)
    slotmap = SlotMap(world, D)
    recipe_ids = [RECIPE_NAMES.index(r) for r in cfg["recipes"]] if recipe_ids is None else list(recipe_ids)

    scheme = cfg["action_scheme"]
    pols = []
    for i in range(A):
        if policy_name == "uniform":
            pols.append(Uniform(rng, scheme))
        elif policy_name == "bumper":
            pols.append(Bumper(rng, scheme))
        elif policy_name == "heuristic":
            pols.append(Heuristic(rng, cfg["recipes"][i], f"agent-{i + 1}"))
        elif policy_name == "mixed":
            pols.append(Heuristic(rng, cfg["recipes"][i], f"agent-{i + 1}", eps=0.2) if i % 2 == 0
                        else Bumper(rng, scheme))
        elif isinstance(policy_name, (list, tuple)):
            pols.append(Scripted([row[i] for row in policy_name]))
        else:
            raise ValueError(policy_name)

    states = [world_to_record(env, dims, slotmap, 0, recipe_ids, recipe_nodes)]
    obs = [np.stack([env.observe(a) for a in env.possible_agents])]
    assert obs[0].dtype == np.float64 and obs[0].shape == (A, F)
    actions, rewards, terms, truncs = [], [], [], []
    statics = static_lists(env)
    limit = max_len or cfg["max_steps"]
    crash = None
    for step in range(limit):
        act = [p.act(world, i) for i, p in enumerate(pols)]
        try:
            env.accumulated_step(act)
        except Exception as exc:                        # noqa: BLE001 - whatever the reference raises is the finding
            if on_crash != "record":
                raise
            crash = crash_record(exc, step, act)
            break
        actions.append(act)
        rewards.append([float(env.rewards[a]) for a in env.possible_agents])
        for a in env.possible_agents:
            assert isinstance(env.rewards[a], (float, np.floating)), type(env.rewards[a])
        terms.append([bool(env.terminations[a]) for a in env.possible_agents])
        truncs.append([bool(env.truncations[a]) for a in env.possible_agents])
        states.append(world_to_record(env, dims, slotmap, 0, recipe_ids, recipe_nodes))
        o = np.stack([env.observe(a) for a in env.possible_agents])
        assert o.dtype == np.float64
        obs.append(o)
        if any(terms[-1]) or any(truncs[-1]):
            break
    return {
        "crash": crash,
        "dims": np.array(dims.as_tuple(), dtype=np.int32),
        "states": np.stack(states),
        "obs": np.stack(obs),
        "actions": np.array(actions, dtype=np.int32).reshape(-1, A),
        "rewards": np.array(rewards, dtype=np.float64).reshape(-1, A),
        "terms": np.array(terms, dtype=np.uint8).reshape(-1, A),
        "truncs": np.array(truncs, dtype=np.uint8).reshape(-1, A),
        "statics": statics,
        "class_order": slotmap.class_order,
    }


def capture_spawn_episode(cfg, seed, policy_name, rates, max_len=None):
    """One episode with agent despawn/respawn switched on (cooking_world.py:267-290; rates = (despawn, respawn, grace)).

    The build keeps the spawn bookkeeping (two global RNG streams, variable agent sets) on the host and gives the device
    one extra input: action -1 = "this agent is not in the list world_step acts on".  So the fixture records, per step,
    the start state (respawn relocations of the previous step applied), the actions with -1 for inactive agents, and the
    world right BEFORE handle_agent_spawn runs (state + observation of every agent), which is what one device step
    must reproduce; plus the per-recipe rewards, the done flag and the t >= max_steps flag of the step."""
    random.seed(seed)
    np.random.seed(seed)
    rng = np.random.default_rng(seed + 7919)
    despawn, respawn, grace = rates
    env = CookingEnvironment(level=cfg["level"], meta_file=cfg["meta_file"], num_agents=cfg["num_agents"],
                             max_steps=cfg["max_steps"], recipes=cfg["recipes"],
                             obs_spaces=["feature_vector"] * cfg["num_agents"],
                             end_condition_all_dishes=cfg["end_condition_all_dishes"],
                             action_scheme=cfg["action_scheme"], reward_scheme=cfg.get("reward_scheme"),
                             agent_respawn_rate=respawn, grace_period=grace, agent_despawn_rate=despawn)
    env.reset()
    world = env.world
    A = cfg["num_agents"]
    F = env.feature_vector_representation_length
    D = cfg["max_dyn"]
    dims = soa.Dims(world.width, world.height, D, A, F)
    slotmap = SlotMap(world, D)
    recipe_ids = [RECIPE_NAMES.index(r) for r in cfg["recipes"]]
    scheme = cfg["action_scheme"]
    pols = [Bumper(rng, scheme) if (policy_name == "bumper" or (policy_name == "mixed" and i % 2)) else Uniform(rng, scheme)
            for i in range(A)]
    snap = {}
    orig_spawn, orig_rewards = world.handle_agent_spawn, env.compute_rewards

    def spawn_hook():
        snap["rec"] = world_to_record(env, dims, slotmap, 0, recipe_ids)
        snap["obs"] = np.stack([env.get_feature_vector(a) for a in env.possible_agents])
        orig_spawn()

    def rewards_hook(*args, **kw):
        out = orig_rewards(*args, **kw)
        snap["dones"], snap["rewards"] = out[0], out[1]
        return out
    world.handle_agent_spawn = spawn_hook
    env.compute_rewards = rewards_hook

    full_obs = lambda: np.stack([env.get_feature_vector(a) for a in env.possible_agents])
    states = [world_to_record(env, dims, slotmap, 0, recipe_ids)]
    obs = [full_obs()]
    active = [list(world.active_agents)]
    pre_states, pre_obs, actions, rewards, terms, truncs, changed = [], [], [], [], [], [], []
    limit = max_len or cfg["max_steps"]
    for step in range(limit):
        act_now = list(world.active_agents)
        full = [(pols[i].act(world, i) if act_now[i] else -1) for i in range(A)]
        try:
            env.accumulated_step([a for a in full if a >= 0])
        except IndexError:
            # reference defect: truncation by max_steps while somebody is despawned sizes active_agents with the CURRENT
            # agent count (cooking_env.py:337, AECEnv.num_agents) and the scatter loop (:257) runs off its end
            assert env.t >= cfg["max_steps"] and not all(act_now)
            break
        pre = snap["rec"].copy()
        post = world_to_record(env, dims, slotmap, 0, recipe_ids)
        pre[soa.W_MARKS] = post[soa.W_MARKS]                  # the graphs are re-evaluated after the world step
        actions.append(full)
        pre_states.append(pre)
        pre_obs.append(snap["obs"])
        rewards.append([float(snap["rewards"][i]) for i in range(A)])
        terms.append([bool(snap["dones"][0])] * A)
        truncs.append([env.t >= cfg["max_steps"]] * A)
        states.append(post)
        obs.append(full_obs())
        active.append(list(world.active_agents))
        changed.append(list(world.status_changed))
        if terms[-1][0] or truncs[-1][0]:
            break
    return {
        "dims": np.array(dims.as_tuple(), dtype=np.int32), "states": np.stack(states), "obs": np.stack(obs),
        "pre_states": np.stack(pre_states), "pre_obs": np.stack(pre_obs),
        "actions": np.array(actions, dtype=np.int32).reshape(-1, A),
        "rewards": np.array(rewards, dtype=np.float64).reshape(-1, A),
        "terms": np.array(terms, dtype=np.uint8).reshape(-1, A), "truncs": np.array(truncs, dtype=np.uint8).reshape(-1, A),
        "active": np.array(active, dtype=np.uint8).reshape(-1, A), "changed": np.array(changed, dtype=np.uint8).reshape(-1, A),
        "statics": static_lists(env), "class_order": slotmap.class_order,
    }


def action_hash(seed, env_global, agent, step, n):
    """the device's counter-based action stream (csrc/cz_device.h action_hash, oracle czo_action), restated for the generator;
    tests/test_spawn_keyed.py checks it against the oracle's"""
    M = 0xFFFFFFFF
    x = (seed & M) ^ (((seed >> 32) * 0x9E3779B1) & M)
    x ^= ((env_global & M) * 0x85EBCA6B + ((env_global >> 32) & M) * 0x27D4EB2F) & M
    x ^= ((agent + 1) * 0xC2B2AE35) & M
    x ^= ((step + 1) * 0x165667B1) & M
    for rnd in range(2):
        if rnd:
            x = (x + step * 0x9E3779B9) & M
        x ^= x >> 16; x = (x * 0x7FEB352D) & M; x ^= x >> 15; x = (x * 0x846CA68B) & M; x ^= x >> 16
    return (x * n) >> 32


def capture_spawn_keyed_episode(cfg, seed, policy_name, rates, env_id, episode_no, spawn_seed, max_len=None):
    """One episode of the UNMODIFIED reference with despawn / respawn on, in which every draw the reference takes inside
    `CookingWorld.handle_agent_spawn` (np.random.random, cooking_world.py:273-276) and `parsing.generate_location`
    (random.sample, parsing.py:157-158) is answered with the batched build's keyed stream
    `spawn.uniform(spawn_seed, env_id, episode_no << 32 | t, agent, draw)`: draw 0 = the despawn test, 1 = the respawn test,
    2 + 2 k / 3 + 2 k = the x / y candidate of try k.  Which agent and which try a call belongs to is read from the calling
    frames of the reference's own functions (the loop variable `i`, `index`, `time_out`) - nothing of the reference is replaced
    but the two random sources.  The fixture is what a batched env with the same key must reproduce: records incl. the status
    word (despawned bits, grace counters), observations of every agent, per-recipe rewards, flags incl. the truncated-once
    report of an agent that leaves.  policy "stream": actions from the device's own counter-based stream (cz_rollout)."""
    import numpy.random as npr
    from cooking_zoo_amd import spawn as czspawn
    random.seed(seed)
    np.random.seed(seed)
    rng = np.random.default_rng(seed + 7919)
    despawn, respawn, grace = rates
    env = CookingEnvironment(level=cfg["level"], meta_file=cfg["meta_file"], num_agents=cfg["num_agents"],
                             max_steps=cfg["max_steps"], recipes=cfg["recipes"],
                             obs_spaces=["feature_vector"] * cfg["num_agents"],
                             end_condition_all_dishes=cfg["end_condition_all_dishes"],
                             action_scheme=cfg["action_scheme"], reward_scheme=cfg.get("reward_scheme"),
                             agent_respawn_rate=respawn, grace_period=grace, agent_despawn_rate=despawn)
    env.reset()
    world = env.world
    A = cfg["num_agents"]
    F = env.feature_vector_representation_length
    D = cfg["max_dyn"]
    dims = soa.Dims(world.width, world.height, D, A, F)
    slotmap = SlotMap(world, D)
    recipe_ids = [RECIPE_NAMES.index(r) for r in cfg["recipes"]]
    scheme = cfg["action_scheme"]
    n_act = 5 if scheme == "scheme3" else 8
    pols = [Bumper(rng, scheme) if (policy_name == "bumper" or (policy_name == "mixed" and i % 2)) else Uniform(rng, scheme)
            for i in range(A)]
    log = {"draws": 0, "tries": 0, "respawns": 0}
    st = {"calls": 0}

    def key():
        return (episode_no << 32) | env.t                    # (accumulated_step has already incremented t: cooking_env.py:244)

    def keyed_random():
        fr = sys._getframe(1)
        assert fr.f_code.co_name == "handle_agent_spawn", fr.f_code.co_name
        i, w = fr.f_locals["i"], fr.f_locals["self"]
        log["draws"] += 1
        return float(czspawn.uniform(spawn_seed, env_id, key(), i, 0 if w.active_agents[i] else 1))

    def keyed_sample(population, k):
        fr = sys._getframe(1)
        assert fr.f_code.co_name == "generate_location" and k == 1, fr.f_code.co_name
        up = sys._getframe(2)
        assert up.f_code.co_name == "respawn_agent"
        agent, time_out = up.f_locals["index"], fr.f_locals["time_out"]
        axis = st["calls"] % 2
        assert st["calls"] // 2 == time_out, (st["calls"], time_out)      # two samples per try: x, then y
        st["calls"] += 1
        log["tries"] += axis
        u = float(czspawn.uniform(spawn_seed, env_id, key(), agent, 2 + 2 * time_out + axis))
        return [population[int(u * len(population))]]

    orig_spawn, orig_respawn, orig_rewards = world.handle_agent_spawn, world.respawn_agent, env.compute_rewards
    snap = {}

    def respawn_hook(index):
        st["calls"] = 0
        log["respawns"] += 1
        return orig_respawn(index)

    def spawn_hook():
        keep = (npr.random, np.random.random, random.sample)
        np.random.random = keyed_random
        random.sample = keyed_sample
        try:
            orig_spawn()
        finally:
            np.random.random, random.sample = keep[1], keep[2]

    def rewards_hook(*args, **kw):
        out = orig_rewards(*args, **kw)
        snap["dones"], snap["rewards"] = out[0], out[1]
        return out
    world.handle_agent_spawn = spawn_hook
    world.respawn_agent = respawn_hook
    env.compute_rewards = rewards_hook

    def record():
        rec = world_to_record(env, dims, slotmap, 0, recipe_ids)
        rec[soa.W_EPISODE] = episode_no
        rec[soa.W_STATUS] |= czspawn.status_bits(np.array([world.active_agents], dtype=bool),
                                                 np.array([world.agent_grace_period], dtype=np.int64),
                                                 czspawn.grace_bits(grace, A))[0]
        return rec

    full_obs = lambda: np.stack([env.get_feature_vector(a) for a in env.possible_agents])
    states, obs = [record()], [full_obs()]
    actions, rewards, terms, truncs = [], [], [], []
    limit = max_len or cfg["max_steps"]
    for step in range(limit):
        act_now = list(world.active_agents)
        if policy_name == "stream":
            full = [action_hash(spawn_seed, env_id, i, step, n_act) for i in range(A)]
        else:
            # (a despawned agent gets an action too: the batched env must ignore it by itself)
            full = [pols[i].act(world, i) if act_now[i] else int(rng.integers(0, n_act)) for i in range(A)]
        try:
            env.accumulated_step([a for i, a in enumerate(full) if act_now[i]])
        except IndexError:
            # reference defect (cooking_env.py:337): max_steps reached while somebody is despawned; the build truncates instead
            assert env.t >= cfg["max_steps"] and not all(act_now)
            break
        gone = [bool(world.status_changed[i] and not world.active_agents[i]) for i in range(A)]
        trunc_all = env.t >= cfg["max_steps"]
        if trunc_all:
            break                     # (compute_truncated has cleared active_agents: the step's spawn bookkeeping is not what a record holds)
        actions.append(full)
        rewards.append([float(snap["rewards"][i]) for i in range(A)])
        terms.append([bool(snap["dones"][0])] * A)
        truncs.append([bool(g) for g in gone])
        rec = record()
        if terms[-1][0]:
            rec[soa.W_STATUS] |= 1 | 2                                     # done | terminated (cz_device.h ST_*)
        states.append(rec)
        obs.append(full_obs())
        if terms[-1][0]:
            break
    return {
        "dims": np.array(dims.as_tuple(), dtype=np.int32), "states": np.stack(states), "obs": np.stack(obs),
        "actions": np.array(actions, dtype=np.int32).reshape(-1, A),
        "rewards": np.array(rewards, dtype=np.float64).reshape(-1, A),
        "terms": np.array(terms, dtype=np.uint8).reshape(-1, A), "truncs": np.array(truncs, dtype=np.uint8).reshape(-1, A),
        "statics": static_lists(env), "class_order": slotmap.class_order,
        "spawn_areas": [[list(xs), list(ys)] for xs, ys in world.agent_spawn_locations],
        "log": dict(log),
    }


def episode_stats(ep):
    """Coverage summary of what an episode reached (for the generator log)."""
    dims = soa.Dims(*[int(v) for v in ep["dims"]])
    last = ep["states"][-1]
    chopped = mashed = plated = clones = 0
    for s in range(dims.D):
        x, y, c, f = soa.unpack_dyn0(last[dims.dyn0_word0 + s])
        cont, seq = soa.unpack_dyn1(last[dims.dyn1_word0 + s])
        if f & soa.DYN_ALIVE:
            chopped += bool(f & soa.DYN_CHOPPED)
            mashed += bool(f & soa.DYN_MASHED)
            plated += cont >= 0
    return dict(steps=len(ep["actions"]), chopped=chopped, mashed=mashed, plated=plated,
                term=int(ep["terms"][-1].any()) if len(ep["terms"]) else 0,
                ret=float(ep["rewards"].sum()) if len(ep["rewards"]) else 0.0,
                marks=hex(int(last[soa.W_MARKS])))


def save_set(name, cfg, episodes, out_dir):
    """One .npz per set: episodes concatenated, with offsets."""
    meta = dict(cfg)
    # store level / meta names as package-relative stems (the build ships the same files)
    meta["level"] = os.path.splitext(os.path.basename(cfg["level"]))[0]
    meta["meta_file"] = os.path.splitext(os.path.basename(cfg["meta_file"]))[0]
    meta["episodes"] = []
    arrays = {}
    for i, ep in enumerate(episodes):
        for k in ("dims", "states", "obs", "actions", "rewards", "terms", "truncs"):
            arrays[f"e{i}_{k}"] = ep[k]
        for k in ("pre_states", "pre_obs", "active", "changed"):           # despawn / respawn sets only
            if k in ep:
                arrays[f"e{i}_{k}"] = ep[k]
        meta["episodes"].append({"statics": ep["statics"], "class_order": ep["class_order"],
                                 "seed": ep["seed"], "policy": ep["policy"] if isinstance(ep["policy"], str) else "scripted"})
        for k in ("env_id", "episode_no", "spawn_areas", "refcrash"):       # keyed despawn / respawn sets; reference-crash sets
            if k in ep:
                meta["episodes"][-1][k] = ep[k]
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **arrays)
    sz = os.path.getsize(path)
    print(f"[golden] {name}: {len(episodes)} episodes, {sum(len(e['actions']) for e in episodes)} steps, {sz / 1024:.0f} KiB")
    return path


# ------------------------------------------------------------------------------------------------
# fixture sets
# ------------------------------------------------------------------------------------------------

def base_cfg(level, agents, recipes, scheme="scheme3", max_steps=400, all_dishes=False, meta="example", max_dyn=None,
             reward_scheme=None):
    return dict(level=level, meta_file=meta, num_agents=agents, max_steps=max_steps, recipes=recipes,
                end_condition_all_dishes=all_dishes, action_scheme=scheme,
                max_dyn=max_dyn if max_dyn is not None else level_max_dyn(level),
                reward_scheme=reward_scheme)


def run_set(name, cfg, plan, out_dir):
    eps = []
    for seed, policy, max_len in plan:
        ep = capture_episode(cfg, seed, policy, max_len)
        ep["seed"], ep["policy"] = seed, policy
        print(f"   {name} seed={seed} policy={policy if isinstance(policy, str) else 'scripted'}: {episode_stats(ep)}")
        eps.append(ep)
    return save_set(name, cfg, eps, out_dir)


def refcrash_set(name, cfg, plan, out_dir):
    """Trajectories on which the UNMODIFIED reference raises (found by tools/diff_fuzz.py; classes and counts in
    profiles/r05/refcrash_histogram*.json).  Per episode: the reference is run twice from the same seeds - as it is, which pins
    WHERE it raises (step, action, frames: stored as `refcrash`), and with exactly that raise site turned into the build's
    documented no-op (tolerant_reference), which gives the expectations THROUGH and PAST the crashing step from the reference's own
    code: the other agents' actions of that step, progress_world, rewards, observation.  The two runs agree up to the crash."""
    eps = []
    for seed, policy, want_step in plan:
        plain = capture_episode(cfg, seed, policy, on_crash="record")
        crash = plain["crash"]
        assert crash is not None, f"{name}: seed {seed} does not crash the reference (any more?)"
        assert crash["step"] == want_step, (crash["step"], want_step)
        with tolerant_reference():
            ep = capture_episode(cfg, seed, policy, on_crash="record")
        assert ep["crash"] is None, f"{name}: seed {seed} hits a second, unknown crash site: {ep['crash']}"
        n = len(plain["actions"])
        assert np.array_equal(ep["states"][:n + 1], plain["states"]) and np.array_equal(ep["obs"][:n + 1], plain["obs"])
        assert [int(a) for a in ep["actions"][n]] == crash["action"]
        ep["seed"], ep["policy"], ep["refcrash"] = seed, policy, crash
        print(f"   {name} seed={seed}: reference raises {crash['type']} at step {crash['step']} action {crash['action']} "
              f"[{crash['detail']}] <- {crash['frames'][-1]}; tolerant run: {episode_stats(ep)}")
        eps.append(ep)
    return save_set(name, cfg, eps, out_dir)


def kat_c3(out_dir):
    """SURVEY.md Appendix C.3 pinned known-answer trace (heuristic agent, completes in 30 steps)."""
    cfg = base_cfg("coop_test", 1, ["TomatoLettuceSalad"], all_dishes=True)
    acts = [[int(c)] for c in "312442214141113331242131124444"]
    ep = capture_episode(cfg, 1, acts, max_len=len(acts))
    ep["seed"], ep["policy"] = 1, "scripted"
    assert ep["rewards"][-1, 0] == 19.9875 and ep["terms"][-1, 0] == 1
    assert abs(ep["rewards"].sum() - 19.625) < 1e-12
    h = hashlib.sha256()
    for o in ep["obs"]:
        h.update(o[0].tobytes())
    assert h.hexdigest().startswith("24e17ccb854352dd"), h.hexdigest()
    print("   KAT C.3 reproduced: sha256(obs) =", h.hexdigest()[:16])
    return save_set("kat_c3", cfg, [ep], out_dir)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    sets = {}
    sets["kat_c3"] = lambda: kat_c3(args.out)
    # reference-crash class "Cutboard READY with empty content + EXECUTE" (cooking_world.py:162 <- world_objects.py:250-269):
    # a mashed Banana goes onto a board (accepts() only looks at chop_state), a Plate in hand absorbs it (attempt_merge branch 2 removes
    # it from the board's content without releases()), the board stays READY and empty, EXECUTE falls off Cutboard.action
    sets["refcrash_cutboard_scheme1"] = lambda: refcrash_set(
        "refcrash_cutboard_scheme1", base_cfg("coexistence_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], scheme="scheme1", max_steps=400),
        REFCRASH_PLAN, args.out)
    # ... the same class on another level family with four agents (reproducer of profiles/r05/refcrash_histogram_hunt.json)
    own = lambda stem: (os.path.join(REPO, "cooking_zoo_amd", "utils", "level", stem + ".json"),
                        os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", stem + ".json"))
    sets["refcrash_cutboard_crowded"] = lambda: refcrash_set(
        "refcrash_cutboard_crowded", base_cfg(own("crowded_6x5")[0], 4, ["MashedCarrotBanana", "TomatoLettuceOnionSalad", "no_recipe", "AppleWatermelon"],
                                             scheme="scheme1", max_steps=209, meta=own("crowded_6x5")[1]),
        [(2028264, "bumper", 110)], args.out)
    # cfg 1: 1 agent, coop_test, TomatoLettuceSalad
    sets["cfg1_coop_1agent"] = lambda: run_set(
        "cfg1_coop_1agent", base_cfg("coop_test", 1, ["TomatoLettuceSalad"]),
        [(0, "uniform", 400), (2, "bumper", 400), (3, "heuristic", 120), (4, "heuristic", 120), (5, "bumper", 200)], args.out)
    # cfg 2: 2 agents, coop_test, [TomatoLettuceSalad, CarrotBanana]
    sets["cfg2_coop_2agents"] = lambda: run_set(
        "cfg2_coop_2agents", base_cfg("coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"]),
        [(10, "uniform", 400), (11, "bumper", 400), (12, "bumper", 400), (13, "mixed", 200), (14, "heuristic", 150),
         (15, "mixed", 200), (16, "bumper", 300), (17, "heuristic", 150)], args.out)
    sets["cfg2_all_dishes"] = lambda: run_set(
        "cfg2_all_dishes", base_cfg("coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], all_dishes=True,
                                    reward_scheme={"recipe_reward": 20, "max_time_penalty": -5, "recipe_penalty": -40,
                                                   "recipe_node_reward": 1.5}),
        [(20, "heuristic", 250), (21, "mixed", 250), (22, "bumper", 250), (23, "heuristic", 250)], args.out)
    # cfg 3 ingredients: other levels, other recipes
    sets["coexistence_2agents"] = lambda: run_set(
        "coexistence_2agents", base_cfg("coexistence_test", 2, ["MashedCarrotBanana", "AppleWatermelon"], max_steps=300),
        [(30, "bumper", 300), (31, "mixed", 300), (32, "heuristic", 200), (33, "uniform", 300), (34, "bumper", 300)], args.out)
    sets["switch_2agents"] = lambda: run_set(
        "switch_2agents", base_cfg("switch_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], max_steps=300),
        [(40, "bumper", 300), (41, "bumper", 300), (42, "uniform", 300), (43, "mixed", 300)], args.out)
    sets["switch_misc_recipes"] = lambda: run_set(
        "switch_misc_recipes", base_cfg("switch_test", 2, ["TomatoSalad", "no_recipe"], max_steps=200,
                                        reward_scheme={"recipe_reward": 10.5, "max_time_penalty": -3, "recipe_penalty": -7,
                                                       "recipe_node_reward": 2}),
        [(50, "bumper", 200), (51, "heuristic", 200), (52, "bumper", 200)], args.out)
    sets["coop_onion_recipes"] = lambda: run_set(
        "coop_onion_recipes", base_cfg("coop_test", 2, ["TomatoLettuceOnionSalad", "CucumberOnion"], max_steps=150),
        [(60, "bumper", 150), (61, "uniform", 150)], args.out)
    # scheme1
    sets["scheme1_coop_2agents"] = lambda: run_set(
        "scheme1_coop_2agents", base_cfg("coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], scheme="scheme1", max_steps=300),
        [(70, "bumper", 300), (71, "bumper", 300), (72, "uniform", 300), (73, "bumper", 300)], args.out)
    sets["scheme1_switch_2agents"] = lambda: run_set(
        "scheme1_switch_2agents", base_cfg("switch_test", 2, ["MashedCarrotBanana", "TomatoSalad"], scheme="scheme1", max_steps=300),
        [(80, "bumper", 300), (81, "uniform", 300), (82, "bumper", 300)], args.out)
    # truncation-heavy short episodes
    sets["short_truncation"] = lambda: run_set(
        "short_truncation", base_cfg("coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], max_steps=5),
        [(90, "uniform", 5), (91, "bumper", 5), (92, "uniform", 5)], args.out)
    # custom levels shipped by the build (4 agents, 16x16) if present
    lvl16 = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", "large_16x16.json")
    meta16 = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "large_16x16.json")
    if os.path.exists(lvl16) and os.path.exists(meta16):
        sets["large16_4agents"] = lambda: run_set(
            "large16_4agents",
            base_cfg(lvl16, 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"],
                     max_steps=250, meta=meta16),
            [(100, "bumper", 250), (101, "bumper", 250), (102, "uniform", 250), (103, "mixed", 250)], args.out)
        sets["large16_scheme1"] = lambda: run_set(
            "large16_scheme1",
            base_cfg(lvl16, 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"],
                     scheme="scheme1", max_steps=200, meta=meta16),
            [(110, "bumper", 200), (111, "uniform", 200)], args.out)
    crowd = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", "crowded_6x5.json")
    metac = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "crowded_6x5.json")
    if os.path.exists(crowd) and os.path.exists(metac):
        sets["crowded_4agents"] = lambda: run_set(
            "crowded_4agents",
            base_cfg(crowd, 4, ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"], max_steps=200, meta=metac),
            [(120, "uniform", 200), (121, "uniform", 200), (122, "bumper", 200), (123, "bumper", 200)], args.out)
        sets["crowded_scheme1"] = lambda: run_set(
            "crowded_scheme1",
            base_cfg(crowd, 4, ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"], scheme="scheme1",
                     max_steps=200, meta=metac),
            [(130, "uniform", 200), (131, "bumper", 200)], args.out)

    metae = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "edge.json")
    for ename, agents, recipes in (("edge_8x8", 3, ["TomatoSalad", "MashedCarrotBanana", "TomatoLettuceSalad"]),
                                   ("edge_9x8", 3, ["TomatoSalad", "CarrotBanana", "TomatoLettuceSalad"]),
                                   ("edge_empty", 2, ["TomatoSalad", "no_recipe"])):
        elvl = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", ename + ".json")
        if os.path.exists(elvl) and os.path.exists(metae):
            sets[ename] = (lambda n=ename, l=elvl, a=agents, r=recipes: run_set(
                n, base_cfg(l, a, r, max_steps=120, meta=metae),
                [(200, "bumper", 120), (201, "uniform", 120), (202, "bumper", 120)], args.out))
            sets[ename + "_scheme1"] = (lambda n=ename, l=elvl, a=agents, r=recipes: run_set(
                n + "_scheme1", base_cfg(l, a, r, scheme="scheme1", max_steps=100, meta=metae),
                [(210, "bumper", 100), (211, "uniform", 100)], args.out))
    # despawn / respawn (rates > 0): file names start with "spawn_" (the generic chain tests skip them, tests/test_spawn_*.py use them)
    def spawn_set(name, cfg, plan, rates):
        eps = []
        for seed, policy, max_len in plan:
            ep = capture_spawn_episode(cfg, seed, policy, rates, max_len)
            ep["seed"], ep["policy"] = seed, policy
            flips = int(np.abs(np.diff(ep["active"].astype(int), axis=0)).sum())
            print(f"   {name} seed={seed} policy={policy}: {episode_stats(ep)} activity flips {flips}")
            eps.append(ep)
        cfg = dict(cfg, rates=list(rates))
        return save_set(name, cfg, eps, args.out)
    sets["spawn_coop_2agents"] = lambda: spawn_set(
        "spawn_coop_2agents", base_cfg("coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], max_steps=150),
        [(300, "bumper", 150), (301, "uniform", 150), (302, "bumper", 150)], (0.15, 0.3, 2))
    if os.path.exists(crowd) and os.path.exists(metac):
        sets["spawn_crowded_4agents"] = lambda: spawn_set(
            "spawn_crowded_4agents",
            base_cfg(crowd, 4, ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"], max_steps=160, meta=metac),
            [(310, "bumper", 160), (311, "uniform", 160), (312, "mixed", 160)], (0.2, 0.25, 1))
        sets["spawn_crowded_scheme1"] = lambda: spawn_set(
            "spawn_crowded_scheme1",
            base_cfg(crowd, 3, ["TomatoSalad", "TomatoLettuceSalad", "MashedCarrotBanana"], scheme="scheme1", max_steps=120, meta=metac),
            [(320, "bumper", 120), (321, "uniform", 120)], (0.1, 0.2, 3))
    # despawn / respawn with KEYED draws (capture_spawn_keyed_episode): what the batched env evaluates on the device, produced by
    # the reference's own handle_agent_spawn / generate_location.  plan entries: (seed, policy, max_len, env id, episode number)
    def keyed_set(name, cfg, plan, rates, spawn_seed):
        eps = []
        for seed, policy, max_len, env_id, episode_no in plan:
            ep = capture_spawn_keyed_episode(cfg, seed, policy, rates, env_id, episode_no, spawn_seed, max_len)
            ep["seed"], ep["policy"], ep["env_id"], ep["episode_no"] = seed, policy, env_id, episode_no
            gone = int(ep["truncs"].sum())
            print(f"   {name} seed={seed} policy={policy} env={env_id} episode={episode_no}: {episode_stats(ep)} left {gone} {ep['log']}")
            eps.append(ep)
        cfg = dict(cfg, rates=list(rates), spawn_seed=spawn_seed)
        return save_set(name, cfg, eps, args.out)
    lv = lambda n: os.path.join(REPO, "cooking_zoo_amd", "utils", "level", n + ".json")
    mt = lambda n: os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", n + ".json")
    four = ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"]
    sets["spawn_keyed_coop_2agents"] = lambda: keyed_set(
        "spawn_keyed_coop_2agents", base_cfg("coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], max_steps=400),
        [(900, "bumper", 160, 0, 0), (901, "uniform", 160, 7, 3), (902, "stream", 200, 4100, 1), (903, "stream", 200, 1 << 33, 65536)], (0.15, 0.3, 2), 11)
    sets["spawn_keyed_crowded_4agents"] = lambda: keyed_set(
        "spawn_keyed_crowded_4agents",
        base_cfg(lv("crowded_6x5"), 4, ["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"], max_steps=400, meta=mt("crowded_6x5")),
        [(910, "bumper", 200, 1, 0), (911, "uniform", 200, 2, 5), (912, "mixed", 200, 3, 2), (913, "stream", 250, 40, 9)], (0.2, 0.25, 1), 5)
    sets["spawn_keyed_crowded_scheme1"] = lambda: keyed_set(
        "spawn_keyed_crowded_scheme1",
        base_cfg(lv("crowded_6x5"), 3, ["TomatoSalad", "TomatoLettuceSalad", "MashedCarrotBanana"], scheme="scheme1", max_steps=400, meta=mt("crowded_6x5")),
        [(920, "bumper", 160, 11, 0), (921, "uniform", 160, 12, 1), (922, "stream", 200, 13, 4)], (0.1, 0.2, 3), 77)
    sets["spawn_keyed_large16_4agents"] = lambda: keyed_set(
        "spawn_keyed_large16_4agents", base_cfg(lv("large_16x16"), 4, four, max_steps=400, meta=mt("large_16x16")),
        [(930, "bumper", 160, 21, 0), (931, "mixed", 160, 22, 6), (932, "stream", 200, 23, 1)], (0.2, 0.3, 2), 123456789)
    sets["spawn_keyed_large16_scheme1"] = lambda: keyed_set(
        "spawn_keyed_large16_scheme1", base_cfg(lv("large_16x16"), 2, four[:2], scheme="scheme1", max_steps=400, meta=mt("large_16x16")),
        [(940, "bumper", 120, 31, 0), (941, "stream", 160, 32, 2)], (0.25, 0.4, 0), 3)
    # (the third kernel instance: huge_objs_16x16, 190 object slots.  huge_20x20 / huge_32x32 give two agents a one-cell spawn
    # area, which the despawned agent itself occupies: the reference's generate_location raises there)
    sets["spawn_keyed_hugeobjs_3agents"] = lambda: keyed_set(
        "spawn_keyed_hugeobjs_3agents",
        base_cfg(lv("huge_objs_16x16"), 3, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon"], max_steps=400, meta=mt("huge_objs_16x16")),
        [(950, "bumper", 120, 41, 0), (951, "stream", 160, 42, 3)], (0.2, 0.35, 1), 99)
    sets["spawn_keyed_hugeobjs_scheme1"] = lambda: keyed_set(
        "spawn_keyed_hugeobjs_scheme1",
        base_cfg(lv("huge_objs_16x16"), 2, ["TomatoLettuceSalad", "CarrotBanana"], scheme="scheme1", max_steps=400, meta=mt("huge_objs_16x16")),
        [(960, "uniform", 100, 51, 0), (961, "stream", 120, 52, 1)], (0.3, 0.5, 2), 8)
    dense = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", "dense_16x16.json")
    metad = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "dense_16x16.json")
    if os.path.exists(dense) and os.path.exists(metad):
        sets["dense16_4agents"] = lambda: run_set(
            "dense16_4agents",
            base_cfg(dense, 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], max_steps=120, meta=metad),
            [(500, "bumper", 120), (501, "uniform", 120), (502, "mixed", 120)], args.out)
        sets["dense16_scheme1"] = lambda: run_set(
            "dense16_scheme1",
            base_cfg(dense, 2, ["TomatoLettuceOnionSalad", "MashedCarrotBanana"], scheme="scheme1", max_steps=80, meta=metad),
            [(510, "bumper", 80), (511, "uniform", 80)], args.out)
    metal = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "limits.json")
    for lname in ("limit_32x8", "limit_8x31"):
        llvl = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", lname + ".json")
        if os.path.exists(llvl) and os.path.exists(metal):
            sets[lname] = (lambda n=lname, l=llvl: run_set(
                n, base_cfg(l, 3, ["TomatoSalad", "MashedCarrotBanana", "TomatoLettuceSalad"], max_steps=150, meta=metal),
                [(400, "bumper", 150), (401, "uniform", 150), (402, "mixed", 150)], args.out))
            sets[lname + "_scheme1"] = (lambda n=lname, l=llvl: run_set(
                n + "_scheme1", base_cfg(l, 2, ["TomatoSalad", "CarrotBanana"], scheme="scheme1", max_steps=100, meta=metal),
                [(410, "bumper", 100), (411, "uniform", 100)], args.out))
    # beyond 128 slots / 256 cells: the third kernel instance (tools/make_levels.py huge_levels)
    lvl = lambda n: os.path.join(REPO, "cooking_zoo_amd", "utils", "level", n + ".json")
    mta = lambda n: os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", n + ".json")
    if os.path.exists(lvl("huge_32x32")):
        sets["huge_32x32_4agents"] = lambda: run_set(
            "huge_32x32_4agents",
            base_cfg(lvl("huge_32x32"), 4, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon", "CucumberOnion"], max_steps=100,
                     meta=mta("huge_32x32")),
            [(800, "bumper", 100), (801, "uniform", 100), (802, "mixed", 100)], args.out)
        sets["huge_32x32_scheme1"] = lambda: run_set(
            "huge_32x32_scheme1",
            base_cfg(lvl("huge_32x32"), 2, ["TomatoLettuceOnionSalad", "MashedCarrotBanana"], scheme="scheme1", max_steps=60,
                     meta=mta("huge_32x32")),
            [(810, "bumper", 60), (811, "uniform", 60)], args.out)
    if os.path.exists(lvl("huge_20x20")):
        sets["huge_20x20"] = lambda: run_set(
            "huge_20x20",
            base_cfg(lvl("huge_20x20"), 3, ["TomatoLettuceSalad", "MashedCarrotBanana", "TomatoSalad"], max_steps=150, meta=mta("huge_20x20")),
            [(820, "bumper", 150), (821, "uniform", 150), (822, "mixed", 150)], args.out)
    if os.path.exists(lvl("huge_objs_16x16")):
        sets["huge_objs_16x16"] = lambda: run_set(
            "huge_objs_16x16",
            base_cfg(lvl("huge_objs_16x16"), 3, ["TomatoLettuceSalad", "CarrotBanana", "AppleWatermelon"], max_steps=120,
                     meta=mta("huge_objs_16x16")),
            [(830, "bumper", 120), (831, "uniform", 120), (832, "mixed", 120)], args.out)
    # an ODD feature length (example meta with one more Tomato slot: F = 283): the last feature has no partner in the
    # two-features-per-lane observation stores
    meta_odd = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "example_odd.json")
    if os.path.exists(meta_odd):
        sets["odd_feature_length"] = lambda: run_set(
            "odd_feature_length", base_cfg("coop_test", 2, ["TomatoLettuceSalad", "CarrotBanana"], max_steps=120, meta=meta_odd),
            [(700, "bumper", 120), (701, "uniform", 120)], args.out)
        sets["odd_feature_length_1agent"] = lambda: run_set(
            "odd_feature_length_1agent", base_cfg("coop_test", 1, ["TomatoSalad"], scheme="scheme1", max_steps=80, meta=meta_odd),
            [(702, "bumper", 80)], args.out)
    sets["api_traces"] = lambda: api_traces(args.out)
    sets["layouts_ref"] = lambda: layout_draws(args.out)
    sets["aec_traces"] = lambda: aec_traces(args.out)
    sets["symbolic_traces"] = lambda: symbolic_traces(args.out)
    sets["custom_recipes"] = lambda: custom_recipes(args.out)          # (registers recipes in the reference: keep last)
    for name, fn in sets.items():
        if args.only and args.only != name:
            continue
        fn()




# ------------------------------------------------------------------------------------------------
# wrapper-level API traces (parallel_env dict-in / dict-out) and level-instantiation layouts
# ------------------------------------------------------------------------------------------------

def api_traces(out_dir):
    """(kwargs, seed, actions) -> the five dicts of every parallel_env.step, incl. a termination and a truncation.
    The parallel wrapper is tools/refshim's paraphrase of pettingzoo's (plumbing only)."""
    cases = [
        dict(seed=1, kwargs=dict(level="coop_test", meta_file="example", num_agents=1, max_steps=400,
                                 recipes=["TomatoLettuceSalad"], obs_spaces=["feature_vector"],
                                 end_condition_all_dishes=True, action_scheme="scheme3"),
             actions=[[int(c)] for c in "312442214141113331242131124444"]),
        dict(seed=7, kwargs=dict(level="coop_test", meta_file="example", num_agents=2, max_steps=5,
                                 recipes=["TomatoLettuceSalad", "CarrotBanana"], obs_spaces=["feature_vector", "feature_vector"],
                                 action_scheme="scheme3"),
             actions=[[1, 2], [3, 4], [0, 1], [2, 2], [4, 3]]),
        dict(seed=11, kwargs=dict(level="switch_test", meta_file="example", num_agents=2, max_steps=30,
                                  recipes=["TomatoSalad", "MashedCarrotBanana"], obs_spaces=["feature_vector", "feature_vector"],
                                  action_scheme="scheme1",
                                  reward_scheme={"recipe_reward": 10, "max_time_penalty": -3, "recipe_penalty": -7,
                                                 "recipe_node_reward": 1}),
             actions=[[int(a), int(b)] for a, b in np.random.default_rng(5).integers(0, 8, size=(30, 2))]),
    ]
    # agent despawn / respawn: the action dict follows env.agents, so it is drawn on the fly (and recorded)
    crowd = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", "crowded_6x5.json")
    metac = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "crowded_6x5.json")
    cases += [
        dict(seed=21, kwargs=dict(level="coop_test", meta_file="example", num_agents=2, max_steps=400,
                                  recipes=["TomatoLettuceSalad", "CarrotBanana"], obs_spaces=["feature_vector", "feature_vector"],
                                  action_scheme="scheme3", agent_respawn_rate=0.25, grace_period=2, agent_despawn_rate=0.15),
             actions=None, steps=140, policy_seed=3),
        dict(seed=22, kwargs=dict(level=crowd, meta_file=metac, num_agents=4, max_steps=400,
                                  recipes=["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"],
                                  obs_spaces=["feature_vector"] * 4, action_scheme="scheme1",
                                  agent_respawn_rate=0.2, grace_period=1, agent_despawn_rate=0.2,
                                  reward_scheme={"recipe_reward": 10, "max_time_penalty": -3, "recipe_penalty": -7,
                                                 "recipe_node_reward": 1}),
             actions=None, steps=160, policy_seed=4),
    ]
    out = []
    for case in cases:
        random.seed(case["seed"])
        np.random.seed(case["seed"])
        env = parallel_env(**case["kwargs"])
        obs, infos = env.reset()
        if case["actions"] is None:
            prng = np.random.default_rng(case["policy_seed"])
            n_act = int(env.action_space("player_0").n)
            case = dict(case, actions=[None] * case["steps"])
        kw = dict(case["kwargs"])
        for key in ("level", "meta_file"):                    # build-shipped files are referred to by stem
            if os.path.isabs(kw[key]):
                kw[key] = os.path.splitext(os.path.basename(kw[key]))[0]
        case = dict(case, kwargs=kw)
        rec = {"seed": case["seed"], "kwargs": case["kwargs"], "possible_agents": list(env.possible_agents),
               "reset_obs": {k: v.tolist() for k, v in obs.items()}, "reset_infos": {k: dict(v) for k, v in infos.items()},
               "obs_shape": list(env.observation_space("player_0").shape), "n_actions": int(env.action_space("player_0").n),
               "steps": []}
        for acts in case["actions"]:
            if not env.agents:
                break
            if acts is None:
                ad = {a: int(prng.integers(n_act)) for a in env.agents}
            else:
                ad = {f"player_{i}": a for i, a in enumerate(acts)}
            o, r, te, tr, inf = env.step(ad)
            rec["steps"].append({
                "actions": acts, "action_dict": ad, "obs": {k: v.tolist() for k, v in o.items()},
                "rewards": {k: float(v) for k, v in r.items()}, "terminations": {k: bool(v) for k, v in te.items()},
                "truncations": {k: bool(v) for k, v in tr.items()},
                "infos": {k: {kk: (vv.tolist() if hasattr(vv, "tolist") else vv) for kk, vv in v.items()} for k, v in inf.items()},
                "agents_after": list(env.agents)})
        out.append(rec)
    path = os.path.join(out_dir, "api_traces.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"[golden] api_traces: {len(out)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


def symbolic_traces(out_dir):
    """obs_spaces "symbolic" / "full" through the reference's parallel_env (cooking_env.py:271-288): per step the action
    dict and the object view of the world as plain data (cooking_zoo_amd.cooking_world.symbolic.physical_view: per
    class, in dict-key and list order, the `physical_state` attributes with object references as [class, index]).  In
    the first case player_0 is driven by the reference's own heuristic CookingAgent fed with that symbolic observation
    (its intended consumer), so the trace contains chopping, plating and a delivery."""
    import gzip
    from cooking_zoo_amd.cooking_world.symbolic import physical_view
    crowd = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", "crowded_6x5.json")
    metac = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "crowded_6x5.json")
    cases = [
        dict(name="coop_heuristic", seed=31, steps=150, policy_seed=1, heuristic="TomatoLettuceSalad",
             kwargs=dict(level="coop_test", meta_file="example", num_agents=2, max_steps=150,
                         recipes=["TomatoLettuceSalad", "CarrotBanana"], obs_spaces=["symbolic", "feature_vector"],
                         end_condition_all_dishes=True, action_scheme="scheme3")),
        dict(name="switch_scheme1", seed=32, steps=80, policy_seed=2, heuristic=None,
             kwargs=dict(level="switch_test", meta_file="example", num_agents=2, max_steps=80,
                         recipes=["MashedCarrotBanana", "TomatoSalad"], obs_spaces=["symbolic", "symbolic"],
                         action_scheme="scheme1")),
        dict(name="crowded_4agents", seed=33, steps=70, policy_seed=3, heuristic=None,
             kwargs=dict(level=crowd, meta_file=metac, num_agents=4, max_steps=70,
                         recipes=["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"],
                         obs_spaces=["symbolic", "feature_vector", "symbolic", "full"], action_scheme="scheme3")),
        dict(name="coop_full", seed=34, steps=25, policy_seed=4, heuristic=None,
             kwargs=dict(level="coop_test", meta_file="example", num_agents=1, max_steps=25, recipes=["TomatoLettuceSalad"],
                         obs_spaces=["full"], action_scheme="scheme3")),
    ]

    def plain(o):
        """one agent's observation as JSON-able data"""
        if isinstance(o, np.ndarray):
            return {"kind": "feature_vector", "sha256": hashlib.sha256(np.ascontiguousarray(o, dtype=np.float64).tobytes()).hexdigest()}
        if isinstance(o, dict) and "agent_location" in o:
            return {"kind": "full", "tensor_shape": list(o["feature_vector"].shape), "tensor_abs_sum": float(np.abs(o["feature_vector"]).sum()),
                    "agent_location": [int(v) for v in o["agent_location"]], "agent_location_dtype": str(o["agent_location"].dtype),
                    "goal_vector": [int(v) for v in o["goal_vector"]]}
        return {"kind": "symbolic", "view": physical_view(o)}

    out = []
    for case in cases:
        random.seed(case["seed"])
        np.random.seed(case["seed"])
        env = parallel_env(**case["kwargs"])
        obs, _ = env.reset()
        prng = np.random.default_rng(case["policy_seed"])
        n_act = int(env.action_space("player_0").n)
        bot = CookingAgent(case["heuristic"], "agent-1") if case["heuristic"] else None
        kw = dict(case["kwargs"])
        for key in ("level", "meta_file"):
            if os.path.isabs(kw[key]):
                kw[key] = os.path.splitext(os.path.basename(kw[key]))[0]
        rec = {"name": case["name"], "seed": case["seed"], "kwargs": kw, "reset_obs": {k: plain(v) for k, v in obs.items()}, "steps": []}
        delivered = False
        for t in range(case["steps"]):
            if not env.agents:
                break
            ad = {a: int(prng.integers(n_act)) for a in env.agents}
            if bot is not None and "player_0" in ad and prng.random() > 0.05:
                ad["player_0"] = int(bot.step(obs["player_0"]))
            obs, r, te, tr, inf = env.step(ad)
            delivered |= any(v > 1 for v in r.values())
            rec["steps"].append({"action_dict": ad, "obs": {k: plain(v) for k, v in obs.items()},
                                 "rewards": {k: float(v) for k, v in r.items()}, "agents_after": list(env.agents)})
        if bot is not None:
            assert delivered, "the heuristic agent was expected to deliver its dish inside the trace"
        out.append(rec)
        print(f"[golden] symbolic_traces/{case['name']}: {len(rec['steps'])} steps" + (" (dish delivered)" if delivered else ""))
    path = os.path.join(out_dir, "symbolic_traces.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(out, sort_keys=False).encode())
    print(f"[golden] symbolic_traces: {len(out)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


def custom_recipes(out_dir):
    """User recipes through the reference's own registry (recipe_drawer.py:19-35: register_recipe switches the env to
    RECIPE_STORE / NUM_GOALS, cooking_env.py:100-105): a 10-node graph (two plates under one Deliversquare, seven
    ingredient leaves -- more than the 8 nodes of the compact tables) and a node with TWO conditions (recipe.py:96-98
    checks every one): a Banana that is chopped but not mashed.  The fixture carries the flattened tables so that the
    build's own RecipeNode / register_recipe mirror can be checked against them."""
    from cooking_zoo.cooking_book import recipe_drawer as rd
    from cooking_zoo.cooking_book.recipe import Recipe as RefRecipe, RecipeNode as RefNode
    from cooking_zoo_amd.cooking_book.recipe import _condition_code
    assert not rd.RECIPE_STORE
    CH, MA, FRESH_BLEND = ("chop_state", ChopFoodStates.CHOPPED), ("blend_state", BlenderFoodStates.MASHED), ("blend_state", BlenderFoodStates.FRESH)
    leaf = lambda name, *conds: RefNode(root_type=name, id_num=rd.get_next_id(), name=name, conditions=list(conds))
    plate = lambda *kids: RefNode(root_type="Plate", id_num=rd.get_next_id(), name="Plate", contains=list(kids))
    deliver = lambda *kids: RefNode(root_type="Deliversquare", id_num=rd.get_next_id(), name="Deliversquare", contains=list(kids))
    feast_root = deliver(plate(leaf("Tomato", CH), leaf("Lettuce", CH), leaf("Apple", CH), leaf("Watermelon", CH)),
                         plate(leaf("Banana", CH, FRESH_BLEND), leaf("Carrot", MA), leaf("Bread", CH)))
    picky_root = deliver(plate(leaf("Banana", CH, FRESH_BLEND)))
    snack_root = deliver(plate(leaf("Bread", CH)))
    roots = {"FruitFeast": feast_root, "PickyBanana": picky_root, "BreadSnack": snack_root}
    for name, root in roots.items():
        rd.register_recipe(RefRecipe(root, rd.NUM_GOALS), name)
    names = list(rd.RECIPE_STORE.keys())
    # cooking_env.py:7 copies NUM_GOALS at import time (`from ... import NUM_GOALS`), so in the reference user recipes
    # only work when their nodes were created BEFORE cooking_env was first imported; emulate that order here
    import cooking_zoo.environment.cooking_env as ref_env_module
    ref_env_module.NUM_GOALS = rd.NUM_GOALS

    def flatten(g, max_nodes):
        wide = max_nodes > soa.NARROW_NODES
        out = [0] * (1 + (2 * soa.MAX_NODES if wide else soa.NARROW_NODES))
        out[0] = len(g.node_list)
        index = {id(n): j for j, n in enumerate(g.node_list)}
        last = {n.id_num: j for j, n in enumerate(g.node_list)}
        for j, n in enumerate(g.node_list):
            kids = sum(1 << index[id(c)] for c in n.contains)
            word = soa.class_node_id(n.name) | (_condition_code(n.conditions, n.name) << 8) | ((1 if last[n.id_num] == j else 0) << 24)
            if wide:
                out[1 + 2 * j], out[2 + 2 * j] = word, kids
            else:
                out[1 + j] = word | (kids << 16)
        return out
    graphs = [rd.RECIPE_STORE[n]() for n in names]
    assert max(len(g.node_list) for g in graphs) == 10
    table = [flatten(g, soa.MAX_NODES) for g in graphs]
    rs = {"recipe_reward": 20, "max_time_penalty": -5, "recipe_penalty": -40, "recipe_node_reward": 1}
    try:
        for set_name, agents, recipes, scheme, plan in (
                ("custom_wide_coop", 2, ["FruitFeast", "PickyBanana"], "scheme3", [(600, "bumper", 300), (601, "uniform", 300), (602, "bumper", 300)]),
                # (more recipes than agents: the third graph's marks live in record word 7)
                ("custom_wide_scheme1", 2, ["BreadSnack", "PickyBanana", "FruitFeast"], "scheme1", [(610, "bumper", 300), (611, "bumper", 300), (612, "uniform", 300)])):
            level = "coop_test" if agents <= 2 else os.path.join(REPO, "cooking_zoo_amd", "utils", "level", "crowded_6x5.json")
            meta = "example" if agents <= 2 else os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "crowded_6x5.json")
            cfg = base_cfg(level, agents, recipes, scheme=scheme, max_steps=300, meta=meta, reward_scheme=rs)
            eps = []
            for seed, policy, max_len in plan:
                ep = capture_episode(cfg, seed, policy, max_len, recipe_ids=[names.index(r) for r in recipes], recipe_nodes=soa.MAX_NODES)
                ep["seed"], ep["policy"] = seed, policy
                marks = ep["states"][:, soa.W_MARKS].astype(np.uint64) | (ep["states"][:, soa.W_MARKS_HI].astype(np.uint64) << np.uint64(32))
                print(f"   {set_name} seed={seed} policy={policy}: {episode_stats(ep)} distinct marks values {len(np.unique(marks))}")
                eps.append(ep)
            cfg = dict(cfg, recipe_table=table, recipe_nodes=soa.MAX_NODES, recipe_store=names, num_goals=int(rd.NUM_GOALS))
            save_set(set_name, cfg, eps, out_dir)
    finally:
        rd.RECIPE_STORE.clear()
        ref_env_module.NUM_GOALS = 0


def aec_traces(out_dir):
    """The raw AEC environment (cooking_env.py:215-241 step, :178-210 reset; gym id cookingZooEnv-v0) driven the
    agent_iter way: before every call the selected agent and its last() tuple, after it the bookkeeping dicts.  A dead
    agent is stepped with None (truncation empties the agent list, termination makes no progress -- SURVEY A.12(5)).
    The turn-taking helpers (agent_selector, AECEnv.last) are tools/refshim's paraphrase of pettingzoo's."""
    from cooking_zoo.environment.cooking_env import CookingEnvironment
    cases = [
        dict(seed=1, kwargs=dict(level="coop_test", meta_file="example", num_agents=1, max_steps=400,
                                 recipes=["TomatoLettuceSalad"], obs_spaces=["feature_vector"],
                                 end_condition_all_dishes=True, action_scheme="scheme3"),
             actions=[[int(c)] for c in "312442214141113331242131124444"]),
        dict(seed=7, kwargs=dict(level="coop_test", meta_file="example", num_agents=2, max_steps=5,
                                 recipes=["TomatoLettuceSalad", "CarrotBanana"], obs_spaces=["feature_vector", "feature_vector"],
                                 action_scheme="scheme3"),
             actions=[[1, 2], [3, 4], [0, 1], [2, 2], [4, 3], [1, 1]]),
        dict(seed=3, kwargs=dict(level="coexistence_test", meta_file="example", num_agents=2, max_steps=12,
                                 recipes=["TomatoSalad", "CarrotBanana"], obs_spaces=["feature_vector", "feature_vector"],
                                 action_scheme="scheme1",
                                 reward_scheme={"recipe_reward": 10, "max_time_penalty": -3, "recipe_penalty": -7,
                                                "recipe_node_reward": 1}),
             actions=[[int(a), int(b)] for a, b in np.random.default_rng(9).integers(0, 8, size=(14, 2))]),
    ]
    # agent despawn / respawn through the agent iterator: the agent list changes under the caller's feet (a despawned
    # agent is reported truncated once and must be acknowledged with step(None), cooking_env.py:216-224); actions are
    # drawn on the fly for whoever is selected and recorded in the calls
    crowd = os.path.join(REPO, "cooking_zoo_amd", "utils", "level", "crowded_6x5.json")
    metac = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files", "crowded_6x5.json")
    cases += [
        dict(seed=41, policy_seed=5, max_calls=420,
             kwargs=dict(level="coop_test", meta_file="example", num_agents=2, max_steps=400,
                         recipes=["TomatoLettuceSalad", "CarrotBanana"], obs_spaces=["feature_vector", "feature_vector"],
                         action_scheme="scheme3", agent_respawn_rate=0.3, grace_period=2, agent_despawn_rate=0.2), actions=None),
        dict(seed=42, policy_seed=6, max_calls=600,
             # (max_steps is never reached: the reference raises IndexError when it is while somebody is despawned)
             kwargs=dict(level=crowd, meta_file=metac, num_agents=4, max_steps=400,
                         recipes=["TomatoSalad", "TomatoLettuceSalad", "no_recipe", "MashedCarrotBanana"],
                         obs_spaces=["feature_vector"] * 4, action_scheme="scheme1",
                         agent_respawn_rate=0.25, grace_period=1, agent_despawn_rate=0.2), actions=None),
    ]
    out = []
    for case in cases:
        random.seed(case["seed"])
        np.random.seed(case["seed"])
        env = CookingEnvironment(**case["kwargs"])
        env.reset()
        kw = dict(case["kwargs"])
        for key in ("level", "meta_file"):                    # build-shipped files are referred to by stem
            if os.path.isabs(kw[key]):
                kw[key] = os.path.splitext(os.path.basename(kw[key]))[0]
        case = dict(case, kwargs=kw)
        prng = np.random.default_rng(case.get("policy_seed", 0))
        n_act = int(env.action_space("player_0").n)
        names = list(env.possible_agents)
        cursor = {a: 0 for a in names}
        rec = {"seed": case["seed"], "kwargs": case["kwargs"], "possible_agents": names,
               "after_reset": {"agent_selection": env.agent_selection, "agents": list(env.agents)}, "calls": []}
        dead_calls = 0
        while env.agents and dead_calls < (3 if case["actions"] is not None else 10 ** 9) and len(rec["calls"]) < case.get("max_calls", 10 ** 9):
            agent = env.agent_selection
            obs, cum, term, trunc, info = env.last()
            if term or trunc:
                action = None
                dead_calls += 1
            elif case["actions"] is None:
                action = int(prng.integers(n_act))
            else:
                idx = names.index(agent)
                if cursor[agent] >= len(case["actions"]):
                    break
                action = case["actions"][cursor[agent]][idx]
                cursor[agent] += 1
            env.step(action)
            rec["calls"].append({
                "agent": agent, "action": action,
                "last": {"obs": np.asarray(obs).tolist(), "reward": float(cum), "termination": bool(term), "truncation": bool(trunc),
                         "info": {kk: (vv.tolist() if hasattr(vv, "tolist") else vv) for kk, vv in info.items()}},
                "after": {"agent_selection": env.agent_selection if env.agents else None, "agents": list(env.agents),
                          "rewards": {k: float(v) for k, v in env.rewards.items()},
                          "cumulative": {k: float(v) for k, v in env._cumulative_rewards.items()},
                          "terminations": {k: bool(v) for k, v in env.terminations.items()},
                          "truncations": {k: bool(v) for k, v in env.truncations.items()}, "t": int(env.t)}})
        out.append(rec)
    import gzip
    path = os.path.join(out_dir, "aec_traces.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())
    print(f"[golden] aec_traces: {len(out)} cases, {sum(len(r['calls']) for r in out)} calls, {os.path.getsize(path) / 1024:.0f} KiB")


def layout_draws(out_dir):
    """random.seed(s) -> the layouts the reference instantiates (two consecutive draws), for the host loader test."""
    from cooking_zoo.cooking_world.cooking_world import CookingWorld
    from cooking_zoo.cooking_world.actions import ActionScheme3
    L = os.path.join(REPO, "cooking_zoo_amd", "utils", "level")
    M = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files")
    cases = [("coop_test", "example", 2), ("coop_test", "example", 1), ("coexistence_test", "example", 2),
             ("switch_test", "example", 2), (os.path.join(L, "large_16x16.json"), os.path.join(M, "large_16x16.json"), 4),
             (os.path.join(L, "crowded_6x5.json"), os.path.join(M, "crowded_6x5.json"), 3)]
    out = []
    for level, meta, A in cases:
        for seed in (0, 1, 2, 3, 12345):
            random.seed(seed)
            draws = []
            for _ in range(2):
                w = CookingWorld(ActionScheme3, meta)
                w.load_level(level, A)
                statics = {k: [[o.location[0], o.location[1]] for o in v] for k, v in w.world_objects.items()
                           if issubclass(wo.StringToClass[k], StaticObject)}
                dyn = [[k, [[o.location[0], o.location[1]] for o in v]] for k, v in w.world_objects.items()
                       if issubclass(wo.StringToClass[k], DynamicObject) and v]
                draws.append({"width": w.width, "height": w.height, "statics": statics, "dynamics": dyn,
                              "agents": [[a.location[0], a.location[1]] for a in w.agents]})
            out.append({"level": os.path.splitext(os.path.basename(level))[0],
                        "meta": os.path.splitext(os.path.basename(meta))[0], "num_agents": A, "seed": seed, "draws": draws,
                        "next_random": random.random()})
    path = os.path.join(out_dir, "layouts_ref.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"[golden] layouts_ref: {len(out)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
