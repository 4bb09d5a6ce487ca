"""Level / meta file loading and level instantiation (host side, runs at reset only).

File formats are the reference's (reference `engine/load_level.py:11-52`,
`engine/parsing.py:5-151`).  Instantiation draws from Python's `random` in exactly the reference's
call order -- one `random.random()` per placement attempt of an OPTIONAL object, then
`random.sample(X, 1)`, `random.sample(Y, 1)` -- so `random.seed(s)` yields the reference's layouts
(pinned by tests/test_level_loading.py against layouts captured from the reference).
The product of instantiation is a flat `Layout`, not an object graph.
"""
from __future__ import annotations

import json
import os
import random as _random
from pathlib import Path

from cooking_zoo_amd import soa
from cooking_zoo_amd.cooking_world.layout import Layout

_UTILS = Path(os.path.dirname(os.path.realpath(__file__))).parent.parent / "utils"


def _resolve(name, sub):
    return name if name.endswith(".json") else str(_UTILS / sub / f"{name}.json")


def load_meta_file(meta_file) -> dict:
    """Ordered {class name: max count}; fixes the feature-vector layout (load_level.py:39-52)."""
    with open(_resolve(meta_file, "meta_files")) as f:
        meta_object = json.load(f)
    return {list(d.keys())[0]: list(d.values())[0] for d in meta_object}


def load_level_file(level) -> dict:
    with open(_resolve(level, "level")) as f:
        return json.load(f)


def level_max_dyn(level_object) -> int:
    """Slot capacity a level can need: every object present, every Bread cloned once."""
    n = 0
    for entry in level_object["DYNAMIC_OBJECTS"]:
        (name, spec), = entry.items()
        n += spec["COUNT"] * (2 if name == "Bread" else 1)
    return max(n, 1)          # the record always has at least one (possibly unused) slot


def instantiate(level_object, meta: dict, num_agents: int, rng=_random) -> Layout:
    """One draw of the level (parsing.py:5-151).  `rng` defaults to the global `random` module,
    like the reference; pass a `random.Random` to keep the global stream untouched."""
    rows = level_object["LEVEL_LAYOUT"].splitlines()
    height = len(rows)
    width = len(rows[-1])
    # parse_level_layout: '-' = Counter, anything else = Floor; width from the last row (parsing.py:8-17)
    cell_type = {}
    lists = {}                       # class name -> list of cells in creation order
    for y, line in enumerate(rows):
        for x, ch in enumerate(line):
            name = "Counter" if ch == "-" else "Floor"
            cell_type[(x, y)] = name
            lists.setdefault(name, []).append((x, y))
    loaded = {}

    def take_meta(name):
        if meta[name] <= loaded.get(name, 0):
            raise ValueError(f"Too many {name} objects loaded")
        loaded[name] = loaded.get(name, 0) + 1

    def draw(spec):
        x = rng.sample(spec["X_POSITION"], 1)[0]
        y = rng.sample(spec["Y_POSITION"], 1)[0]
        if x < 0 or y < 0 or x > width or y > height:
            raise ValueError(f"Position {x} {y} is out of bounds set by the level layout!")
        return x, y

    # parse_static_objects (parsing.py:21-76): a static replaces the Counter or Floor at the drawn cell
    for entry in level_object["STATIC_OBJECTS"]:
        name = list(entry.keys())[0]
        spec = entry[name]
        for _ in range(spec["COUNT"]):
            time_out = 0
            while True:
                if "OPTIONAL" in spec and spec["OPTIONAL"] <= rng.random():
                    break
                x, y = draw(spec)
                here = cell_type.get((x, y))
                if here in ("Counter", "Floor"):
                    take_meta(name)
                    lists[here].remove((x, y))
                    cell_type[(x, y)] = name
                    lists.setdefault(name, []).append((x, y))
                    break
                time_out += 1
                if time_out > 10000:
                    raise ValueError(f"Can't find valid position for object: {entry} in {time_out} steps")

    # parse_dynamic_objects (parsing.py:79-115): needs a plain Counter with nothing on it, not excluded
    excluded = level_object["DYNAMIC_EXCLUDED_POSITIONS"]
    dyn = {}                         # class name -> [(x, y)] in creation order; dict order = key order
    occupied = set()
    for entry in level_object["DYNAMIC_OBJECTS"]:
        name = list(entry.keys())[0]
        spec = entry[name]
        for _ in range(spec["COUNT"]):
            time_out = 0
            while True:
                if "OPTIONAL" in spec and spec["OPTIONAL"] <= rng.random():
                    break
                x, y = draw(spec)
                if cell_type.get((x, y)) == "Counter" and (x, y) not in occupied and [x, y] not in excluded:
                    take_meta(name)
                    dyn.setdefault(name, []).append((x, y))
                    occupied.add((x, y))
                    break
                time_out += 1
                if time_out > 10000:
                    raise ValueError(f"Can't find valid position for object: {entry} in {time_out} steps")

    # parse_agents (parsing.py:118-151): a Floor cell no other agent stands on
    agents = []
    agent_idx = 0
    done = False
    for agent_object in level_object["AGENTS"]:
        if done:
            break
        for _ in range(agent_object["MAX_COUNT"]):
            agent_idx += 1
            if agent_idx > num_agents:
                done = True
                break
            time_out = 0
            while True:
                x, y = draw(agent_object)
                if (x, y) not in agents and cell_type.get((x, y)) == "Floor":
                    take_meta("Agent")
                    agents.append((int(x), int(y)))
                    break
                time_out += 1
                if time_out > 1000:
                    raise ValueError(f"Can't find valid position for agent: {agent_object} in {time_out} steps")

    if len(lists.get("Switch", [])) > 1:
        # every LinkedObject shares group None: a second Switch makes the reference raise
        # AttributeError ('Switch' has no switch_state) on the first press
        raise ValueError("levels with more than one Switch are not supported (the reference crashes on them)")
    if len(agents) > soa.MAX_AGENTS:
        raise ValueError("at most 4 agents (reference COLORS has 4 entries, cooking_world.py:21)")

    cells = [0] * (width * height)
    for (x, y), name in cell_type.items():
        cells[y * width + x] = soa.STATIC_CLASSES.index(name)
    static_lists = {k: [y * width + x for (x, y) in v] for k, v in lists.items()}
    dyn_classes = [(soa.DYNAMIC_CLASSES.index(k), len(v)) for k, v in dyn.items()]
    dyn_xy = [p for v in dyn.values() for p in v]
    return Layout(width, height, cells, static_lists, dyn_classes, dyn_xy, agents)
