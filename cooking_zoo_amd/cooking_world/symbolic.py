"""Host-side materialiser: one flat env record -> the object view the reference calls the "symbolic" observation.

The reference serves `obs_spaces=["symbolic"]` as a deep copy of `world.world_objects` plus the agent list
(`cooking_env.py:279-284`): a dict  class name -> list of Python objects, which is what its heuristic agent walks
(`cooking_agents/base_agent.py:53-58,64-199`, `cooking_agent.py:9-122`: `observation["Agent"]`, `observation["Floor"]`,
`observation[appliance]`, `.location`, `.content`, `.holding`, `.chop_state`, `.blend_state`).  On the device there are no
objects, only the record of cooking_zoo_amd/soa.py; this module rebuilds the view from it:

* keys in `world_objects` first-insertion order (layout characters, statics in level-file order, dynamic classes in
  level-file order), then `"Agent"`; list order inside a class = the reference's list order (= slot order);
* every object carries the attributes of the reference's `physical_state` list (`abstract_classes.py:68-92`) that exist
  for its class: `location`, `walkable`, `movable`, `content`, `status`, `toggle`, `switch_active`, `free`,
  `chop_state`, `blend_state`, `holding`, `orientation` (+ `name`, `color` for agents);
* references are real references inside one view: `agent.holding` is the very object listed under its class, a held
  plate's `content` are the listed food objects, a static's `content` are the objects lying on it.

Not materialised: `unique_id` (a process-global counter upstream) and `interacts_with` (transient, not part of the step
state).  `content` of a walkable cell (Floor, Switch, Block) is "the agents standing there, in agent order"; the
reference's list can go stale when three or more agents shared a cell (it clears the whole list when ONE agent
leaves, `action_scheme3.py:31`), which no consumer of the symbolic view reads.

`physical_view` turns a view (this module's, or the reference's own objects) into plain JSON-able data with object
references replaced by (class, index) pairs; the golden fixtures under tests/golden/symbolic_*.json are that form.
"""
from __future__ import annotations

from collections import defaultdict

from cooking_zoo_amd import soa
from cooking_zoo_amd.cooking_world.constants import ActionObjectState, BlenderFoodStates, ChopFoodStates

COLORS = ['blue', 'magenta', 'yellow', 'green']          # cooking_world.py:21
# state_length() of every reference class (world_objects.py): the depth of the (constant zero) "full" tensor
STATE_LENGTH = {"Agent": 5, "Apple": 2, "Banana": 2, "Blender": 1, "Block": 1, "Bread": 3, "Carrot": 2, "Counter": 1,
                "Cucumber": 2, "Cutboard": 1, "Deliversquare": 1, "Floor": 1, "Lettuce": 2, "Onion": 2, "Plate": 1,
                "Switch": 1, "Tomato": 2, "Watermelon": 2}
GRAPH_REPRESENTATION_LENGTH = sum(STATE_LENGTH.values())     # cooking_env.py:110


class WorldObject:
    """A plain attribute bag; `type(obj).__name__` is the reference's class name."""
    __slots__ = ("__dict__",)

    def name(self):
        return type(self).__name__

    def done(self):
        """ChopFood.done / BlenderFood.done (abstract_classes.py:256-257,275-277)"""
        chopped = getattr(self, "chop_state", None) == ChopFoodStates.CHOPPED
        mashed = getattr(self, "blend_state", None) == BlenderFoodStates.MASHED
        return chopped or mashed

    def __repr__(self):
        return f"{type(self).__name__}{getattr(self, 'location', '')}"


_CLASSES = {n: type(n, (WorldObject,), {}) for n in soa.STATIC_CLASSES + soa.DYNAMIC_CLASSES + ["Agent"]}


def _make(_class_name, **attrs):
    o = _CLASSES[_class_name]()
    o.__dict__.update(attrs)
    return o


def materialize(dims: soa.Dims, layout, record):
    """-> defaultdict(list): class name -> objects, "Agent" last.  `layout` supplies the list orders (static_lists,
    dyn_classes: cooking_world/layout.py); `record` is the env's current record (uint32[RW])."""
    W = dims.W
    cells = soa.record_cells(dims, record)
    out = defaultdict(list)
    static_at = {}
    for name, cell_list in layout.static_lists.items():
        for c in cell_list:
            b = int(cells[c])
            o = _make(name, location=(c % W, c // W), movable=False, content=[],
                      walkable=(name in ("Floor", "Switch")) or (name == "Block" and bool(b & soa.CELL_WALK)))
            if name in ("Cutboard", "Blender"):
                o.status = ActionObjectState.READY if b & soa.CELL_READY else ActionObjectState.NOT_USABLE
            if name == "Blender":
                o.toggle = bool(b & soa.CELL_TOGGLE)
            if name == "Switch":
                o.switch_active = bool(b & soa.CELL_ACTIVE)
            out[name].append(o)
            static_at[c] = o
    by_slot = {}
    inside = []                                   # (plate slot, seq, object)
    for cls, _n in layout.dyn_classes:
        name = soa.DYNAMIC_CLASSES[cls]
        base = layout.slot_base[cls]
        for s in range(base, base + layout.slot_cap[cls]):
            x, y, c, flags = soa.unpack_dyn0(record[dims.dyn0_word0 + s])
            if not flags & soa.DYN_ALIVE:
                continue
            o = _make(name, location=(x, y), movable=True, walkable=False, free=bool(flags & soa.DYN_FREE))
            if cls == soa.PLATE:
                o.content = []
            else:
                o.chop_state = ChopFoodStates.CHOPPED if flags & soa.DYN_CHOPPED else ChopFoodStates.FRESH
                if cls in soa.BLENDER_FOOD:
                    o.blend_state = BlenderFoodStates.MASHED if flags & soa.DYN_MASHED else BlenderFoodStates.FRESH
            out[name].append(o)
            by_slot[s] = o
            plate, seq = soa.unpack_dyn1(record[dims.dyn1_word0 + s])
            if plate >= 0:
                inside.append((plate, seq, o))
    for plate, _seq, o in sorted(inside, key=lambda t: (t[0], t[1])):
        by_slot[plate].content.append(o)
    agents = []
    held = set()
    for a in range(dims.A):
        x, y, orient, h = soa.unpack_agent(record[soa.AGENT_WORD0 + a])
        ag = _make("Agent", location=(x, y), orientation=orient, holding=by_slot.get(h) if h >= 0 else None, movable=False,
                   walkable=False, color=COLORS[a], name=f"agent-{a + 1}")
        agents.append(ag)
        if h >= 0:
            held.add(h)
    in_plate = {id(o) for _p, _s, o in inside}
    for s in sorted(by_slot):                     # a static's content: what lies on it directly, in slot order
        o = by_slot[s]
        if s in held or id(o) in in_plate:
            continue
        st = static_at.get(o.location[1] * W + o.location[0])
        if st is not None and not st.walkable:
            st.content.append(o)
    for ag in agents:                             # walkable cells hold the agents standing on them
        st = static_at.get(ag.location[1] * W + ag.location[0])
        if st is not None and type(st).__name__ in ("Floor", "Switch", "Block"):
            st.content.append(ag)
    out["Agent"] = agents
    return out


_ENUM_ATTRS = ("chop_state", "blend_state", "status")
_PLAIN_ATTRS = ("free", "toggle", "walkable", "movable", "switch_active", "orientation", "name", "color")


def physical_view(objects, skip_walkable_content=False):
    """{class: [objects]} (this module's or the reference's own) -> {class: [{attr: plain value}]}, empty classes left
    out, object references as [class, index in that class's list]."""
    index = {}
    for cls, lst in objects.items():
        for i, o in enumerate(lst):
            index[id(o)] = [cls, i]
    out = {}
    for cls, lst in objects.items():
        if not lst:
            continue
        rows = []
        for o in lst:
            row = {"location": [int(v) for v in o.location]}
            for attr in _ENUM_ATTRS:
                if hasattr(o, attr):
                    row[attr] = getattr(o, attr).value
            for attr in _PLAIN_ATTRS:
                if hasattr(o, attr):
                    v = getattr(o, attr)
                    if callable(v):                   # the reference's `name` of non-agents is a method
                        continue
                    row[attr] = v if isinstance(v, str) else (bool(v) if isinstance(v, bool) else int(v))
            if hasattr(o, "content") and not (skip_walkable_content and cls in ("Floor", "Switch", "Block")):
                row["content"] = [index[id(c)] for c in o.content]
            if hasattr(o, "holding"):
                row["holding"] = index[id(o.holding)] if o.holding is not None else None
            rows.append(row)
        out[cls] = rows
    return out
