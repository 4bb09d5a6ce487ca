"""Recipe graphs, host side.

Same public shape as the reference's `RecipeNode` / `Recipe` (cooking_book/recipe.py:12-40,74-75) so
that user code building custom recipes keeps working, plus `flatten()` which turns a graph into the
fixed-width node table the kernels evaluate (recipe.py:77-104 is evaluated on the device, not here).
"""
from __future__ import annotations

import numpy as np

from cooking_zoo_amd import soa
from cooking_zoo_amd.cooking_world.constants import ChopFoodStates, BlenderFoodStates


class RecipeNode:
    def __init__(self, root_type, id_num, name, parent=None, conditions=None, contains=None, objects_to_seek=None):
        self.parent = parent
        self.marked = False
        self.id_num = id_num
        self.root_type = root_type
        self.conditions = conditions or []
        self.contains = contains or []
        self.world_objects = []
        self.name = name
        self.child_nodes = objects_to_seek or []

    def is_leaf(self):
        return not bool(self.contains)


_BLENDER_FOOD = ("Carrot", "Banana")                     # the only classes with a blend_state (abstract_classes.py:257-264)


def _condition_code(conditions, class_name=""):
    """(attr, value) pairs -> one device condition code (include/cookingzoo.h, cz_load_recipes).

    An object's recipe-visible state is two bits (chopped, mashed), so ANY number of conditions on chop_state /
    blend_state (recipe.py:96-98 loops over all of them) is exactly a set of accepted states
    s = chopped | mashed << 1.  One condition keeps the legacy codes 1..4; several become 0x10 | accept mask.
    Deviations: a blend_state condition on a class without that attribute (the reference raises AttributeError as soon
    as such an object exists) and BlenderFoodStates.IN_PROGRESS (never observable between steps) match nothing."""
    if not conditions:
        return soa.COND_NONE
    accept = 0xF
    single = None
    for attr, value in conditions:
        value = getattr(value, "value", value)
        if attr == "chop_state" and value == ChopFoodStates.CHOPPED.value:
            accept &= 0xA; single = soa.COND_CHOPPED
        elif attr == "chop_state" and value == ChopFoodStates.FRESH.value:
            accept &= 0x5; single = soa.COND_NOT_CHOPPED
        elif attr == "blend_state" and value == BlenderFoodStates.MASHED.value:
            accept &= 0xC if class_name in _BLENDER_FOOD else 0; single = soa.COND_MASHED
        elif attr == "blend_state" and value == BlenderFoodStates.FRESH.value:
            accept &= 0x3 if class_name in _BLENDER_FOOD else 0; single = soa.COND_NOT_MASHED
        elif attr == "blend_state" and value == BlenderFoodStates.IN_PROGRESS.value:
            accept = 0; single = None
        else:
            raise ValueError(f"unsupported recipe condition {(attr, value)!r}")
    if len(conditions) == 1 and single is not None:
        return single
    return 0x10 | accept


class Recipe:
    def __init__(self, root_node: RecipeNode, num_goals: int, name: str = ""):
        self.root_node = root_node
        self.name = name
        self.node_list = [root_node] + self.expand_child_nodes(root_node)
        if len(self.node_list) > soa.MAX_NODES:
            raise ValueError(f"recipe graphs are limited to {soa.MAX_NODES} nodes")
        self.num_goals = num_goals
        self.marks = 0           # bit j = node_list[j].marked, refreshed from the device after each step
        self.goal_encoding = self.goals_completed(num_goals)

    def expand_child_nodes(self, node: RecipeNode):
        child_nodes = []
        for child in node.contains:
            child_nodes.extend(self.expand_child_nodes(child))
        return node.contains + child_nodes

    def set_marks(self, marks: int):
        self.marks = int(marks)
        for j, node in enumerate(self.node_list):
            node.marked = bool((self.marks >> j) & 1)

    def goals_completed(self, num_goals):
        goals = np.zeros(num_goals, dtype=np.int32)
        for j, node in enumerate(self.node_list):
            goals[node.id_num] = int(not (self.marks >> j) & 1)
        return goals

    def completed(self):
        return bool(self.marks & 1)

    def flatten(self, max_nodes: int = soa.NARROW_NODES) -> np.ndarray:
        """The node table row the kernels evaluate.  `counts` marks the last node carrying a given goal id (goals[id] is
        overwritten by later nodes, recipe.py:38-39, so only that one contributes to sum(goals)).
        max_nodes = 8  -> uint32[9]:  n_nodes, then per node  cls | cond<<8 | child_mask<<16 | counts<<24
        max_nodes = 16 -> uint32[33]: n_nodes, then per node two words  cls | cond<<8 | counts<<24,  child_mask (16 bits)"""
        if len(self.node_list) > max_nodes:
            raise ValueError(f"recipe graph has {len(self.node_list)} nodes, the table row holds {max_nodes}")
        wide = max_nodes > soa.NARROW_NODES
        out = np.zeros(1 + (2 * soa.MAX_NODES if wide else soa.NARROW_NODES), dtype=np.uint32)
        out[0] = len(self.node_list)
        index = {id(n): j for j, n in enumerate(self.node_list)}
        last_with_id = {}
        for j, n in enumerate(self.node_list):
            last_with_id[n.id_num] = j
        for j, n in enumerate(self.node_list):
            child_mask = 0
            for c in n.contains:
                child_mask |= 1 << index[id(c)]
            counts = 1 if last_with_id[n.id_num] == j else 0
            word = soa.class_node_id(n.name) | (_condition_code(n.conditions, n.name) << 8) | (counts << 24)
            if wide:
                out[1 + 2 * j], out[2 + 2 * j] = word, child_mask
            else:
                out[1 + j] = word | (child_mask << 16)
        return out
