"""Flattened (struct-of-arrays) data model of one CookingZoo world.

This module is the single definition of the binary contract between the Python
host, the C-ABI (include/cookingzoo.h), the HIP kernels and the test oracle.
It replaces the reference's object graph (`world_objects.py`, `abstract_classes.py`)
with fixed-width integer fields; every field cites the reference attribute it stands for.

Per-env *record* (array of little-endian u32 words, `record_words(cfg)` long):

  word 0      t                      cooking_env.py:244  (self.t)
  word 1      marks                  recipe.py:16  node.marked, bit (8*r + j) = node j of recipe r
  word 2      layout_id              index into the layout pool (initial state + obs descriptor table)
  word 3      status                 bit0 = episode over (terminated|truncated), awaiting reset
  word 4      episode                episodes completed by this env slot
  word 5      recipe ids             4 x u8, index into the recipe table, 0xFF = unused
  word 6      layout pool            base | count<<16: the slice of the layout pool this env redraws from on
                                     auto-reset (0 = whole pool); lets one batch mix levels
  word 7      marks of recipes 2, 3 when the recipe tables are wide (more than 8 nodes per graph: 16 bits per recipe,
              recipes 0, 1 in word 1); otherwise 0
  word 8..11  agents[4]              x | y<<8 | orientation<<16 | (held slot+1)<<24   world_objects.py:776-783
  word 12..19 ret[4]                 float64 running return of the episode in flight, per agent slot
                                     (cooking_env.py:265 _cumulative_rewards; device-side statistics only)
  then        cells[CW]              W*H bytes, 4 per word:  type | READY<<3 | TOGGLE<<4 | ACTIVE<<5 | WALK<<6
  then        dyn0[D]                x | y<<8 | cls<<16 | flags<<24
  then        dyn1[D]                (plate slot+1) | seq<<8     (ContentObject.content membership + position)

Static `content` lists are not stored: for a non-walkable cell they are, by construction, the
objects at that cell that are neither inside a plate nor held (the golden
generator asserts this against the reference at every step).
"""
from __future__ import annotations

import numpy as np

# ---- static cell types (world_objects.py:17,57,101,144,195,242,314)
FLOOR, COUNTER, DELIVERSQUARE, SWITCH, BLOCK, CUTBOARD, BLENDER = range(7)
STATIC_CLASSES = ["Floor", "Counter", "Deliversquare", "Switch", "Block", "Cutboard", "Blender"]
CELL_TYPE_MASK = 0x07
CELL_READY = 0x08      # ActionObject.status == READY          abstract_classes.py:99
CELL_TOGGLE = 0x10     # ToggleObject.toggle                   abstract_classes.py:110
CELL_ACTIVE = 0x20     # Switch.switch_active                  world_objects.py:150
CELL_WALK = 0x40       # Block.walkable                        world_objects.py:199,216

# ---- dynamic classes (world_objects.py:386-771)
PLATE, ONION, TOMATO, LETTUCE, CARROT, CUCUMBER, BANANA, APPLE, WATERMELON, BREAD = range(10)
DYNAMIC_CLASSES = ["Plate", "Onion", "Tomato", "Lettuce", "Carrot", "Cucumber", "Banana", "Apple",
                   "Watermelon", "Bread"]
BLENDER_FOOD = (CARROT, BANANA)                     # abstract_classes.py:257 subclasses
DYN_ALIVE, DYN_CHOPPED, DYN_MASHED, DYN_FREE = 1, 2, 4, 8
PLATE_MAX_CONTENT = 64                              # world_objects.py:391

# feature lengths (feature_vector_length of every class)
FEATURE_LEN = {"Floor": 0, "Counter": 3, "Deliversquare": 3, "Switch": 4, "Block": 4, "Cutboard": 3,
               "Blender": 3, "Plate": 3, "Onion": 5, "Tomato": 5, "Lettuce": 5, "Carrot": 6,
               "Cucumber": 5, "Banana": 6, "Apple": 5, "Watermelon": 5, "Bread": 5, "Agent": 7}

# ---- unified class ids used by recipe nodes
NODE_CLS_STATIC0 = 0          # + static type
NODE_CLS_DYN0 = 16            # + dynamic class
NODE_CLS_NONE = 255           # class with no objects (e.g. an unknown name)
COND_NONE, COND_CHOPPED, COND_MASHED, COND_NOT_CHOPPED, COND_NOT_MASHED = range(5)
MAX_NODES = 16               # node capacity of a recipe graph (the reference has none, recipe.py:29-34)
NARROW_NODES = 8             # graphs up to this size use the compact tables: rows of 9 words, marks 8 bits per recipe in
                             # record word 1; larger ones ("wide"): rows of 33 words, marks 16 bits per recipe in words 1 and 7
MAX_RECIPES_PER_ENV = 4
MAX_AGENTS = 4

# ---- observation descriptor: one u32 per feature =  (image halfword index * 2) | (axis code * 4) << 16
# The kernel keeps, per env, an LDS "image" of halfwords; every halfword is a byte offset into a 256-entry table of
# doubles (lut):  lut[i] = (i-(W-1))/W for i < 2W-1,  lut[63+i] = (i-(H-1))/H,  lut[126] = 0.0,  lut[127] = 1.0,
# lut[128..255] = 0.0 (the "absent" zone: every halfword of a dead slot points at entry 255, and stays inside the
# zone after an agent coordinate is subtracted).  Image layout (halfword indices):
#   slot s    IMG_OBJ0 + 6 s : x+W-1 | y+63+H-1 | 126+!done | 126+chopped | 126+mashed | 127
#   cell c    IMG_CELL0 + 4 c: x+W-1 | y+63+H-1 | 126+(switch_active or block walkable) | 127
#   agent a   IMG_AG0 + 8 a  : x+W-1 | y+63+H-1 | 126+(o==1) | .. | 126+(o==4) | 127 | pad
#   IMG_ZERO                 : 255
# A feature value for observer a is  lut[image[hw] - sub[a][code]]  with the axis code choosing what is subtracted:
#   0 nothing, 1 ax, 2 ay, 4+2j / 5+2j: ax / ay unless a == j (an agent's own position is absolute, cooking_env.py:366-368)
IMG_OBJ0, IMG_CELL0, IMG_AG0, IMG_ZERO, IMG_HALFWORDS = 0, 768, 1792, 1824, 1826
# Batches beyond 128 slots or 256 cells run on the "huge" kernel instance (up to 255 slots, 32 x 32 cells), whose image
# has room for 256 slots and 1024 cells (cz_kernels.h Img<16>):
HUGE_IMG_OBJ0, HUGE_IMG_CELL0, HUGE_IMG_AG0, HUGE_IMG_ZERO = 0, 1536, 5632, 5664
MAX_DYN, MAX_W, MAX_H = 255, 32, 32          # (slot + 1) is an 8-bit field; the quotient table holds 2W-1 <= 63 and 2H-1 <= 63 entries
AX_NONE, AX_X, AX_Y = 0, 1, 2
LUT_Y0, LUT_ZERO, LUT_ONE, LUT_ABSENT, LUT_SIZE = 63, 126, 127, 255, 256


def desc_word(hw, code=0):
    return ((hw * 2) & 0xFFFF) | ((code * 4) << 16)


def ax_self(agent, axis):
    """axis code of agent `agent`'s own x (axis 0) / y (axis 1) feature"""
    return 4 + 2 * agent + axis


HDR_WORDS = 8
W_T, W_MARKS, W_LAYOUT, W_STATUS, W_EPISODE, W_RECIPES, W_POOL, W_RES1 = range(8)
W_MARKS_HI = W_RES1          # wide recipe tables only: marks of recipes 2 and 3
AGENT_WORD0 = HDR_WORDS
RET_WORD0 = AGENT_WORD0 + 4          # 4 float64 = 8 words
STATUS_DONE, STATUS_TERM, STATUS_TRUNC = 1, 2, 4


def class_node_id(name: str) -> int:
    if name in STATIC_CLASSES:
        return NODE_CLS_STATIC0 + STATIC_CLASSES.index(name)
    if name in DYNAMIC_CLASSES:
        return NODE_CLS_DYN0 + DYNAMIC_CLASSES.index(name)
    return NODE_CLS_NONE


class Dims:
    """Capacities of one batch: grid W x H, D dynamic slots, A agents, F features."""

    def __init__(self, width: int, height: int, max_dyn: int, num_agents: int, feat_len: int):
        self.W, self.H, self.D, self.A, self.F = int(width), int(height), int(max_dyn), int(num_agents), int(feat_len)
        self.C = self.W * self.H
        self.CW = (self.C + 3) // 4
        self.cells_word0 = RET_WORD0 + 2 * MAX_AGENTS
        self.dyn0_word0 = self.cells_word0 + self.CW
        self.dyn1_word0 = self.dyn0_word0 + self.D
        used = self.dyn1_word0 + self.D
        self.RW = (used + 15) // 16 * 16            # 64-byte multiple

    def as_tuple(self):
        return (self.W, self.H, self.D, self.A, self.F)

    @property
    def huge(self):
        return self.D > 128 or self.C > 256

    def img_layout(self):
        """(OBJ0, CELL0, AG0, ZERO) halfword bases of the LDS image this batch's kernel instance keeps"""
        if self.huge:
            return HUGE_IMG_OBJ0, HUGE_IMG_CELL0, HUGE_IMG_AG0, HUGE_IMG_ZERO
        return IMG_OBJ0, IMG_CELL0, IMG_AG0, IMG_ZERO


def pack_agent(x, y, orient, holding_slot):
    """holding_slot: -1 for empty hands."""
    return (x & 0xFF) | ((y & 0xFF) << 8) | ((orient & 0xFF) << 16) | (((holding_slot + 1) & 0xFF) << 24)


def unpack_agent(w):
    w = int(w)
    return (w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF, ((w >> 24) & 0xFF) - 1)


def pack_dyn0(x, y, cls, flags):
    return (x & 0xFF) | ((y & 0xFF) << 8) | ((cls & 0xFF) << 16) | ((flags & 0xFF) << 24)


def unpack_dyn0(w):
    w = int(w)
    return (w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF, (w >> 24) & 0xFF)


def pack_dyn1(container_slot, seq):
    """container_slot: -1 when not inside a plate."""
    return ((container_slot + 1) & 0xFF) | ((seq & 0xFF) << 8)


def unpack_dyn1(w):
    w = int(w)
    return ((w & 0xFF) - 1, (w >> 8) & 0xFF)


def new_record(dims: Dims) -> np.ndarray:
    return np.zeros(dims.RW, dtype=np.uint32)


def record_cells(dims: Dims, rec: np.ndarray) -> np.ndarray:
    """Byte view (length C) of the cell array of one record."""
    return rec[dims.cells_word0:dims.cells_word0 + dims.CW].view(np.uint8)[:dims.C]


def describe_record(dims: Dims, rec: np.ndarray) -> str:
    """Human-readable dump used by test failure messages."""
    out = [f"t={int(rec[W_T])} marks={int(rec[W_MARKS]):#x} layout={int(rec[W_LAYOUT])} status={int(rec[W_STATUS])}"]
    for a in range(dims.A):
        out.append(f"  agent{a}: x,y,o,hold={unpack_agent(rec[AGENT_WORD0 + a])}")
    cells = record_cells(dims, rec)
    for y in range(dims.H):
        out.append("  " + " ".join(f"{int(c):02x}" for c in cells[y * dims.W:(y + 1) * dims.W]))
    for s in range(dims.D):
        x, y, c, f = unpack_dyn0(rec[dims.dyn0_word0 + s])
        cont, seq = unpack_dyn1(rec[dims.dyn1_word0 + s])
        if f & DYN_ALIVE or c:
            out.append(f"  slot{s}: {DYNAMIC_CLASSES[c] if c < 10 else c} ({x},{y}) flags={f:#x} in={cont} seq={seq}")
    return "\n".join(out)
