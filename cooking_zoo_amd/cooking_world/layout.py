"""A `Layout` is one instantiated level: the initial world of an episode in flat form.

It carries exactly what the device needs for reset and observation order, and nothing object-shaped:

* `cells`        uint8[W*H]      static type per cell (mutable bits start cleared)
* `static_lists` {class: [cell]} cell indices per static class in `world_objects[class]` LIST order
                                 (creation order minus replaced counters/floors) -- fixes obs order
* `dyn_classes`  [(class id, count)] dynamic classes in `world_objects` KEY order (first creation)
* `dyn_xy`       [(x, y)]         initial position per dynamic object, class-major / list order
* `agents`       [(x, y)]

Slot rule (soa.py): slots are class-major in key order, list order inside a class, and a Bread class
gets count extra slots right after its originals for the clones `Bread.chop` creates
(reference world_objects.py:738-745, cooking_world.py:168-169) -- so that slot order always equals the
reference's `get_objects_at` order (cooking_world.py:232-241).
"""
from __future__ import annotations

import numpy as np

from cooking_zoo_amd import soa


class Layout:
    def __init__(self, width, height, cells, static_lists, dyn_classes, dyn_xy, agents):
        self.width, self.height = int(width), int(height)
        self.cells = np.asarray(cells, dtype=np.uint8)
        self.static_lists = {k: list(v) for k, v in static_lists.items()}
        self.dyn_classes = [(int(c), int(n)) for c, n in dyn_classes]
        self.dyn_xy = [(int(x), int(y)) for x, y in dyn_xy]
        self.agents = [(int(x), int(y)) for x, y in agents]
        # slot table
        self.slot_base = {}
        self.slot_cap = {}
        s = 0
        for cls, n in self.dyn_classes:
            self.slot_base[cls] = s
            self.slot_cap[cls] = 2 * n if cls == soa.BREAD else n
            s += self.slot_cap[cls]
        self.slots_used = s

    def n_switches(self):
        return len(self.static_lists.get("Switch", []))

    # ------------------------------------------------------------------ record
    def init_record(self, dims: soa.Dims, layout_id=0, recipe_ids=()) -> np.ndarray:
        """Initial per-env record (t = 0; marks are evaluated by the device at reset)."""
        if self.slots_used > dims.D:
            raise ValueError(f"layout needs {self.slots_used} dynamic slots, batch capacity is {dims.D}")
        if (self.width, self.height) != (dims.W, dims.H):
            raise ValueError("layout size differs from the batch grid size")
        rec = soa.new_record(dims)
        rec[soa.W_LAYOUT] = layout_id
        rid = [0xFF] * 4
        for i, v in enumerate(recipe_ids):
            rid[i] = int(v)
        rec[soa.W_RECIPES] = rid[0] | (rid[1] << 8) | (rid[2] << 16) | (rid[3] << 24)
        for a, (x, y) in enumerate(self.agents[:dims.A]):
            rec[soa.AGENT_WORD0 + a] = soa.pack_agent(x, y, 1, -1)          # orientation 1, empty hands
        soa.record_cells(dims, rec)[:] = self.cells
        i = 0
        for cls, n in self.dyn_classes:
            base = self.slot_base[cls]
            for k in range(n):
                x, y = self.dyn_xy[i]
                rec[dims.dyn0_word0 + base + k] = soa.pack_dyn0(x, y, cls, soa.DYN_ALIVE | soa.DYN_FREE)
                i += 1
            for k in range(n, self.slot_cap[cls]):                           # clone head-room, not alive
                rec[dims.dyn0_word0 + base + k] = soa.pack_dyn0(0, 0, cls, 0)
        return rec

    # ------------------------------------------------------------------ observation descriptor
    def obs_descriptor(self, meta: dict, dims: soa.Dims) -> np.ndarray:
        """uint32[F]: one descriptor word (soa.desc_word) per feature, in the order cooking_env.py:352-373 emits
        them: meta-file class order, list order inside a class, zero padding up to the meta count."""
        d = []
        dw = soa.desc_word
        img_obj0, img_cell0, img_ag0, img_zero = dims.img_layout()
        zero = dw(img_zero)
        for name, num in meta.items():
            flen = soa.FEATURE_LEN[name]
            emitted = 0
            if name in soa.STATIC_CLASSES:
                for c in self.static_lists.get(name, []):
                    if flen == 0:
                        continue
                    i = img_cell0 + 4 * c
                    d += [dw(i, soa.AX_X), dw(i + 1, soa.AX_Y)]
                    if name in ("Switch", "Block"):
                        d.append(dw(i + 2))                    # switch_active / int(walkable)
                    d.append(dw(i + 3))
                    emitted += 1
                if flen and emitted > num:
                    raise ValueError(f"level has {emitted} {name} objects, meta file allows {num}")
            elif name in soa.DYNAMIC_CLASSES:
                cls = soa.DYNAMIC_CLASSES.index(name)
                base, cap = self.slot_base.get(cls, 0), self.slot_cap.get(cls, 0)
                # slots beyond the meta count cannot be encoded (the reference would emit an over-long vector)
                for k in range(min(cap, num)):
                    i = img_obj0 + 6 * (base + k)
                    d += [dw(i, soa.AX_X), dw(i + 1, soa.AX_Y)]
                    if cls != soa.PLATE:
                        d.append(dw(i + 2))                                    # int(not done())
                        if cls in soa.BLENDER_FOOD:
                            d += [dw(i + 3), dw(i + 4)]                        # chopped, mashed
                        else:
                            d.append(dw(i + 3))                                # int(done()) == chopped
                    d.append(dw(i + 5))
                    emitted += 1
            elif name == "Agent":
                for a in range(min(dims.A, num)):
                    i = img_ag0 + 8 * a
                    d += [dw(i, soa.ax_self(a, 0)), dw(i + 1, soa.ax_self(a, 1))]
                    d += [dw(i + 2 + o) for o in range(4)]
                    d.append(dw(i + 6))
                    emitted += 1
            else:
                raise KeyError(name)
            d += [zero] * ((num - emitted) * flen)
        out = np.asarray(d, dtype=np.uint32)
        if out.size != dims.F:
            raise ValueError(f"descriptor length {out.size} != feature length {dims.F}")
        return out

    def static_table(self):
        """(offsets int32[8], cells int16[n]) per static class in soa.STATIC_CLASSES order (oracle input)."""
        off = [0]
        cells = []
        for name in soa.STATIC_CLASSES:
            cells += self.static_lists.get(name, [])
            off.append(len(cells))
        return np.asarray(off, dtype=np.int32), np.asarray(cells, dtype=np.int16)

    def key(self):
        return (self.cells.tobytes(), tuple(sorted((k, tuple(v)) for k, v in self.static_lists.items())),
                tuple(self.dyn_classes), tuple(self.dyn_xy), tuple(self.agents))


def feature_length(meta: dict) -> int:
    """cooking_env.py:114-117"""
    return sum(soa.FEATURE_LEN[name] * num for name, num in meta.items())
