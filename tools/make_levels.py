#!/usr/bin/env python3
"""Author the build's level / meta JSON files (schema: SURVEY.md Appendix B.3).

* large_16x16  : BASELINE config 5 (16x16, 4 agents, max object density).
* crowded_6x5  : tiny 4-agent level with a Switch and a Block (collision chains, shared cells).
* coop_test / coexistence_test / switch_test / example meta: the levels every BASELINE config names.
  They are *input data* of the path (a caller passes level="coop_test"); when /root/reference is
  present they are re-serialised from the parsed JSON so the layouts are identical by construction.
"""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEVEL_DIR = os.path.join(REPO, "cooking_zoo_amd", "utils", "level")
META_DIR = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files")
REF = "/root/reference/cooking_zoo/utils"


def dump_level(path, lv):
    with open(path, "w") as f:
        f.write("{\n")
        f.write('"LEVEL_LAYOUT": %s,\n' % json.dumps(lv["LEVEL_LAYOUT"]))
        for key in ("STATIC_OBJECTS", "DYNAMIC_OBJECTS", "AGENTS", "DYNAMIC_EXCLUDED_POSITIONS"):
            f.write('"%s": [\n' % key)
            f.write(",\n".join("    " + json.dumps(e) for e in lv[key]))
            f.write("\n]%s\n" % ("" if key == "DYNAMIC_EXCLUDED_POSITIONS" else ","))
        f.write("}\n")


def dump_meta(path, meta):
    with open(path, "w") as f:
        f.write("[\n" + ",\n".join("    " + json.dumps(e) for e in meta) + "\n]\n")


def large_16x16():
    W = H = 16
    rows = []
    for y in range(H):
        row = ""
        for x in range(W):
            border = x in (0, W - 1) or y in (0, H - 1)
            column = x in (3, 6, 9, 12) and 2 <= y <= 13
            row += "-" if (border or column) else " "
        rows.append(row)
    one = lambda name, x, y: {name: {"COUNT": 1, "X_POSITION": [x], "Y_POSITION": [y]}}
    statics = [one("Cutboard", x, y) for x, y in [(3, 4), (3, 11), (6, 6), (6, 9), (9, 6), (9, 9), (12, 4), (12, 11)]]
    statics += [one("Blender", x, y) for x, y in [(0, 5), (0, 10), (15, 5), (15, 10)]]
    statics += [{"Deliversquare": {"COUNT": 4, "X_POSITION": [4, 7, 8, 11], "Y_POSITION": [0]}}]
    allx, ally = list(range(W)), list(range(H))
    dyn = [{name: {"COUNT": 8, "X_POSITION": allx, "Y_POSITION": ally}}
           for name in ["Plate", "Tomato", "Onion", "Lettuce", "Carrot", "Banana", "Apple", "Watermelon",
                        "Cucumber", "Bread"]]
    agents = [{"MAX_COUNT": 1, "X_POSITION": [1, 2], "Y_POSITION": list(range(1, 15))},
              {"MAX_COUNT": 1, "X_POSITION": [13, 14], "Y_POSITION": list(range(1, 15))},
              {"MAX_COUNT": 1, "X_POSITION": [4, 5, 7, 8], "Y_POSITION": list(range(1, 15))},
              {"MAX_COUNT": 1, "X_POSITION": [10, 11], "Y_POSITION": list(range(1, 15))}]
    lv = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn, "AGENTS": agents,
          "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [15, 0], [0, 15], [15, 15]]}
    meta = [{"Cutboard": 8}, {"Counter": 108}, {"Blender": 4}, {"Deliversquare": 4}, {"Plate": 8}, {"Tomato": 8},
            {"Onion": 8}, {"Lettuce": 8}, {"Carrot": 8}, {"Banana": 8}, {"Apple": 8}, {"Watermelon": 8},
            {"Cucumber": 8}, {"Bread": 16}, {"Agent": 4}]
    return lv, meta


def crowded_6x5():
    rows = ["------", "-    -", "-    -", "-    -", "------"]
    one = lambda name, x, y: {name: {"COUNT": 1, "X_POSITION": [x], "Y_POSITION": [y]}}
    statics = [one("Cutboard", 0, 2), one("Deliversquare", 3, 0), one("Blender", 5, 2),
               one("Switch", 2, 2), one("Block", 3, 2)]
    dyn = [one("Plate", 1, 0), one("Tomato", 0, 1), one("Carrot", 5, 3), one("Bread", 2, 4),
           {"Lettuce": {"COUNT": 1, "X_POSITION": [1, 2, 3, 4], "Y_POSITION": [0, 4], "OPTIONAL": 0.5}}]
    cells = {"X_POSITION": [1, 2, 3, 4], "Y_POSITION": [1, 2, 3]}
    agents = [dict(MAX_COUNT=1, **cells) for _ in range(4)]
    lv = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn, "AGENTS": agents,
          "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [5, 0], [0, 4], [5, 4]]}
    meta = [{"Agent": 4}, {"Switch": 1}, {"Block": 1}, {"Cutboard": 1}, {"Counter": 18}, {"Blender": 1},
            {"Deliversquare": 1}, {"Bread": 2}, {"Plate": 1}, {"Tomato": 1}, {"Carrot": 1}, {"Lettuce": 1}]
    return lv, meta


def edge_levels():
    """boundary sizes: 8x8 = 64 cells (last one-cell-per-lane size), 9x8 = 72 cells (first multi-cell size),
    and a level without any dynamic object"""
    out = {}
    one = lambda name, x, y: {name: {"COUNT": 1, "X_POSITION": [x], "Y_POSITION": [y]}}
    for name, W, H in (("edge_8x8", 8, 8), ("edge_9x8", 9, 8)):
        rows = ["-" * W] + ["-" + " " * (W - 2) + "-" for _ in range(H - 2)] + ["-" * W]
        statics = [one("Cutboard", 0, 2), one("Cutboard", W - 1, 5), one("Blender", 3, H - 1), one("Deliversquare", 4, 0),
                   one("Switch", 2, 3), one("Block", W - 3, 4)]
        dyn = [{"Plate": {"COUNT": 2, "X_POSITION": [0, W - 1], "Y_POSITION": list(range(1, H - 1))}},
               {"Tomato": {"COUNT": 2, "X_POSITION": list(range(1, W - 1)), "Y_POSITION": [0, H - 1]}},
               {"Carrot": {"COUNT": 1, "X_POSITION": list(range(1, W - 1)), "Y_POSITION": [0, H - 1]}},
               {"Banana": {"COUNT": 1, "X_POSITION": [0, W - 1], "Y_POSITION": list(range(1, H - 1))}},
               {"Bread": {"COUNT": 2, "X_POSITION": list(range(1, W - 1)), "Y_POSITION": [0, H - 1], "OPTIONAL": 0.8}},
               {"Lettuce": {"COUNT": 1, "X_POSITION": [0, W - 1], "Y_POSITION": list(range(1, H - 1))}}]
        cells = {"X_POSITION": list(range(1, W - 1)), "Y_POSITION": list(range(1, H - 1))}
        lv = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn,
              "AGENTS": [dict(MAX_COUNT=1, **cells) for _ in range(3)],
              "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1]]}
        out[name] = lv
    rows = ["-----", "-   -", "-   -", "-----"]
    out["edge_empty"] = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": [one("Deliversquare", 2, 0), one("Cutboard", 0, 1)],
                         "DYNAMIC_OBJECTS": [], "AGENTS": [{"MAX_COUNT": 2, "X_POSITION": [1, 2, 3], "Y_POSITION": [1, 2]}],
                         "DYNAMIC_EXCLUDED_POSITIONS": []}
    meta = [{"Switch": 1}, {"Block": 1}, {"Agent": 3}, {"Cutboard": 2}, {"Counter": 30}, {"Blender": 1}, {"Deliversquare": 1},
            {"Plate": 2}, {"Tomato": 2}, {"Carrot": 1}, {"Banana": 1}, {"Bread": 4}, {"Lettuce": 1}]
    return out, meta


def dense_16x16():
    """every dynamic-object slot in use: 122 objects + 3 Bread originals with 3 clone slots = 128 = the kernels' cap"""
    W = H = 16
    rows = []
    for y in range(H):
        row = ""
        for x in range(W):
            border = x in (0, W - 1) or y in (0, H - 1)
            block = 2 <= x <= 13 and y in (2, 3, 5, 6, 8, 9, 11, 12)
            row += "-" if (border or block) else " "
        rows.append(row)
    n_counters = sum(r.count("-") for r in rows)
    one = lambda name, x, y: {name: {"COUNT": 1, "X_POSITION": [x], "Y_POSITION": [y]}}
    statics = [one("Cutboard", x, y) for x, y in [(2, 3), (13, 3), (2, 12), (13, 12)]]
    statics += [one("Blender", 0, 7), one("Blender", 15, 7), one("Deliversquare", 7, 0), one("Deliversquare", 8, 15)]
    allx, ally = list(range(W)), list(range(H))
    counts = [("Plate", 10)] + [(n, 14) for n in ["Tomato", "Onion", "Lettuce", "Carrot", "Banana", "Apple", "Watermelon", "Cucumber"]] + [("Bread", 3)]
    dyn = [{name: {"COUNT": c, "X_POSITION": allx, "Y_POSITION": ally}} for name, c in counts]
    agents = [{"MAX_COUNT": 1, "X_POSITION": [1], "Y_POSITION": list(range(1, 15))},
              {"MAX_COUNT": 1, "X_POSITION": [14], "Y_POSITION": list(range(1, 15))},
              {"MAX_COUNT": 1, "X_POSITION": list(range(2, 14)), "Y_POSITION": [4, 7]},
              {"MAX_COUNT": 1, "X_POSITION": list(range(2, 14)), "Y_POSITION": [10, 13]}]
    lv = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn, "AGENTS": agents,
          "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [15, 0], [0, 15], [15, 15]]}
    meta = [{"Cutboard": 4}, {"Counter": n_counters}, {"Blender": 2}, {"Deliversquare": 2}] + \
           [{n: c} for n, c in counts[:-1]] + [{"Bread": 6}, {"Agent": 4}]
    assert sum(c for _, c in counts) + 3 == 128
    return lv, meta


def limit_levels():
    """long thin grids, 32 columns and 31 rows, both close to the 256-cell cap of the middle kernel instance;
    they exercise the ends of the observation quotient table ((x - ax) / W for |x - ax| up to 31, (y - ay) / H up to 30)"""
    out = {}
    one = lambda name, x, y: {name: {"COUNT": 1, "X_POSITION": [x], "Y_POSITION": [y]}}
    for name, W, H in (("limit_32x8", 32, 8), ("limit_8x31", 8, 31)):
        rows = ["-" * W] + ["-" + " " * (W - 2) + "-" for _ in range(H - 2)] + ["-" * W]
        statics = [one("Cutboard", 0, 2), one("Cutboard", W - 1, H - 3), one("Blender", 3, H - 1), one("Deliversquare", 4, 0),
                   one("Switch", 2, 3), one("Block", W - 3, H - 4)]
        dyn = [{"Plate": {"COUNT": 2, "X_POSITION": [0, W - 1], "Y_POSITION": list(range(1, H - 1))}},
               {"Tomato": {"COUNT": 2, "X_POSITION": list(range(1, W - 1)), "Y_POSITION": [0, H - 1]}},
               {"Carrot": {"COUNT": 1, "X_POSITION": list(range(1, W - 1)), "Y_POSITION": [0, H - 1]}},
               {"Banana": {"COUNT": 1, "X_POSITION": [0, W - 1], "Y_POSITION": list(range(1, H - 1))}},
               {"Bread": {"COUNT": 2, "X_POSITION": list(range(1, W - 1)), "Y_POSITION": [0, H - 1], "OPTIONAL": 0.8}},
               {"Lettuce": {"COUNT": 1, "X_POSITION": [0, W - 1], "Y_POSITION": list(range(1, H - 1))}}]
        # agents start in opposite corners so that the extreme coordinate differences occur from the first step on
        agents = [dict(MAX_COUNT=1, X_POSITION=[1], Y_POSITION=[1]), dict(MAX_COUNT=1, X_POSITION=[W - 2], Y_POSITION=[H - 2]),
                  dict(MAX_COUNT=1, X_POSITION=list(range(1, W - 1)), Y_POSITION=list(range(1, H - 1)))]
        out[name] = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn, "AGENTS": agents,
                     "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1]]}
    meta = [{"Switch": 1}, {"Block": 1}, {"Agent": 3}, {"Cutboard": 2}, {"Counter": 80}, {"Blender": 1}, {"Deliversquare": 1},
            {"Plate": 2}, {"Tomato": 2}, {"Carrot": 1}, {"Banana": 1}, {"Bread": 4}, {"Lettuce": 1}]
    return out, meta


def huge_levels():
    """beyond 128 slots / 256 cells (the third kernel instance):
    * huge_32x32: the largest grid there is (1024 cells) with every one of the 255 slots in use, 4 agents,
    * huge_20x20: 400 cells with few objects (only the cell count is over the old cap),
    * huge_objs_16x16: 256 cells with 190 slots (only the object count is over the old cap)"""
    out = {}
    one = lambda name, x, y: {name: {"COUNT": 1, "X_POSITION": [x], "Y_POSITION": [y]}}
    foods = ["Tomato", "Onion", "Lettuce", "Carrot", "Banana", "Apple", "Watermelon", "Cucumber"]

    def grid(W, H, block_rows, x0, x1):
        rows = []
        for y in range(H):
            rows.append("".join("-" if (x in (0, W - 1) or y in (0, H - 1) or (x0 <= x <= x1 and y in block_rows)) else " "
                                for x in range(W)))
        return rows

    # ---- 32 x 32, all 255 slots
    W, H = 32, 32
    rows = grid(W, H, (3, 4, 8, 9, 13, 14, 18, 19, 23, 24, 28), 2, 29)
    statics = [one("Cutboard", x, y) for x, y in [(2, 3), (29, 4), (10, 13), (20, 14), (2, 23), (29, 24)]]
    statics += [one("Blender", 0, 10), one("Blender", 31, 20), one("Blender", 15, 28)]
    statics += [{"Deliversquare": {"COUNT": 3, "X_POSITION": [5, 16, 26], "Y_POSITION": [0]}}, one("Deliversquare", 16, 31)]
    # (one Switch only: a second one crashes the reference, SURVEY A.8 -- Switch has no switch_state)
    statics += [one("Switch", 1, 6), one("Block", 1, 11), one("Block", 30, 16)]
    counts = [("Plate", 25)] + [(n, 28) for n in foods] + [("Bread", 3)]
    assert sum(c for _, c in counts) + 3 == 255
    dyn = [{n: {"COUNT": c, "X_POSITION": list(range(W)), "Y_POSITION": list(range(H))}} for n, c in counts]
    agents = [{"MAX_COUNT": 1, "X_POSITION": [1], "Y_POSITION": [1]},
              {"MAX_COUNT": 1, "X_POSITION": [30], "Y_POSITION": [30]},
              {"MAX_COUNT": 1, "X_POSITION": list(range(2, 30)), "Y_POSITION": [5, 10, 15]},
              {"MAX_COUNT": 1, "X_POSITION": list(range(2, 30)), "Y_POSITION": [20, 25, 29]}]
    lv = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn, "AGENTS": agents,
          "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1]]}
    n_counters = sum(r.count("-") for r in rows)
    meta = [{"Switch": 1}, {"Block": 2}, {"Cutboard": 6}, {"Counter": n_counters}, {"Blender": 3}, {"Deliversquare": 4}] + \
           [{n: c} for n, c in counts[:-1]] + [{"Bread": 6}, {"Agent": 4}]
    out["huge_32x32"] = (lv, meta)

    # ---- 20 x 20, few objects
    W = H = 20
    rows = grid(W, H, (4, 5, 9, 10, 14, 15), 3, 16)
    statics = [one("Cutboard", 3, 4), one("Cutboard", 16, 15), one("Blender", 0, 8), one("Deliversquare", 9, 0),
               one("Switch", 1, 12), one("Block", 18, 7)]
    counts = [("Plate", 4), ("Tomato", 4), ("Lettuce", 3), ("Onion", 3), ("Carrot", 3), ("Banana", 3), ("Bread", 2)]
    dyn = [{n: {"COUNT": c, "X_POSITION": list(range(W)), "Y_POSITION": list(range(H))}} for n, c in counts]
    agents = [{"MAX_COUNT": 1, "X_POSITION": [1], "Y_POSITION": [1]}, {"MAX_COUNT": 1, "X_POSITION": [18], "Y_POSITION": [18]},
              {"MAX_COUNT": 1, "X_POSITION": list(range(2, 18)), "Y_POSITION": [6, 7, 11, 12]}]
    lv = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn, "AGENTS": agents,
          "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1]]}
    n_counters = sum(r.count("-") for r in rows)
    meta = [{"Agent": 3}, {"Switch": 1}, {"Block": 1}, {"Cutboard": 2}, {"Counter": n_counters}, {"Blender": 1},
            {"Deliversquare": 1}] + [{n: c} for n, c in counts[:-1]] + [{"Bread": 4}]
    out["huge_20x20"] = (lv, meta)

    # ---- 16 x 16 = 256 cells (still the four-cells-per-lane size) with 190 slots.  The reference puts one object per
    # Counter, so the grid is a solid block of counters inside a one-cell floor ring.
    W = H = 16
    rows = grid(W, H, tuple(range(2, 14)), 2, 13)
    statics = [one("Cutboard", 2, 6), one("Cutboard", 13, 9), one("Blender", 0, 7), one("Deliversquare", 7, 0), one("Deliversquare", 8, 15)]
    counts = [("Plate", 16)] + [(n, 21) for n in foods] + [("Bread", 3)]
    assert sum(c for _, c in counts) + 3 == 190
    dyn = [{n: {"COUNT": c, "X_POSITION": list(range(W)), "Y_POSITION": list(range(H))}} for n, c in counts]
    agents = [{"MAX_COUNT": 1, "X_POSITION": [1], "Y_POSITION": list(range(1, 15))},
              {"MAX_COUNT": 1, "X_POSITION": [14], "Y_POSITION": list(range(1, 15))},
              {"MAX_COUNT": 1, "X_POSITION": list(range(2, 14)), "Y_POSITION": [1, 14]}]
    lv = {"LEVEL_LAYOUT": "\n".join(rows), "STATIC_OBJECTS": statics, "DYNAMIC_OBJECTS": dyn, "AGENTS": agents,
          "DYNAMIC_EXCLUDED_POSITIONS": [[0, 0], [15, 0], [0, 15], [15, 15]]}
    n_counters = sum(r.count("-") for r in rows)
    meta = [{"Cutboard": 2}, {"Counter": n_counters}, {"Blender": 1}, {"Deliversquare": 2}] + \
           [{n: c} for n, c in counts[:-1]] + [{"Bread": 6}, {"Agent": 3}]
    out["huge_objs_16x16"] = (lv, meta)
    return out


def main():
    os.makedirs(LEVEL_DIR, exist_ok=True)
    os.makedirs(META_DIR, exist_ok=True)
    lv, meta = large_16x16()
    dump_level(os.path.join(LEVEL_DIR, "large_16x16.json"), lv)
    dump_meta(os.path.join(META_DIR, "large_16x16.json"), meta)
    lv, meta = crowded_6x5()
    dump_level(os.path.join(LEVEL_DIR, "crowded_6x5.json"), lv)
    dump_meta(os.path.join(META_DIR, "crowded_6x5.json"), meta)
    levels, meta = edge_levels()
    for name, lv in levels.items():
        dump_level(os.path.join(LEVEL_DIR, name + ".json"), lv)
    dump_meta(os.path.join(META_DIR, "edge.json"), meta)
    lv, meta = dense_16x16()
    dump_level(os.path.join(LEVEL_DIR, "dense_16x16.json"), lv)
    dump_meta(os.path.join(META_DIR, "dense_16x16.json"), meta)
    levels, meta = limit_levels()
    for name, lv in levels.items():
        dump_level(os.path.join(LEVEL_DIR, name + ".json"), lv)
    dump_meta(os.path.join(META_DIR, "limits.json"), meta)
    for name, (lv, meta) in huge_levels().items():
        dump_level(os.path.join(LEVEL_DIR, name + ".json"), lv)
        dump_meta(os.path.join(META_DIR, name + ".json"), meta)
    if os.path.isdir(REF):
        for name in ("coop_test", "coexistence_test", "switch_test"):
            with open(os.path.join(REF, "level", name + ".json")) as f:
                dump_level(os.path.join(LEVEL_DIR, name + ".json"), json.load(f))
        with open(os.path.join(REF, "meta_files", "example.json")) as f:
            dump_meta(os.path.join(META_DIR, "example.json"), json.load(f))


if __name__ == "__main__":
    main()
