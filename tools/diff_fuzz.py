#!/usr/bin/env python3
"""Differential fuzz: unmodified reference vs. the C oracle, many seeds/policies, nothing stored.

Build-container only (needs /root/reference).  Complements the committed golden fixtures with far
more coverage: every episode is captured from the reference exactly like tools/gen_golden.py does and
replayed through oracle/cz_oracle.c; any bit difference in state / obs / reward / flags aborts.

A reference that RAISES is a finding, not an abort: the episode is cut at the last good step (which the oracle must
still reproduce), the exception is recorded as `(type, reference frames, config, seed, step, action, what the innermost
frame was using)`, and - where gen_golden.tolerant_reference knows the site - the episode is captured a second time from
the reference with exactly that raise site turned into the build's documented no-op, and the oracle must reproduce THAT
bit for bit through and past the crashing step.  `--histogram FILE` writes the crash classes with counts and one
reproducer each (profiles/r05/refcrash_histogram.json is such a file; DESIGN.md section 2's deviation list is made from it).

Usage: python tools/diff_fuzz.py [--episodes N] [--seed0 S] [--families base|all] [--schemes mix|scheme1] [--policies ...]
                                 [--jobs J] [--histogram FILE]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import gen_golden as gg  # noqa: E402  (imports the reference through the shim)
from cooking_zoo_amd import soa  # noqa: E402
from cooking_zoo_amd.cooking_world.engine.load_level import load_meta_file  # noqa: E402
from golden_io import recipe_table  # noqa: E402
from oracle_binding import Oracle  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def check(cfg, ep):
    dims = soa.Dims(*[int(v) for v in ep["dims"]])
    meta = load_meta_file(cfg["meta_file"])
    W = dims.W
    off, cells = [0], []
    for name in soa.STATIC_CLASSES:
        cells += [y * W + x for x, y in ep["statics"].get(name, [])]
        off.append(len(cells))
    orc = Oracle(dims, meta, recipe_table(), [(ep["states"][0], np.asarray(off, np.int32), np.asarray(cells, np.int16))],
                 scheme=3 if cfg["action_scheme"] == "scheme3" else 1, max_steps=cfg["max_steps"],
                 end_condition_all=cfg["end_condition_all_dishes"], num_recipes=len(cfg["recipes"]),
                 reward_scheme=cfg.get("reward_scheme"))
    rec = ep["states"][0].copy()
    assert np.array_equal(bits(orc.observe(rec)), bits(ep["obs"][0]))
    for t in range(len(ep["actions"])):
        err, obs, rew, term, trunc = orc.step_env(rec, ep["actions"][t])
        a, b = rec.copy(), ep["states"][t + 1].copy()
        a[soa.W_STATUS] = b[soa.W_STATUS] = 0
        ok = (err == 0 and np.array_equal(a, b) and np.array_equal(bits(obs), bits(ep["obs"][t + 1]))
              and np.array_equal(bits(rew), bits(ep["rewards"][t])) and np.array_equal(term, ep["terms"][t])
              and np.array_equal(trunc, ep["truncs"][t]))
        if not ok:
            print("MISMATCH", cfg, "seed", ep["seed"], "step", t, "err", err)
            print(soa.describe_record(dims, rec))
            print(soa.describe_record(dims, ep["states"][t + 1]))
            return False
    return True


def crash_class(c):
    """Crash class = exception type + the reference frame that raised + the frame that caused it (one above) + what it held."""
    fr = c["frames"]
    return f"{c['type']} @ {fr[-1] if fr else '?'} <- {fr[-2] if len(fr) > 1 else '?'} [{c['detail']}]"


def draw_case(args, i, rng):
    L = os.path.join(REPO, "cooking_zoo_amd", "utils", "level")
    M = os.path.join(REPO, "cooking_zoo_amd", "utils", "meta_files")
    all_recipes = gg.RECIPE_NAMES
    kind = i % 6
    if args.schemes == "scheme1":
        scheme = "scheme1"
    else:
        scheme = "scheme1" if (i // 6) % 3 == 2 else "scheme3"
    own = lambda name, meta_name: (os.path.join(L, name + ".json"), os.path.join(M, meta_name + ".json"))
    extra = [("edge_8x8", "edge", 3), ("edge_9x8", "edge", 3), ("edge_empty", "edge", 2), ("limit_32x8", "limits", 3),
             ("limit_8x31", "limits", 3), ("dense_16x16", "dense_16x16", 4), ("huge_32x32", "huge_32x32", 4),
             ("huge_20x20", "huge_20x20", 3), ("huge_objs_16x16", "huge_objs_16x16", 3)]
    if args.families == "all" and i % 2 == 1:
        name, meta_name, max_agents = extra[(i // 2) % len(extra)]
        (lvl, meta), A = own(name, meta_name), int(rng.integers(1, max_agents + 1))
    elif kind in (0, 1):
        lvl, meta, A = "coop_test", "example", 1 + (i % 2)
    elif kind == 2:
        lvl, meta, A = "coexistence_test", "example", 2
    elif kind == 3:
        lvl, meta, A = "switch_test", "example", 2
    elif kind == 4:
        lvl, meta, A = os.path.join(L, "crowded_6x5.json"), os.path.join(M, "crowded_6x5.json"), int(rng.integers(2, 5))
    else:
        lvl, meta, A = os.path.join(L, "large_16x16.json"), os.path.join(M, "large_16x16.json"), int(rng.integers(1, 5))
    recipes = [all_recipes[int(rng.integers(len(all_recipes)))] for _ in range(A)]
    rs = None
    if i % 4 == 1:
        rs = {"recipe_reward": float(rng.integers(1, 40)) / 2, "max_time_penalty": -float(rng.integers(0, 10)),
              "recipe_penalty": -float(rng.integers(0, 50)), "recipe_node_reward": float(rng.integers(0, 8)) / 4}
    cfg = gg.base_cfg(lvl, A, recipes, scheme=scheme, max_steps=int(rng.integers(20, args.max_steps + 1)),
                      all_dishes=bool(i % 2), meta=meta, reward_scheme=rs)
    pols = args.policies.split(",")
    policy = pols[int(rng.integers(len(pols)))]
    if scheme == "scheme1" and policy in ("mixed", "heuristic"):
        policy = "bumper"
    return cfg, policy


def run_range(args, lo, hi):
    """Episodes [lo, hi) of the run; returns (steps, events, crashes, ok)."""
    steps, crashes = 0, []
    events = dict(term=0, chopped=0, mashed=0, plated=0)
    for i in range(lo, hi):
        seed = args.seed0 + i
        rng = np.random.default_rng([args.seed0, i])            # per-episode stream: results do not depend on --jobs
        cfg, policy = draw_case(args, i, rng)
        ep = gg.capture_episode(cfg, seed, policy, on_crash="record")
        ep["seed"], ep["policy"] = seed, policy
        if not check(cfg, ep):                                   # the prefix up to the last good step
            return steps, events, crashes, False
        c = ep["crash"]
        if c is not None:
            c = dict(c, seed=seed, policy=policy, cls=crash_class(c), tolerated=False,
                     cfg={k: (os.path.splitext(os.path.basename(v))[0] if k in ("level", "meta_file") else v)
                          for k, v in cfg.items()})
            # the reference with that raise site as the build's no-op: the oracle must follow it past the crash
            with gg.tolerant_reference():
                ep2 = gg.capture_episode(cfg, seed, policy, on_crash="record")
            ep2["seed"], ep2["policy"] = seed, policy
            longer = len(ep2["actions"]) > len(ep["actions"])
            if longer and np.array_equal(ep2["actions"][len(ep["actions"])], c["action"]):
                if not check(cfg, ep2):
                    print("MISMATCH past a tolerated reference crash:", c)
                    return steps, events, crashes, False
                c["tolerated"] = True
                c["steps_past_crash"] = len(ep2["actions"]) - len(ep["actions"])
                ep = ep2 if ep2["crash"] is None else ep
            crashes.append(c)
        st = gg.episode_stats(ep)
        steps += st["steps"]
        for k in events:
            events[k] += st[k]
    return steps, events, crashes, True


def _worker(a):
    args, lo, hi = a
    return run_range(args, lo, hi)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=200)
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--families", default="base", choices=["base", "all"],
                    help="base: the BASELINE levels; all: also the edge / limit / dense / huge levels (slow in the reference)")
    ap.add_argument("--schemes", default="mix", choices=["mix", "scheme1"])
    ap.add_argument("--policies", default="bumper,mixed,uniform,heuristic,bumper,bumper")
    ap.add_argument("--max-steps", type=int, default=249)
    ap.add_argument("--jobs", type=int, default=1)
    ap.add_argument("--histogram", default=None)
    args = ap.parse_args()
    t0 = time.time()
    if args.jobs > 1:
        import multiprocessing as mp
        chunk = max(1, (args.episodes + 8 * args.jobs - 1) // (8 * args.jobs))
        parts = [(args, lo, min(lo + chunk, args.episodes)) for lo in range(0, args.episodes, chunk)]
        with mp.get_context("fork").Pool(args.jobs) as pool:
            results = pool.map(_worker, parts)
    else:
        results = [run_range(args, 0, args.episodes)]
    steps = sum(r[0] for r in results)
    events = {k: sum(r[1][k] for r in results) for k in results[0][1]}
    crashes = [c for r in results for c in r[2]]
    if not all(r[3] for r in results):
        sys.exit(1)
    hist = {}
    for c in crashes:
        h = hist.setdefault(c["cls"], {"count": 0, "tolerated": 0, "by_level": {}, "reproducer": None})
        h["count"] += 1
        h["tolerated"] += bool(c["tolerated"])
        h["by_level"][c["cfg"]["level"]] = h["by_level"].get(c["cfg"]["level"], 0) + 1
        if h["reproducer"] is None or c["step"] < h["reproducer"]["step"]:
            h["reproducer"] = {k: c[k] for k in ("cfg", "seed", "policy", "step", "action", "frames", "message", "detail")}
    print(f"diff_fuzz OK: {args.episodes} episodes, {steps} steps bit-exact vs reference in {time.time() - t0:.0f}s; "
          f"end-state totals {events}; reference crashes {len(crashes)} in {len(hist)} classes")
    for k, h in sorted(hist.items(), key=lambda kv: -kv[1]["count"]):
        print(f"   {h['count']:5d} (oracle follows the tolerant reference past it in {h['tolerated']})  {k}")
    if args.histogram:
        with open(args.histogram, "w") as f:
            json.dump({"command": " ".join(sys.argv), "episodes": args.episodes, "steps": steps, "end_state_totals": events,
                       "crashes": len(crashes), "classes": hist}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
