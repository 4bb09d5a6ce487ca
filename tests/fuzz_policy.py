"""Test infrastructure: a biased ("bumper") action source over flat records, and event accounting from record diffs.

Uniform random actions almost never reach the deep transitions of the step path (a plated multi-item dish, a chopped + mashed
object, a plate absorbing from a counter, a delivery, a Bread clone, PICK_UP_SPECIAL out of a loaded plate): the device
side of the differential fuzz therefore also runs under this policy - walk towards a cell that holds an object or is an
action object and use it, the numpy counterpart (vectorised over the batch, greedy instead of a path search) of the
`Bumper` policy tools/gen_golden.py drives the reference with - and counts which transitions of
cooking_world.py:114-170 (resolve_primary_interaction, pick_up_special, execute) and :243-261 (attempt_merge, three
branches) a run has exercised, from the oracle's records before / after every step.

Nothing here is a model of the dynamics: the policy only chooses actions, the counters only read states the oracle made.
"""
from __future__ import annotations

import numpy as np

from cooking_zoo_amd import soa

DX = np.array([0, -1, 1, 0, 0], np.int64)         # action 1 left, 2 right, 3 down (+y), 4 up (cooking_world.py:172-184)
DY = np.array([0, 0, 0, 1, -1], np.int64)
BIG = 16000

EVENTS = ["pick_up", "put_down", "chop", "bread_clone", "blend", "blender_toggle", "plate_add", "plate_absorb",
          "static_accepts", "pick_up_special", "switch_press", "delivery", "marks_changed", "termination", "truncation",
          "despawn", "respawn", "plate_with_2plus", "chopped_and_mashed"]


class Fields:
    """Vectorised views of a batch of records [n, RW]."""

    def __init__(self, dims: soa.Dims, recs: np.ndarray):
        d = dims
        self.ag = recs[:, soa.AGENT_WORD0:soa.AGENT_WORD0 + d.A].astype(np.int64)
        self.ax, self.ay = self.ag & 0xFF, (self.ag >> 8) & 0xFF
        self.ao, self.held = (self.ag >> 16) & 0xFF, ((self.ag >> 24) & 0xFF) - 1
        self.cells = np.ascontiguousarray(recs[:, d.cells_word0:d.cells_word0 + d.CW]).view(np.uint8)[:, :d.C].astype(np.int64)
        d0 = recs[:, d.dyn0_word0:d.dyn0_word0 + d.D].astype(np.int64)
        d1 = recs[:, d.dyn1_word0:d.dyn1_word0 + d.D].astype(np.int64)
        self.ox, self.oy, self.ocls, self.oflags = d0 & 0xFF, (d0 >> 8) & 0xFF, (d0 >> 16) & 0xFF, (d0 >> 24) & 0xFF
        self.alive = (self.oflags & soa.DYN_ALIVE) != 0
        self.cont = (d1 & 0xFF) - 1                                   # plate slot the object is inside, -1: none
        self.status = recs[:, soa.W_STATUS].astype(np.int64)
        self.episode = recs[:, soa.W_EPISODE].astype(np.int64)
        self.marks = recs[:, soa.W_MARKS].astype(np.int64) | (recs[:, soa.W_MARKS_HI].astype(np.int64) << 32)


class BumperActions:
    """act(records) -> int32 [n, A].  Per (env, agent) a goal cell drawn with weights that depend on what the agent holds
    (nothing: food, loaded boards / blenders, plates; fresh food: a free cutboard or blender; processed food: a plate - or
    another machine; a plate: processed food to absorb, the deliver square); greedy steps towards it; adjacent = bump
    (scheme3) or face + PRIMARY / EXECUTE / PICK_UP_SPECIAL chosen by what stands there (scheme1)."""

    def __init__(self, dims: soa.Dims, scheme: int, rng: np.random.Generator, eps=0.08, regoal=0.03):
        self.d, self.scheme, self.rng, self.eps, self.regoal = dims, scheme, rng, eps, regoal
        self.n_act = 5 if scheme == 3 else 8
        self.goal = None

    def _cell_maps(self, f: Fields):
        d, n = self.d, f.cells.shape[0]
        rows = np.repeat(np.arange(n), d.D).reshape(n, d.D)
        held_any = np.zeros((n, d.D), bool)
        for a in range(d.A):
            h = f.held[:, a]
            ok = h >= 0
            held_any[np.nonzero(ok)[0], h[ok]] = True
        lying = f.alive & ~held_any & (f.cont < 0)
        cell_of = np.clip(f.oy * d.W + f.ox, 0, d.C - 1)
        done = (f.oflags & (soa.DYN_CHOPPED | soa.DYN_MASHED)) != 0
        is_plate = f.ocls == soa.PLATE

        def at(mask):
            m = np.zeros((n, d.C))
            np.add.at(m, (rows[mask], cell_of[mask]), 1.0)
            return m > 0
        loaded = np.zeros((n, d.D), bool)                       # plates with something inside
        inside = f.alive & (f.cont >= 0)
        loaded[rows[inside], np.clip(f.cont, 0, d.D - 1)[inside]] = True
        return dict(fresh=at(lying & ~is_plate & ~done), done=at(lying & ~is_plate & done), plate=at(lying & is_plate),
                    loaded_plate=at(lying & is_plate & loaded), any=at(lying), held_any=held_any, loaded=loaded)

    def _draw_goals(self, f: Fields, need, maps=None):
        if not need.any():
            return
        d, n = self.d, f.cells.shape[0]
        m = maps or self._cell_maps(f)
        ty = f.cells & soa.CELL_TYPE_MASK
        solid = ~((ty == soa.FLOOR) | (ty == soa.SWITCH) | ((ty == soa.BLOCK) & ((f.cells & soa.CELL_WALK) != 0)))
        board, blender, deliver = ty == soa.CUTBOARD, ty == soa.BLENDER, ty == soa.DELIVERSQUARE
        free = solid & ~m["any"]
        base = 0.25 * solid + 1.5 * (ty == soa.SWITCH) + 0.5 * (ty == soa.BLOCK)
        w_empty = base + 4.0 * m["fresh"] + 3.0 * m["done"] + 3.0 * m["plate"] + 6.0 * ((board | blender) & m["any"])
        w_fresh = base + 8.0 * (board & free) + 4.0 * (blender & free) + 1.5 * m["plate"] + 0.5 * free
        w_done = base + 8.0 * m["plate"] + 3.0 * ((board | blender) & free) + 0.5 * free
        w_plate = base + 8.0 * m["done"] + 1.0 * m["fresh"] + 0.5 * free
        for a in range(d.A):
            sel = need[:, a]
            if not sel.any():
                continue
            h = f.held[:, a]
            hc = np.clip(h, 0, d.D - 1)
            hcls = np.take_along_axis(f.ocls, hc[:, None], 1)[:, 0]
            hdone = (np.take_along_axis(f.oflags, hc[:, None], 1)[:, 0] & (soa.DYN_CHOPPED | soa.DYN_MASHED)) != 0
            hloaded = np.take_along_axis(m["loaded"], hc[:, None], 1)[:, 0]
            w = np.where((h < 0)[:, None], w_empty,
                         np.where((hcls == soa.PLATE)[:, None], w_plate + (6.0 * hloaded + 0.5)[:, None] * deliver,
                                  np.where(hdone[:, None], w_done, w_fresh)))[sel]
            # only goals the agent can get to: cells of its own walkable region, and solid cells that border it
            mine = self._distances(solid[sel], (f.ay[sel, a] * d.W + f.ax[sel, a]), from_cell=True) < BIG
            mm = mine.reshape(-1, d.H, d.W)
            near = mm.copy()
            near[:, 1:, :] |= mm[:, :-1, :]
            near[:, :-1, :] |= mm[:, 1:, :]
            near[:, :, 1:] |= mm[:, :, :-1]
            near[:, :, :-1] |= mm[:, :, 1:]
            w = w * near.reshape(-1, d.C)
            tot = w.sum(1)
            cum = np.cumsum(w, axis=1)
            r = self.rng.random(int(sel.sum())) * np.maximum(tot, 1e-9)
            g = np.minimum((cum <= r[:, None]).sum(1), d.C - 1)
            g = np.where(tot > 0, g, self.rng.integers(d.C, size=g.shape))
            self.goal[sel, a] = g
            self.dist[sel, a] = self._distances(solid[sel], g)

    def _distances(self, solid, goal, max_iter=40, from_cell=False):
        """walking distance of every floor cell to the goal's side (a cell next to a solid goal, or the goal itself when it can be
        walked on), by relaxation over the whole sub-batch at once; BIG where the goal is further than max_iter steps or unreachable"""
        d = self.d
        m = solid.shape[0]
        walk = (~solid).reshape(m, d.H, d.W)
        gy, gx = goal // d.W, goal % d.W
        seed = np.zeros((m, d.H, d.W), bool)
        r = np.arange(m)
        gsolid = solid[r, goal] & (not from_cell)
        seed[r[~gsolid], gy[~gsolid], gx[~gsolid]] = True
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            x, y = gx + dx, gy + dy
            ok = gsolid & (x >= 0) & (x < d.W) & (y >= 0) & (y < d.H)
            seed[r[ok], y[ok], x[ok]] = True
        if not from_cell:
            seed &= walk
        else:
            walk = walk | seed                          # (an agent may stand on a Block that has closed under it)
        dist = np.where(seed, 0, BIG).astype(np.int16)
        for _ in range(max_iter):
            nb = np.full_like(dist, BIG)
            nb[:, 1:, :] = np.minimum(nb[:, 1:, :], dist[:, :-1, :])
            nb[:, :-1, :] = np.minimum(nb[:, :-1, :], dist[:, 1:, :])
            nb[:, :, 1:] = np.minimum(nb[:, :, 1:], dist[:, :, :-1])
            nb[:, :, :-1] = np.minimum(nb[:, :, :-1], dist[:, :, 1:])
            new = np.where(walk, np.minimum(dist, nb + 1), BIG).astype(np.int16)
            if np.array_equal(new, dist):
                break
            dist = new
        return dist.reshape(m, d.C)

    def act(self, recs: np.ndarray) -> np.ndarray:
        d, rng = self.d, self.rng
        f = Fields(d, recs)
        n = recs.shape[0]
        maps = self._cell_maps(f)
        if self.goal is None:
            self.goal = np.zeros((n, d.A), np.int64)
            self.dist = np.full((n, d.A, d.C), BIG, np.int16)
            need = np.ones((n, d.A), bool)
        else:
            need = rng.random((n, d.A)) < self.regoal
        self._draw_goals(f, need, maps)
        gx, gy = self.goal % d.W, self.goal // d.W
        dx, dy = gx - f.ax, gy - f.ay
        man = np.abs(dx) + np.abs(dy)
        # the step that shortens the larger remaining axis (ties broken at random)
        use_x = (np.abs(dx) > np.abs(dy)) | ((np.abs(dx) == np.abs(dy)) & (rng.random((n, d.A)) < 0.5))
        use_x &= dx != 0
        use_x |= (dy == 0) & (dx != 0)
        direction = np.where(use_x, np.where(dx < 0, 1, 2), np.where(dy > 0, 3, 4))
        towards = direction
        # follow the distance map where it knows a way (ties broken at random); the greedy step otherwise
        best = np.full((n, d.A), BIG + 1, np.int64)
        pick = direction.copy()
        order = rng.permuted(np.tile(np.arange(1, 5), (n * d.A, 1)), axis=1).reshape(n, d.A, 4)
        for k in range(4):
            a_k = order[:, :, k]
            nx, ny = f.ax + DX[a_k], f.ay + DY[a_k]
            inside = (nx >= 0) & (nx < d.W) & (ny >= 0) & (ny < d.H)
            cell = np.clip(ny, 0, d.H - 1) * d.W + np.clip(nx, 0, d.W - 1)
            dk = np.where(inside, np.take_along_axis(self.dist, cell[:, :, None], axis=2)[:, :, 0].astype(np.int64), BIG)
            better = dk < best
            best, pick = np.where(better, dk, best), np.where(better, a_k, pick)
        here = np.take_along_axis(self.dist, (f.ay * d.W + f.ax)[:, :, None], axis=2)[:, :, 0]
        direction = np.where((best < BIG) & (here > 0), pick, direction)
        act = direction.copy()
        gcell = np.take_along_axis(f.cells, self.goal, axis=1)
        gty = gcell & soa.CELL_TYPE_MASK
        solid = ~((gty == soa.FLOOR) | (gty == soa.SWITCH))
        adjacent = (man == 1) & solid
        used = adjacent                                                  # scheme3: the bump is the interaction
        if self.scheme != 3:
            # facing it: PRIMARY / EXECUTE / PICK_UP_SPECIAL by what stands there (raised probabilities for 6 and 7); else turn
            facing = adjacent & (f.ao == towards)
            machine = ((gty == soa.CUTBOARD) | (gty == soa.BLENDER)) & np.take_along_axis(maps["any"], self.goal, axis=1)
            lp = np.take_along_axis(maps["loaded_plate"], self.goal, axis=1) & (f.held < 0)
            u = rng.random((n, d.A))
            use = np.where(machine, np.where(u < 0.6, 7, np.where(u < 0.9, 5, 6)),
                           np.where(lp, np.where(u < 0.45, 6, np.where(u < 0.9, 5, 7)),
                                    np.where(u < 0.84, 5, np.where(u < 0.92, 6, 7))))
            act = np.where(facing, use, act)
            used = facing
        arrived = (man == 0) | (used & (rng.random((n, d.A)) < 0.8))
        rnd = rng.random((n, d.A)) < self.eps
        act = np.where(rnd | (man == 0), rng.integers(0, self.n_act, size=(n, d.A)), act)
        # a new goal next time for whoever arrived - chosen for what the agent will hold once this step's action has worked
        self._pending = arrived
        return act.astype(np.int32)

    def observe_result(self, recs: np.ndarray):
        """after the step: agents that used their goal draw the next one from the new state"""
        if getattr(self, "_pending", None) is not None and self._pending.any():
            self._draw_goals(Fields(self.d, recs), self._pending)
        self._pending = None


class EventCounter:
    """Counts transitions from (records before, records after) of one step of the whole batch.  Envs that were finished before
    the step (the step was their auto-reset pass) are left out."""

    def __init__(self, dims: soa.Dims):
        self.d = dims
        self.counts = {k: 0 for k in EVENTS}
        self.steps = 0

    def update(self, before: np.ndarray, after: np.ndarray, terms=None, truncs=None):
        d = self.d
        b, a = Fields(d, before), Fields(d, after)
        live = ((b.status & soa.STATUS_DONE) == 0) & (b.episode == a.episode)
        self.steps += int(live.sum())
        c = self.counts
        L = live[:, None]
        c["pick_up"] += int(((b.held < 0) & (a.held >= 0) & L).sum())
        c["put_down"] += int(((b.held >= 0) & (a.held < 0) & L).sum())
        chopped_b, chopped_a = (b.oflags & soa.DYN_CHOPPED) != 0, (a.oflags & soa.DYN_CHOPPED) != 0
        mashed_b, mashed_a = (b.oflags & soa.DYN_MASHED) != 0, (a.oflags & soa.DYN_MASHED) != 0
        c["chop"] += int((b.alive & a.alive & ~chopped_b & chopped_a & L).sum())
        c["bread_clone"] += int((~b.alive & a.alive & L).sum())
        c["blend"] += int((b.alive & a.alive & ~mashed_b & mashed_a & L).sum())
        c["chopped_and_mashed"] += int((a.alive & chopped_a & mashed_a & ~(chopped_b & mashed_b) & L).sum())
        ty = b.cells & soa.CELL_TYPE_MASK
        c["blender_toggle"] += int(((ty == soa.BLENDER) & (((b.cells ^ a.cells) & soa.CELL_TOGGLE) != 0) & L).sum())
        c["switch_press"] += int(((ty == soa.SWITCH) & (((b.cells ^ a.cells) & soa.CELL_ACTIVE) != 0) & L).sum())
        entered = (b.cont < 0) & (a.cont >= 0) & a.alive & L                 # objects that went into a plate this step
        n = before.shape[0]
        held_b = np.zeros((n, d.D), bool)                                    # was in somebody's hands before the step
        plate_held_a = np.zeros((n, d.D), bool)                              # plate slots in somebody's hands before AND after
        for ag in range(d.A):
            hb, ha = b.held[:, ag], a.held[:, ag]
            ok = hb >= 0
            held_b[np.nonzero(ok)[0], hb[ok]] = True
            keep = ok & (ha == hb)
            plate_held_a[np.nonzero(keep)[0], hb[keep]] = True
        into_held_plate = entered & np.take_along_axis(plate_held_a, np.clip(a.cont, 0, d.D - 1), axis=1)
        c["plate_add"] += int((entered & held_b).sum())                      # attempt_merge branch 1 (cooking_world.py:245-249)
        c["plate_absorb"] += int((into_held_plate & ~held_b).sum())          # branch 2 (:250-256)
        # branch 3 (:257-261): a held object lands on a ContentObject static (cutboard / blender / deliversquare / counter)
        moved_out = held_b & ~entered & L & a.alive
        still_held = np.zeros((n, d.D), bool)
        for ag in range(d.A):
            ha = a.held[:, ag]
            ok = ha >= 0
            still_held[np.nonzero(ok)[0], ha[ok]] = True
        landed = moved_out & ~still_held & (a.cont < 0)
        c["static_accepts"] += int(landed.sum())
        cell_a = np.clip(a.oy * d.W + a.ox, 0, d.C - 1)
        on_deliver = (np.take_along_axis(a.cells, cell_a, axis=1) & soa.CELL_TYPE_MASK) == soa.DELIVERSQUARE
        c["delivery"] += int((landed & on_deliver).sum())
        # PICK_UP_SPECIAL (:131-147): an object leaves a plate into empty hands
        left = (b.cont >= 0) & (a.cont < 0) & a.alive & L
        c["pick_up_special"] += int((left & still_held & ~held_b).sum())
        c["marks_changed"] += int(((b.marks != a.marks) & live).sum())
        if terms is not None:
            c["termination"] += int((np.asarray(terms).any(axis=1) & live).sum())
        if truncs is not None:
            c["truncation"] += int((np.asarray(truncs).any(axis=1) & live).sum())
        for ag in range(d.A):
            bit = 1 << (8 + ag)
            c["despawn"] += int((((b.status & bit) == 0) & ((a.status & bit) != 0) & live).sum())
            c["respawn"] += int((((b.status & bit) != 0) & ((a.status & bit) == 0) & live).sum())
        # a plate that holds two or more objects after the step and did not before
        cnt_b = _per_plate_counts(b, d)
        cnt_a = _per_plate_counts(a, d)
        c["plate_with_2plus"] += int(((cnt_a >= 2) & (cnt_b < 2) & L).sum())

    def table(self) -> str:
        return "\n".join(f"  {k:20s} {v}" for k, v in self.counts.items()) + f"\n  {'(live env-steps)':20s} {self.steps}"

    def merge(self, other: "EventCounter"):
        for k in self.counts:
            self.counts[k] += other.counts[k]
        self.steps += other.steps


def _per_plate_counts(f: Fields, d: soa.Dims):
    n = f.cont.shape[0]
    cnt = np.zeros((n, d.D), np.int64)
    inside = (f.cont >= 0) & f.alive
    rows = np.repeat(np.arange(n), d.D).reshape(n, d.D)
    np.add.at(cnt, (rows[inside], np.clip(f.cont, 0, d.D - 1)[inside]), 1)
    return cnt
