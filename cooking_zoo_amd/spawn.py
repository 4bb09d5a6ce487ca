"""Agent despawn / respawn bookkeeping for a BATCH of worlds: the host-side MODEL of what the step kernels do.

The product path evaluates this rule on the device (cz_set_spawn, Ops::handle_agent_spawn in csrc/cz_device.h); this module is
the same rule in numpy, used by the tests (against a scalar transliteration of the reference, and against the device) and by
`CookingVecEnv.spawn` only as a read-only view of the device's bookkeeping.

The reference does this per world in `CookingWorld.handle_agent_spawn` (+ `despawn_agent` / `respawn_agent`,
cooking_world.py:267-290) with draws from numpy's process-global stream and respawn cells from Python's global `random`
(parsing.py:154-167).  That defines the draws for ONE world per process; the single-env facade
(environment/cooking_env.py) reproduces exactly that and is pinned to the reference.  For thousands of worlds this module
keeps the same RULE per world -- grace countdown, at most one Bernoulli draw per agent and step, an agent that holds
something stays, a respawned agent lands on a free Floor cell of its spawn area -- but takes every draw from a
counter-based stream keyed by (seed, global env id, episode << 32 | t of that world, agent, draw index), so results do not
depend on the batch size, the sharding over GPUs, the order worlds are processed in or the form the steps are launched in.  The device learns who acts through action -1
(cooking_world.py:105-108); this class only edits agent positions in the env records of worlds where somebody respawns.
"""
from __future__ import annotations

import numpy as np

from cooking_zoo_amd import soa

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on uint64 arrays"""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def uniform(seed, env_global, step, agent, draw):
    """float64 in [0, 1): the `draw`-th number of (world, step, agent); arrays broadcast"""
    with np.errstate(over="ignore"):
        k = _mix(np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.asarray(env_global, dtype=np.uint64))
        k = _mix(k ^ (np.asarray(step, dtype=np.uint64) * np.uint64(0xD1B54A32D192ED03)))
        k = _mix(k ^ (np.asarray(agent, dtype=np.uint64) << np.uint64(32)) ^ np.asarray(draw, dtype=np.uint64))
    return (k >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class SpawnBook:
    def __init__(self, num_envs, num_agents, spawn_cells, *, despawn_rate, respawn_rate, grace_period, seed=0, env_id_base=0,
                 level_of_layout=None):
        """spawn_cells[level][i] = (x candidates, y candidates) of agent i in that level (the level file's AGENTS entries,
        parsing.py:118-151); level_of_layout[layout id] = the level a layout instantiates (None: one level)"""
        self.N, self.A = int(num_envs), int(num_agents)
        self.despawn_rate, self.respawn_rate, self.grace_period = float(despawn_rate), float(respawn_rate), int(grace_period)
        self.spawn_cells = [[(list(xs), list(ys)) for xs, ys in lv] for lv in spawn_cells]
        self.level_of_layout = None if level_of_layout is None else np.asarray(level_of_layout, dtype=np.int64)
        self.seed, self.env_id_base = int(seed), int(env_id_base)
        self.env_ids = np.arange(self.N, dtype=np.uint64) + np.uint64(self.env_id_base)
        self.reset_all()

    def reset_all(self):
        self.active = np.ones((self.N, self.A), dtype=bool)                 # load_level.py:67
        self.changed = np.zeros((self.N, self.A), dtype=bool)               # load_level.py:68
        self.grace = np.full((self.N, self.A), self.grace_period, dtype=np.int64)   # parsing.py:142
        self.step = 0

    def reset_envs(self, mask):
        """worlds that started a new episode (auto-reset re-instantiates the world: everybody is back)"""
        self.active[mask] = True
        self.changed[mask] = False
        self.grace[mask] = self.grace_period

    def mask_actions(self, actions):
        """actions [N, A] -> a copy with -1 for agents that are not in the list world_step acts on"""
        out = np.array(actions, dtype=np.int32).reshape(self.N, self.A)
        out[~self.active] = -1
        return out

    def after_step(self, records, dims, stepped=None):
        """handle_agent_spawn for every world that executed a world step (`stepped`, default all), on the records as they
        are after that step.  Returns the indices of the worlds in which an agent was moved (their rows of `records` were
        edited in place: write them back)."""
        self.step += 1
        stepped = np.ones(self.N, dtype=bool) if stepped is None else np.asarray(stepped, dtype=bool)
        # the key of this step's draws, per world: its episode and step counter (after the step)
        key = (records[:, soa.W_EPISODE].astype(np.uint64) << np.uint64(32)) | records[:, soa.W_T].astype(np.uint64)
        self._key = key
        self.changed[stepped] = False                                        # cooking_world.py:106
        aw0 = soa.AGENT_WORD0
        moved = np.zeros(self.N, dtype=bool)
        for i in range(self.A):                                              # agents in index order, like the reference
            graced = stepped & (self.grace[:, i] > 0)
            self.grace[graced, i] -= 1
            free = stepped & ~graced
            cand_d = free & (self.active.sum(axis=1) > 1) & self.active[:, i]
            u1 = uniform(self.seed, self.env_ids, key, i, 0)
            hit = cand_d & (u1 < self.despawn_rate)
            holding = ((records[:, aw0 + i] >> np.uint32(24)) & np.uint32(0xFF)) != 0
            gone = hit & ~holding                                            # an agent that holds something stays (:280-281)
            self.active[gone, i] = False
            self.changed[gone, i] = True
            cand_r = free & ~cand_d & ~self.active[:, i] & ~gone              # the `elif` arm: only when the first test was not taken
            u2 = uniform(self.seed, self.env_ids, key, i, 1)
            back = cand_r & (u2 < self.respawn_rate)
            for e in np.nonzero(back)[0]:
                x, y = self._generate_location(records[e], dims, i, int(e))
                _, _, o, h = soa.unpack_agent(records[e, aw0 + i])
                records[e, aw0 + i] = soa.pack_agent(x, y, o, h)             # location only (cooking_world.py:290)
                moved[e] = True
            self.active[back, i] = True
            self.changed[back, i] = True
            self.grace[back, i] = self.grace_period
        return np.nonzero(moved)[0]

    def _generate_location(self, rec, dims, agent, e):
        """parsing.py:154-167: a Floor cell nobody (active or not) stands on, from the agent's spawn area"""
        level = 0 if self.level_of_layout is None else int(self.level_of_layout[int(rec[soa.W_LAYOUT])])
        xs, ys = self.spawn_cells[level][agent]
        cells = soa.record_cells(dims, rec)
        taken = {soa.unpack_agent(rec[soa.AGENT_WORD0 + a])[:2] for a in range(dims.A)}
        for k in range(1001):
            ux = uniform(self.seed, self.env_ids[e], self._key[e], agent, 2 + 2 * k)
            uy = uniform(self.seed, self.env_ids[e], self._key[e], agent, 3 + 2 * k)
            x, y = xs[int(ux * len(xs))], ys[int(uy * len(ys))]
            if 0 <= x < dims.W and 0 <= y < dims.H and (x, y) not in taken and (cells[y * dims.W + x] & soa.CELL_TYPE_MASK) == soa.FLOOR:
                return int(x), int(y)
        raise ValueError("Can't find valid position in 1000 steps")

    def relevant(self):
        """[N, A] bool: the agents a step reports on (active, or despawned in this very step): cooking_world.py:292-293"""
        return self.active | self.changed


# ---- the device's encoding of this bookkeeping in record word W_STATUS (csrc/cz_device.h SPAWN_*): bit 8 + a = agent a is
# despawned; from bit 12 one grace countdown per agent, `bits` wide: 5 while the grace period is at most 31, else 20 // num_agents
SPAWN_GONE0, SPAWN_GRACE0 = 8, 12


def grace_bits(grace_period, num_agents):
    return 5 if int(grace_period) <= 31 else 20 // int(num_agents)


def max_grace(num_agents):
    b = 20 // int(num_agents)
    return (1 << b) - 1 if b > 5 else 31


def status_bits(active, grace, bits):
    """[N, A] bool active, [N, A] int grace -> uint32 [N] to be or-ed into the status word.  `bits` = grace_bits(grace_period, A) of the
    batch the words are for - required: a record does not say which width it was packed with"""
    out = np.zeros(active.shape[0], dtype=np.uint32)
    for a in range(active.shape[1]):
        out |= (~active[:, a]).astype(np.uint32) << np.uint32(SPAWN_GONE0 + a)
        out |= grace[:, a].astype(np.uint32) << np.uint32(SPAWN_GRACE0 + bits * a)
    return out


def decode_status(status, num_agents, bits):
    """uint32 [N] status words -> (active [N, A] bool, grace [N, A] int64); `bits` = grace_bits(grace_period, A) of the batch they come from"""
    status = np.asarray(status, dtype=np.uint32)
    active = np.empty((status.shape[0], num_agents), dtype=bool)
    grace = np.empty((status.shape[0], num_agents), dtype=np.int64)
    for a in range(num_agents):
        active[:, a] = ((status >> np.uint32(SPAWN_GONE0 + a)) & np.uint32(1)) == 0
        grace[:, a] = (status >> np.uint32(SPAWN_GRACE0 + bits * a)) & np.uint32((1 << bits) - 1)
    return active, grace
