// cz_device.h -- CookingZoo step() for gfx950 (CDNA4), one 64-lane wavefront per env instance.
//
// Mapping (DESIGN.md "Kernels"):
//   * lane l owns dynamic-object slots l, l+64, ... (OPL per lane) and grid cells l, l+64, ... (CPL per
//     lane), held in VGPRs for the whole step; agents, header words and every decision are wave-uniform
//     (SGPR) values.
//   * every lookup the reference does by a linear scan over Python objects
//     (cooking_world.py:232-241 get_objects_at, :223 square_walkable, "first free object", "agent at
//     location") is a lane-parallel compare + wave ballot; picking a list element is ffs/clz on the
//     ballot; a field of one object is a v_readlane; a mutation is a predicated write by the owning lane.
//   * agents are resolved serially in index order (action_scheme3.py:15-16) -- semantics demand it --
//     but each agent's work is O(1) wave instructions instead of O(objects) scans.
//   * agents live in one VGPR (lane a = agent a): orientation, target cell, bounds, walkability (ds_bpermute into
//     the cell lanes) and collisions are computed for all agents at once; all walking happens at once too, because
//     walking and interacting commute (see perform_agent_actions); only interacting agents are taken serially.
//   * the launch is latency-bound at a few thousand envs (profiles/r01), so the common path is kept short: agent
//     count and action scheme are compile-time, and the recipe graphs / free-flag normalisation are re-evaluated only
//     when an object they depend on actually changed (both are pure functions of object state, so skipping them on
//     untouched steps is exact).
//   * the feature-vector encode (cooking_env.py:352-373) walks a per-layout descriptor table: lanes stride over the F
//     output doubles two at a time, read one halfword of an LDS image of the final state (a byte offset into a table
//     of correctly rounded quotients d/W, d/H and the constants 0.0 / 1.0, computed on the host), subtract the
//     observer's coordinate where the feature is relative, and store coalesced 16-byte values (cz_kernels.h).
// Integer / indexing work only; no MFMA.  No CPU fallback exists in this file or its callers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cz {

constexpr int MAX_AGENTS = 4;
constexpr int MAX_NODES = 8;            // compact recipe tables: rows of 1 + 8 words, marks 8 bits per recipe
constexpr int WIDE_NODES = 16;          // wide tables (a graph with more than 8 nodes in the book): rows of 1 + 2 * 16 words
constexpr int HDR_WORDS = 8;
constexpr int AGENT_WORD0 = HDR_WORDS;
constexpr int RET_WORD0 = HDR_WORDS + MAX_AGENTS;          // 4 x f64 running episode returns
constexpr int CELL_WORD0 = RET_WORD0 + 2 * MAX_AGENTS;

enum : uint32_t { FLOOR = 0, COUNTER, DELIVERSQUARE, SWITCH, BLOCK, CUTBOARD, BLENDER };
enum : uint32_t { PLATE = 0, ONION, TOMATO, LETTUCE, CARROT, CUCUMBER, BANANA, APPLE, WATERMELON, BREAD };
enum : uint32_t { CELL_TYPE = 7, CELL_READY = 8, CELL_TOGGLE = 16, CELL_ACTIVE = 32, CELL_WALK = 64 };
// dyn0 = x | y<<8 | cls<<16 | flags<<24
enum : uint32_t { D_ALIVE = 1u << 24, D_CHOPPED = 2u << 24, D_MASHED = 4u << 24, D_FREE = 8u << 24, D_DONE = 6u << 24 };
enum : uint32_t { COND_NONE = 0, COND_CHOPPED, COND_MASHED, COND_NOT_CHOPPED, COND_NOT_MASHED };
enum : uint32_t { W_T = 0, W_MARKS, W_LAYOUT, W_STATUS, W_EPISODE, W_RECIPES, W_POOL, W_MARKS_HI };   // word 7: wide tables only
enum : uint32_t { ST_DONE = 1, ST_TERM = 2, ST_TRUNC = 4 };
// observation descriptor (one u32 per feature): (image halfword index * 2) | (axis code * 4) << 16  -- soa.py
constexpr int LUT_Y0 = 63, LUT_ZERO = 126, LUT_ONE = 127;
// truth tables of the flag features, indexed by raw state bits (cz_create fills them): not done / chopped / mashed by the
// object's chopped | mashed << 1; the cell flag by ACTIVE | WALK << 1; orientation == k at LUT_OR0 + 8 (k - 1) + orientation.
// Everything else up to 255 is 0.0: the "absent" zone (entry 255 minus an agent coordinate stays inside 224..255).
constexpr int LUT_NDONE0 = 128, LUT_CH0 = 132, LUT_MA0 = 136, LUT_CF0 = 140, LUT_OR0 = 144;
// per-env statistics: u32 words and doubles
enum : uint32_t { SU_EPISODES = 0, SU_STEPS, SU_LENSUM, SU_TRUNC, SU_TERM, SU_COMPLETED0, SU_WORDS = 16 };
enum : uint32_t { SF_CUR0 = 0, SF_SUM0 = 4, SF_WORDS = 8 };

struct Params {
    uint32_t *state;               // [N][RW]
    const uint32_t *lay_init;      // [L][RW]
    const uint32_t *lay_desc;      // [L][F]
    const uint32_t *recipes;       // [n][1 + MAX_NODES]
    const int32_t *actions;        // [N][A] (step) or nullptr (rollout: on-device stream)
    double *obs;                   // [N][A][F] / [T][N][A][F] / nullptr
    double *rewards;               // [N][A] / [T][N][A] / nullptr
    uint8_t *term, *trunc;         // likewise
    uint32_t *marks_out;           // [N] recipe-node marks after the step (host-pointer cz_step only), or nullptr
    uint8_t *codes;                // [N][A][Fp] compact observation (cz_step_device_compact): one table index per feature, or nullptr
    uint32_t *stat_u;              // [N][SU_WORDS]
    double *stat_f;                // [N][SF_WORDS]
    int64_t env_id_base;
    uint64_t seed;
    double recipe_reward, recipe_penalty, node_reward, time_penalty_step;
    double reward_idle;            // the reward formula evaluated with no goal change (host, same op order)
    const double *lut;             // [256] host-computed quotients: [i] (i-(W-1))/W, [63+i] (i-(H-1))/H, [126] 0, [127] 1, rest 0
    int32_t N, A, W, H, D, F, RW, scheme, max_steps, end_all, R, auto_reset, L;
    int32_t T;                     // fused steps per launch (1 for cz_step)
    uint32_t step0;
    int32_t dyn0_off, dyn1_off;    // word offsets inside a record
    int32_t wt;                    // 1: observation stores are write-through (sc1); chosen per launch by the host
    int32_t walk_touches;          // 1 if carrying an object across cells can change a recipe mark (see cz_load_recipes)
    int32_t wide;                  // 1: wide recipe tables (up to 16 nodes per graph, marks in record words 1 and 7)
    // (the argument block: 56 bytes of leading scalars + this struct, within five 64-byte lines; a sixth line costs every launch)
#ifdef CZ_PROFILE
    unsigned long long *stamps;    // diagnostic build only: [N][8] s_memtime stamps
#endif
#ifdef CZ_TIMELINE
    unsigned long long *timeline;  // timeline build only (make timeline): [N][2] entry / exit stamps of this launch's waves, or nullptr
#endif
};
#if !defined(CZ_TIMELINE) && !defined(CZ_PROFILE)
static_assert(sizeof(Params) <= 264 && sizeof(Params) % 8 == 0, "argument block: see the note above");
#endif

#ifndef CZ_PRIO
#define CZ_PRIO 1
#endif
#if CZ_PRIO
#define CZ_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define CZ_SETPRIO(p) do { } while (0)
#endif
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// the inverse: a wave-uniform lane mask as a per-lane predicate.  Costs no instruction - the SGPR pair becomes the condition
// operand of the select that consumes it
__device__ __forceinline__ bool lanes(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// v_writelane_b32: clang 22 / ROCm 7.2 has no builtin for it; bind the LLVM intrinsic the way the HIP headers do
extern "C" __device__ uint32_t __cz_writelane(uint32_t, uint32_t, uint32_t) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t wrl(uint32_t value, int lane, uint32_t into) { return __cz_writelane(value, (uint32_t)lane, into); }

// counter-based action stream (host mirror: cz_action in cz_api.hip, oracle mirror: czo_action)
__host__ __device__ inline uint32_t action_hash(uint64_t seed, int64_t env_global, int agent, uint32_t step, uint32_t n) {
    /* counter-based: two rounds of a 32-bit avalanche mixer over (seed, env, agent, step) */
    uint32_t x = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u);
    x ^= (uint32_t)env_global * 0x85EBCA6Bu + (uint32_t)((uint64_t)env_global >> 32) * 0x27D4EB2Fu;
    x ^= ((uint32_t)agent + 1u) * 0xC2B2AE35u;
    x ^= (step + 1u) * 0x165667B1u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    x += step * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (uint32_t)(((uint64_t)x * (uint64_t)n) >> 32);
}
// The layout an env draws for its episode: keyed by the GLOBAL env id (sharding does not change it), inside the env's slice
// of the pool (pool_word = base | count << 16, 0 = whole pool).  `groups` > 1 (cz_set_layout_group): the slice is cut into
// that many equal parts and only part `active` is drawn from - the others can be refreshed meanwhile (cz_update_layouts).
__host__ __device__ inline uint32_t next_layout(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t n_layouts,
                                                uint32_t groups = 1u, uint32_t active = 0u) {
    uint32_t base = pool_word & 0xFFFFu, count = pool_word >> 16;
    if (count == 0) { base = 0; count = n_layouts; }
    if (groups > 1u) { count /= groups; base += active * count; }
    return base + (uint32_t)(((uint64_t)env_global + (uint64_t)episode * 7919u) % count);
}
// control words in front of the layout records (lay_init - LAY_CTL_WORDS): which part of their slices the envs draw from
constexpr int LAY_CTL_WORDS = 16;
enum : uint32_t { LC_GROUPS = 0, LC_ACTIVE = 1 };

// ---- agent despawn / respawn on the device (cooking_world.py:267-290 handle_agent_spawn, parsing.py:154-167 generate_location).
// The reference draws from numpy's process-global stream, which defines the draws of ONE world per process; a batch takes every
// draw from a counter-based stream keyed by (seed, global env id, episode << 32 | t, agent, draw index) instead - the same
// function as cooking_zoo_amd/spawn.py `uniform` - so results depend neither on the batch size nor on the sharding nor on how
// the steps are launched (one per launch, fused).  The record's status word carries one "despawned" bit per agent
// (bit 8 + a) and a grace countdown each (from bit 12; field width: spawn_grace_bits).  The parameters live in device memory behind the
// quotient table.
struct SpawnCfg {
    uint64_t seed;
    double despawn_rate, respawn_rate;
    uint32_t grace_period;
    uint32_t stride;                   // bytes per candidate list
    // the level files' spawn areas (parsing.py:118-151), one block per (level, agent): n_x, n_y (two uint16), then `stride` x
    // candidates and `stride` y candidates (bytes); which level a world belongs to follows from its layout
    const uint8_t *areas;              // [n_levels][MAX_AGENTS][4 + 2 * stride]
    const uint8_t *level_of_layout;    // [L]
    uint32_t *exhausted;               // counts respawns that found no free cell in 1001 tries (the reference raises there)
};
static_assert(sizeof(SpawnCfg) == 56, "layout of the block behind Params::lut");
constexpr uint32_t SPAWN_CFG_OFFSET = 256 * 8 + 64 * 4;      // bytes behind Params::lut: the 256 doubles and the 64 submask words
// ... then two more per-lane constant tables (cz_create): the cell-coordinate image words of cells 0..1023, and for lane l
// which word of which recipe row it holds (load_recipe_rows): 8 r | 4 i << 8 for word i of the env's r-th recipe
constexpr uint32_t COORD_TABLE_OFFSET = SPAWN_CFG_OFFSET + 56, ROWSEL_TABLE_OFFSET = COORD_TABLE_OFFSET + 1024 * 4;
constexpr uint32_t LUT_BLOCK_BYTES = ROWSEL_TABLE_OFFSET + 64 * 4;
// status word: bit 8 + a = agent a is despawned; from bit 12: one grace countdown per agent, `bits` wide - 5 bits while the grace period
// is at most 31 (the layout every fixture has), else 20 / n_agents bits (one agent 20, two 10, three 6; four agents stay at 5).
// SpawnCfg::grace_period carries the period in its low 24 bits and the field width in bits 24..31.
constexpr int SPAWN_GONE0 = 8, SPAWN_GRACE0 = 12;
__host__ __device__ inline uint32_t spawn_grace_bits(uint32_t grace_period, int n_agents) { return grace_period <= 31u ? 5u : 20u / (uint32_t)n_agents; }
__host__ __device__ inline uint32_t spawn_max_grace(int n_agents) { const uint32_t b = 20u / (uint32_t)n_agents; return b > 5u ? (1u << b) - 1u : 31u; }
__host__ __device__ inline uint32_t spawn_initial_status(uint32_t packed_grace, int n_agents) {   // everybody present, grace running
    const uint32_t grace = packed_grace & 0xFFFFFFu, bits = packed_grace >> 24;
    uint32_t st = 0;
    for (int a = 0; a < n_agents; ++a) st |= grace << (SPAWN_GRACE0 + bits * a);
    return st;
}
__host__ __device__ inline uint64_t spawn_mix(uint64_t x) {                      // splitmix64 finaliser
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline double spawn_uniform(uint64_t seed, uint64_t env_global, uint64_t step_key, uint32_t agent, uint32_t draw) {
    uint64_t k = spawn_mix(seed + 0x9E3779B97F4A7C15ull * env_global);
    k = spawn_mix(k ^ (step_key * 0xD1B54A32D192ED03ull));
    k = spawn_mix(k ^ ((uint64_t)agent << 32) ^ (uint64_t)draw);
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

// wave-uniform bit set over N*64 positions (object slots or grid cells)
template <int N>
struct Mask {
    uint64_t w[N];
    __device__ __forceinline__ bool any() const {
        uint64_t o = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) o |= w[k];
        return o != 0;
    }
    __device__ __forceinline__ int count() const {
        int c = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) c += __popcll(w[k]);
        return c;
    }
    __device__ __forceinline__ int first() const {   // lowest set position, -1 if empty
        int r = -1;
#pragma unroll
        for (int k = N - 1; k >= 0; --k)
            if (w[k]) r = 64 * k + __ffsll((unsigned long long)w[k]) - 1;
        return r;
    }
    __device__ __forceinline__ int last() const {    // highest set position, -1 if empty
        int r = -1;
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (w[k]) r = 64 * k + 63 - __clzll((long long)w[k]);
        return r;
    }
    // (word / set / clear by arithmetic on every word, not "if (j == k) touch w[j]": the optimiser turns such a chain into a
    // dynamically indexed access, which sends the whole set - scalar registers - to a scratch-memory array)
    __device__ __forceinline__ uint64_t word(int k) const {
        if (N == 1) return w[0];
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) v |= w[j] & (0ull - (uint64_t)(j == k));
        return v;
    }
    __device__ __forceinline__ bool test(int s) const { return s >= 0 && ((word(s >> 6) >> (s & 63)) & 1); }
    __device__ __forceinline__ void set(int s) {
#pragma unroll
        for (int j = 0; j < N; ++j) w[j] |= (uint64_t)(N == 1 || j == (s >> 6)) << (s & 63);
    }
    __device__ __forceinline__ void clear(int s) {
#pragma unroll
        for (int j = 0; j < N; ++j) w[j] &= ~((uint64_t)(N == 1 || j == (s >> 6)) << (s & 63));
    }
    __device__ __forceinline__ Mask operator&(const Mask &o) const {
        Mask r;
#pragma unroll
        for (int k = 0; k < N; ++k) r.w[k] = w[k] & o.w[k];
        return r;
    }
    __device__ __forceinline__ Mask andnot(const Mask &o) const {
        Mask r;
#pragma unroll
        for (int k = 0; k < N; ++k) r.w[k] = w[k] & ~o.w[k];
        return r;
    }
    static __device__ __forceinline__ Mask zero() {
        Mask r;
#pragma unroll
        for (int k = 0; k < N; ++k) r.w[k] = 0;
        return r;
    }
};

// One env's world, resident in registers for the whole step (or the whole fused rollout).
template <int OPL, int CPL, int NA>
struct Env {
    uint32_t d0[OPL], d1[OPL];     // per lane: slots lane + 64k
    uint32_t cell[CPL];            // per lane: cells lane + 64k
    uint32_t agw;                  // lane a < NA: agent a as x | y<<8 | orientation<<16 | (held slot+1)<<24 (record word 8+a)
    uint32_t t, marks, layout, status, episode, recipes, pool;            // uniform header
    uint32_t marks_hi;             // wide recipe tables only (marks of recipes 2, 3); 0 otherwise
};

struct Ctx {                       // wave-uniform geometry + this lane's index
    int W, H, D, C, lane;
};

struct Dirty {                     // what this step changed (uniform)
    uint32_t pressed;              // a Switch was stepped on                    -> Blocks flip at end of step
    uint32_t touched;              // some object moved or changed state         -> recipe marks must be re-evaluated
    uint32_t interacted;           // containment / free flags may have changed  -> free-flag normalisation needed
    uint32_t cells;                // some mutable cell bit changed              -> the cell bytes must be written back
    uint32_t moved;                // a held object was carried to another cell  -> objects must be written back
    uint32_t statechg;             // some object changed its chop / blend state or was created (not a mere move)
    uint64_t kinds;                // bit 4c + s: an object of dynamic class c in state s = chopped | mashed << 1 moved or
                                   // changed (both its old and its new state are recorded); ~0 = anything (a plate drags
                                   // its content along).  Recipe filter, see step_env.
};

// dx + 1 / dy + 1 of get_target_location (cooking_world.py:172-184), two bits per action code 0..7
constexpr uint32_t DX_TABLE = 0x5561u, DY_TABLE = 0x5495u;

template <int OPL, int CPL, int NA, int SCHEME>
struct Ops {
    using E = Env<OPL, CPL, NA>;
    using OM = Mask<OPL>;
    using CM = Mask<CPL>;

    // Coding rule of this file (it decides the instruction count): a per-lane predicate is either ONE comparison handed to
    // ballot() - which is then a single v_cmp writing an SGPR pair - or a wave-uniform 64-bit lane mask combined with
    // scalar and / or / andn2 and handed back to the lanes with lanes(mask), which costs nothing (the mask is the condition
    // operand of the v_cndmask).  ballot(a && b) on the other hand makes the compiler materialise 0 / 1 in a VGPR and
    // compare again (two more VALU instructions per use).

    // ---- uniform reads / predicated writes of one slot / one cell ---------------------------------------
    static __device__ __forceinline__ uint32_t slot_d0(const E &e, int s) {
        if (OPL == 1) return rdl(e.d0[0], s);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < OPL; ++k)
            if ((s >> 6) == k) v = rdl(e.d0[k], s & 63);
        return v;
    }
    static __device__ __forceinline__ uint32_t slot_d1(const E &e, int s) {
        if (OPL == 1) return rdl(e.d1[0], s);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < OPL; ++k)
            if ((s >> 6) == k) v = rdl(e.d1[k], s & 63);
        return v;
    }
    static __device__ __forceinline__ uint32_t cell_at(const E &e, int c) {
        if (CPL == 1) return rdl(e.cell[0], c);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            if ((c >> 6) == k) v = rdl(e.cell[k], c & 63);
        return v;
    }
    static __device__ __forceinline__ void cell_update(E &e, const Ctx &cx, int c, uint32_t clear_bits, uint32_t xor_bits, Dirty &dt) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const bool mine = (cx.lane + 64 * k) == c;
            e.cell[k] = mine ? ((e.cell[k] & ~clear_bits) ^ xor_bits) : e.cell[k];
        }
        dt.cells = 1;
    }
    // cooking_world.py:223-227 square_walkable: Floor (0) and Switch (3) always, Block (4) by its bit (only Blocks
    // ever carry CELL_WALK), everything else never
    static __device__ __forceinline__ bool walkable(uint32_t cv) { return (((0x9u >> (cv & CELL_TYPE)) | (cv >> 6)) & 1u) != 0u; }
    // statics that can hold objects: Counter 1, Deliversquare 2, Cutboard 5, Blender 6.  Interactions aimed at a Floor,
    // Switch or Block cell are no-ops in every case (nothing is ever placed there; the reference's agent-at-location
    // guard, cooking_world.py:116,140,158, only ever fires for such cells), so they return early on the type.
    static __device__ __forceinline__ bool holds_objects(uint32_t ty) { return (0x66u >> ty) & 1u; }

    // ---- lane masks over the object slots: one comparison each -------------------------------------------
    static __device__ __forceinline__ OM m_d0(const E &e, uint32_t mask, uint32_t val) {       // (d0 & mask) == val
        OM m;
#pragma unroll
        for (int k = 0; k < OPL; ++k) m.w[k] = ballot((e.d0[k] & mask) == val);
        return m;
    }
    static __device__ __forceinline__ OM m_d0_any(const E &e, uint32_t bits) {                  // (d0 & bits) != 0
        OM m;
#pragma unroll
        for (int k = 0; k < OPL; ++k) m.w[k] = ballot((e.d0[k] & bits) != 0u);
        return m;
    }
    static __device__ __forceinline__ OM m_d1(const E &e, uint32_t mask, uint32_t val) {       // (d1 & mask) == val
        OM m;
#pragma unroll
        for (int k = 0; k < OPL; ++k) m.w[k] = ballot((e.d1[k] & mask) == val);
        return m;
    }
    // Plate.content membership.  Invariant of every record (checked by cz_set_state / cz_load_layouts, kept by every
    // operation below): a slot that is not alive carries d1 == 0, so a container tag (never 0) implies "alive".
    static __device__ __forceinline__ OM content_of(const E &e, int plate) { return m_d1(e, 0xFFu, (uint32_t)(plate + 1)); }
    static __device__ __forceinline__ OM outside_plates(const E &e) { return m_d1(e, 0xFFu, 0u); }   // (also dead slots: and it with alive ones)
    // static.content of a cell that holds objects == objects there that are not inside a plate (agents never stand on
    // such a cell, so nothing there is "held")
    static __device__ __forceinline__ OM direct_at(const E &e, uint32_t xy) {
        return m_d0(e, D_ALIVE | 0xFFFFu, D_ALIVE | xy) & outside_plates(e);
    }
    // Object.move_to / Plate.move_to (abstract_classes.py:20, world_objects.py:393-396): slot s and, when it is a
    // plate, everything inside it
    static __device__ __forceinline__ void move_obj(E &e, const Ctx &cx, int s, uint32_t xy) {
        OM m = content_of(e, s);
        m.set(s);
#pragma unroll
        for (int k = 0; k < OPL; ++k) e.d0[k] = lanes(m.w[k]) ? ((e.d0[k] & 0xFFFF0000u) | xy) : e.d0[k];
    }
    static __device__ __forceinline__ void slot_or(E &e, const Ctx &cx, int s, uint32_t bits) {
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            const bool mine = (cx.lane + 64 * k) == s;
            e.d0[k] = mine ? (e.d0[k] | bits) : e.d0[k];
        }
    }
    // Plate.add_content (world_objects.py:398-406): append s to plate p: every item free=False, the new last True
    static __device__ __forceinline__ void plate_add(E &e, const Ctx &cx, int p, int s, int cnt) {
        const uint32_t tag = (uint32_t)(p + 1);
        const OM inside = content_of(e, p);
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            const bool mine = (cx.lane + 64 * k) == s;
            const uint32_t d = lanes(inside.w[k]) ? (e.d0[k] & ~D_FREE) : e.d0[k];
            e.d0[k] = mine ? (d | D_FREE) : d;
            e.d1[k] = mine ? (tag | ((uint32_t)cnt << 8)) : e.d1[k];
        }
    }

    // One agent's view while it acts (uniform scalars, written back by the caller)
    struct Me { uint32_t xy; int h; };

    // record that the object with dyn0 word `w` moved: bit 4 * class + state of dt.kinds (a plate drags its content
    // along: step_env widens a set with a Plate bit to "anything")
    static __device__ __forceinline__ void touch(Dirty &dt, uint32_t w) {
        dt.touched = 1;
        dt.kinds |= 1ull << (((w >> 14) & 0x3Cu) | ((w >> 25) & 3u));
    }
    // ... or changed from `before` to `after` (chopped, mashed, created)
    static __device__ __forceinline__ void touch_change(Dirty &dt, uint32_t before, uint32_t after) {
        touch(dt, before);
        touch(dt, after);
        dt.statechg = 1;
    }

    // cooking_world.py:243-261 attempt_merge (first matching branch only, no fall-through on refusal).
    // dyn = objects at the target cell, sv = its cell byte, lxy the cell's x | y << 8, c its index
    static __device__ __forceinline__ void attempt_merge(E &e, const Ctx &cx, Me &me, const OM &dyn, uint32_t lxy, int c,
                                                         uint32_t sv, Dirty &dt) {
        const int held = me.h;
        const uint32_t hw = slot_d0(e, held);
        const uint32_t hcls = (hw >> 16) & 0xFF;
        const OM plates = dyn & m_d0(e, 0xFF0000u, PLATE << 16);
        const int np = plates.count();
        if (np == 1) {
            const int p = plates.first();
            const int cnt = content_of(e, p).count();
            // Plate.accepts world_objects.py:408-409: Food, done(), < 64 items
            if (hcls != PLATE && (hw & D_DONE) && cnt < 64) {
                plate_add(e, cx, p, held, cnt);
                move_obj(e, cx, held, lxy);                      // Agent.put_down world_objects.py:789-791
                me.h = -1;
                touch(dt, hw);
            }
        } else if (hcls == PLATE && dyn.any()) {
            const int o = dyn.last();                             // pick_index = -1
            const uint32_t ow = slot_d0(e, o);
            const int cnt = content_of(e, held).count();
            if ((ow & 0xFF0000u) != (PLATE << 16) && (ow & D_DONE) && cnt < 64) {
                plate_add(e, cx, held, o, cnt);
                move_obj(e, cx, o, me.xy);
                touch(dt, ow);
                // static_object.content.remove(o) is implicit: o now carries a container tag
            }
        } else {
            const int ncontent = (dyn & outside_plates(e)).count();
            const uint32_t ty = sv & CELL_TYPE;
            bool ok = false;
            if (ty == COUNTER || ty == DELIVERSQUARE) {                            // world_objects.py:64-66,107-108
                ok = ncontent < 1;
            } else if (ty == CUTBOARD) {                                           // :271-273 (every food is a ChopFood)
                ok = hcls != PLATE && ncontent < 1 && !(hw & D_CHOPPED);
            } else if (ty == BLENDER) {                                            // :337-338
                ok = (hcls == CARROT || hcls == BANANA) && !(sv & CELL_TOGGLE) && ncontent + 1 <= 1 && !(hw & D_MASHED);
            }
            if (ok) {
                if (ty >= CUTBOARD) cell_update(e, cx, c, CELL_READY, CELL_READY, dt);
                // add_content: the content list was empty, so the new item is its last element: free=True
                slot_or(e, cx, held, D_FREE);
                move_obj(e, cx, held, lxy);
                me.h = -1;
                touch(dt, hw);
            }
        }
    }

    // cooking_world.py:114-136 resolve_primary_interaction on the cell lxy = c whose byte is sv (a cell that holds objects)
    static __device__ __forceinline__ void primary(E &e, const Ctx &cx, Me &me, uint32_t lxy, int c, uint32_t sv, const OM &dyn,
                                                   Dirty &dt) {
        if (me.h < 0) {
            if (!dyn.any()) return;
            dt.interacted = 1;
            const OM direct = dyn & outside_plates(e);
            const int ncontent = direct.count();
            const uint32_t ty = sv & CELL_TYPE;
            // static_object.releases() with its side effects: world_objects.py:117-118,275-278,340-346
            bool rel = true;
            if (ty == DELIVERSQUARE) rel = false;
            else if (ty == CUTBOARD) {
                if (ncontent == 1 && (sv & CELL_READY)) cell_update(e, cx, c, CELL_READY, 0, dt);
            } else if (ty == BLENDER) {
                rel = !(sv & CELL_TOGGLE);
                if (rel && ncontent - 1 == 0 && (sv & CELL_READY)) cell_update(e, cx, c, CELL_READY, 0, dt);
            }
            if (rel) {
                const OM fr = dyn & m_d0_any(e, D_FREE);
                const int grab = fr.any() ? fr.first() : dyn.last();
                if (direct.test(grab)) {                        // object_to_grab in static_object.content
                    me.h = grab;                                // Agent.grab world_objects.py:785-787
                    touch(dt, slot_d0(e, grab));
                    move_obj(e, cx, grab, me.xy);
                }
            }
        } else {
            dt.interacted = 1;
            attempt_merge(e, cx, me, dyn, lxy, c, sv, dt);
        }
    }

    // cooking_world.py:138-154 resolve_interaction_pick_up_special (scheme1)
    static __device__ __forceinline__ void pick_up_special(E &e, const Ctx &cx, Me &me, const OM &dyn, Dirty &dt) {
        if (me.h >= 0 || !dyn.any()) return;
        const OM plates = dyn & m_d0(e, 0xFF0000u, PLATE << 16);
        if (plates.count() != 1) return;
        const int p = plates.first();
        const OM cm = content_of(e, p);
        const int cnt = cm.count();
        if (cnt == 0) return;                               // IndexError swallowed
        const OM lastm = cm & m_d1(e, 0xFF00u, (uint32_t)(cnt - 1) << 8);
        const int s = lastm.first();
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            const bool mine = (cx.lane + 64 * k) == s;
            e.d1[k] = mine ? 0u : e.d1[k];                 // content.pop(-1)
        }
        me.h = s;
        touch(dt, slot_d0(e, s));
        move_obj(e, cx, s, me.xy);
        dt.interacted = 1;
    }

    // cooking_world.py:156-170 resolve_execute_action; Cutboard.action world_objects.py:250-269;
    // ChopFood.chop abstract_classes.py:250-254; Bread.chop world_objects.py:738-745; Blender.action :356-360
    static __device__ __forceinline__ void execute(E &e, const Ctx &cx, uint32_t lxy, int c, uint32_t sv, const OM &dyn, Dirty &dt) {
        const uint32_t ty = sv & CELL_TYPE;
        if (ty == CUTBOARD) {
            if (!(sv & CELL_READY)) return;
            const OM fresh = dyn & outside_plates(e) & m_d0(e, D_CHOPPED, 0u);
            const int f = fresh.first();                    // first content item whose chop() executes
            // No such item on a READY board: the reference falls off Cutboard.action -> None -> TypeError at cooking_world.py:162.
            // REACHABLE (a mashed Banana on the board, absorbed by a Plate in hand through attempt_merge branch 2, leaves it READY and
            // empty); the build's answer is a no-op, pinned by tests/golden/refcrash_cutboard_scheme1.npz.
            if (f < 0) return;
            const uint32_t fw = slot_d0(e, f);
            slot_or(e, cx, f, D_CHOPPED);
            if ((fw & 0xFF0000u) == (BREAD << 16)) {
                // the clone: first not-alive Bread slot (head-room follows the originals), born chopped and free
                const OM spare = m_d0(e, D_ALIVE | 0xFF0000u, BREAD << 16);
                const int n = spare.first();
#pragma unroll
                for (int k = 0; k < OPL; ++k) {
                    const bool mine = (cx.lane + 64 * k) == n;      // n == -1 (no head-room left): nobody
                    e.d0[k] = mine ? (lxy | (BREAD << 16) | D_ALIVE | D_CHOPPED | D_FREE) : e.d0[k];
                    e.d1[k] = mine ? 0u : e.d1[k];
                }
            }
            cell_update(e, cx, c, CELL_READY, 0, dt);
            touch_change(dt, fw, fw | D_CHOPPED);               // (a Bread clone is a chopped Bread too)
            dt.interacted = 1;
        } else if (ty == BLENDER) {
            if (sv & CELL_READY) cell_update(e, cx, c, 0, CELL_TOGGLE, dt);
        }
    }

    // action_scheme3.py:4-43 / action_scheme1.py:4-40 perform_agent_actions (+ check_inbounds
    // cooking_world.py:192-204, check_collisions :206-221).  `act` : lane a = raw action of agent a.
    // The pre-pass and both filters run for all agents at once (lane a = agent a); the execution loop is serial
    // in agent order (action_scheme3.py:15-16) and pulls one agent out of the vectors with v_readlane.
    // bit `c` of a cell-set mask, per lane
    static __device__ __forceinline__ bool cell_bit(const CM &m, uint32_t c) {
        if constexpr (CPL == 1) {
            return ((m.w[0] >> (c & 63u)) & 1ull) != 0ull;
        } else {
            // Word by word, each as two 32-bit scalars: "pick the word by c >> 6, then shift" is a select chain over the set's
            // words, which the optimiser turns into a dynamically indexed load - and the set, eight scalar registers, into a
            // scratch-memory array: a store and a dependent load from memory in the middle of every step, and 6 KB of scratch
            // written per wave (the "WRITE_SIZE = 1.17 x algorithmic" of config 5 up to round 3; profiles/r04/README.md).
            uint32_t hit = 0u;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const uint32_t lo = (uint32_t)m.w[k], hi = (uint32_t)(m.w[k] >> 32);
                const uint32_t half = (c & 32u) ? hi : lo;
                hit |= ((c >> 6) == (uint32_t)k) ? (half >> (c & 31u)) : 0u;
            }
            return (hit & 1u) != 0u;
        }
    }
    // the value of lane `lane ^ 1` / of the next lanes of the quad (agents live in lanes 0..3: DPP, no LDS, no SGPR trip)
    template <int CTRL>
    static __device__ __forceinline__ uint32_t quad(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
    }

    // what the walking half hands to the interacting half (lane a = agent a; `inter`: the agents that interact)
    struct Pre { uint64_t inter; uint32_t txy, c, act; };

    static __device__ __forceinline__ void perform_agent_actions(E &e, const Ctx &cx, uint32_t act_raw, Dirty &dt) {
        const Pre pre = agents_walk(e, cx, act_raw, dt);
        agents_interact(e, cx, pre, dt);
    }

    // pre-pass, both filters and all walking: after this the agents' positions and orientations are final for the step
    static __device__ __forceinline__ Pre agents_walk(E &e, const Ctx &cx, uint32_t act_raw, Dirty &dt) {
        const uint32_t W = (uint32_t)cx.W, H = (uint32_t)cx.H;
        // (despawned agents, cz_set_spawn: bits 8.. of the status word, all 0 in batches without despawn / respawn)
        const uint64_t present_m = ~(uint64_t)((e.status >> SPAWN_GONE0) & 0xFu);
        const uint32_t agw = e.agw;                                        // 0 on lanes >= NA
        const uint32_t x = agw & 0xFFu, y = (agw >> 8) & 0xFFu, xy = agw & 0xFFFFu;
        // a negative action = the agent is despawned: it is not in the list world_step acts on (cooking_world.py:105-108:
        // no turn, no move, no part in the collision filter) but keeps its place in the world
        const uint64_t agent_m = ballot((int32_t)act_raw >= 0) & ((1ull << NA) - 1ull) & present_m;
        const uint32_t act0 = lanes(agent_m) ? (act_raw & 7u) : 0u;        // (lanes >= NA carry a copy of the last agent's action)
        const uint32_t sh = 2u * act0;
        const uint32_t tx = x + ((DX_TABLE >> sh) & 3u) - 1u, ty = y + ((DY_TABLE >> sh) & 3u) - 1u;   // negative wraps to huge
        // change_orientation before any filtering (lanes that are not agents have action 0 here: their word stays 0)
        uint32_t ag = lanes(ballot(act0 - 1u < 4u)) ? ((agw & 0xFF00FFFFu) | (act0 << 16)) : agw;
        // check_inbounds (0 and 5 pass; 6 and 7 aim at the own cell, always inside): outside -> action 0
        const uint64_t inb_m = ballot(tx < W) & ballot(ty < H);
        const uint32_t own = __umul24(y, W) + x;
        uint32_t txy = lanes(inb_m) ? (tx | (ty << 8)) : xy;
        uint32_t c = lanes(inb_m) ? (__umul24(ty, W) + tx) : own;
        uint64_t moving_m = ballot(act0 != 0u) & inb_m;                    // lanes whose action is not 0 (any more)
        // walkable cells and Switch cells as wave-uniform cell sets: what an agent lane needs to know about its target
        // and its own cell is a bit test (the cell bytes live in the cell lanes; a ballot is cheaper than two LDS permutes)
        CM walk_set, switch_set;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            walk_set.w[k] = ballot(walkable(e.cell[k]));
            switch_set.w[k] = ballot((e.cell[k] & CELL_TYPE) == SWITCH);
        }
        const uint64_t wk_m = ballot(cell_bit(walk_set, c));
        // check_collisions: an agent is cancelled iff its end cell equals another agent's end cell and its own
        // target was walkable.  Non-agents carry a value nobody else has.
        if (NA > 1) {
            const uint32_t endxy = lanes(wk_m) ? txy : xy;
            const uint32_t exy = lanes(agent_m) ? endxy : (0xFFFF0000u | (uint32_t)cx.lane);
            // (every DPP move runs with all lanes enabled: a lane switched off by a short-circuit would read as 0 elsewhere)
            uint64_t clash_m;
            if (NA == 2) {
                clash_m = ballot(exy == quad<0xB1>(exy));                                       // lanes 0 <-> 1
            } else {                                                                            // the other three lanes of the quad
                const uint32_t r1 = quad<0x39>(exy), r2 = quad<0x4E>(exy), r3 = quad<0x93>(exy);
                clash_m = ballot(exy == r1) | ballot(exy == r2) | ballot(exy == r3);
            }
            const uint64_t cancel_m = clash_m & wk_m & moving_m;           // now "walks" onto its own cell, with action 0
            txy = lanes(cancel_m) ? xy : txy;
            c = lanes(cancel_m) ? own : c;
            moving_m &= ~cancel_m;
        }
        const uint32_t act = lanes(moving_m) ? act0 : 0u;                  // the action after both filters (scheme1 looks at it)
        // ---- execution.  The reference resolves agents one after the other (action_scheme3.py:15-16), but a walking
        // agent only changes its own position, the position of what it carries and Switch bits, while an interacting
        // agent only changes objects on cells that hold objects, its own hands and READY/TOGGLE bits; collisions were
        // resolved above.  The two kinds of effects commute, so all walking happens at once (lane a = agent a) and
        // only the interacting agents are taken serially, in index order.
        uint64_t walks_m = agent_m & wk_m;                                 // action_scheme3.py:26-34 (scheme3: also action 0)
        if (SCHEME != 3) walks_m &= ballot(act - 1u < 4u);
        ag = lanes(walks_m) ? ((ag & 0xFFFF0000u) | txy) : ag;             // Agent.move_to world_objects.py:793-796
        e.agw = ag;
        uint64_t drag = walks_m & moving_m & ballot(ag > 0x00FFFFFFu);     // ... which drags what the agent holds
        if (drag) {
            dt.moved = 1;
            do {
                const int a = __builtin_ctzll(drag);
                drag &= drag - 1;
                const uint32_t A = rdl(e.agw, a);
                move_obj(e, cx, (int)(A >> 24) - 1, A & 0xFFFFu);
            } while (drag);
        }
        uint64_t press = walks_m & ballot(cell_bit(switch_set, c));        // Switch.add_content world_objects.py:159-163
        if (press) {
            dt.pressed = 1;
            do {
                const int a = __builtin_ctzll(press);
                press &= press - 1;
                cell_update(e, cx, (int)rdl(c, a), 0, CELL_ACTIVE, dt);
            } while (press);
        }
        return Pre{agent_m & ~walks_m & (SCHEME == 3 ? moving_m : ballot(act >= 5u)), txy, c, act};
    }

    // the interacting agents, serially in index order
    static __device__ __forceinline__ void agents_interact(E &e, const Ctx &cx, const Pre &pre, Dirty &dt) {
        const uint32_t W = (uint32_t)cx.W, H = (uint32_t)cx.H;
        const uint32_t txy = pre.txy, c = pre.c, act = pre.act;
        uint64_t inter = pre.inter;
        // A launch lasts as long as its slowest wave, and the slowest waves are the ones with the most to resolve: they
        // take the issue slots first (s_setprio: 1 with one interacting agent, 2 with more, 3 while recipe graphs are
        // evaluated in step_env), the quick majority fills in behind them.
        if (inter & (inter - 1)) CZ_SETPRIO(2);
        else if (inter) CZ_SETPRIO(1);
#ifndef CZ_INTER_UNROLL
#define CZ_INTER_UNROLL 0
#endif
#if CZ_INTER_UNROLL
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (!((inter >> a) & 1ull)) continue;
#else
        while (inter) {
            const int a = __builtin_ctzll(inter);
            inter &= inter - 1;
#endif
            const uint32_t A = rdl(e.agw, a);
            Me me{A & 0xFFFFu, (int)(A >> 24) - 1};
            // the cell in front (scheme3: the bumped cell; scheme1: by orientation, may be off-grid)
            uint32_t fxy;
            int fc;
            bool ok = true;
            if (SCHEME == 3) {
                fxy = rdl(txy, a);
                fc = (int)rdl(c, a);
            } else {
                const uint32_t o = (A >> 16) & 0xFFu;
                const uint32_t fx = (A & 0xFFu) + ((DX_TABLE >> (2u * o)) & 3u) - 1u;
                const uint32_t fy = ((A >> 8) & 0xFFu) + ((DY_TABLE >> (2u * o)) & 3u) - 1u;
                ok = fx < W && fy < H;                                      // reference: IndexError; build: no-op
                fxy = fx | (fy << 8);
                fc = (int)(fy * W + fx);
            }
            const uint32_t sv = ok ? cell_at(e, fc) : 0u;                   // READY / TOGGLE may have been changed by an earlier agent
            if (ok && holds_objects(sv & CELL_TYPE)) {
                // get_objects_at(location, DynamicObject) cooking_world.py:232-241 as a slot mask
                const OM dyn = m_d0(e, D_ALIVE | 0xFFFFu, D_ALIVE | fxy);
                if (SCHEME == 3) {
                    // action_scheme3.py:37-43: ActionObject with a not-done item -> execute, else primary
                    bool exec = false;
                    if ((sv & CELL_TYPE) >= CUTBOARD) exec = (dyn & m_d0(e, D_DONE, 0u)).any();
                    if (exec) execute(e, cx, fxy, fc, sv, dyn, dt);
                    else primary(e, cx, me, fxy, fc, sv, dyn, dt);
                } else {
                    const int ac = (int)rdl(act, a);
                    if (ac == 5) primary(e, cx, me, fxy, fc, sv, dyn, dt);
                    else if (ac == 6) pick_up_special(e, cx, me, dyn, dt);
                    else execute(e, cx, fxy, fc, sv, dyn, dt);
                }
                e.agw = wrl((A & 0x00FFFFFFu) | ((uint32_t)((me.h + 1) & 0xFF) << 24), a, e.agw);   // only the hands can change
            }
        }
    }

    // cooking_world.py:267-290 handle_agent_spawn for this world, agents in index order; returns the agents that left in this
    // step (bit a; they are reported truncated once, cooking_env.py:344-349).  Everything here is wave-uniform scalar work,
    // executed only by batches that have despawn / respawn switched on (cz_set_spawn).
    static __device__ __forceinline__ uint32_t handle_agent_spawn(E &e, const Ctx &cx, const SpawnCfg *cfg_generic, int64_t env_global) {
        typedef const __attribute__((address_space(4))) SpawnCfg *kcfg;
        const kcfg cfg = (kcfg)cfg_generic;
        const uint64_t seed = cfg->seed, key = ((uint64_t)e.episode << 32) | (uint64_t)e.t;
        const double despawn_rate = cfg->despawn_rate, respawn_rate = cfg->respawn_rate;
        uint32_t st = e.status, gone = 0u;
        const uint32_t packed_grace = cfg->grace_period, gbits = packed_grace >> 24, gmask = (1u << gbits) - 1u;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const uint32_t gsh = (uint32_t)SPAWN_GRACE0 + gbits * (uint32_t)a;
            if ((st >> gsh) & gmask) { st -= 1u << gsh; continue; }                  // agent_grace_period[i] -= 1
            const bool despawned = (st >> (SPAWN_GONE0 + a)) & 1u;
            if (!despawned) {
                const int n_active = NA - __popc((st >> SPAWN_GONE0) & 0xFu);
                if (n_active > 1 && spawn_uniform(seed, (uint64_t)env_global, key, (uint32_t)a, 0u) < despawn_rate) {
                    if ((rdl(e.agw, a) >> 24) == 0u) {                                // despawn_agent: an agent that holds something stays
                        st |= 1u << (SPAWN_GONE0 + a);
                        gone |= 1u << a;
                    }
                }
            } else if (spawn_uniform(seed, (uint64_t)env_global, key, (uint32_t)a, 1u) < respawn_rate) {
                // respawn_agent: back, grace period restarted, on a Floor cell of its spawn area that nobody - active or
                // not, itself included - stands on (generate_location: up to 1001 tries; the reference raises ValueError when
                // they are used up or a candidate lies beyond the grid, here such candidates are skipped, the agent comes
                // back where it stood and the handle counts the event: cz_spawn_exhausted)
                st = (st & ~(1u << (SPAWN_GONE0 + a))) | ((packed_grace & 0xFFFFFFu) << gsh);
                typedef const __attribute__((address_space(4))) uint8_t *kbytes;
                const uint32_t stride = cfg->stride, level = ((kbytes)cfg->level_of_layout)[e.layout];
                const kbytes blk = (kbytes)cfg->areas + (size_t)(level * (uint32_t)MAX_AGENTS + (uint32_t)a) * (4u + 2u * stride);
                const uint32_t nx = (uint32_t)blk[0] | ((uint32_t)blk[1] << 8), ny = (uint32_t)blk[2] | ((uint32_t)blk[3] << 8);
                bool placed = false;
#pragma nounroll
                for (uint32_t k = 0; k < 1001u; ++k) {
                    const uint32_t ix = (uint32_t)(spawn_uniform(seed, (uint64_t)env_global, key, (uint32_t)a, 2u + 2u * k) * (double)nx);
                    const uint32_t iy = (uint32_t)(spawn_uniform(seed, (uint64_t)env_global, key, (uint32_t)a, 3u + 2u * k) * (double)ny);
                    const uint32_t x = blk[4u + ix], y = blk[4u + stride + iy];
                    if (x >= (uint32_t)cx.W || y >= (uint32_t)cx.H) continue;
                    if ((cell_at(e, (int)(y * (uint32_t)cx.W + x)) & CELL_TYPE) != FLOOR) continue;
                    const uint32_t xy = x | (y << 8);
                    if (ballot((e.agw & 0xFFFFu) == xy) & ((1ull << NA) - 1ull)) continue;
                    e.agw = wrl((rdl(e.agw, a) & 0xFFFF0000u) | xy, a, e.agw);         // the location only (cooking_world.py:290)
                    placed = true;
                    break;
                }
                if (!placed && cx.lane == 0) atomicAdd(cfg->exhausted, 1u);
            }
        }
        e.status = st;
        return gone;
    }

    // cooking_world.py:77-88 progress_world (+ Blender.process world_objects.py:321-335, BlenderFood.blend
    // abstract_classes.py:266-273) and :90-92 resolve_linked_interactions
    static __device__ __forceinline__ void progress_and_link(E &e, const Ctx &cx, Dirty &dt) {
        // running blenders (rare: a toggle set in this step, or a scheme1 blender stuck on)
        CM running;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            running.w[k] = ballot((e.cell[k] & (CELL_TYPE | CELL_TOGGLE)) == (BLENDER | CELL_TOGGLE));
        while (running.any()) {
            int c = running.first();
            running.clear(c);
            uint32_t y = (uint32_t)c / (uint32_t)cx.W;
            uint32_t xy = ((uint32_t)c - y * (uint32_t)cx.W) | (y << 8);
            const OM content = direct_at(e, xy);
            if (content.any()) {
                const OM fresh = content & m_d0(e, D_DONE, 0u);
#pragma unroll
                for (int k = 0; k < OPL; ++k) e.d0[k] = lanes(fresh.w[k]) ? (e.d0[k] | D_MASHED) : e.d0[k];
                const OM mashed = content & m_d0_any(e, D_MASHED);
                if (mashed.count() == content.count()) cell_update(e, cx, c, CELL_READY | CELL_TOGGLE, 0, dt);
                dt.touched = 1;
                dt.statechg = 1;
                // the only BlenderFood classes; fresh (state 0) items become mashed (state 2)
                dt.kinds |= (5ull << (4u * CARROT)) | (5ull << (4u * BANANA));
                dt.interacted = 1;
            }
        }
        // free-flag normalisation (every container with content: all False, last True).  A pure function of the
        // containment relations, so it is a no-op on steps without an interaction and skipped there.
        //  * on a static: the item is alone (free) except on a Cutboard carrying Bread + clone
        //  * inside a plate: free iff it is the last appended (seq == count-1)
        //  * held directly by an agent: untouched
        if (dt.interacted) {
            // The pass over the statics can only change something if an object lying directly on a static is not free, or
            // if a Bread and its clone may share a Cutboard; both are rare, so test for them first (exact for any state).
            const OM outside = outside_plates(e);
            const OM unfree = m_d0(e, D_ALIVE | D_FREE, D_ALIVE) & outside;
            const OM breads = m_d0(e, D_ALIVE | D_CHOPPED | 0xFF0000u, D_ALIVE | D_CHOPPED | (BREAD << 16)) & outside;
            if (unfree.any() || breads.count() > 1) {
                OM held = OM::zero();
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int hs = (int)(rdl(e.agw, a) >> 24) - 1;
                    if (hs >= 0) held.set(hs);
                }
                // every object lying directly on a static is the last (only) item of that static's content ...
                const OM on_static = (m_d0_any(e, D_ALIVE) & outside).andnot(held);
#pragma unroll
                for (int k = 0; k < OPL; ++k) e.d0[k] = lanes(on_static.w[k]) ? (e.d0[k] | D_FREE) : e.d0[k];
                // ... except on a Cutboard that carries a Bread and its clone (the only way a static gets two items,
                // world_objects.py:738-745): both are chopped Breads lying directly on the same cell
                const OM twins = breads.andnot(held);
                if (twins.count() > 1) {
                    OM it = twins;
                    while (it.any()) {
                        const int s = it.first();
                        const uint32_t xy = slot_d0(e, s) & 0xFFFFu;
                        const OM content = direct_at(e, xy).andnot(held);
                        it = it.andnot(content);
                        if (content.count() > 1) {
                            OM rest = content;
                            rest.clear(content.last());
#pragma unroll
                            for (int k = 0; k < OPL; ++k) e.d0[k] = lanes(rest.w[k]) ? (e.d0[k] & ~D_FREE) : e.d0[k];
                        }
                    }
                }
            }
            // inside a plate: free iff last appended
            OM inside = m_d0_any(e, D_ALIVE).andnot(outside);
            while (inside.any()) {
                const int s = inside.first();
                const int p = (int)(slot_d1(e, s) & 0xFFu) - 1;
                const OM content = content_of(e, p);
                inside = inside.andnot(content);
                const int cnt = content.count();
                const OM lastone = content & m_d1(e, 0xFF00u, (uint32_t)(cnt - 1) << 8);
                const OM others = content.andnot(lastone);
#pragma unroll
                for (int k = 0; k < OPL; ++k) {
                    const uint32_t d = lanes(lastone.w[k]) ? (e.d0[k] | D_FREE) : e.d0[k];
                    e.d0[k] = lanes(others.w[k]) ? (d & ~D_FREE) : d;
                }
            }
        }
        // Switch.process_linked_objects world_objects.py:165-169 -> Block.switch_state :215-216 (all linked)
        if (dt.pressed) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const bool blk = (e.cell[k] & CELL_TYPE) == BLOCK;
                e.cell[k] = blk ? (e.cell[k] ^ CELL_WALK) : e.cell[k];
            }
        }
    }

    // objects a recipe node's class / state conditions accept: class and alive in one comparison, the accept mask over the
    // state index chopped | mashed << 1 in another
    static __device__ __forceinline__ OM node_matches(const E &e, uint32_t cval, uint32_t acc) {
        OM m;
#pragma unroll
        for (int k = 0; k < OPL; ++k)
            m.w[k] = ballot((e.d0[k] & 0x01FF0000u) == cval) & ballot(((acc >> ((e.d0[k] >> 25) & 3u)) & 1u) != 0u);
        return m;
    }
    // ... of which those that sit on a cell of the set `allow`
    static __device__ __forceinline__ OM on_cells(const E &e, const Ctx &cx, const OM &cand, const CM &allow) {
        OM m;
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            const uint32_t mycell = __umul24((e.d0[k] >> 8) & 0xFFu, (uint32_t)cx.W) + (e.d0[k] & 0xFFu);
            m.w[k] = cand.w[k] & ballot(cell_bit(allow, mycell));
        }
        return m;
    }
    // recipe.py:77-104 update_recipe_state for one recipe graph; returns the marks byte (bit j = node j marked).
    // Device node word (built by cz_load_recipes from the host table):
    //   bits 0..7   child mask                 bit 8       counts in the goal sum
    //   bit 9       static class, bits 10..12 = cell type (7 = matches nothing)
    //   bit 13      dynamic class: bits 16..23 = class id (as in dyn0), bits 24..27 = the accept mask over the object's
    //               state index chopped | mashed << 1 -- every combination of (attr, value) conditions on chop_state /
    //               blend_state (recipe.py:96-98) is one such mask, so a node may carry any number of conditions
    // A node's matched set is kept as a bit set of CELLS (LDS scratch `locs`), because the only thing a parent asks of
    // a child's matches is location equality (recipe.py:103).  Children always follow their parent in node_list, so
    // one pass from the last node to the first suffices.  Rolled: this is a cold path (steps that changed an object the
    // recipe can see, see step_env) and must not bloat the hot path's registers.
    static __device__ __forceinline__ uint32_t recipe_marks(const E &e, const Ctx &cx, uint32_t rowv, int rbase,
                                                            uint64_t *__restrict__ locs) {
        const int n = (int)(rdl(rowv, rbase) & 0xFFu);
        uint32_t marks = 0;
#pragma nounroll
        for (int j = n - 1; j >= 0; --j) {
            const uint32_t w = rdl(rowv, rbase + 1 + j);
            const uint32_t children = w & 0xFFu;
            if ((marks & children) != children) continue;                 // all(contains.marked)
            const uint32_t cval = (w & 0x00FF0000u) | D_ALIVE, acc = (w >> 24) & 0xFu;
            CM here = CM::zero();
            bool any;
            if (children == 0u && !(w & 0x200u)) {
                // leaf of a dynamic class (the common case): no location constraint
                OM m = node_matches(e, cval, acc);
                any = m.any();
                while (j > 0 && m.any()) {                                 // record where the matches are (a handful at most)
                    const int s = m.first();
                    m.clear(s);
                    const uint32_t o = slot_d0(e, s);
                    here.set((int)(((o >> 8) & 0xFFu) * (uint32_t)cx.W + (o & 0xFFu)));
                }
            } else {
                // intersection of the children's location sets
                CM allow;
#pragma unroll
                for (int q = 0; q < CPL; ++q) allow.w[q] = ~0ull;
                uint32_t ch = children;
                while (ch) {
                    const int c2 = __ffs((int)ch) - 1;
                    ch &= ch - 1;
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        const uint64_t lw = locs[c2 * CPL + q];
                        allow.w[q] &= ((uint64_t)rfl((uint32_t)(lw >> 32)) << 32) | rfl((uint32_t)lw);
                    }
                }
                if (w & 0x200u) {                                         // a static class: candidates are cells
                    const uint32_t cls = (w >> 10) & 7u;
#pragma unroll
                    for (int k = 0; k < CPL; ++k)
                        here.w[k] = ballot((e.cell[k] & CELL_TYPE) == cls) & ballot((cx.lane + 64 * k) < cx.C) & allow.w[k];
                    any = here.any();
                } else {                                                  // a dynamic class with children
                    OM m = on_cells(e, cx, node_matches(e, cval, acc), allow);
                    any = m.any();
                    while (j > 0 && m.any()) {
                        const int s = m.first();
                        m.clear(s);
                        const uint32_t o = slot_d0(e, s);
                        here.set((int)(((o >> 8) & 0xFFu) * (uint32_t)cx.W + (o & 0xFFu)));
                    }
                }
            }
            if (any) {
                marks |= 1u << j;
                if (j > 0 && cx.lane == 0) {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) locs[j * CPL + q] = here.w[q];
                }
            }
        }
        return marks;
    }

    // The same evaluation for WIDE recipe tables (graphs of up to 16 nodes): the node words come from the table in memory
    // (row: n, then per node  the device node word without its child field, the 16-bit child mask), the result has 16
    // bits.  A cold path by construction: batches whose book fits the compact tables never come here.
    // The same for ALL compact recipe graphs of the env in one pass (instances with up to 4 cells per lane; the 32x32 instance
    // keeps recipe_marks above, whose scratch is 16x smaller).  A node asks two things of a cell: "is there an object of my
    // class in an accepted state here / is this cell of my static class", and "are all my children satisfied HERE"
    // (recipe.py:103: a child's matches count only where the parent's candidate is); node j is marked iff both hold on some
    // cell.  `which` = bit r: evaluate recipe r (rows 9r.. of rowv); returns the marks bytes of those recipes (bits 8r + j), 0
    // for the others.
    // bitwise OR over the wave (DPP row shifts and row broadcasts; the result is wave-uniform)
    static __device__ __forceinline__ uint32_t wave_or(uint32_t x) {
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);      // row_shr:1
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);      // row_shr:2
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);      // row_shr:4
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);      // row_shr:8: lane 15 of a row = the row
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xF, 0xF, true);      // row_bcast:15
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xF, 0xF, true);      // row_bcast:31
        return rdl(x, 63);
    }
    // Node-parallel where the nodes are (their lanes), object-parallel where the objects are, cell-parallel for the rest:
    //   B. every node lane (lane 9r + 1 + j of rowv) ORs its bit 8r + j into NK[kind] for each kind its class / accept mask
    //      takes (dynamic class) or into NS[cell type] (static class);
    //   C. every live object reads NK[its kind] - the nodes that accept it - and ORs that into U[its cell];
    //   D. every cell lane reads U[cell] | NS[its type]: the nodes whose class condition holds on this cell;
    //   E. S (a bit per node, per cell lane): S_j = U_j && (S & children_j) == children_j, nodes from last to first;
    //   F. marks = OR of S over the cells.
    // Four LDS round trips of one or two instructions each, no per-match and no per-object loops; the only serial part is E,
    // five vector and four scalar instructions per node.
    static __device__ __forceinline__ uint32_t recipe_marks_cells(const E &e, const Ctx &cx, uint32_t rowv, uint32_t which, int R,
                                                                  uint64_t *__restrict__ tbl64) {
        static_assert(CPL <= 4, "scratch: 64 * CPL + 8 words of 64 bits");
        uint32_t *const U = reinterpret_cast<uint32_t *>(tbl64);       // [64 * CPL] node bits per cell
        uint32_t *const NK = U + 64 * CPL;                              // [64] node bits per object kind (4 * class + state)
        uint32_t *const NS = NK + 64;                                   // [8] node bits per static cell type
        // (the lane number through an opaque move: what is computed from it below - node lane or not, which recipe, which node -
        // would otherwise be hoisted to the top of the kernel and live in scalar register pairs through every step)
        uint32_t lane = (uint32_t)cx.lane;
        asm volatile("" : "+v"(lane));
        which &= (1u << R) - 1u;
#pragma unroll
        for (int q = 0; q < CPL; ++q) U[lane + 64 * q] = 0u;
        NK[lane] = 0u;
        if (lane < 8u) NS[lane] = 0u;
        // (B and C without branches: a lane with nothing to add ORs 0 into a word of its own, so that idle lanes neither switch
        // exec nor pile up on one address, where an LDS atomic would take them one after the other)
        {   // B
            const uint32_t r_of = (lane * 57u) >> 9, j_of = lane - 9u * r_of - 1u;          // lane / 9, lane % 9 - 1
            const bool mine = r_of < (uint32_t)R && j_of < (uint32_t)MAX_NODES && ((which >> r_of) & 1u) != 0u;
            const uint32_t w = mine ? rowv : 0u;                                            // (rows are zero-padded behind their nodes)
            const uint32_t nb = 1u << ((8u * r_of + j_of) & 31u);
            // (selects by arithmetic on 0 / ~0 words: lane masks would occupy a pair of scalar registers each)
            const uint32_t base = (w >> 14) & 0x3Cu, acc = ((w >> 24) & 0xFu) & (0u - ((w >> 13) & 1u));
#pragma unroll
            for (uint32_t st = 0; st < 4u; ++st) {
                const uint32_t on = 0u - ((acc >> st) & 1u);
                __hip_atomic_fetch_or(NK + lane + ((base + st - lane) & on), nb & on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            const uint32_t stat = 0u - ((w >> 9) & 1u);                                     // NS follows NK: word 64 + type
            __hip_atomic_fetch_or(NK + lane + ((64u + ((w >> 10) & 7u) - lane) & stat), nb & stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        // One wave owns this LDS region and the DS operations of a wave execute in order, so no hardware barrier is needed -
        // but the compiler must not forward a lane's own earlier stores to its loads below (what other lanes wrote in between
        // is invisible to its per-thread view of memory).
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < OPL; ++k) {   // C
            const uint32_t w = e.d0[k];
            const uint32_t alive = 0u - ((w >> 24) & 1u);
            static_assert(D_ALIVE == 1u << 24, "bit 24");
            const uint32_t kind = ((w >> 14) & 0x3Cu) | ((w >> 25) & 3u);
            const uint32_t cell = __umul24((w >> 8) & 0xFFu, (uint32_t)cx.W) + (w & 0xFFu);
            const uint32_t nodes = NK[kind];
            __hip_atomic_fetch_or(U + lane + ((cell - lane) & alive), nodes & alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        uint32_t Uq[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {   // D
            const uint32_t st = NS[e.cell[q] & CELL_TYPE];
            Uq[q] = U[lane + 64 * q] | ((lane + 64u * q) < (uint32_t)cx.C ? st : 0u);
        }
        // E.  A wave that gets here is, more often than not, the last one of its launch (0.3 % of the waves of the bench
        // workload; a launch lasts as long as its slowest wave), so this is written for latency, not for instruction count: no
        // loops, no branches, no lane reads inside the dependent chain.  Every recipe has its own 8-bit S (a node's children
        // are nodes of the same recipe), so the chains of different recipes are independent; nodes run from 7 (or 3)
        // down to 0 whatever the graph's size (rows are zero-padded: a node that does not exist has no U bit and adds
        // nothing), three dependent vector instructions per node: S & need, == need, select(S, S | U & bit).  Only the recipes
        // in `which` run (a wave-uniform branch each), graphs of up to four nodes run four levels.  (Until round 4: a loop over the recipes and their nodes with two lane reads,
        // a shift and a branch per node in front of the same arithmetic: 0.9 us for two five-node graphs, now ~0.3.)
        uint32_t all = 0u;
        const auto chain = [&](int r, auto from_tag) {           // nodes FROM - 1 .. 0 of recipe r
            constexpr int FROM = decltype(from_tag)::value;
            uint32_t need[FROM];
#pragma unroll
            for (int j = 0; j < FROM; ++j) need[j] = rdl(rowv, 9 * r + 1 + j) & 0xFFu;
            uint32_t Ur[CPL], S[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) { Ur[q] = (Uq[q] >> (8 * r)) & 0xFFu; S[q] = 0u; }
#pragma unroll
            for (int j = FROM - 1; j >= 0; --j)
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const uint32_t with = S[q] | (Ur[q] & (1u << j));
                    S[q] = ((S[q] & need[j]) == need[j]) ? with : S[q];
                }
#pragma unroll
            for (int q = 0; q < CPL; ++q) all |= S[q] << (8 * r);
        };
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!((which >> r) & 1u)) continue;                  // (wave-uniform: `which` has no bits past R)
            if ((rdl(rowv, 9 * r) & 0xFFu) > 4u) chain(r, std::integral_constant<int, MAX_NODES>{});
            else chain(r, std::integral_constant<int, 4>{});
        }
        return wave_or(all);              // F
    }
    static __device__ __forceinline__ uint32_t recipe_marks_wide(const E &e, const Ctx &cx, const uint32_t *__restrict__ row,
                                                              uint64_t *__restrict__ locs) {
        const int n = (int)(rfl(row[0]) & 0xFFu);
        uint32_t marks = 0;
#pragma nounroll
        for (int j = n - 1; j >= 0; --j) {
            const uint32_t w = rfl(row[1 + 2 * j]), children = rfl(row[2 + 2 * j]) & 0xFFFFu;
            if ((marks & children) != children) continue;                 // all(contains.marked)
            const uint32_t cval = (w & 0x00FF0000u) | D_ALIVE, acc = (w >> 24) & 0xFu;
            CM allow;
#pragma unroll
            for (int q = 0; q < CPL; ++q) allow.w[q] = ~0ull;
            uint32_t ch = children;
            while (ch) {
                const int c2 = __ffs((int)ch) - 1;
                ch &= ch - 1;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const uint64_t lw = locs[c2 * CPL + q];
                    allow.w[q] &= ((uint64_t)rfl((uint32_t)(lw >> 32)) << 32) | rfl((uint32_t)lw);
                }
            }
            CM here = CM::zero();
            if (w & 0x200u) {                                             // a static class: candidates are cells
                const uint32_t cls = (w >> 10) & 7u;
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    here.w[k] = ballot((e.cell[k] & CELL_TYPE) == cls) & ballot((cx.lane + 64 * k) < cx.C) & allow.w[k];
            } else if (w & 0x2000u) {                                     // a dynamic class
                OM m = on_cells(e, cx, node_matches(e, cval, acc), allow);
                while (m.any()) {
                    const int sl = m.first();
                    m.clear(sl);
                    const uint32_t o = slot_d0(e, sl);
                    here.set((int)(((o >> 8) & 0xFFu) * (uint32_t)cx.W + (o & 0xFFu)));
                }
            }
            if (here.any()) {
                marks |= 1u << j;
                if (cx.lane == 0) {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) locs[j * CPL + q] = here.w[q];
                }
            }
        }
        return marks;
    }
};

}  // namespace cz
