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
constexpr int LUT_Y0 = 63, LUT_ZERO = 126, LUT_ONE = 127;   // + entries 128..255 = 0.0, the "absent" zone
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
    uint32_t *stat_u;              // [N][SU_WORDS]
    double *stat_f;                // [N][SF_WORDS]
    int64_t env_id_base;
    uint64_t seed;
    double recipe_reward, recipe_penalty, node_reward, time_penalty_step;
    double reward_idle;            // the reward formula evaluated with no goal change (host, same op order)
    const double *lut;             // [256] host-computed quotients: [i] (i-(W-1))/W, [63+i] (i-(H-1))/H, [126] 0, [127] 1, rest 0
    uint32_t inv_w;                // ceil(65536 / W): c / W == (c * inv_w) >> 16 for every cell index c
    int32_t N, A, W, H, D, F, RW, scheme, max_steps, end_all, R, auto_reset, L;
    int32_t T;                     // fused steps per launch (1 for cz_step)
    uint32_t step0;
    int32_t dyn0_off, dyn1_off;    // word offsets inside a record
    int32_t wt;                    // 1: observation stores are write-through (sc1); chosen per launch by the host
    int32_t walk_touches;          // 1 if carrying an object across cells can change a recipe mark (see cz_load_recipes)
    int32_t wide;                  // 1: wide recipe tables (up to 16 nodes per graph, marks in record words 1 and 7)
    int32_t stop;                  // ablation build only: phase index after which the kernel returns (CZ_STOP), else -1
    // (the argument block is 56 + 264 bytes = five 64-byte lines exactly; one more field costs every launch a sixth)
#ifdef CZ_PROFILE
    unsigned long long *stamps;    // diagnostic build only: [N][8] s_memtime stamps (that build does not overlap launches)
#else
    uint32_t *chain_err;           // pinned host word: an overlapped launch whose hand-off never came sets it (cz_sync fails on it)
#endif
    uint32_t seq;                  // per-env hand-off of overlapped launches (SEQ_*); travels as a leading scalar argument
};
static_assert(sizeof(Params) == 264, "argument block: see the note above");
// Overlapped ("chained") launches: consecutive step kernels of a run go to two streams alternately, so a kernel may start
// while its predecessor still runs; what orders them is a sequence word per env (64 B apart, right behind the records):
// a wave waits until its env's word equals the launch's number, steps, and publishes number + 1.  The launch boundary
// (1.6 us + start skew on this part) then hides behind the other kernel's work.  Everything a step reads of its
// predecessor's (the record, the per-env statistics) therefore moves with device-scope loads and write-through stores:
// the L2 of another XCD is not coherent for ordinary accesses between launch boundaries.
constexpr uint32_t SEQ_PUBLISH = 1u << 31, SEQ_WAIT = 1u << 30, SEQ_MASK = SEQ_WAIT - 1u;
constexpr int SEQ_STRIDE_WORDS = 16;

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
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
__host__ __device__ inline uint32_t next_layout(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t n_layouts) {
    uint32_t base = pool_word & 0xFFFFu, count = pool_word >> 16;
    if (count == 0) { base = 0; count = n_layouts; }
    return base + (uint32_t)(((uint64_t)env_global + (uint64_t)episode * 7919u) % count);
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
    __device__ __forceinline__ uint64_t word(int k) const {
        if (N == 1) return w[0];
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j == k) v = w[j];
        return v;
    }
    __device__ __forceinline__ bool test(int s) const { return s >= 0 && ((word(s >> 6) >> (s & 63)) & 1); }
    __device__ __forceinline__ void set(int s) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (N == 1 || j == (s >> 6)) w[j] |= 1ull << (s & 63);
    }
    __device__ __forceinline__ void clear(int s) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (N == 1 || j == (s >> 6)) w[j] &= ~(1ull << (s & 63));
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

    // ---- uniform reads / predicated writes of one slot / one cell ---------------------------------------
    static __device__ __forceinline__ uint32_t slot_d0(const E &e, int s) {
        if (OPL == 1) return rdl(e.d0[0], s);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < OPL; ++k)
            if ((s >> 6) == k) v = rdl(e.d0[k], s & 63);
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
        for (int k = 0; k < CPL; ++k)
            if ((CPL == 1 || (c >> 6) == k) && cx.lane == (c & 63)) e.cell[k] = (e.cell[k] & ~clear_bits) ^ xor_bits;
        dt.cells = 1;
    }
    // cooking_world.py:223-227 square_walkable: Floor (0) and Switch (3) always, Block (4) by its bit (only Blocks
    // ever carry CELL_WALK), everything else never
    static __device__ __forceinline__ bool walkable(uint32_t cv) { return ((0x9u >> (cv & CELL_TYPE)) | (cv >> 6)) & 1u; }
    // statics that can hold objects: Counter 1, Deliversquare 2, Cutboard 5, Blender 6.  Interactions aimed at a Floor,
    // Switch or Block cell are no-ops in every case (nothing is ever placed there; the reference's agent-at-location
    // guard, cooking_world.py:116,140,158, only ever fires for such cells), so they return early on the type.
    static __device__ __forceinline__ bool holds_objects(uint32_t ty) { return (0x66u >> ty) & 1u; }

    // ---- ballots over object lanes ----------------------------------------------------------------------
    template <class F>
    static __device__ __forceinline__ OM oballot(const E &e, F pred) {
        OM m;
#pragma unroll
        for (int k = 0; k < OPL; ++k) m.w[k] = ballot(pred(e.d0[k], e.d1[k]));
        return m;
    }
    static __device__ __forceinline__ OM content_of(const E &e, int plate) {   // Plate.content membership
        uint32_t tag = (uint32_t)(plate + 1);
        return oballot(e, [=](uint32_t a, uint32_t b) { return (a & D_ALIVE) && (b & 0xFFu) == tag; });
    }
    // static.content of a cell that holds objects == objects there that are not inside a plate (agents never stand on
    // such a cell, so nothing there is "held")
    static __device__ __forceinline__ OM direct_at(const E &e, uint32_t xy) {
        return oballot(e, [=](uint32_t a, uint32_t b) { return (a & (D_ALIVE | 0xFFFFu)) == (D_ALIVE | xy) && (b & 0xFFu) == 0; });
    }
    // Object.move_to / Plate.move_to (abstract_classes.py:20, world_objects.py:393-396): slot s and, when it is a
    // plate, everything inside it
    static __device__ __forceinline__ void move_obj(E &e, const Ctx &cx, int s, uint32_t xy) {
        uint32_t tag = (uint32_t)(s + 1);
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            bool me = (cx.lane + 64 * k) == s || ((e.d0[k] & D_ALIVE) && (e.d1[k] & 0xFFu) == tag);
            if (me) e.d0[k] = (e.d0[k] & 0xFFFF0000u) | xy;
        }
    }
    static __device__ __forceinline__ void slot_or(E &e, const Ctx &cx, int s, uint32_t bits) {
#pragma unroll
        for (int k = 0; k < OPL; ++k)
            if (cx.lane + 64 * k == s) e.d0[k] |= bits;
    }
    // Plate.add_content (world_objects.py:398-406): append s to plate p: every item free=False, the new last True
    static __device__ __forceinline__ void plate_add(E &e, const Ctx &cx, int p, int s, int cnt) {
        uint32_t tag = (uint32_t)(p + 1);
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            if ((e.d0[k] & D_ALIVE) && (e.d1[k] & 0xFFu) == tag) e.d0[k] &= ~D_FREE;
            if (cx.lane + 64 * k == s) {
                e.d1[k] = tag | ((uint32_t)cnt << 8);
                e.d0[k] |= D_FREE;
            }
        }
    }

    // One agent's view while it acts (uniform scalars, written back by the caller)
    struct Me { int x, y, o, h; };

    // record that the object with dyn0 word `w` moved; a plate drags its content along -> anything
    static __device__ __forceinline__ void touch(Dirty &dt, uint32_t w) {
        const uint32_t cls = (w >> 16) & 0xFFu;
        dt.touched = 1;
        dt.kinds |= (cls == PLATE) ? ~0ull : (1ull << (4u * cls + ((w >> 25) & 3u)));
    }
    // ... or changed from `before` to `after` (chopped, mashed, created)
    static __device__ __forceinline__ void touch_change(Dirty &dt, uint32_t before, uint32_t after) {
        touch(dt, before);
        touch(dt, after);
        dt.statechg = 1;
    }

    // cooking_world.py:243-261 attempt_merge (first matching branch only, no fall-through on refusal).
    // dyn = objects at the target cell, sv = its cell byte, (lx, ly) the cell
    static __device__ __forceinline__ void attempt_merge(E &e, const Ctx &cx, Me &me, const OM &dyn, int lx, int ly, int c,
                                                         uint32_t sv, Dirty &dt) {
        const uint32_t lxy = (uint32_t)lx | ((uint32_t)ly << 8);
        const int held = me.h;
        const uint32_t hw = slot_d0(e, held);
        const uint32_t hcls = (hw >> 16) & 0xFF;
        OM plates = dyn & oballot(e, [](uint32_t a, uint32_t) { return (a & 0xFF0000u) == (PLATE << 16); });
        int np = plates.count();
        if (np == 1) {
            int p = plates.first();
            int cnt = content_of(e, p).count();
            // Plate.accepts world_objects.py:408-409: Food, done(), < 64 items
            if (hcls != PLATE && (hw & D_DONE) && cnt < 64) {
                plate_add(e, cx, p, held, cnt);
                move_obj(e, cx, held, lxy);                      // Agent.put_down world_objects.py:789-791
                me.h = -1;
                touch(dt, hw);
            }
        } else if (hcls == PLATE && dyn.any()) {
            int o = dyn.last();                                   // pick_index = -1
            uint32_t ow = slot_d0(e, o);
            int cnt = content_of(e, held).count();
            if ((ow & 0xFF0000u) != (PLATE << 16) && (ow & D_DONE) && cnt < 64) {
                plate_add(e, cx, held, o, cnt);
                move_obj(e, cx, o, (uint32_t)me.x | ((uint32_t)me.y << 8));
                touch(dt, ow);
                // static_object.content.remove(o) is implicit: o now carries a container tag
            }
        } else {
            int ncontent = (dyn & oballot(e, [](uint32_t, uint32_t b) { return (b & 0xFFu) == 0; })).count();
            uint32_t ty = sv & CELL_TYPE;
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

    // cooking_world.py:114-136 resolve_primary_interaction on cell (lx, ly) = c whose byte is sv (a cell that holds objects)
    static __device__ __forceinline__ void primary(E &e, const Ctx &cx, Me &me, int lx, int ly, int c, uint32_t sv, const OM &dyn,
                                                   Dirty &dt) {
        if (me.h < 0) {
            if (!dyn.any()) return;
            dt.interacted = 1;
            OM direct = dyn & oballot(e, [](uint32_t, uint32_t b) { return (b & 0xFFu) == 0; });
            int ncontent = direct.count();
            uint32_t ty = sv & CELL_TYPE;
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
                OM fr = dyn & oballot(e, [](uint32_t a, uint32_t) { return (a & D_FREE) != 0; });
                int grab = fr.any() ? fr.first() : dyn.last();
                if (direct.test(grab)) {                        // object_to_grab in static_object.content
                    me.h = grab;                                // Agent.grab world_objects.py:785-787
                    touch(dt, slot_d0(e, grab));
                    move_obj(e, cx, grab, (uint32_t)me.x | ((uint32_t)me.y << 8));
                }
            }
        } else {
            dt.interacted = 1;
            attempt_merge(e, cx, me, dyn, lx, ly, c, sv, dt);
        }
    }

    // cooking_world.py:138-154 resolve_interaction_pick_up_special (scheme1)
    static __device__ __forceinline__ void pick_up_special(E &e, const Ctx &cx, Me &me, const OM &dyn, Dirty &dt) {
        if (me.h >= 0 || !dyn.any()) return;
        OM plates = dyn & oballot(e, [](uint32_t a, uint32_t) { return (a & 0xFF0000u) == (PLATE << 16); });
        if (plates.count() != 1) return;
        int p = plates.first();
        OM cm = content_of(e, p);
        int cnt = cm.count();
        if (cnt == 0) return;                               // IndexError swallowed
        uint32_t want = (uint32_t)(cnt - 1);
        OM lastm = cm & oballot(e, [=](uint32_t, uint32_t b) { return ((b >> 8) & 0xFFu) == want; });
        int s = lastm.first();
#pragma unroll
        for (int k = 0; k < OPL; ++k)
            if (cx.lane + 64 * k == s) e.d1[k] = 0;        // content.pop(-1)
        me.h = s;
        touch(dt, slot_d0(e, s));
        move_obj(e, cx, s, (uint32_t)me.x | ((uint32_t)me.y << 8));
        dt.interacted = 1;
    }

    // cooking_world.py:156-170 resolve_execute_action; Cutboard.action world_objects.py:250-269;
    // ChopFood.chop abstract_classes.py:250-254; Bread.chop world_objects.py:738-745; Blender.action :356-360
    static __device__ __forceinline__ void execute(E &e, const Ctx &cx, int lx, int ly, int c, uint32_t sv, const OM &dyn, Dirty &dt) {
        const uint32_t ty = sv & CELL_TYPE;
        if (ty == CUTBOARD) {
            if (!(sv & CELL_READY)) return;
            const uint32_t lxy = (uint32_t)lx | ((uint32_t)ly << 8);
            OM fresh = dyn & oballot(e, [](uint32_t a, uint32_t b) { return (b & 0xFFu) == 0 && !(a & D_CHOPPED); });
            int f = fresh.first();                          // first content item whose chop() executes
            if (f < 0) return;                              // reference falls off Cutboard.action (TypeError); unreachable
            uint32_t fw = slot_d0(e, f);
            slot_or(e, cx, f, D_CHOPPED);
            if ((fw & 0xFF0000u) == (BREAD << 16)) {
                // the clone: first not-alive Bread slot (head-room follows the originals), born chopped and free
                OM spare = oballot(e, [](uint32_t a, uint32_t) { return (a & (D_ALIVE | 0xFF0000u)) == (BREAD << 16); });
                int n = spare.first();
#pragma unroll
                for (int k = 0; k < OPL; ++k)
                    if (n >= 0 && cx.lane + 64 * k == n) {
                        e.d0[k] = lxy | (BREAD << 16) | D_ALIVE | D_CHOPPED | D_FREE;
                        e.d1[k] = 0;
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
        uint64_t w = m.w[0];
#pragma unroll
        for (int k = 1; k < CPL; ++k)
            if ((c >> 6) == (uint32_t)k) w = m.w[k];
        return ((w >> (c & 63u)) & 1ull) != 0ull;
    }
    // the value of lane `lane ^ 1` / of the next lanes of the quad (agents live in lanes 0..3: DPP, no LDS, no SGPR trip)
    template <int CTRL>
    static __device__ __forceinline__ uint32_t quad(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
    }

    static __device__ __forceinline__ void perform_agent_actions(E &e, const Ctx &cx, uint32_t act, Dirty &dt) {
        const uint32_t W = (uint32_t)cx.W, H = (uint32_t)cx.H;
        uint32_t x = e.agw & 0xFFu, y = (e.agw >> 8) & 0xFFu, o = (e.agw >> 16) & 0xFFu;
        // a negative action = the agent is despawned: it is not in the list world_step acts on (cooking_world.py:105-108:
        // no turn, no move, no part in the collision filter) but keeps its place in the world
        const bool live = (int32_t)act >= 0;
        act = live ? (act & 7u) : 0u;
        if (act - 1u < 4u) o = act;                                         // change_orientation before any filtering
        uint32_t tx = x + ((DX_TABLE >> (2u * act)) & 3u) - 1u, ty = y + ((DY_TABLE >> (2u * act)) & 3u) - 1u;
        // check_inbounds (0 and 5 pass; 6 and 7 aim at the own cell, always inside); negative wraps to huge
        if (tx >= W || ty >= H) { act = 0; tx = x; ty = y; }
        const uint32_t own = y * W + x;
        uint32_t c = ty * W + tx;
        // walkable cells and Switch cells as wave-uniform cell sets: what an agent lane needs to know about its target
        // and its own cell is a bit test (the cell bytes live in the cell lanes; a ballot is cheaper than two LDS permutes)
        CM walk_set, switch_set;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            walk_set.w[k] = ballot(walkable(e.cell[k]));
            switch_set.w[k] = ballot((e.cell[k] & CELL_TYPE) == SWITCH);
        }
        const bool wk = cell_bit(walk_set, c);
        // check_collisions: an agent is cancelled iff its end cell equals another agent's end cell and its own
        // target was walkable.  Non-agents carry a value nobody else has.
        const bool is_agent = cx.lane < NA && live;
        const uint32_t exy = !is_agent ? (0xFFFF0000u | (uint32_t)cx.lane) : (wk ? (tx | (ty << 8)) : (x | (y << 8)));
        // (every DPP move runs with all lanes enabled: a lane switched off by a short-circuit would read as 0 elsewhere)
        bool clash = false;
        if (NA == 2) clash = exy == quad<0xB1>(exy);                                            // lanes 0 <-> 1
        if (NA > 2) {                                                                           // the other three lanes of the quad
            const uint32_t r1 = quad<0x39>(exy), r2 = quad<0x4E>(exy), r3 = quad<0x93>(exy);
            clash = (exy == r1) | (exy == r2) | (exy == r3);
        }
        if (NA > 1 && clash && wk && act != 0u) { act = 0; tx = x; ty = y; c = own; }   // now "walks" onto its own cell
        if (is_agent) e.agw = (e.agw & 0xFF00FFFFu) | (o << 16);              // lanes >= NA stay 0 (unused agent words)
        // ---- execution.  The reference resolves agents one after the other (action_scheme3.py:15-16), but a walking
        // agent only changes its own position, the position of what it carries and Switch bits, while an interacting
        // agent only changes objects on cells that hold objects, its own hands and READY/TOGGLE bits; collisions were
        // resolved above.  The two kinds of effects commute, so all walking happens at once (lane a = agent a) and
        // only the interacting agents are taken serially, in index order.
        const bool is_walk_act = SCHEME == 3 || (act - 1u < 4u);
        const bool walks = is_agent && is_walk_act && wk;                 // action_scheme3.py:26-34 (scheme3: also action 0)
        const uint32_t txy = tx | (ty << 8);
        if (walks) e.agw = (e.agw & 0xFFFF0000u) | txy;                    // Agent.move_to world_objects.py:793-796
        uint64_t drag = ballot(walks && act != 0u && (e.agw >> 24) != 0u); // ... which drags what the agent holds
        while (drag) {
            const int a = __ffsll((unsigned long long)drag) - 1;
            drag &= drag - 1;
            const uint32_t A = rdl(e.agw, a);
            move_obj(e, cx, (int)(A >> 24) - 1, A & 0xFFFFu);
            dt.moved = 1;
        }
        uint64_t press = ballot(walks && cell_bit(switch_set, c));         // Switch.add_content world_objects.py:159-163
        while (press) {
            const int a = __ffsll((unsigned long long)press) - 1;
            press &= press - 1;
            cell_update(e, cx, (int)rdl(c, a), 0, CELL_ACTIVE, dt);
            dt.pressed = 1;
        }
        uint64_t inter = ballot(is_agent && !walks && (SCHEME == 3 ? act != 0u : act >= 5u));
        while (inter) {
            const int a = __ffsll((unsigned long long)inter) - 1;
            inter &= inter - 1;
            const uint32_t A = rdl(e.agw, a);
            const int ac = (int)rdl(act, a);
            Me me{(int)(A & 0xFFu), (int)((A >> 8) & 0xFFu), (int)((A >> 16) & 0xFFu), (int)(A >> 24) - 1};
            // the cell in front (scheme3: the bumped cell; scheme1: by orientation, may be off-grid)
            int fx, fy;
            bool ok = true;
            if (SCHEME == 3) {
                const uint32_t lxy = rdl(txy, a);
                fx = (int)(lxy & 0xFFu); fy = (int)(lxy >> 8);
            } else {
                fx = me.x + (int)((DX_TABLE >> (2 * me.o)) & 3u) - 1;
                fy = me.y + (int)((DY_TABLE >> (2 * me.o)) & 3u) - 1;
                ok = (uint32_t)fx < W && (uint32_t)fy < H;              // reference: IndexError; build: no-op
            }
            const int fc = fy * cx.W + fx;
            const uint32_t sv = ok ? cell_at(e, fc) : 0u;               // READY / TOGGLE may have been changed by an earlier agent
            if (ok && holds_objects(sv & CELL_TYPE)) {
                const uint32_t fxy = (uint32_t)fx | ((uint32_t)fy << 8);
                // get_objects_at(location, DynamicObject) cooking_world.py:232-241 as a slot mask
                OM dyn = oballot(e, [=](uint32_t w, uint32_t) { return (w & (D_ALIVE | 0xFFFFu)) == (D_ALIVE | fxy); });
                if (SCHEME == 3) {
                    // action_scheme3.py:37-43: ActionObject with a not-done item -> execute, else primary
                    bool exec = false;
                    if ((sv & CELL_TYPE) >= CUTBOARD)
                        exec = (dyn & oballot(e, [](uint32_t w, uint32_t) { return !(w & D_DONE); })).any();
                    if (exec) execute(e, cx, fx, fy, fc, sv, dyn, dt);
                    else primary(e, cx, me, fx, fy, fc, sv, dyn, dt);
                } else {
                    if (ac == 5) primary(e, cx, me, fx, fy, fc, sv, dyn, dt);
                    else if (ac == 6) pick_up_special(e, cx, me, dyn, dt);
                    else execute(e, cx, fx, fy, fc, sv, dyn, dt);
                }
                e.agw = wrl((A & 0x00FFFFFFu) | ((uint32_t)((me.h + 1) & 0xFF) << 24), a, e.agw);   // only the hands can change
            }
        }
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
            OM content = direct_at(e, xy);
            if (content.any()) {
#pragma unroll
                for (int k = 0; k < OPL; ++k)
                    if (((content.w[k] >> cx.lane) & 1) && !(e.d0[k] & D_DONE)) e.d0[k] |= D_MASHED;
                OM mashed = content & oballot(e, [](uint32_t a, uint32_t) { return (a & D_MASHED) != 0; });
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
            const OM unfree = oballot(e, [](uint32_t a, uint32_t b) { return (a & (D_ALIVE | D_FREE)) == D_ALIVE && (b & 0xFFu) == 0; });
            const OM breads = oballot(e, [](uint32_t a, uint32_t b) {
                return (a & (D_ALIVE | D_CHOPPED | 0xFF0000u)) == (D_ALIVE | D_CHOPPED | (BREAD << 16)) && (b & 0xFFu) == 0;
            });
            if (unfree.any() || breads.count() > 1) {
                OM held = OM::zero();
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int hs = (int)(rdl(e.agw, a) >> 24) - 1;
                    if (hs >= 0) held.set(hs);
                }
                // every object lying directly on a static is the last (only) item of that static's content ...
#pragma unroll
                for (int k = 0; k < OPL; ++k) {
                    bool on_static = (e.d0[k] & D_ALIVE) && (e.d1[k] & 0xFFu) == 0 && !((held.w[k] >> cx.lane) & 1);
                    if (on_static) e.d0[k] |= D_FREE;
                }
                // ... except on a Cutboard that carries a Bread and its clone (the only way a static gets two items,
                // world_objects.py:738-745): both are chopped Breads lying directly on the same cell
                OM twins = oballot(e, [](uint32_t a, uint32_t b) {
                    return (a & (D_ALIVE | D_CHOPPED | 0xFF0000u)) == (D_ALIVE | D_CHOPPED | (BREAD << 16)) && (b & 0xFFu) == 0;
                }).andnot(held);
                if (twins.count() > 1) {
                    OM it = twins;
                    while (it.any()) {
                        const int s = it.first();
                        const uint32_t xy = slot_d0(e, s) & 0xFFFFu;
                        OM content = direct_at(e, xy).andnot(held);
                        it = it.andnot(content);
                        if (content.count() > 1) {
                            const int last = content.last();
#pragma unroll
                            for (int k = 0; k < OPL; ++k)
                                if (((content.w[k] >> cx.lane) & 1) && (cx.lane + 64 * k) != last) e.d0[k] &= ~D_FREE;
                        }
                    }
                }
            }
            // inside a plate: free iff last appended
            OM inside = oballot(e, [](uint32_t a, uint32_t b) { return (a & D_ALIVE) && (b & 0xFFu) != 0; });
            while (inside.any()) {
                const int s = inside.first();
                uint32_t b1 = 0;
#pragma unroll
                for (int k = 0; k < OPL; ++k)
                    if (OPL == 1 || (s >> 6) == k) b1 = rdl(e.d1[k], s & 63);
                const int p = (int)(b1 & 0xFFu) - 1;
                OM content = content_of(e, p);
                inside = inside.andnot(content);
                const int cnt = content.count();
#pragma unroll
                for (int k = 0; k < OPL; ++k)
                    if ((content.w[k] >> cx.lane) & 1) {
                        bool lastone = ((e.d1[k] >> 8) & 0xFFu) == (uint32_t)(cnt - 1);
                        e.d0[k] = lastone ? (e.d0[k] | D_FREE) : (e.d0[k] & ~D_FREE);
                    }
            }
        }
        // Switch.process_linked_objects world_objects.py:165-169 -> Block.switch_state :215-216 (all linked)
        if (dt.pressed) {
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                if ((e.cell[k] & CELL_TYPE) == BLOCK) e.cell[k] ^= CELL_WALK;
        }
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
            auto matches = [=](uint32_t a) { return (a & 0x01FF0000u) == cval && ((acc >> ((a >> 25) & 3u)) & 1u) != 0u; };
            CM here = CM::zero();
            bool any;
            if (children == 0u && !(w & 0x200u)) {
                // leaf of a dynamic class (the common case): no location constraint
                OM m = oballot(e, [=](uint32_t a, uint32_t) { return matches(a); });
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
                        here.w[k] = ballot((e.cell[k] & CELL_TYPE) == cls && (cx.lane + 64 * k) < cx.C) & allow.w[k];
                    any = here.any();
                } else {                                                  // a dynamic class with children
                    OM m;
#pragma unroll
                    for (int k = 0; k < OPL; ++k) {
                        const uint32_t mycell = ((e.d0[k] >> 8) & 0xFFu) * (uint32_t)cx.W + (e.d0[k] & 0xFFu);
                        uint64_t wsel = allow.w[0];
                        if (CPL > 1) {
#pragma unroll
                            for (int q = 1; q < CPL; ++q)
                                if ((mycell >> 6) == (uint32_t)q) wsel = allow.w[q];
                        }
                        m.w[k] = ballot(matches(e.d0[k]) && ((wsel >> (mycell & 63)) & 1));
                    }
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
    static __device__ __forceinline__ uint32_t recipe_marks_wide(const E &e, const Ctx &cx, const uint32_t *__restrict__ row,
                                                              uint64_t *__restrict__ locs) {
        const int n = (int)(rfl(row[0]) & 0xFFu);
        uint32_t marks = 0;
#pragma nounroll
        for (int j = n - 1; j >= 0; --j) {
            const uint32_t w = rfl(row[1 + 2 * j]), children = rfl(row[2 + 2 * j]) & 0xFFFFu;
            if ((marks & children) != children) continue;                 // all(contains.marked)
            const uint32_t cval = (w & 0x00FF0000u) | D_ALIVE, acc = (w >> 24) & 0xFu;
            auto matches = [=](uint32_t a) { return (a & 0x01FF0000u) == cval && ((acc >> ((a >> 25) & 3u)) & 1u) != 0u; };
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
                    here.w[k] = ballot((e.cell[k] & CELL_TYPE) == cls && (cx.lane + 64 * k) < cx.C) & allow.w[k];
            } else if (w & 0x2000u) {                                     // a dynamic class
                OM m;
#pragma unroll
                for (int k = 0; k < OPL; ++k) {
                    const uint32_t mycell = ((e.d0[k] >> 8) & 0xFFu) * (uint32_t)cx.W + (e.d0[k] & 0xFFu);
                    uint64_t wsel = allow.w[0];
#pragma unroll
                    for (int q = 1; q < CPL; ++q)
                        if ((mycell >> 6) == (uint32_t)q) wsel = allow.w[q];
                    m.w[k] = ballot(matches(e.d0[k]) && ((wsel >> (mycell & 63)) & 1));
                }
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
