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
//   * the feature-vector encode (cooking_env.py:352-373) is a per-layout descriptor table walk: lanes
//     stride over the F output doubles, gather the referenced slot/cell from an LDS image of the final
//     state, and store coalesced 8-byte values; x/W, y/H come from an LDS table of correctly rounded
//     quotients (IEEE f64 division, done once per wave).
// Integer / indexing work only; no MFMA.  No CPU fallback exists in this file or its callers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cz {

constexpr int MAX_AGENTS = 4;
constexpr int MAX_NODES = 8;
constexpr int HDR_WORDS = 8;
constexpr int AGENT_WORD0 = HDR_WORDS;
constexpr int CELL_WORD0 = HDR_WORDS + MAX_AGENTS;

enum : uint32_t { FLOOR = 0, COUNTER, DELIVERSQUARE, SWITCH, BLOCK, CUTBOARD, BLENDER };
enum : uint32_t { PLATE = 0, ONION, TOMATO, LETTUCE, CARROT, CUCUMBER, BANANA, APPLE, WATERMELON, BREAD };
enum : uint32_t { CELL_TYPE = 7, CELL_READY = 8, CELL_TOGGLE = 16, CELL_ACTIVE = 32, CELL_WALK = 64 };
// dyn0 = x | y<<8 | cls<<16 | flags<<24
enum : uint32_t { D_ALIVE = 1u << 24, D_CHOPPED = 2u << 24, D_MASHED = 4u << 24, D_FREE = 8u << 24, D_DONE = 6u << 24 };
enum : uint32_t { COND_NONE = 0, COND_CHOPPED, COND_MASHED, COND_NOT_CHOPPED, COND_NOT_MASHED };
enum : uint32_t { W_T = 0, W_MARKS, W_LAYOUT, W_STATUS, W_EPISODE, W_RECIPES, W_POOL, W_RES1 };
enum : uint32_t { ST_DONE = 1, ST_TERM = 2, ST_TRUNC = 4 };
enum : uint32_t {
    OP_ZERO = 0, OP_ONE, OP_CONST_X, OP_CONST_Y, OP_CELL_ACTIVE, OP_CELL_WALK, OP_DYN_X, OP_DYN_Y, OP_DYN_NOTDONE,
    OP_DYN_DONE, OP_DYN_CHOPPED, OP_DYN_MASHED, OP_DYN_ONE, OP_AG_X, OP_AG_Y, OP_AG_O1, OP_AG_O2, OP_AG_O3, OP_AG_O4,
    OP_AG_ONE
};
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
    uint32_t *stat_u;              // [N][SU_WORDS]
    double *stat_f;                // [N][SF_WORDS]
    int64_t env_id_base;
    uint64_t seed;
    double recipe_reward, recipe_penalty, node_reward, time_penalty_step;
    int32_t N, A, W, H, D, F, RW, scheme, max_steps, end_all, R, auto_reset, L;
    int32_t T;                     // fused steps per launch (1 for cz_step)
    uint32_t step0;
    int32_t dyn0_off, dyn1_off;    // word offsets inside a record
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// counter-based action stream (host mirror: cz_action in cz_api.hip, oracle mirror: czo_action)
__host__ __device__ inline uint32_t action_hash(uint64_t seed, int64_t env_global, int agent, uint32_t step, uint32_t n) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)env_global * 4u + (uint64_t)agent + 1u) +
                 0xD1B54A32D192ED03ull * ((uint64_t)step + 1u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(((z >> 32) * (uint64_t)n) >> 32);
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
            if (j == (s >> 6)) w[j] |= 1ull << (s & 63);
    }
    __device__ __forceinline__ void clear(int s) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j == (s >> 6)) w[j] &= ~(1ull << (s & 63));
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
template <int OPL, int CPL>
struct Env {
    uint32_t d0[OPL], d1[OPL];     // per lane: slots lane + 64k
    uint32_t cell[CPL];            // per lane: cells lane + 64k
    int ax[MAX_AGENTS], ay[MAX_AGENTS], ao[MAX_AGENTS], ah[MAX_AGENTS];   // uniform; ah = held slot or -1
    uint32_t t, marks, layout, status, episode, recipes, pool;            // uniform header
};

struct Ctx {                       // wave-uniform geometry + this lane's index
    int A, W, H, D, lane;
};

template <int OPL, int CPL>
struct Ops {
    using E = Env<OPL, CPL>;
    using OM = Mask<OPL>;
    using CM = Mask<CPL>;

    // ---- uniform reads / predicated writes of one slot / one cell ---------------------------------------
    static __device__ __forceinline__ uint32_t slot_d0(const E &e, int s) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < OPL; ++k)
            if ((s >> 6) == k) v = rdl(e.d0[k], s & 63);
        return v;
    }
    static __device__ __forceinline__ uint32_t cell_at(const E &e, int c) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            if ((c >> 6) == k) v = rdl(e.cell[k], c & 63);
        return v;
    }
    static __device__ __forceinline__ void cell_update(E &e, const Ctx &cx, int c, uint32_t clear_bits, uint32_t xor_bits) {
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            if ((c >> 6) == k && cx.lane == (c & 63)) e.cell[k] = (e.cell[k] & ~clear_bits) ^ xor_bits;
    }
    // cooking_world.py:223-227 square_walkable (Floor, Switch: yes; Block: its bit; everything else: no)
    static __device__ __forceinline__ bool walkable(uint32_t cv) {
        uint32_t ty = cv & CELL_TYPE;
        return ty == FLOOR || ty == SWITCH || (ty == BLOCK && (cv & CELL_WALK));
    }

    // ---- ballots over object lanes ----------------------------------------------------------------------
    template <class F>
    static __device__ __forceinline__ OM oballot(const E &e, F pred) {
        OM m;
#pragma unroll
        for (int k = 0; k < OPL; ++k) m.w[k] = ballot(pred(e.d0[k], e.d1[k]));
        return m;
    }
    // get_objects_at(location, DynamicObject) cooking_world.py:232-241 as a slot mask (slot order == list order)
    static __device__ __forceinline__ OM dyn_at(const E &e, uint32_t xy) {
        return oballot(e, [=](uint32_t a, uint32_t) { return (a & D_ALIVE) && (a & 0xFFFFu) == xy; });
    }
    static __device__ __forceinline__ OM content_of(const E &e, int plate) {   // Plate.content membership
        uint32_t tag = (uint32_t)(plate + 1);
        return oballot(e, [=](uint32_t a, uint32_t b) { return (a & D_ALIVE) && (b & 0xFFu) == tag; });
    }
    static __device__ __forceinline__ OM held_mask(const E &e, const Ctx &cx) {
        OM m = OM::zero();
#pragma unroll
        for (int a = 0; a < MAX_AGENTS; ++a)
            if (a < cx.A && e.ah[a] >= 0) m.set(e.ah[a]);
        return m;
    }
    // static.content == objects on the cell that are neither inside a plate nor held (soa.py header)
    static __device__ __forceinline__ OM direct_at(const E &e, const Ctx &cx, uint32_t xy) {
        OM m = oballot(e, [=](uint32_t a, uint32_t b) { return (a & D_ALIVE) && (a & 0xFFFFu) == xy && (b & 0xFFu) == 0; });
        return m.andnot(held_mask(e, cx));
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
    static __device__ __forceinline__ bool agent_at(const E &e, const Ctx &cx, int x, int y) {
        bool r = false;
#pragma unroll
        for (int a = 0; a < MAX_AGENTS; ++a)
            if (a < cx.A && e.ax[a] == x && e.ay[a] == y) r = true;
        return r;
    }
    // cooking_world.py:172-184 get_target_location
    static __device__ __forceinline__ void target(int x, int y, int action, int &tx, int &ty) {
        tx = x + (action == 2) - (action == 1);
        ty = y + (action == 3) - (action == 4);
    }
    static __device__ __forceinline__ bool in_bounds(const Ctx &cx, int x, int y) {
        return x >= 0 && y >= 0 && x < cx.W && y < cx.H;
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

    // cooking_world.py:243-261 attempt_merge (first matching branch only, no fall-through on refusal)
    static __device__ __forceinline__ void attempt_merge(E &e, const Ctx &cx, int i, const OM &dyn, int lx, int ly,
                                                         uint32_t sv) {
        const uint32_t lxy = (uint32_t)lx | ((uint32_t)ly << 8);
        const int held = e.ah[i];
        const uint32_t hw = slot_d0(e, held);
        const uint32_t hcls = (hw >> 16) & 0xFF;
        OM plates = dyn & oballot(e, [](uint32_t a, uint32_t) { return ((a >> 16) & 0xFF) == PLATE; });
        int np = plates.count();
        if (np == 1) {
            int p = plates.first();
            int cnt = content_of(e, p).count();
            // Plate.accepts world_objects.py:408-409: Food, done(), < 64 items
            if (hcls != PLATE && (hw & D_DONE) && cnt < 64) {
                plate_add(e, cx, p, held, cnt);
                move_obj(e, cx, held, lxy);                      // Agent.put_down world_objects.py:789-791
                e.ah[i] = -1;
            }
        } else if (hcls == PLATE && dyn.any()) {
            int o = dyn.last();                                   // pick_index = -1
            uint32_t ow = slot_d0(e, o);
            int cnt = content_of(e, held).count();
            if (((ow >> 16) & 0xFF) != PLATE && (ow & D_DONE) && cnt < 64) {
                plate_add(e, cx, held, o, cnt);
                move_obj(e, cx, o, (uint32_t)e.ax[i] | ((uint32_t)e.ay[i] << 8));
                // static_object.content.remove(o) is implicit: o now carries a container tag
            }
        } else {
            int ncontent = direct_at(e, cx, lxy).count();
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
                if (ty == CUTBOARD || ty == BLENDER) cell_update(e, cx, ly * cx.W + lx, CELL_READY, CELL_READY);
                // add_content: the content list was empty, so the new item is its last element: free=True
                slot_or(e, cx, held, D_FREE);
                move_obj(e, cx, held, lxy);
                e.ah[i] = -1;
            }
        }
    }

    // cooking_world.py:114-136 resolve_primary_interaction
    static __device__ __forceinline__ void primary(E &e, const Ctx &cx, int i) {
        int lx, ly;
        target(e.ax[i], e.ay[i], e.ao[i], lx, ly);
        if (!in_bounds(cx, lx, ly)) return;             // reference: IndexError (scheme1 facing off-grid); build: no-op
        if (agent_at(e, cx, lx, ly)) return;
        const uint32_t lxy = (uint32_t)lx | ((uint32_t)ly << 8);
        OM dyn = dyn_at(e, lxy);
        const int c = ly * cx.W + lx;
        const uint32_t sv = cell_at(e, c);
        if (e.ah[i] < 0) {
            if (!dyn.any()) return;
            OM direct = direct_at(e, cx, lxy);
            int ncontent = direct.count();
            uint32_t ty = sv & CELL_TYPE;
            // static_object.releases() with its side effects: world_objects.py:117-118,275-278,340-346
            bool rel = true;
            if (ty == DELIVERSQUARE) rel = false;
            else if (ty == CUTBOARD) {
                if (ncontent == 1) cell_update(e, cx, c, CELL_READY, 0);
            } else if (ty == BLENDER) {
                rel = !(sv & CELL_TOGGLE);
                if (rel && ncontent - 1 == 0) cell_update(e, cx, c, CELL_READY, 0);
            }
            if (rel) {
                OM fr = dyn & oballot(e, [](uint32_t a, uint32_t) { return (a & D_FREE) != 0; });
                int grab = fr.any() ? fr.first() : dyn.last();
                if (direct.test(grab)) {                        // object_to_grab in static_object.content
                    e.ah[i] = grab;                             // Agent.grab world_objects.py:785-787
                    move_obj(e, cx, grab, (uint32_t)e.ax[i] | ((uint32_t)e.ay[i] << 8));
                }
            }
        } else {
            attempt_merge(e, cx, i, dyn, lx, ly, sv);
        }
    }

    // cooking_world.py:138-154 resolve_interaction_pick_up_special (scheme1)
    static __device__ __forceinline__ void pick_up_special(E &e, const Ctx &cx, int i) {
        int lx, ly;
        target(e.ax[i], e.ay[i], e.ao[i], lx, ly);
        if (!in_bounds(cx, lx, ly)) return;
        if (agent_at(e, cx, lx, ly)) return;
        OM dyn = dyn_at(e, (uint32_t)lx | ((uint32_t)ly << 8));
        if (e.ah[i] >= 0 || !dyn.any()) return;
        OM plates = dyn & oballot(e, [](uint32_t a, uint32_t) { return ((a >> 16) & 0xFF) == PLATE; });
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
        e.ah[i] = s;
        move_obj(e, cx, s, (uint32_t)e.ax[i] | ((uint32_t)e.ay[i] << 8));
    }

    // cooking_world.py:156-170 resolve_execute_action; Cutboard.action world_objects.py:250-269;
    // ChopFood.chop abstract_classes.py:250-254; Bread.chop world_objects.py:738-745; Blender.action :356-360
    static __device__ __forceinline__ void execute(E &e, const Ctx &cx, int i) {
        int lx, ly;
        target(e.ax[i], e.ay[i], e.ao[i], lx, ly);
        if (!in_bounds(cx, lx, ly)) return;
        if (agent_at(e, cx, lx, ly)) return;
        const int c = ly * cx.W + lx;
        const uint32_t sv = cell_at(e, c);
        const uint32_t ty = sv & CELL_TYPE;
        if (ty == CUTBOARD) {
            if (!(sv & CELL_READY)) return;
            const uint32_t lxy = (uint32_t)lx | ((uint32_t)ly << 8);
            OM fresh = direct_at(e, cx, lxy) & oballot(e, [](uint32_t a, uint32_t) { return !(a & D_CHOPPED); });
            int f = fresh.first();                          // first content item whose chop() executes
            if (f < 0) return;                              // reference falls off Cutboard.action (TypeError); unreachable
            uint32_t fw = slot_d0(e, f);
            slot_or(e, cx, f, D_CHOPPED);
            if (((fw >> 16) & 0xFF) == BREAD) {
                // the clone: first not-alive Bread slot (head-room follows the originals), born chopped and free
                OM spare = oballot(e, [](uint32_t a, uint32_t) { return !(a & D_ALIVE) && ((a >> 16) & 0xFF) == BREAD; });
                int n = spare.first();
#pragma unroll
                for (int k = 0; k < OPL; ++k)
                    if (n >= 0 && cx.lane + 64 * k == n) {
                        e.d0[k] = lxy | (BREAD << 16) | D_ALIVE | D_CHOPPED | D_FREE;
                        e.d1[k] = 0;
                    }
            }
            cell_update(e, cx, c, CELL_READY, 0);
        } else if (ty == BLENDER) {
            if (sv & CELL_READY) cell_update(e, cx, c, 0, CELL_TOGGLE);
        }
    }

    // action_scheme3.py:26-34 resolve_walking_action; returns whether the agent "moved" (also true for a == 0 on a
    // walkable cell: move_to(own cell) creates a new tuple, action_scheme3.py:22)
    static __device__ __forceinline__ bool walk(E &e, const Ctx &cx, int i, int action, uint32_t &pressed) {
        int tx, ty;
        target(e.ax[i], e.ay[i], action, tx, ty);
        const int c = ty * cx.W + tx;
        const uint32_t cv = cell_at(e, c);
        if (!walkable(cv)) return false;
        e.ax[i] = tx;
        e.ay[i] = ty;                                                   // Agent.move_to world_objects.py:793-796
        if (e.ah[i] >= 0) move_obj(e, cx, e.ah[i], (uint32_t)tx | ((uint32_t)ty << 8));
        if ((cv & CELL_TYPE) == SWITCH) {                               // Switch.add_content :159-163
            cell_update(e, cx, c, 0, CELL_ACTIVE);
            pressed = 1;
        }
        return true;
    }

    // action_scheme3.py:4-43 / action_scheme1.py:4-40 perform_agent_actions (+ check_inbounds
    // cooking_world.py:192-204, check_collisions :206-221)
    static __device__ __forceinline__ void perform_agent_actions(E &e, const Ctx &cx, const int (&raw)[MAX_AGENTS], int scheme,
                                                                 uint32_t &pressed) {
        int cleaned[MAX_AGENTS], coll[MAX_AGENTS], ex[MAX_AGENTS], ey[MAX_AGENTS];
        bool wk[MAX_AGENTS];
#pragma unroll
        for (int a = 0; a < MAX_AGENTS; ++a) {
            cleaned[a] = 0; coll[a] = 0; ex[a] = -1; ey[a] = -1; wk[a] = false;
            if (a >= cx.A) continue;
            int act = raw[a];
            if (act >= 1 && act <= 4) e.ao[a] = act;                    // change_orientation before any filtering
            // check_inbounds
            if (act != 0 && act != 5) {
                int tx, ty;
                target(e.ax[a], e.ay[a], act, tx, ty);
                if (tx > cx.W - 1 || tx < 0) act = 0;
                if (ty > cx.H - 1 || ty < 0) act = 0;
            }
            cleaned[a] = act;
            int tx, ty;
            target(e.ax[a], e.ay[a], act, tx, ty);
            wk[a] = walkable(cell_at(e, ty * cx.W + tx));
            ex[a] = wk[a] ? tx : e.ax[a];
            ey[a] = wk[a] ? ty : e.ay[a];
        }
#pragma unroll
        for (int a = 0; a < MAX_AGENTS; ++a) {
            if (a >= cx.A) continue;
            bool clash = false;
#pragma unroll
            for (int b = 0; b < MAX_AGENTS; ++b)
                if (b != a && b < cx.A && ex[b] == ex[a] && ey[b] == ey[a]) clash = true;
            coll[a] = (clash && wk[a]) ? 0 : cleaned[a];
        }
#pragma unroll
        for (int a = 0; a < MAX_AGENTS; ++a) {
            if (a >= cx.A) continue;
            const int act = coll[a];
            if (scheme == 3) {
                bool moved = walk(e, cx, a, act, pressed);
                if (!moved && act != 0) {
                    // action_scheme3.py:37-43 resolve_interaction: ActionObject with a not-done item -> execute
                    int tx, ty;
                    target(e.ax[a], e.ay[a], act, tx, ty);
                    uint32_t sv = cell_at(e, ty * cx.W + tx);
                    uint32_t ty_ = sv & CELL_TYPE;
                    OM dyn = dyn_at(e, (uint32_t)tx | ((uint32_t)ty << 8));
                    OM notdone = dyn & oballot(e, [](uint32_t w, uint32_t) { return !(w & D_DONE); });
                    if ((ty_ == CUTBOARD || ty_ == BLENDER) && notdone.any()) execute(e, cx, a);
                    else primary(e, cx, a);
                }
            } else {
                if (act >= 1 && act <= 4) walk(e, cx, a, act, pressed);
                else if (act == 5) primary(e, cx, a);
                else if (act == 6) pick_up_special(e, cx, a);
                else if (act == 7) execute(e, cx, a);
            }
        }
    }

    // cooking_world.py:77-88 progress_world (+ Blender.process world_objects.py:321-335, BlenderFood.blend
    // abstract_classes.py:266-273) and :90-92 resolve_linked_interactions
    static __device__ __forceinline__ void progress_and_link(E &e, const Ctx &cx, uint32_t pressed) {
        // running blenders
        CM running;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            running.w[k] = ballot((e.cell[k] & CELL_TYPE) == BLENDER && (e.cell[k] & CELL_TOGGLE) &&
                                  (cx.lane + 64 * k) < cx.W * cx.H);
        while (running.any()) {
            int c = running.first();
            running.clear(c);
            uint32_t xy = (uint32_t)(c % cx.W) | ((uint32_t)(c / cx.W) << 8);
            OM content = direct_at(e, cx, xy);
            if (content.any()) {
#pragma unroll
                for (int k = 0; k < OPL; ++k)
                    if (((content.w[k] >> cx.lane) & 1) && !(e.d0[k] & D_DONE)) e.d0[k] |= D_MASHED;
                OM mashed = content & oballot(e, [](uint32_t a, uint32_t) { return (a & D_MASHED) != 0; });
                if (mashed.count() == content.count()) cell_update(e, cx, c, CELL_READY | CELL_TOGGLE, 0);
            }
        }
        // free-flag normalisation: every container with content: all False, last True.
        //  * on a static: the item is alone (free) except on a Cutboard carrying Bread + clone
        //  * inside a plate: free iff it is the last appended (seq == count-1)
        //  * held directly by an agent: untouched
        OM held = held_mask(e, cx);
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            bool on_static = (e.d0[k] & D_ALIVE) && (e.d1[k] & 0xFFu) == 0 && !((held.w[k] >> cx.lane) & 1);
            if (on_static) e.d0[k] |= D_FREE;
        }
        CM boards;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            boards.w[k] = ballot((e.cell[k] & CELL_TYPE) == CUTBOARD && (cx.lane + 64 * k) < cx.W * cx.H);
        while (boards.any()) {
            int c = boards.first();
            boards.clear(c);
            uint32_t xy = (uint32_t)(c % cx.W) | ((uint32_t)(c / cx.W) << 8);
            OM content = direct_at(e, cx, xy);
            if (content.count() > 1) {
                int last = content.last();
#pragma unroll
                for (int k = 0; k < OPL; ++k)
                    if (((content.w[k] >> cx.lane) & 1) && (cx.lane + 64 * k) != last) e.d0[k] &= ~D_FREE;
            }
        }
        OM plates = oballot(e, [](uint32_t a, uint32_t) { return (a & D_ALIVE) && ((a >> 16) & 0xFF) == PLATE; });
        while (plates.any()) {
            int p = plates.first();
            plates.clear(p);
            OM content = content_of(e, p);
            int cnt = content.count();
            if (cnt == 0) continue;
#pragma unroll
            for (int k = 0; k < OPL; ++k)
                if ((content.w[k] >> cx.lane) & 1) {
                    bool lastone = ((e.d1[k] >> 8) & 0xFFu) == (uint32_t)(cnt - 1);
                    e.d0[k] = lastone ? (e.d0[k] | D_FREE) : (e.d0[k] & ~D_FREE);
                }
        }
        // Switch.process_linked_objects world_objects.py:165-169 -> Block.switch_state :215-216 (all linked)
        if (pressed) {
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                if ((e.cell[k] & CELL_TYPE) == BLOCK) e.cell[k] ^= CELL_WALK;
        }
    }

    // recipe.py:77-104 update_recipe_state for one recipe graph; returns the marks byte (bit j = node j marked).
    // A node's matched set is kept as a bit set of CELLS, because the only thing a parent asks of a child's
    // matches is location equality (recipe.py:103).
    static __device__ __forceinline__ uint32_t recipe_marks(const E &e, const Ctx &cx, const uint32_t *__restrict__ rp) {
        const int n = (int)rfl(rp[0]);
        CM loc[MAX_NODES];
        uint32_t marks = 0;
        uint32_t mycell[OPL];
#pragma unroll
        for (int k = 0; k < OPL; ++k) mycell[k] = ((e.d0[k] >> 8) & 0xFFu) * (uint32_t)cx.W + (e.d0[k] & 0xFFu);
#pragma unroll
        for (int j = MAX_NODES - 1; j >= 0; --j) {
            loc[j] = CM::zero();
            if (j >= n) continue;
            const uint32_t nd = rfl(rp[1 + j]);
            const uint32_t cls = nd & 0xFF, cond = (nd >> 8) & 0xFF, children = (nd >> 16) & 0xFF;
            if ((marks & children) != children) continue;                 // all(contains.marked)
            if (cls < 16) {                                               // a static class: candidates are cells
                CM m;
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    m.w[k] = ballot((e.cell[k] & CELL_TYPE) == cls && (cx.lane + 64 * k) < cx.W * cx.H);
#pragma unroll
                for (int c2 = 0; c2 < MAX_NODES; ++c2)
                    if (c2 > j && (children >> c2) & 1) m = m & loc[c2];
                loc[j] = m;
                if (m.any()) marks |= 1u << j;
            } else if (cls < 32) {                                        // a dynamic class: candidates are slots
                const uint32_t dc = cls - 16;
                OM m;
#pragma unroll
                for (int k = 0; k < OPL; ++k) {
                    uint32_t a = e.d0[k];
                    bool ok = (a & D_ALIVE) && ((a >> 16) & 0xFF) == dc;
                    if (cond == COND_CHOPPED) ok = ok && (a & D_CHOPPED);
                    else if (cond == COND_MASHED) ok = ok && (a & D_MASHED);
                    else if (cond == COND_NOT_CHOPPED) ok = ok && !(a & D_CHOPPED);
                    else if (cond == COND_NOT_MASHED) ok = ok && !(a & D_MASHED);
#pragma unroll
                    for (int c2 = 0; c2 < MAX_NODES; ++c2)
                        if (c2 > j && (children >> c2) & 1) {
                            uint64_t wsel = loc[c2].word((int)0);
                            if (CPL > 1) {
                                wsel = 0;
#pragma unroll
                                for (int q = 0; q < CPL; ++q)
                                    if ((mycell[k] >> 6) == (uint32_t)q) wsel = loc[c2].w[q];
                            }
                            ok = ok && ((wsel >> (mycell[k] & 63)) & 1);
                        }
                    m.w[k] = ballot(ok);
                }
                if (m.any()) {
                    marks |= 1u << j;
                    // scatter matched objects to their cells (a handful of objects at most)
                    OM it = m;
                    while (it.any()) {
                        int s = it.first();
                        it.clear(s);
                        uint32_t w = slot_d0(e, s);
                        loc[j].set((int)(((w >> 8) & 0xFFu) * (uint32_t)cx.W + (w & 0xFFu)));
                    }
                }
            }
        }
        return marks;
    }
};

}  // namespace cz
