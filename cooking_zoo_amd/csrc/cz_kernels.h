// cz_kernels.h -- kernel templates of the step path (included by the per-size instantiation units).
#pragma once
#include "cz_device.h"

namespace cz {

// Diagnostic build only (make prof -> libcookingzoo_hip_prof.so, -DCZ_PROFILE): s_memtime stamps at phase boundaries,
// written to a buffer of their own (Params::stamps); the shipped library contains none of this.
#ifdef CZ_PROFILE
#define CZ_STAMP(i)                                                                                    \
    do {                                                                                               \
        unsigned long long _t;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                     \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (lane == 0 && P.stamps) P.stamps[(size_t)env * 8 + (i)] = _t;                               \
    } while (0)
#else
#define CZ_STAMP(i) do { } while (0)
#endif

// LDS image of one env (halfwords, cooking_zoo_amd/soa.py IMG_*): every halfword is a BYTE offset into `lut`
constexpr int IMG_OBJ0 = 0, IMG_CELL0 = 768, IMG_AG0 = 1792, IMG_ZERO = 1824, IMG_HALFWORDS = 1832;
constexpr int LUT_ABSENT = 255, LUT_SIZE = 256;
constexpr int OBS_CHUNK = 8;       // descriptor words prefetched per lane (8 x 64 = 512 features per chunk)

struct Lds {
    double lut[LUT_SIZE];          // [0..2W-2] (i-(W-1))/W | [64..64+2H-2] (i-(H-1))/H | [126] 0.0 | [127] 1.0 | [128..255] 0.0
    uint16_t img[IMG_HALFWORDS];   // objects 6 hw each | cells 4 hw each | agents 8 hw each | the "absent" halfword
    int32_t sub[MAX_AGENTS][16];   // per observer: what to subtract (x8) for each axis code
    uint64_t locs[MAX_NODES * 4];  // recipe evaluation scratch: matched-location bit sets per node (CPL <= 4 words)
};

// Loads are clamped instead of exec-masked (no branches): every address stays inside the record.
template <int OPL, int CPL, int NA>
__device__ __forceinline__ void load_env(const Params &P, Env<OPL, CPL, NA> &e, const Ctx &cx, const uint32_t *__restrict__ rec) {
    const uint32_t h = rec[cx.lane & 15];                                  // header + agents (record is >= 32 words)
    const uint8_t *cb = reinterpret_cast<const uint8_t *>(rec + CELL_WORD0);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = cx.lane + 64 * k;
        uint32_t v = cb[min(c, cx.C - 1)];
        e.cell[k] = (c < cx.C) ? v : 0u;
    }
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
        int s = cx.lane + 64 * k, sc = min(s, cx.D - 1);
        uint32_t a = rec[P.dyn0_off + sc], b = rec[P.dyn1_off + sc];
        e.d0[k] = (s < cx.D) ? a : 0u;
        e.d1[k] = (s < cx.D) ? b : 0u;
    }
    e.t = rdl(h, W_T); e.marks = rdl(h, W_MARKS); e.layout = rdl(h, W_LAYOUT); e.status = rdl(h, W_STATUS);
    e.episode = rdl(h, W_EPISODE); e.recipes = rdl(h, W_RECIPES); e.pool = rdl(h, W_POOL);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        uint32_t w = rdl(h, AGENT_WORD0 + a);
        e.ax[a] = (int)(w & 0xFF); e.ay[a] = (int)((w >> 8) & 0xFF); e.ao[a] = (int)((w >> 16) & 0xFF);
        e.ah[a] = (int)(w >> 24) - 1;
    }
}

// header + agents in one 16-lane store (v_writelane assembles the words), cells / objects only when they changed
template <int OPL, int CPL, int NA>
__device__ __forceinline__ void store_env(const Params &P, const Env<OPL, CPL, NA> &e, const Ctx &cx, uint32_t *__restrict__ rec,
                                          bool cells_dirty, bool objs_dirty) {
    uint32_t h = 0;
    h = wrl(e.t, W_T, h);
    h = wrl(e.marks, W_MARKS, h);
    h = wrl(e.layout, W_LAYOUT, h);
    h = wrl(e.status, W_STATUS, h);
    h = wrl(e.episode, W_EPISODE, h);
    h = wrl(e.recipes, W_RECIPES, h);
    h = wrl(e.pool, W_POOL, h);
#pragma unroll
    for (int a = 0; a < NA; ++a)
        h = wrl((uint32_t)e.ax[a] | ((uint32_t)e.ay[a] << 8) | ((uint32_t)e.ao[a] << 16) |
                                           ((uint32_t)((e.ah[a] + 1) & 0xFF) << 24), AGENT_WORD0 + a, h);
    if (cx.lane < RET_WORD0) rec[cx.lane] = h;
    if (cells_dirty) {
        uint8_t *cb = reinterpret_cast<uint8_t *>(rec + CELL_WORD0);
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            int c = cx.lane + 64 * k;
            if (c < cx.C) cb[c] = (uint8_t)e.cell[k];
        }
    }
    if (objs_dirty) {
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            int s = cx.lane + 64 * k;
            if (s < cx.D) {
                rec[P.dyn0_off + s] = e.d0[k];
                rec[P.dyn1_off + s] = e.d1[k];
            }
        }
    }
}

// the recipe rows of this env, one word per lane: lane 9r + i = word i of the row of recipe r
__device__ __forceinline__ uint32_t load_recipe_rows(const Params &P, uint32_t recipes, int lane) {
    const int r = lane / 9, i = lane - 9 * r;
    const uint32_t id = (recipes >> (8 * (r & 3))) & 0xFFu;
    return (r < P.R) ? P.recipes[(size_t)id * (1 + MAX_NODES) + i] : 0u;
}

template <int OPL, int CPL, int NA>
__device__ __forceinline__ uint32_t all_marks(const Params &P, const Env<OPL, CPL, NA> &e, const Ctx &cx, uint32_t rowv, Lds &s) {
    uint32_t marks = 0;
#pragma nounroll
    for (int r = 0; r < P.R; ++r) marks |= Ops<OPL, CPL, NA, 3>::recipe_marks(e, cx, rowv, 9 * r, s.locs) << (8 * r);
    return marks;
}

// once per kernel: the quotient table and the constant part of the image (cell coordinates)
template <int CPL>
__device__ __forceinline__ void init_lds(const Params &P, const Ctx &cx, Lds &s) {
    const int lane = cx.lane;
    s.lut[lane] = P.lut[lane];
    s.lut[64 + lane] = P.lut[64 + lane];
    s.lut[128 + lane] = 0.0;
    s.lut[192 + lane] = 0.0;
    if (lane == 0) s.img[IMG_ZERO] = (uint16_t)(LUT_ABSENT * 8);
    uint32_t *img32 = reinterpret_cast<uint32_t *>(s.img);
    const uint32_t c01 = (uint32_t)((P.W - 1) * 8) | ((uint32_t)((LUT_Y0 + P.H - 1) * 8) << 16);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        uint32_t c = (uint32_t)(lane + 64 * k);
        uint32_t y = (c * P.inv_w) >> 16;                    // exact for c < 1024 (checked on the host)
        uint32_t x = c - y * (uint32_t)P.W;
        img32[(IMG_CELL0 >> 1) + 2 * c] = ((x << 3) | (y << 19)) + c01;
    }
}

// the descriptor words of this lane for features [chunk*512, chunk*512 + 512)
__device__ __forceinline__ void load_desc(const Params &P, uint32_t layout, int chunk, int lane, uint32_t (&dsc)[OBS_CHUNK]) {
    const uint32_t *__restrict__ desc = P.lay_desc + (size_t)layout * P.F;
#pragma unroll
    for (int i = 0; i < OBS_CHUNK; ++i) dsc[i] = desc[min(chunk * 64 * OBS_CHUNK + 64 * i + lane, P.F - 1)];
}

// cooking_env.py:352-373 get_feature_vector for every agent of the env: out[a][f] = lut[img[desc.hw] - sub[a][desc.code]]
template <int OPL, int CPL, int NA>
__device__ __forceinline__ void observe(const Params &P, const Env<OPL, CPL, NA> &e, const Ctx &cx, Lds &s,
                                        uint32_t (&dsc)[OBS_CHUNK], double *__restrict__ out /* [A][F] of this env */) {
    uint32_t *img32 = reinterpret_cast<uint32_t *>(s.img);
    const uint32_t dead = (uint32_t)(LUT_ABSENT * 8) * 0x10001u;
    const uint32_t c01 = (uint32_t)((P.W - 1) * 8) | ((uint32_t)((LUT_Y0 + P.H - 1) * 8) << 16);
    const uint32_t f0 = (uint32_t)(LUT_ZERO * 8) * 0x10001u;        // two "0.0" flags
    // ---- objects: 3 dwords per slot
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
        const uint32_t w = e.d0[k];
        const bool alive = (w & D_ALIVE) != 0u;
        const uint32_t ch = (w >> 25) & 1u, ma = (w >> 26) & 1u;
        uint32_t q0 = (((w & 0xFFu) << 3) | ((w & 0xFF00u) << 11)) + c01;
        uint32_t q1 = f0 + (((ch | ma) ^ 1u) << 3) + (ch << 19);
        uint32_t q2 = ((uint32_t)(LUT_ZERO * 8) + (ma << 3)) | ((uint32_t)(LUT_ONE * 8) << 16);
        q0 = alive ? q0 : dead; q1 = alive ? q1 : dead; q2 = alive ? q2 : dead;
        const int slot = cx.lane + 64 * k;
        img32[(IMG_OBJ0 >> 1) + 3 * slot] = q0;
        img32[(IMG_OBJ0 >> 1) + 3 * slot + 1] = q1;
        img32[(IMG_OBJ0 >> 1) + 3 * slot + 2] = q2;
    }
    // ---- cells: the mutable flag (switch_active / block walkable) + the constant 1
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const uint32_t fa = ((e.cell[k] >> 5) | (e.cell[k] >> 6)) & 1u;
        img32[(IMG_CELL0 >> 1) + 2 * (cx.lane + 64 * k) + 1] = ((uint32_t)(LUT_ZERO * 8) + (fa << 3)) | ((uint32_t)(LUT_ONE * 8) << 16);
    }
    // ---- agents: 4 dwords each, and the per-observer subtrahend table
    {
        uint32_t aw = 0;
        int subv = 0;
        const uint32_t code = (uint32_t)cx.lane & 15u;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const uint32_t t = 1u << e.ao[a];
            aw = wrl((((uint32_t)e.ax[a] << 3) | ((uint32_t)e.ay[a] << 19)) + c01, 4 * a, aw);
            aw = wrl(f0 + (((t >> 1) & 1u) << 3) + (((t >> 2) & 1u) << 19), 4 * a + 1, aw);
            aw = wrl(f0 + (((t >> 3) & 1u) << 3) + (((t >> 4) & 1u) << 19), 4 * a + 2, aw);
            aw = wrl((uint32_t)(LUT_ONE * 8), 4 * a + 3, aw);
            // axis codes: 1 -> x, 2 -> y, 4+2j -> x unless j == a, 5+2j -> y unless j == a
            const uint32_t mx = 0x552u & ~(1u << (4 + 2 * a)), my = 0xAA4u & ~(1u << (5 + 2 * a));
            const int v = (((mx >> code) & 1u) ? (e.ax[a] << 3) : 0) + (((my >> code) & 1u) ? (e.ay[a] << 3) : 0);
            if ((cx.lane >> 4) == a) subv = v;
        }
        if (cx.lane < 4 * NA) img32[(IMG_AG0 >> 1) + cx.lane] = aw;
        if (cx.lane < 16 * NA) (&s.sub[0][0])[cx.lane] = subv;
    }
    __builtin_amdgcn_wave_barrier();             // one wave owns this LDS region: DS ops of a wave execute in order
    const char *lutb = reinterpret_cast<const char *>(s.lut);
    const char *imgb = reinterpret_cast<const char *>(s.img);
    const char *subb = reinterpret_cast<const char *>(s.sub);
    for (int chunk = 0; chunk * 64 * OBS_CHUNK < P.F; ++chunk) {
        if (chunk > 0) load_desc(P, e.layout, chunk, cx.lane, dsc);
#pragma unroll
        for (int i = 0; i < OBS_CHUNK; ++i) {
            const int fbase = chunk * 64 * OBS_CHUNK + 64 * i;
            if (fbase >= P.F) break;
            const int f = fbase + cx.lane;
            const uint32_t d = dsc[i];
            const uint32_t base = *reinterpret_cast<const uint16_t *>(imgb + (d & 0xFFFFu));
            const bool live = f < P.F;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int sub = *reinterpret_cast<const int32_t *>(subb + 64 * a + (d >> 16));
                const double v = *reinterpret_cast<const double *>(lutb + ((int)base - sub));
                if (live) out[(size_t)a * P.F + f] = v;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

struct StepOut {
    double rew[MAX_AGENTS];
    uint32_t term, trunc;
    bool stepped, finished;        // a world step was executed / it ended the episode
};

// One accumulated_step (cooking_env.py:243-269) of one env held in registers.
template <int OPL, int CPL, int NA, int SCHEME>
__device__ __forceinline__ void step_env(const Params &P, Env<OPL, CPL, NA> &e, const Ctx &cx, const int (&acts)[NA],
                                         int64_t env_global, uint32_t &rowv, Lds &lds, uint32_t (&dsc)[OBS_CHUNK], Dirty &dt, StepOut &o) {
    using O = Ops<OPL, CPL, NA, SCHEME>;
#pragma unroll
    for (int a = 0; a < MAX_AGENTS; ++a) o.rew[a] = 0.0;
    o.term = 0; o.trunc = 0; o.stepped = false; o.finished = false;
    if (e.status & ST_DONE) {
        if (P.auto_reset) {
            // next-step autoreset: reset() of cooking_env.py:178-210 from the layout pool
            e.episode += 1;
            uint32_t lay = next_layout(env_global, e.episode, e.pool, (uint32_t)P.L);
            uint32_t recipes = e.recipes, episode = e.episode, pool = e.pool;
            load_env(P, e, cx, P.lay_init + (size_t)lay * P.RW);
            e.t = 0; e.layout = lay; e.status = 0; e.episode = episode; e.recipes = recipes; e.pool = pool;
            e.marks = all_marks(P, e, cx, rowv, lds);
            if (P.obs) load_desc(P, lay, 0, cx.lane, dsc);
            dt.cells = 1; dt.touched = 1; dt.interacted = 1;         // everything must be written back
        } else {
            o.term = (e.status & ST_TERM) ? 1u : 0u;
            o.trunc = (e.status & ST_TRUNC) ? 1u : 0u;
        }
        return;
    }
    o.stepped = true;
    e.t += 1;                                                        // cooking_env.py:244
#ifdef CZ_PROFILE
    const int lane = cx.lane; const long long env = env_global - P.env_id_base;
#endif
    O::perform_agent_actions(e, cx, acts, dt);                      // cooking_world.py:104-108
    CZ_STAMP(2);
    O::progress_and_link(e, cx, dt);
    CZ_STAMP(3);                                // :109-110 (handle_agent_spawn: neutral at rate 0)
    // compute_rewards cooking_env.py:290-315
    const bool truncated = (int)e.t >= P.max_steps;                 // compute_truncated :333-350
    const uint32_t before = e.marks;
    uint32_t after = before;
#pragma unroll
    for (int a = 0; a < NA; ++a) o.rew[a] = P.reward_idle;
    if (dt.touched | (dt.moved & (uint32_t)P.walk_touches)) {
        after = 0;
#pragma nounroll
        for (int r = 0; r < P.R; ++r) {
            const uint32_t ma = O::recipe_marks(e, cx, rowv, 9 * r, lds.locs);
            const uint32_t mb = (before >> (8 * r)) & 0xFF;
            after |= ma << (8 * r);
            if (ma != mb) {
                // goals_completed sums (recipe.py:36-40): open goal slots before / after
                uint32_t countmask = 0;
                const int n = (int)rdl(rowv, 9 * r);
#pragma unroll
                for (int j = 0; j < MAX_NODES; ++j)
                    if (j < n && ((rdl(rowv, 9 * r + 1 + j) >> 24) & 1)) countmask |= 1u << j;
                const int goals_before = __popc(~mb & countmask), goals_after = __popc(~ma & countmask);
                const bool completed = ma & 1, completion_before = mb & 1;
                const bool malus = !completed && completion_before, bonus = completed && !completion_before;
                double x = 0.0;
                x += (double)(goals_before - goals_after) * P.node_reward;
                x += (bonus ? 1.0 : 0.0) * P.recipe_reward;
                x += (malus ? 1.0 : 0.0) * P.recipe_penalty;
                x += P.time_penalty_step;
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    if (r == a) o.rew[a] = x;
            }
        }
    }
    e.marks = after;
    // recipe roots are bit 0 of each marks byte
    const uint32_t roots = after & 0x01010101u & (P.R >= 4 ? 0xFFFFFFFFu : ((1u << (8 * P.R)) - 1u));
    const bool done = P.end_all ? (__popc(roots) == P.R) : (roots != 0u);
    o.term = done ? 1u : 0u;
    o.trunc = truncated ? 1u : 0u;
    if (done || truncated) {
        e.status |= ST_DONE | (done ? ST_TERM : 0u) | (truncated ? ST_TRUNC : 0u);
        o.finished = true;
    }
}

// Four envs per 256-thread workgroup (one wavefront each, no cross-wave communication).
// FUSED = false: one step, actions from memory.  FUSED = true: P.T steps, on-device action stream, outputs [t][env].
template <int OPL, int CPL, int NA, int SCHEME, bool FUSED>
__global__ __launch_bounds__(256) void k_step(const Params P) {
    __shared__ Lds lds_all[4];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)rfl(threadIdx.x >> 6);
    const int env = (int)blockIdx.x * 4 + wave;
    if (env >= P.N) return;
    Lds &lds = lds_all[wave];
    Ctx cx{P.W, P.H, P.D, P.W * P.H, lane};
    CZ_STAMP(0);
    uint32_t *rec = P.state + (size_t)env * P.RW;
    double *retp = reinterpret_cast<double *>(rec + RET_WORD0);
    // ---- every load of the step is issued here, before anything waits
    int av = 0;
    if (!FUSED) av = P.actions[(size_t)env * NA + min(lane, NA - 1)];
    double ret = retp[lane & 3];                                           // running episode return, lane a = agent a
    Env<OPL, CPL, NA> e;
    load_env(P, e, cx, rec);
    init_lds<CPL>(P, cx, lds);
    uint32_t rowv = load_recipe_rows(P, e.recipes, lane);
    uint32_t dsc[OBS_CHUNK];
    if (P.obs) load_desc(P, e.layout, 0, lane, dsc);
    const int64_t env_global = P.env_id_base + env;
    bool cells_dirty = false, objs_dirty = false;
    CZ_STAMP(1);

    const int T = FUSED ? P.T : 1;
#pragma nounroll
    for (int t = 0; t < T; ++t) {
        int acts[NA];
        if (!FUSED) {
#pragma unroll
            for (int a = 0; a < NA; ++a) acts[a] = (int)rdl((uint32_t)av, a) & 7;
        } else {
            const uint32_t nact = SCHEME == 3 ? 5u : 8u;
#pragma unroll
            for (int a = 0; a < NA; ++a) acts[a] = (int)action_hash(P.seed, env_global, a, P.step0 + (uint32_t)t, nact);
        }
        Dirty dt{0, 0, 0, 0, 0};
        StepOut o;
        step_env<OPL, CPL, NA, SCHEME>(P, e, cx, acts, env_global, rowv, lds, dsc, dt, o);
        CZ_STAMP(4);
        cells_dirty |= dt.cells != 0;
        objs_dirty |= (dt.touched | dt.interacted | dt.moved) != 0;
        // ---- running return (lane a = agent a) and, at episode end only, the per-env statistics
        double myrew = 0.0;
#pragma unroll
        for (int a = 0; a < NA; ++a)
            if (lane == a) myrew = o.rew[a];
        if (o.stepped) ret += myrew;
        if (o.finished) {
            uint32_t *su = P.stat_u + (size_t)env * SU_WORDS;
            double *sf = P.stat_f + (size_t)env * SF_WORDS;
            if (lane == 0) {
                su[SU_EPISODES] += 1; su[SU_LENSUM] += e.t; su[SU_TRUNC] += o.trunc; su[SU_TERM] += o.term;
            }
            if (lane < NA) {
                su[SU_COMPLETED0 + lane] += (e.marks >> (8 * lane)) & 1u;
                sf[SF_SUM0 + lane] += ret;
            }
            ret = 0.0;
        }
        // ---- outputs of this step
        const size_t row = FUSED ? ((size_t)t * P.N + env) : (size_t)env;
        if (lane < NA) {
            if (P.rewards) P.rewards[row * NA + lane] = myrew;
            if (P.term) P.term[row * NA + lane] = (uint8_t)o.term;
            if (P.trunc) P.trunc[row * NA + lane] = (uint8_t)o.trunc;
        }
        CZ_STAMP(5);
        if (P.obs) observe(P, e, cx, lds, dsc, P.obs + row * (size_t)NA * P.F);
        CZ_STAMP(6);
    }
    store_env(P, e, cx, rec, cells_dirty, objs_dirty);
    if (lane < NA) retp[lane] = ret;
    CZ_STAMP(7);
}

// reset(): cooking_env.py:178-210 for envs [env_begin, env_begin + count)
template <int OPL, int CPL, int NA>
__global__ __launch_bounds__(64) void k_reset(const Params P, int64_t env_begin, const int32_t *__restrict__ layout_ids,
                                              const uint32_t *__restrict__ recipe_words, const uint32_t *__restrict__ pool_words,
                                              double *obs_out) {
    __shared__ Lds lds;
    const int i = blockIdx.x;
    const int64_t env = env_begin + i;
    const int lane = (int)threadIdx.x;
    Ctx cx{P.W, P.H, P.D, P.W * P.H, lane};
    init_lds<CPL>(P, cx, lds);
    Env<OPL, CPL, NA> e;
    uint32_t *rec = P.state + (size_t)env * P.RW;
    const uint32_t lay = rfl((uint32_t)layout_ids[i]);
    const uint32_t old_episode = rfl(rec[W_EPISODE]);
    load_env(P, e, cx, P.lay_init + (size_t)lay * P.RW);
    e.t = 0; e.layout = lay; e.status = 0; e.episode = old_episode; e.recipes = rfl(recipe_words[i]);
    e.pool = rfl(pool_words[i]);
    uint32_t rowv = load_recipe_rows(P, e.recipes, lane);
    e.marks = all_marks(P, e, cx, rowv, lds);
    store_env(P, e, cx, rec, true, true);
    if (lane < MAX_AGENTS) reinterpret_cast<double *>(rec + RET_WORD0)[lane] = 0.0;
    if (obs_out) {
        uint32_t dsc[OBS_CHUNK];
        load_desc(P, e.layout, 0, lane, dsc);
        observe(P, e, cx, lds, dsc, obs_out + (size_t)i * NA * P.F);
    }
}

// observe() only (after cz_set_state)
template <int OPL, int CPL, int NA>
__global__ __launch_bounds__(64) void k_observe(const Params P, int64_t env_begin, double *obs_out) {
    __shared__ Lds lds;
    const int i = blockIdx.x;
    const int lane = (int)threadIdx.x;
    Ctx cx{P.W, P.H, P.D, P.W * P.H, lane};
    init_lds<CPL>(P, cx, lds);
    Env<OPL, CPL, NA> e;
    load_env(P, e, cx, P.state + (size_t)(env_begin + i) * P.RW);
    uint32_t dsc[OBS_CHUNK];
    load_desc(P, e.layout, 0, lane, dsc);
    observe(P, e, cx, lds, dsc, obs_out + (size_t)i * NA * P.F);
}

// launchers exported by each instantiation unit
struct Launchers {
    hipError_t (*step)(const Params &, hipStream_t);
    hipError_t (*reset)(const Params &, hipStream_t, int64_t, int, const int32_t *, const uint32_t *, const uint32_t *, double *);
    hipError_t (*observe)(const Params &, hipStream_t, int64_t, int, double *);
};

template <int OPL, int CPL>
struct Inst {
    template <int NA>
    static hipError_t step_na(const Params &P, hipStream_t st) {
        const dim3 grid((unsigned)((P.N + 3) / 4)), block(256);
        if (P.actions) {
            if (P.scheme == 3) hipLaunchKernelGGL((k_step<OPL, CPL, NA, 3, false>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((k_step<OPL, CPL, NA, 1, false>), grid, block, 0, st, P);
        } else {
            if (P.scheme == 3) hipLaunchKernelGGL((k_step<OPL, CPL, NA, 3, true>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((k_step<OPL, CPL, NA, 1, true>), grid, block, 0, st, P);
        }
        return hipGetLastError();
    }
    static hipError_t step(const Params &P, hipStream_t st) {
        switch (P.A) {
        case 1: return step_na<1>(P, st);
        case 2: return step_na<2>(P, st);
        case 3: return step_na<3>(P, st);
        default: return step_na<4>(P, st);
        }
    }
    static hipError_t reset(const Params &P, hipStream_t st, int64_t b, int n, const int32_t *lay, const uint32_t *rec,
                            const uint32_t *pool, double *obs) {
        switch (P.A) {
        case 1: hipLaunchKernelGGL((k_reset<OPL, CPL, 1>), dim3(n), dim3(64), 0, st, P, b, lay, rec, pool, obs); break;
        case 2: hipLaunchKernelGGL((k_reset<OPL, CPL, 2>), dim3(n), dim3(64), 0, st, P, b, lay, rec, pool, obs); break;
        case 3: hipLaunchKernelGGL((k_reset<OPL, CPL, 3>), dim3(n), dim3(64), 0, st, P, b, lay, rec, pool, obs); break;
        default: hipLaunchKernelGGL((k_reset<OPL, CPL, 4>), dim3(n), dim3(64), 0, st, P, b, lay, rec, pool, obs); break;
        }
        return hipGetLastError();
    }
    static hipError_t observe(const Params &P, hipStream_t st, int64_t b, int n, double *obs) {
        switch (P.A) {
        case 1: hipLaunchKernelGGL((k_observe<OPL, CPL, 1>), dim3(n), dim3(64), 0, st, P, b, obs); break;
        case 2: hipLaunchKernelGGL((k_observe<OPL, CPL, 2>), dim3(n), dim3(64), 0, st, P, b, obs); break;
        case 3: hipLaunchKernelGGL((k_observe<OPL, CPL, 3>), dim3(n), dim3(64), 0, st, P, b, obs); break;
        default: hipLaunchKernelGGL((k_observe<OPL, CPL, 4>), dim3(n), dim3(64), 0, st, P, b, obs); break;
        }
        return hipGetLastError();
    }
};

Launchers launchers_small();   // D <= 64 slots, W*H <= 64 cells
Launchers launchers_large();   // D <= 128 slots, W*H <= 256 cells

}  // namespace cz
