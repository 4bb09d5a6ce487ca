// cz_api.hip -- kernels + C-ABI (include/cookingzoo.h) of the MI355X-native CookingZoo step path.
// gfx950 only.  One wavefront (64 lanes, one 64-thread workgroup) per env instance; see cz_device.h.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/cookingzoo.h"
#include "cz_device.h"

using namespace cz;

// ======================================================================================================
// device: load / store / reset / observe / one full step
// ======================================================================================================

template <int OPL, int CPL>
__device__ __forceinline__ void load_env(const Params &P, Env<OPL, CPL> &e, const Ctx &cx, const uint32_t *__restrict__ rec) {
    uint32_t h = (cx.lane < CELL_WORD0) ? rec[cx.lane] : 0u;
    e.t = rdl(h, W_T); e.marks = rdl(h, W_MARKS); e.layout = rdl(h, W_LAYOUT); e.status = rdl(h, W_STATUS);
    e.episode = rdl(h, W_EPISODE); e.recipes = rdl(h, W_RECIPES); e.pool = rdl(h, W_POOL);
#pragma unroll
    for (int a = 0; a < MAX_AGENTS; ++a) {
        uint32_t w = rdl(h, AGENT_WORD0 + a);
        e.ax[a] = (int)(w & 0xFF); e.ay[a] = (int)((w >> 8) & 0xFF); e.ao[a] = (int)((w >> 16) & 0xFF);
        e.ah[a] = (int)((w >> 24) & 0xFF) - 1;
    }
    const uint8_t *cb = reinterpret_cast<const uint8_t *>(rec + CELL_WORD0);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = cx.lane + 64 * k;
        e.cell[k] = (c < cx.W * cx.H) ? (uint32_t)cb[c] : 0u;
    }
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
        int s = cx.lane + 64 * k;
        e.d0[k] = (s < cx.D) ? rec[P.dyn0_off + s] : 0u;
        e.d1[k] = (s < cx.D) ? rec[P.dyn1_off + s] : 0u;
    }
}

template <int OPL, int CPL>
__device__ __forceinline__ void store_env(const Params &P, const Env<OPL, CPL> &e, const Ctx &cx, uint32_t *__restrict__ rec) {
    uint32_t h = 0;
    const int l = cx.lane;
    if (l == W_T) h = e.t;
    if (l == W_MARKS) h = e.marks;
    if (l == W_LAYOUT) h = e.layout;
    if (l == W_STATUS) h = e.status;
    if (l == W_EPISODE) h = e.episode;
    if (l == W_RECIPES) h = e.recipes;
    if (l == W_POOL) h = e.pool;
#pragma unroll
    for (int a = 0; a < MAX_AGENTS; ++a)
        if (l == AGENT_WORD0 + a)
            h = (a < cx.A) ? ((uint32_t)e.ax[a] | ((uint32_t)e.ay[a] << 8) | ((uint32_t)e.ao[a] << 16) |
                              ((uint32_t)((e.ah[a] + 1) & 0xFF) << 24))
                           : 0u;
    if (l < CELL_WORD0) rec[l] = h;
    uint8_t *cb = reinterpret_cast<uint8_t *>(rec + CELL_WORD0);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = l + 64 * k;
        if (c < cx.W * cx.H) cb[c] = (uint8_t)e.cell[k];
    }
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
        int s = l + 64 * k;
        if (s < cx.D) {
            rec[P.dyn0_off + s] = e.d0[k];
            rec[P.dyn1_off + s] = e.d1[k];
        }
    }
}

template <int OPL, int CPL>
__device__ __forceinline__ uint32_t all_marks(const Params &P, const Env<OPL, CPL> &e, const Ctx &cx) {
    uint32_t marks = 0;
#pragma unroll
    for (int r = 0; r < MAX_AGENTS; ++r) {
        if (r >= P.R) continue;
        uint32_t id = (e.recipes >> (8 * r)) & 0xFF;
        marks |= Ops<OPL, CPL>::recipe_marks(e, cx, P.recipes + (size_t)id * (1 + MAX_NODES)) << (8 * r);
    }
    return marks;
}

// LDS image of one env (per wave): what the feature encode gathers from
template <int OPL, int CPL>
struct Lds {
    uint32_t d0[OPL * 64];
    uint32_t cell[CPL * 64];
    double lutx[64], luty[64];     // (i - (W-1)) / W and (i - (H-1)) / H, i in [0, 2W-2]: exact IEEE quotients
};

// cooking_env.py:352-373 get_feature_vector for every agent of the env, through the layout's descriptor table
template <int OPL, int CPL>
__device__ __forceinline__ void observe(const Params &P, const Env<OPL, CPL> &e, const Ctx &cx, Lds<OPL, CPL> &s,
                                        double *__restrict__ out /* [A][F] of this env */) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OPL; ++k) s.d0[cx.lane + 64 * k] = e.d0[k];
#pragma unroll
    for (int k = 0; k < CPL; ++k) s.cell[cx.lane + 64 * k] = e.cell[k];
    __syncthreads();
    const uint32_t *__restrict__ desc = P.lay_desc + (size_t)e.layout * P.F;
    for (int f = cx.lane; f < P.F; f += 64) {
        const uint32_t dsc = desc[f];
        const uint32_t op = dsc & 0xFF, ref = dsc >> 8;
        int kind = 0;            // 0: constant v, 1: x-like coordinate, 2: y-like coordinate
        int coord = 0, self = -1;
        double v = 0.0;
        if (op == OP_ONE) v = 1.0;
        else if (op == OP_CONST_X) { kind = 1; coord = (int)ref; }
        else if (op == OP_CONST_Y) { kind = 2; coord = (int)ref; }
        else if (op == OP_CELL_ACTIVE) v = (s.cell[ref] & CELL_ACTIVE) ? 1.0 : 0.0;
        else if (op == OP_CELL_WALK) v = (s.cell[ref] & CELL_WALK) ? 1.0 : 0.0;
        else if (op >= OP_DYN_X && op <= OP_DYN_ONE) {
            const uint32_t w = s.d0[ref];
            if (w & D_ALIVE) {
                if (op == OP_DYN_X) { kind = 1; coord = (int)(w & 0xFF); }
                else if (op == OP_DYN_Y) { kind = 2; coord = (int)((w >> 8) & 0xFF); }
                else if (op == OP_DYN_NOTDONE) v = (w & D_DONE) ? 0.0 : 1.0;
                else if (op == OP_DYN_DONE) v = (w & D_DONE) ? 1.0 : 0.0;
                else if (op == OP_DYN_CHOPPED) v = (w & D_CHOPPED) ? 1.0 : 0.0;
                else if (op == OP_DYN_MASHED) v = (w & D_MASHED) ? 1.0 : 0.0;
                else v = 1.0;
            }
        } else if (op >= OP_AG_X) {
            int gx = 0, gy = 0, go = 0;
#pragma unroll
            for (int a = 0; a < MAX_AGENTS; ++a)
                if ((int)ref == a) { gx = e.ax[a]; gy = e.ay[a]; go = e.ao[a]; }
            if (op == OP_AG_X) { kind = 1; coord = gx; self = (int)ref; }
            else if (op == OP_AG_Y) { kind = 2; coord = gy; self = (int)ref; }
            else if (op == OP_AG_ONE) v = 1.0;
            else v = (go == (int)(op - OP_AG_O1) + 1) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int a = 0; a < MAX_AGENTS; ++a) {
            if (a >= cx.A) continue;
            double val = v;
            if (kind == 1) val = s.lutx[coord - (self == a ? 0 : e.ax[a]) + (cx.W - 1)];
            else if (kind == 2) val = s.luty[coord - (self == a ? 0 : e.ay[a]) + (cx.H - 1)];
            out[(size_t)a * P.F + f] = val;
        }
    }
}

struct StepOut {
    double rew[MAX_AGENTS];
    uint32_t term, trunc, was_reset;
};

// One accumulated_step (cooking_env.py:243-269) of one env held in registers.
template <int OPL, int CPL>
__device__ __forceinline__ void step_env(const Params &P, Env<OPL, CPL> &e, const Ctx &cx, const int (&acts)[MAX_AGENTS],
                                         int64_t env_global, StepOut &o) {
    using O = Ops<OPL, CPL>;
#pragma unroll
    for (int a = 0; a < MAX_AGENTS; ++a) o.rew[a] = 0.0;
    o.term = 0; o.trunc = 0; o.was_reset = 0;
    if (e.status & ST_DONE) {
        if (P.auto_reset) {
            // next-step autoreset: reset() of cooking_env.py:178-210 from the layout pool
            e.episode += 1;
            uint32_t lay = next_layout(env_global, e.episode, e.pool, (uint32_t)P.L);
            uint32_t recipes = e.recipes, episode = e.episode, pool = e.pool;
            load_env(P, e, cx, P.lay_init + (size_t)lay * P.RW);
            e.t = 0; e.layout = lay; e.status = 0; e.episode = episode; e.recipes = recipes; e.pool = pool;
            e.marks = all_marks(P, e, cx);
            o.was_reset = 1;
        } else {
            o.term = (e.status & ST_TERM) ? 1u : 0u;
            o.trunc = (e.status & ST_TRUNC) ? 1u : 0u;
        }
        return;
    }
    e.t += 1;                                                        // cooking_env.py:244
    uint32_t pressed = 0;
    O::perform_agent_actions(e, cx, acts, P.scheme, pressed);       // cooking_world.py:104-108
    O::progress_and_link(e, cx, pressed);                           // :109-110 (handle_agent_spawn: neutral at rate 0)
    // compute_rewards cooking_env.py:290-315
    const bool truncated = (int)e.t >= P.max_steps;                 // compute_truncated :333-350
    const uint32_t before = e.marks;
    uint32_t after = 0;
    int n_completed = 0;
#pragma unroll
    for (int r = 0; r < MAX_AGENTS; ++r) {
        if (r >= P.R) continue;
        const uint32_t id = (e.recipes >> (8 * r)) & 0xFF;
        const uint32_t *rp = P.recipes + (size_t)id * (1 + MAX_NODES);
        const uint32_t ma = O::recipe_marks(e, cx, rp);
        const uint32_t mb = (before >> (8 * r)) & 0xFF;
        after |= ma << (8 * r);
        // goals_completed sums (recipe.py:36-40): open goal slots before / after
        uint32_t countmask = 0;
        const int n = (int)rfl(rp[0]);
#pragma unroll
        for (int j = 0; j < MAX_NODES; ++j)
            if (j < n && ((rfl(rp[1 + j]) >> 24) & 1)) countmask |= 1u << j;
        const int goals_before = __popc(~mb & countmask), goals_after = __popc(~ma & countmask);
        const bool completed = ma & 1, completion_before = mb & 1;
        const bool malus = !completed && completion_before, bonus = completed && !completion_before;
        double x = 0.0;
        x += (double)(goals_before - goals_after) * P.node_reward;
        x += (bonus ? 1.0 : 0.0) * P.recipe_reward;
        x += (malus ? 1.0 : 0.0) * P.recipe_penalty;
        x += P.time_penalty_step;
        if (r < cx.A) o.rew[r] = x;
        n_completed += completed ? 1 : 0;
    }
    e.marks = after;
    const bool done = P.end_all ? (n_completed == P.R) : (n_completed > 0);
    o.term = done ? 1u : 0u;
    o.trunc = truncated ? 1u : 0u;
    if (done || truncated) e.status |= ST_DONE | (done ? ST_TERM : 0u) | (truncated ? ST_TRUNC : 0u);
}

template <int OPL, int CPL>
__global__ __launch_bounds__(64) void k_step(Params P) {
    __shared__ Lds<OPL, CPL> lds;
    const int env = blockIdx.x;
    Ctx cx{P.A, P.W, P.H, P.D, (int)threadIdx.x};
    if (cx.lane < 2 * P.W - 1) lds.lutx[cx.lane] = (double)(cx.lane - (P.W - 1)) / (double)P.W;
    if (cx.lane < 2 * P.H - 1) lds.luty[cx.lane] = (double)(cx.lane - (P.H - 1)) / (double)P.H;
    Env<OPL, CPL> e;
    uint32_t *rec = P.state + (size_t)env * P.RW;
    load_env(P, e, cx, rec);
    const int64_t env_global = P.env_id_base + env;

    // statistics registers (lane-replicated uniform values)
    uint32_t *su = P.stat_u + (size_t)env * SU_WORDS;
    double *sf = P.stat_f + (size_t)env * SF_WORDS;
    uint32_t s_steps = 0, s_episodes = 0, s_lensum = 0, s_trunc = 0, s_term = 0;
    uint32_t s_completed[MAX_AGENTS] = {0, 0, 0, 0};
    double s_cur[MAX_AGENTS], s_sum[MAX_AGENTS] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int a = 0; a < MAX_AGENTS; ++a) s_cur[a] = (a < P.A) ? sf[SF_CUR0 + a] : 0.0;

    for (int t = 0; t < P.T; ++t) {
        int acts[MAX_AGENTS] = {0, 0, 0, 0};
        if (P.actions) {
            int v = (cx.lane < P.A) ? P.actions[(size_t)env * P.A + cx.lane] : 0;
#pragma unroll
            for (int a = 0; a < MAX_AGENTS; ++a) acts[a] = (int)rdl((uint32_t)v, a);
        } else {
            const uint32_t nact = P.scheme == 3 ? 5u : 8u;
#pragma unroll
            for (int a = 0; a < MAX_AGENTS; ++a)
                acts[a] = (a < P.A) ? (int)action_hash(P.seed, env_global, a, P.step0 + (uint32_t)t, nact) : 0;
        }
        const bool was_done = (e.status & ST_DONE) != 0;
        StepOut o;
        step_env(P, e, cx, acts, env_global, o);
        // ---- statistics
        if (!was_done) {
            s_steps += 1;
#pragma unroll
            for (int a = 0; a < MAX_AGENTS; ++a) s_cur[a] += o.rew[a];
            if (e.status & ST_DONE) {
                s_episodes += 1;
                s_lensum += e.t;
                s_trunc += o.trunc;
                s_term += o.term;
#pragma unroll
                for (int a = 0; a < MAX_AGENTS; ++a) {
                    if (a >= P.A) continue;
                    s_sum[a] += s_cur[a];
                    s_cur[a] = 0.0;
                    s_completed[a] += (e.marks >> (8 * a)) & 1u;
                }
            }
        }
        // ---- outputs of this step
        const size_t row = (size_t)t * P.N + env;
        if (P.rewards && cx.lane < P.A) {
            double r = 0.0;
#pragma unroll
            for (int a = 0; a < MAX_AGENTS; ++a)
                if (cx.lane == a) r = o.rew[a];
            P.rewards[row * P.A + cx.lane] = r;
        }
        if (P.term && cx.lane < P.A) P.term[row * P.A + cx.lane] = (uint8_t)o.term;
        if (P.trunc && cx.lane < P.A) P.trunc[row * P.A + cx.lane] = (uint8_t)o.trunc;
        if (P.obs) observe(P, e, cx, lds, P.obs + row * (size_t)P.A * P.F);
    }
    store_env(P, e, cx, rec);
    if (cx.lane == 0) {
        su[SU_STEPS] += s_steps; su[SU_EPISODES] += s_episodes; su[SU_LENSUM] += s_lensum;
        su[SU_TRUNC] += s_trunc; su[SU_TERM] += s_term;
    }
    if (cx.lane < P.A) {
        uint32_t c = 0;
        double cur = 0.0, sum = 0.0;
#pragma unroll
        for (int a = 0; a < MAX_AGENTS; ++a)
            if (cx.lane == a) { c = s_completed[a]; cur = s_cur[a]; sum = s_sum[a]; }
        su[SU_COMPLETED0 + cx.lane] += c;
        sf[SF_CUR0 + cx.lane] = cur;
        sf[SF_SUM0 + cx.lane] += sum;
    }
}

// reset(): cooking_env.py:178-210 for envs [env_begin, env_begin + count)
template <int OPL, int CPL>
__global__ __launch_bounds__(64) void k_reset(Params P, int64_t env_begin, const int32_t *__restrict__ layout_ids,
                                              const uint32_t *__restrict__ recipe_words, const uint32_t *__restrict__ pool_words, double *obs_out) {
    __shared__ Lds<OPL, CPL> lds;
    const int i = blockIdx.x;
    const int64_t env = env_begin + i;
    Ctx cx{P.A, P.W, P.H, P.D, (int)threadIdx.x};
    if (cx.lane < 2 * P.W - 1) lds.lutx[cx.lane] = (double)(cx.lane - (P.W - 1)) / (double)P.W;
    if (cx.lane < 2 * P.H - 1) lds.luty[cx.lane] = (double)(cx.lane - (P.H - 1)) / (double)P.H;
    Env<OPL, CPL> e;
    uint32_t *rec = P.state + (size_t)env * P.RW;
    const uint32_t lay = rfl((uint32_t)layout_ids[i]);
    const uint32_t old_episode = rfl(rec[W_EPISODE]);
    load_env(P, e, cx, P.lay_init + (size_t)lay * P.RW);
    e.t = 0; e.layout = lay; e.status = 0; e.episode = old_episode; e.recipes = rfl(recipe_words[i]);
    e.pool = rfl(pool_words[i]);
    e.marks = all_marks(P, e, cx);
    store_env(P, e, cx, rec);
    if (cx.lane < P.A) P.stat_f[(size_t)env * SF_WORDS + SF_CUR0 + cx.lane] = 0.0;
    if (obs_out) observe(P, e, cx, lds, obs_out + (size_t)i * P.A * P.F);
}

// observe() only (used after cz_set_state by the host API)
template <int OPL, int CPL>
__global__ __launch_bounds__(64) void k_observe(Params P, int64_t env_begin, double *obs_out) {
    __shared__ Lds<OPL, CPL> lds;
    const int i = blockIdx.x;
    Ctx cx{P.A, P.W, P.H, P.D, (int)threadIdx.x};
    if (cx.lane < 2 * P.W - 1) lds.lutx[cx.lane] = (double)(cx.lane - (P.W - 1)) / (double)P.W;
    if (cx.lane < 2 * P.H - 1) lds.luty[cx.lane] = (double)(cx.lane - (P.H - 1)) / (double)P.H;
    Env<OPL, CPL> e;
    load_env(P, e, cx, P.state + (size_t)(env_begin + i) * P.RW);
    observe(P, e, cx, lds, obs_out + (size_t)i * P.A * P.F);
}

// deterministic reduction of the per-env statistics into one cz_stats (fixed thread->env mapping, fixed tree)
__global__ __launch_bounds__(256) void k_stats_reduce(const uint32_t *__restrict__ su, const double *__restrict__ sf, int N,
                                                      cz_stats *out) {
    __shared__ unsigned long long su64[256][9];
    __shared__ double sd[256][4];
    unsigned long long acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double ret[4] = {0.0, 0.0, 0.0, 0.0};
    for (int e = threadIdx.x; e < N; e += 256) {
        const uint32_t *p = su + (size_t)e * SU_WORDS;
        acc[0] += p[SU_STEPS]; acc[1] += p[SU_EPISODES]; acc[2] += p[SU_LENSUM]; acc[3] += p[SU_TRUNC]; acc[4] += p[SU_TERM];
        for (int a = 0; a < 4; ++a) acc[5 + a] += p[SU_COMPLETED0 + a];
        for (int a = 0; a < 4; ++a) ret[a] += sf[(size_t)e * SF_WORDS + SF_SUM0 + a];
    }
    for (int j = 0; j < 9; ++j) su64[threadIdx.x][j] = acc[j];
    for (int a = 0; a < 4; ++a) sd[threadIdx.x][a] = ret[a];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) {
            for (int j = 0; j < 9; ++j) su64[threadIdx.x][j] += su64[threadIdx.x + stride][j];
            for (int a = 0; a < 4; ++a) sd[threadIdx.x][a] += sd[threadIdx.x + stride][a];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out->env_steps = su64[0][0]; out->episodes = su64[0][1]; out->length_sum = su64[0][2];
        out->truncations = su64[0][3]; out->terminations = su64[0][4];
        for (int a = 0; a < 4; ++a) { out->recipes_completed[a] = su64[0][5 + a]; out->return_sum[a] = sd[0][a]; }
    }
}

// ======================================================================================================
// host: handle + C-ABI
// ======================================================================================================

struct cz_handle_s {
    cz_config cfg;
    Params P;
    hipStream_t stream = nullptr;
    int opl = 1, cpl = 1;
    int n_layouts = 0, n_recipes = 0;
    uint32_t *d_state = nullptr, *d_lay_init = nullptr, *d_lay_desc = nullptr, *d_recipes = nullptr;
    uint32_t *d_stat_u = nullptr;
    double *d_stat_f = nullptr;
    cz_stats *d_stats_out = nullptr;
    // staging for the host-pointer API
    int32_t *d_actions = nullptr;
    double *d_obs = nullptr, *d_rew = nullptr;
    uint8_t *d_term = nullptr, *d_trunc = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // kernel timing
    bool ktime = false;
    std::vector<hipEvent_t> kev;
    size_t kev_used = 0;
    double ktime_ms = 0.0;
    int64_t klaunches = 0;
    // RCCL (lazy)
    void *rccl = nullptr;
    void *comm = nullptr;
    int n_ranks = 1, rank = 0;
    cz_stats *d_gather = nullptr;
    std::string err;
};

static thread_local std::string g_err;

static int fail(cz_handle h, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_err = buf;
    return 1;
}

#define HIPCHK(h, call)                                                                      \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess) return fail(h, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)

extern "C" const char *cz_last_error(cz_handle h) { return h ? h->err.c_str() : g_err.c_str(); }
extern "C" int32_t cz_abi_version(void) { return 1; }
extern "C" int32_t cz_sizeof_config(void) { return (int32_t)sizeof(cz_config); }
extern "C" int32_t cz_sizeof_stats(void) { return (int32_t)sizeof(cz_stats); }
extern "C" uint32_t cz_action(uint64_t seed, int64_t env_global, int32_t agent, uint32_t step, uint32_t n_actions) {
    return action_hash(seed, env_global, agent, step, n_actions);
}
extern "C" uint32_t cz_next_layout(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t n_layouts) {
    return next_layout(env_global, episode, pool_word, n_layouts);
}

extern "C" int cz_create(const cz_config *cfg, cz_handle *out) {
    if (!cfg || !out) return fail(nullptr, "cz_create: null argument");
    *out = nullptr;
    const int C = cfg->width * cfg->height;
    if (cfg->num_envs < 1) return fail(nullptr, "cz_create: num_envs must be >= 1");
    if (cfg->num_agents < 1 || cfg->num_agents > MAX_AGENTS) return fail(nullptr, "cz_create: num_agents must be 1..4");
    if (cfg->num_recipes < cfg->num_agents || cfg->num_recipes > MAX_AGENTS)
        return fail(nullptr, "cz_create: need num_agents <= num_recipes <= 4 (one recipe per agent, cooking_env.py:329)");
    if (cfg->width < 1 || cfg->height < 1 || cfg->width > 32 || cfg->height > 32 || C > 256)
        return fail(nullptr, "cz_create: grid %dx%d unsupported (W,H <= 32 and W*H <= 256)", cfg->width, cfg->height);
    if (cfg->max_dyn < 1 || cfg->max_dyn > 128) return fail(nullptr, "cz_create: max_dyn must be 1..128");
    if (cfg->action_scheme != 1 && cfg->action_scheme != 3)
        return fail(nullptr, "cz_create: action_scheme must be 1 or 3 (scheme2 is unusable in the reference)");
    if (cfg->max_steps < 1 || cfg->feat_len < 1) return fail(nullptr, "cz_create: max_steps and feat_len must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, "cz_create: no HIP device available (this library has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, "cz_create: device_id %d out of range", cfg->device_id);
    cz_handle h = new cz_handle_s();
    h->cfg = *cfg;
    HIPCHK(nullptr, hipSetDevice(cfg->device_id));
    HIPCHK(nullptr, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(nullptr, hipEventCreate(&h->ev0));
    HIPCHK(nullptr, hipEventCreate(&h->ev1));
    Params &P = h->P;
    memset(&P, 0, sizeof P);
    P.N = cfg->num_envs; P.A = cfg->num_agents; P.W = cfg->width; P.H = cfg->height; P.D = cfg->max_dyn; P.F = cfg->feat_len;
    const int CW = (C + 3) / 4;
    P.dyn0_off = CELL_WORD0 + CW;
    P.dyn1_off = P.dyn0_off + P.D;
    P.RW = (P.dyn1_off + P.D + 15) / 16 * 16;
    P.scheme = cfg->action_scheme; P.max_steps = cfg->max_steps; P.end_all = cfg->end_condition_all ? 1 : 0;
    P.R = cfg->num_recipes; P.auto_reset = cfg->auto_reset ? 1 : 0; P.env_id_base = cfg->env_id_base;
    P.recipe_reward = cfg->recipe_reward; P.recipe_penalty = cfg->recipe_penalty; P.node_reward = cfg->recipe_node_reward;
    P.time_penalty_step = cfg->max_time_penalty / (double)cfg->max_steps;       // cooking_env.py:307
    P.T = 1;
    h->opl = (P.D + 63) / 64;
    h->cpl = (C <= 64) ? 1 : 4;
    if (h->opl == 2 || h->cpl == 4) { h->opl = 2; h->cpl = 4; }
    const size_t N = (size_t)P.N;
    HIPCHK(nullptr, hipMalloc(&h->d_state, N * P.RW * 4));
    HIPCHK(nullptr, hipMemsetAsync(h->d_state, 0, N * P.RW * 4, h->stream));
    HIPCHK(nullptr, hipMalloc(&h->d_stat_u, N * SU_WORDS * 4));
    HIPCHK(nullptr, hipMalloc(&h->d_stat_f, N * SF_WORDS * 8));
    HIPCHK(nullptr, hipMemsetAsync(h->d_stat_u, 0, N * SU_WORDS * 4, h->stream));
    HIPCHK(nullptr, hipMemsetAsync(h->d_stat_f, 0, N * SF_WORDS * 8, h->stream));
    HIPCHK(nullptr, hipMalloc(&h->d_stats_out, sizeof(cz_stats)));
    HIPCHK(nullptr, hipStreamSynchronize(h->stream));
    P.state = h->d_state; P.stat_u = h->d_stat_u; P.stat_f = h->d_stat_f;
    *out = h;
    return 0;
}

extern "C" int cz_destroy(cz_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device_id);
    (void)hipStreamSynchronize(h->stream);
    if (h->comm && h->rccl) {
        typedef int (*destroy_t)(void *);
        destroy_t f = (destroy_t)dlsym(h->rccl, "ncclCommDestroy");
        if (f) f(h->comm);
    }
    void *ptrs[] = {h->d_state, h->d_lay_init, h->d_lay_desc, h->d_recipes, h->d_stat_u, h->d_stat_f, h->d_stats_out,
                    h->d_actions, h->d_obs, h->d_rew, h->d_term, h->d_trunc, h->d_gather};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : h->kev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

extern "C" int32_t cz_record_words(cz_handle h) { return h ? h->P.RW : 0; }
extern "C" int cz_sync(cz_handle h) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cz_load_recipes(cz_handle h, const uint32_t *table, int32_t n) {
    if (!h || !table || n < 1 || n > 255) return fail(h, "cz_load_recipes: bad arguments");
    for (int i = 0; i < n; ++i)
        if (table[(size_t)i * (1 + MAX_NODES)] > MAX_NODES) return fail(h, "cz_load_recipes: recipe %d has more than 8 nodes", i);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (h->d_recipes) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(h->d_recipes)); }
    size_t bytes = (size_t)n * (1 + MAX_NODES) * 4;
    HIPCHK(h, hipMalloc(&h->d_recipes, bytes));
    HIPCHK(h, hipMemcpyAsync(h->d_recipes, table, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_recipes = n;
    h->P.recipes = h->d_recipes;
    return 0;
}

extern "C" int cz_load_layouts(cz_handle h, const uint32_t *init_records, const uint32_t *obs_desc, int32_t n) {
    if (!h || !init_records || !obs_desc || n < 1) return fail(h, "cz_load_layouts: bad arguments");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_lay_init) HIPCHK(h, hipFree(h->d_lay_init));
    if (h->d_lay_desc) HIPCHK(h, hipFree(h->d_lay_desc));
    // validate descriptors: refs must stay inside the LDS image
    const int C = h->P.W * h->P.H;
    for (size_t i = 0; i < (size_t)n * h->P.F; ++i) {
        uint32_t op = obs_desc[i] & 0xFF, ref = obs_desc[i] >> 8;
        bool ok = op <= OP_AG_ONE;
        if (op == OP_CELL_ACTIVE || op == OP_CELL_WALK) ok = ok && (int)ref < C;
        if (op >= OP_DYN_X && op <= OP_DYN_ONE) ok = ok && (int)ref < h->P.D;
        if (op >= OP_AG_X) ok = ok && (int)ref < h->P.A;
        if (op == OP_CONST_X) ok = ok && (int)ref < h->P.W;
        if (op == OP_CONST_Y) ok = ok && (int)ref < h->P.H;
        if (!ok) return fail(h, "cz_load_layouts: bad observation descriptor %#x at %zu", obs_desc[i], i);
    }
    size_t b0 = (size_t)n * h->P.RW * 4, b1 = (size_t)n * h->P.F * 4;
    HIPCHK(h, hipMalloc(&h->d_lay_init, b0));
    HIPCHK(h, hipMalloc(&h->d_lay_desc, b1));
    HIPCHK(h, hipMemcpyAsync(h->d_lay_init, init_records, b0, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_lay_desc, obs_desc, b1, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_layouts = n;
    h->P.lay_init = h->d_lay_init; h->P.lay_desc = h->d_lay_desc; h->P.L = n;
    return 0;
}

static int check_range(cz_handle h, int64_t b, int64_t c) {
    if (!h) return fail(nullptr, "null handle");
    if (b < 0 || c < 0 || b + c > h->P.N) return fail(h, "env range [%lld, %lld) outside [0, %d)", (long long)b, (long long)(b + c), h->P.N);
    return 0;
}

extern "C" int cz_set_state(cz_handle h, int64_t b, int64_t c, const uint32_t *records) {
    if (check_range(h, b, c)) return 1;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMemcpyAsync(h->d_state + (size_t)b * h->P.RW, records, (size_t)c * h->P.RW * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cz_get_state(cz_handle h, int64_t b, int64_t c, uint32_t *records) {
    if (check_range(h, b, c)) return 1;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMemcpyAsync(records, h->d_state + (size_t)b * h->P.RW, (size_t)c * h->P.RW * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int ready(cz_handle h) {
    if (!h) return fail(nullptr, "null handle");
    if (!h->P.recipes) return fail(h, "recipes not loaded (cz_load_recipes)");
    if (!h->P.lay_init) return fail(h, "layouts not loaded (cz_load_layouts)");
    return 0;
}

template <class F1, class F2>
static void dispatch(cz_handle h, F1 small, F2 large) {
    if (h->opl == 1 && h->cpl == 1) small(); else large();
}

static int launch_step(cz_handle h, const Params &P) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->ktime) {
        while (h->kev.size() < h->kev_used + 2) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e));
            h->kev.push_back(e);
        }
        e0 = h->kev[h->kev_used]; e1 = h->kev[h->kev_used + 1];
        h->kev_used += 2;
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    dispatch(h, [&] { hipLaunchKernelGGL((k_step<1, 1>), dim3(P.N), dim3(64), 0, h->stream, P); },
             [&] { hipLaunchKernelGGL((k_step<2, 4>), dim3(P.N), dim3(64), 0, h->stream, P); });
    HIPCHK(h, hipGetLastError());
    if (h->ktime) HIPCHK(h, hipEventRecord(e1, h->stream));
    return 0;
}

extern "C" int cz_reset(cz_handle h, int64_t b, int64_t c, const int32_t *layout_ids, const uint8_t *recipe_ids,
                        const uint32_t *pool_words, double *obs) {
    if (ready(h) || check_range(h, b, c)) return 1;
    if (c == 0) return 0;
    if (!layout_ids || !recipe_ids) return fail(h, "cz_reset: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    std::vector<uint32_t> pools((size_t)c, 0u);
    for (int64_t i = 0; i < c; ++i) {
        if (layout_ids[i] < 0 || layout_ids[i] >= h->n_layouts) return fail(h, "cz_reset: layout id %d out of range", layout_ids[i]);
        for (int r = 0; r < h->P.R; ++r)
            if (recipe_ids[i * 4 + r] >= h->n_recipes) return fail(h, "cz_reset: recipe id %d out of range", recipe_ids[i * 4 + r]);
        if (pool_words) {
            uint32_t base = pool_words[i] & 0xFFFFu, count = pool_words[i] >> 16;
            if (count && (int)(base + count) > h->n_layouts) return fail(h, "cz_reset: layout pool slice [%u,%u) out of range", base, base + count);
            pools[(size_t)i] = pool_words[i];
        }
    }
    int32_t *d_lay = nullptr;
    uint32_t *d_rec = nullptr, *d_pool = nullptr;
    double *d_obs = nullptr;
    HIPCHK(h, hipMalloc(&d_lay, (size_t)c * 4));
    HIPCHK(h, hipMalloc(&d_rec, (size_t)c * 4));
    HIPCHK(h, hipMalloc(&d_pool, (size_t)c * 4));
    HIPCHK(h, hipMemcpyAsync(d_lay, layout_ids, (size_t)c * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_rec, recipe_ids, (size_t)c * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_pool, pools.data(), (size_t)c * 4, hipMemcpyHostToDevice, h->stream));
    size_t ob = (size_t)c * h->P.A * h->P.F * 8;
    if (obs) HIPCHK(h, hipMalloc(&d_obs, ob));
    Params P = h->P;
    dispatch(h, [&] { hipLaunchKernelGGL((k_reset<1, 1>), dim3((unsigned)c), dim3(64), 0, h->stream, P, b, d_lay, d_rec, d_pool, d_obs); },
             [&] { hipLaunchKernelGGL((k_reset<2, 4>), dim3((unsigned)c), dim3(64), 0, h->stream, P, b, d_lay, d_rec, d_pool, d_obs); });
    HIPCHK(h, hipGetLastError());
    if (obs) HIPCHK(h, hipMemcpyAsync(obs, d_obs, ob, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(d_lay); (void)hipFree(d_rec); (void)hipFree(d_pool);
    if (d_obs) (void)hipFree(d_obs);
    return 0;
}

// observe() of the current state (after cz_set_state), host buffer [count][A][F]
extern "C" int cz_observe(cz_handle h, int64_t b, int64_t c, double *obs) {
    if (ready(h) || check_range(h, b, c)) return 1;
    if (c == 0) return 0;
    if (!obs) return fail(h, "cz_observe: null buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    double *d_obs = nullptr;
    size_t ob = (size_t)c * h->P.A * h->P.F * 8;
    HIPCHK(h, hipMalloc(&d_obs, ob));
    Params P = h->P;
    dispatch(h, [&] { hipLaunchKernelGGL((k_observe<1, 1>), dim3((unsigned)c), dim3(64), 0, h->stream, P, b, d_obs); },
             [&] { hipLaunchKernelGGL((k_observe<2, 4>), dim3((unsigned)c), dim3(64), 0, h->stream, P, b, d_obs); });
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(obs, d_obs, ob, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(d_obs);
    return 0;
}

extern "C" int cz_step_device(cz_handle h, const int32_t *d_actions, double *d_obs, double *d_rewards, uint8_t *d_term,
                              uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (!d_actions) return fail(h, "cz_step_device: actions pointer is null");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    Params P = h->P;
    P.actions = d_actions; P.obs = d_obs; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc; P.T = 1;
    return launch_step(h, P);
}

extern "C" int cz_rollout(cz_handle h, int32_t T, uint64_t seed, uint32_t step0, double *d_obs, double *d_rewards,
                          uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (T < 1) return fail(h, "cz_rollout: T must be >= 1");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    Params P = h->P;
    P.actions = nullptr; P.obs = d_obs; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc;
    P.T = T; P.seed = seed; P.step0 = step0;
    return launch_step(h, P);
}

extern "C" int cz_step(cz_handle h, const int32_t *actions, double *obs, double *rewards, uint8_t *term, uint8_t *trunc) {
    if (ready(h)) return 1;
    if (!actions || !rewards || !term || !trunc) return fail(h, "cz_step: null buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t NA = (size_t)h->P.N * h->P.A;
    if (!h->d_actions) {
        HIPCHK(h, hipMalloc(&h->d_actions, NA * 4));
        HIPCHK(h, hipMalloc(&h->d_rew, NA * 8));
        HIPCHK(h, hipMalloc(&h->d_term, NA));
        HIPCHK(h, hipMalloc(&h->d_trunc, NA));
    }
    if (obs && !h->d_obs) HIPCHK(h, hipMalloc(&h->d_obs, NA * h->P.F * 8));
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, NA * 4, hipMemcpyHostToDevice, h->stream));
    if (cz_step_device(h, h->d_actions, obs ? h->d_obs : nullptr, h->d_rew, h->d_term, h->d_trunc)) return 1;
    if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, NA * h->P.F * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(rewards, h->d_rew, NA * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(term, h->d_term, NA, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(trunc, h->d_trunc, NA, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- device memory + timing helpers --------------------------------------------------------------------
extern "C" void *cz_dev_alloc(cz_handle h, size_t bytes) {
    if (!h) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(h->cfg.device_id) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
        fail(h, "cz_dev_alloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
extern "C" int cz_dev_free(cz_handle h, void *p) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(p));
    return 0;
}
extern "C" int cz_memcpy_h2d(cz_handle h, void *d, const void *s, size_t n) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int cz_memcpy_d2h(cz_handle h, void *d, const void *s, size_t n) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int cz_timer_start(cz_handle h) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    return 0;
}
extern "C" int cz_timer_stop(cz_handle h, float *ms) {
    if (!h || !ms) return fail(h, "cz_timer_stop: null argument");
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}
extern "C" int cz_kernel_time_reset(cz_handle h, int32_t enable) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->ktime = enable != 0;
    h->kev_used = 0;
    h->ktime_ms = 0.0;
    h->klaunches = 0;
    return 0;
}
extern "C" int cz_kernel_time_read(cz_handle h, double *total_ms, int64_t *launches) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i + 1 < h->kev_used; i += 2) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->kev[i], h->kev[i + 1]));
        h->ktime_ms += ms;
        h->klaunches += 1;
    }
    h->kev_used = 0;
    if (total_ms) *total_ms = h->ktime_ms;
    if (launches) *launches = h->klaunches;
    return 0;
}

// ---- statistics + RCCL ----------------------------------------------------------------------------------
extern "C" int cz_get_stats(cz_handle h, cz_stats *out) {
    if (!h || !out) return fail(h, "cz_get_stats: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(256), 0, h->stream, h->d_stat_u, h->d_stat_f, h->P.N, h->d_stats_out);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->d_stats_out, sizeof(cz_stats), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int cz_reset_stats(cz_handle h) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipMemsetAsync(h->d_stat_u, 0, (size_t)h->P.N * SU_WORDS * 4, h->stream));
    // keep the running return of the episode in flight (SF_CUR), clear the finished-episode sums
    HIPCHK(h, hipMemset2DAsync(h->d_stat_f + SF_SUM0, SF_WORDS * 8, 0, 4 * 8, (size_t)h->P.N, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// RCCL is loaded lazily so that the library also loads on machines without it (CPU build check)
typedef struct { char internal[128]; } cz_nccl_id;
static void *rccl_lib(cz_handle h) {
    static void *lib = nullptr;
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) fail(h, "cannot load librccl.so: %s", dlerror());
    return lib;
}
extern "C" int cz_comm_unique_id(uint8_t id[128]) {
    void *lib = rccl_lib(nullptr);
    if (!lib) return 1;
    typedef int (*fn_t)(cz_nccl_id *);
    fn_t f = (fn_t)dlsym(lib, "ncclGetUniqueId");
    if (!f) return fail(nullptr, "ncclGetUniqueId not found");
    cz_nccl_id u;
    int r = f(&u);
    if (r) return fail(nullptr, "ncclGetUniqueId failed: %d", r);
    memcpy(id, u.internal, 128);
    return 0;
}
extern "C" int cz_comm_init(cz_handle h, int32_t n_ranks, int32_t rank, const uint8_t id[128]) {
    if (!h) return fail(nullptr, "null handle");
    h->rccl = rccl_lib(h);
    if (!h->rccl) return 1;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    typedef int (*fn_t)(void **, int, cz_nccl_id, int);
    fn_t f = (fn_t)dlsym(h->rccl, "ncclCommInitRank");
    if (!f) return fail(h, "ncclCommInitRank not found");
    cz_nccl_id u;
    memcpy(u.internal, id, 128);
    int r = f(&h->comm, n_ranks, u, rank);
    if (r) return fail(h, "ncclCommInitRank failed: %d", r);
    h->n_ranks = n_ranks; h->rank = rank;
    HIPCHK(h, hipMalloc(&h->d_gather, sizeof(cz_stats) * (size_t)n_ranks));
    return 0;
}
extern "C" int cz_stats_allgather(cz_handle h, cz_stats *out) {
    if (!h || !out) return fail(h, "cz_stats_allgather: null argument");
    if (!h->comm) return fail(h, "cz_stats_allgather: communicator not initialised (cz_comm_init)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(256), 0, h->stream, h->d_stat_u, h->d_stat_f, h->P.N, h->d_stats_out);
    HIPCHK(h, hipGetLastError());
    typedef int (*fn_t)(const void *, void *, size_t, int, void *, hipStream_t);
    fn_t f = (fn_t)dlsym(h->rccl, "ncclAllGather");
    if (!f) return fail(h, "ncclAllGather not found");
    int r = f(h->d_stats_out, h->d_gather, sizeof(cz_stats), /*ncclInt8*/ 0, h->comm, h->stream);
    if (r) return fail(h, "ncclAllGather failed: %d", r);
    HIPCHK(h, hipMemcpyAsync(out, h->d_gather, sizeof(cz_stats) * (size_t)h->n_ranks, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
