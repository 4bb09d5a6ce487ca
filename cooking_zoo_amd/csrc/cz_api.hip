// cz_api.hip -- C-ABI (include/cookingzoo.h) of the MI355X-native CookingZoo step path: handle, tables, launches,
// statistics, RCCL.  The kernels live in cz_kernels.h / cz_device.h (one wavefront per env instance) and are
// instantiated in cz_inst_small.hip / cz_inst_large.hip.  gfx950 only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <vector>

#include "../../include/cookingzoo.h"
#include "cz_kernels.h"

using namespace cz;

// Deterministic reduction of the per-env statistics into one cz_stats, in two launches.
// env_steps = finished-episode lengths + steps of the episode in flight + the signed correction word SU_STEPS
// (steps of episodes aborted by cz_reset, minus what was in flight at cz_reset_stats).
// Summation order (fixed, independent of the grid): 256 chains, chain c adds envs c, c + 256, c + 512, ... one after
// the other; the 256 chain sums are then combined by a fixed binary tree.  Stage 1 spreads the chains over 64
// single-wave workgroups (4 chains each, 16 lanes per chain: lane j of a chain owns output column j), so the records
// are pulled in by 64 CUs instead of one; stage 2 is the tree.  Integer columns are exact in any order; the float64
// columns are bitwise reproducible because the order never changes.
constexpr int STAT_CHAINS = 256, STAT_COLS = 13;      // columns: 9 integer sums (u64), 4 return sums (f64)
__global__ __launch_bounds__(64) void k_stats_chains(const uint32_t *__restrict__ su, const double *__restrict__ sf,
                                                     const uint32_t *__restrict__ state, int RW, int N,
                                                     unsigned long long *__restrict__ part /* [256][16] */) {
    const int chain = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 4), col = (int)threadIdx.x & 15;
    // which word of the per-env counters this lane's column sums (branch-free loop body: every load of a batch of envs is
    // issued before anything is added, so the chain pays one memory round trip per batch, not per env)
    const int widx = col == 0 ? (int)SU_STEPS : col == 1 ? (int)SU_EPISODES : col == 2 ? (int)SU_LENSUM : col == 3 ? (int)SU_TRUNC
                   : col == 4 ? (int)SU_TERM : col < 9 ? (int)SU_COMPLETED0 + col - 5 : 0;
    const int fidx = (col >= 9 && col < STAT_COLS) ? (int)SF_SUM0 + col - 9 : (int)SF_SUM0;
    unsigned long long acc = 0;
    double ret = 0.0;
    constexpr int B = 8;
    for (int e0 = chain; e0 < N; e0 += STAT_CHAINS * B) {
        uint32_t a[B], len[B], st[B], t[B];
        double f[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int e = min(e0 + j * STAT_CHAINS, N - 1);                    // clamped: the value is dropped below
            const uint32_t *p = su + (size_t)e * SU_WORDS;
            const uint32_t *rec = state + (size_t)e * RW;
            a[j] = p[widx]; len[j] = p[SU_LENSUM]; st[j] = rec[W_STATUS]; t[j] = rec[W_T];
            f[j] = sf[(size_t)e * SF_WORDS + fidx];
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (e0 + j * STAT_CHAINS >= N) break;
            const long long steps = (long long)(int)a[j] + (long long)len[j] + ((st[j] & ST_DONE) ? 0 : (long long)t[j]);
            acc += col == 0 ? (unsigned long long)steps : (unsigned long long)a[j];
            ret += f[j];                                                        // env order within the chain: fixed
        }
    }
    part[(size_t)chain * 16 + col] = (col >= 9 && col < STAT_COLS) ? (unsigned long long)__double_as_longlong(ret) : acc;
}
__global__ __launch_bounds__(256) void k_stats_tree(const unsigned long long *__restrict__ part, cz_stats *out) {
    __shared__ unsigned long long su64[256][9];
    __shared__ double sd[256][4];
    for (int j = 0; j < 9; ++j) su64[threadIdx.x][j] = part[(size_t)threadIdx.x * 16 + j];
    for (int a = 0; a < 4; ++a) sd[threadIdx.x][a] = __longlong_as_double((long long)part[(size_t)threadIdx.x * 16 + 9 + a]);
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

// cz_reset_stats: zero the counters; steps already taken by episodes in flight must not be counted again
__global__ void k_stats_clear(uint32_t *su, double *sf, const uint32_t *__restrict__ state, int RW, int N) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    uint32_t *p = su + (size_t)e * SU_WORDS;
    const uint32_t *rec = state + (size_t)e * RW;
    for (int j = 0; j < (int)SU_WORDS; ++j) p[j] = 0;
    p[SU_STEPS] = (rec[W_STATUS] & ST_DONE) ? 0u : (uint32_t)(-(int)rec[W_T]);
    for (int a = 0; a < 4; ++a) sf[(size_t)e * SF_WORDS + SF_SUM0 + a] = 0.0;
}

// cz_reset on envs whose episode is still running: their steps stay counted as env-steps
// measurement aid (cz_probe_output_only): a grid of the step kernel's shape that does nothing but write `bytes` with the
// same write-through 16-byte stores the observation encode uses -- the floor of any launch that has to emit that much
__global__ __launch_bounds__(64 * ENVS_PER_WG) void k_probe_fill(void *dst, uint32_t bytes) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)bytes, 0x00020000);
    const uint32_t stride = gridDim.x * blockDim.x * 16u;
    for (uint32_t off = (blockIdx.x * blockDim.x + threadIdx.x) * 16u; off + 16u <= bytes; off += stride)
        __builtin_amdgcn_raw_buffer_store_b128(u4{0, 0, 0, 0}, rs, off, 0, 16);
}

// measurement aid (cz_probe_closed_loop): the smallest "policy" there is - every agent's next action is a hash of a few
// doubles of the observation it was just given - so that step k + 1 depends on step k's observation through a kernel of the
// caller, as in any reinforcement-learning loop
__global__ void k_probe_policy(const double *__restrict__ obs, int32_t *__restrict__ actions, int n_rows, int F, uint32_t n_actions) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;               // (env, agent)
    if (row >= n_rows) return;
    const unsigned long long *o = reinterpret_cast<const unsigned long long *>(obs) + (size_t)row * F;
    unsigned long long x = o[0] ^ (o[F / 3] * 3ull) ^ (o[(2 * F) / 3] * 5ull) ^ (o[F - 1] * 7ull) ^ ((unsigned long long)row << 17);
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33;
    actions[row] = (int32_t)(((x & 0xFFFFFFFFull) * n_actions) >> 32);
}

// ... the same policy for a consumer of the COMPACT observation (cz_step_device_compact): the four features come as table
// indices, the table turns them into the very doubles k_probe_policy reads - so both loops take the same actions
__global__ void k_probe_policy_codes(const uint8_t *__restrict__ codes, const double *__restrict__ table, int32_t *__restrict__ actions,
                                     int n_rows, int F, int Fp, uint32_t n_actions) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;               // (env, agent)
    if (row >= n_rows) return;
    const uint8_t *c = codes + (size_t)row * Fp;
    const unsigned long long *t = reinterpret_cast<const unsigned long long *>(table);
    unsigned long long x = t[c[0]] ^ (t[c[F / 3]] * 3ull) ^ (t[c[(2 * F) / 3]] * 5ull) ^ (t[c[F - 1]] * 7ull) ^ ((unsigned long long)row << 17);
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33;
    actions[row] = (int32_t)(((x & 0xFFFFFFFFull) * n_actions) >> 32);
}

// cz_load_layouts with a smaller pool: how many resident records still point past the new pool (layout id or redraw slice)
__global__ void k_layout_misfits(const uint32_t *__restrict__ state, int RW, int N, uint32_t n_new, unsigned long long *count) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const uint32_t *rec = state + (size_t)e * RW;
    const uint32_t base = rec[W_POOL] & 0xFFFFu, cnt = rec[W_POOL] >> 16;
    if (rec[W_LAYOUT] >= n_new || (cnt && base + cnt > n_new)) atomicAdd(count, 1ull);
}

// cz_set_spawn across the 31-step boundary: the countdown fields of the status word change their width (spawn_grace_bits); the
// resident records are re-packed so that no env reads another agent's bits (a countdown that does not fit the narrower field is
// clamped to its maximum, 31 - the new period is at most 31 then, so the agent is protected for at most one period)
__global__ void k_repack_grace(uint32_t *state, int RW, int N, int n_agents, uint32_t old_bits, uint32_t new_bits) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    uint32_t *rec = state + (size_t)e * RW;
    const uint32_t st = rec[W_STATUS];
    uint32_t out = st & ((1u << SPAWN_GRACE0) - 1u);
    for (int a = 0; a < n_agents; ++a) {
        const uint32_t g = (st >> (SPAWN_GRACE0 + old_bits * a)) & ((1u << old_bits) - 1u);
        out |= min(g, (1u << new_bits) - 1u) << (SPAWN_GRACE0 + new_bits * a);
    }
    rec[W_STATUS] = out;
}

__global__ void k_count_aborted(uint32_t *su, const uint32_t *__restrict__ state, int RW, long long env_begin, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *rec = state + (size_t)(env_begin + i) * RW;
    if (!(rec[W_STATUS] & ST_DONE)) su[(size_t)(env_begin + i) * SU_WORDS + SU_STEPS] += rec[W_T];
}

// ======================================================================================================
// host: handle + C-ABI
// ======================================================================================================

struct cz_handle_s {
    cz_config cfg;
    Params P;
    hipStream_t stream = nullptr;      // the stream every call of this handle is ordered on
    hipStream_t own_stream = nullptr;  // the one cz_create made (cz_set_stream may point `stream` at a caller's)
    Launchers kl;
    int n_layouts = 0, n_recipes = 0;
    uint32_t *d_state = nullptr, *d_lay_init = nullptr, *d_lay_desc = nullptr, *d_recipes = nullptr;
    uint32_t *d_lay_block = nullptr;       // the allocation behind d_lay_init: LAY_CTL_WORDS control words, then the records
    char *h_lay_stage = nullptr;           // pinned staging of cz_update_layouts: [L] records, then [L] descriptors
    hipStream_t copy_stream = nullptr;     // cz_update_layouts copies run here, beside the steps
    hipEvent_t ev_copy_done = nullptr, ev_steps_issued = nullptr;
    bool copy_pending = false;             // a copy has been issued that no cz_set_layout_group has waited for yet
    struct SlotRange { int32_t first, count; };
    std::vector<SlotRange> upd_ranges;     // staged by cz_update_layouts, not copied yet (flush_updates)
    std::vector<SlotRange> staged_on_copy, staged_on_main;   // slots of copies issued on the copy stream / in order on the handle's stream whose completion nobody has seen yet
    hipEvent_t ev_main_copy_done = nullptr;
    int32_t lay_groups = 1, lay_active = 0;
    int64_t n_layout_updates = 0;
    void *d_spawn_tables = nullptr;        // cz_set_spawn: exhausted-respawn counter | spawn areas per (level, agent) | level of every layout
    int spawn_layouts = 0;                 // the pool size those tables were made for
    uint32_t spawn_bits = 5;               // width of the countdown fields the resident records are packed with (spawn_grace_bits)
    uint32_t *d_stat_u = nullptr;
    double *d_stat_f = nullptr;
    cz_stats *d_stats_out = nullptr;
    unsigned long long *d_stats_part = nullptr;   // [256 chains][16 columns] between the two stages of the reduction
    double *d_lut = nullptr;
    double obs_table[LUT_SIZE];        // host copy of the quotient table (cz_obs_table: what the compact observation's codes index)
    int32_t *d_reset_words = nullptr;  // [3][N]: layout ids, recipe words, pool words of a cz_reset call
    void *d_codes_stage = nullptr;     // cz_step_compact with pageable host memory: device staging of the codes
    void *d_dump = nullptr;            // [N][4] doubles: where a one-step launch writes an output array the caller passed as NULL
    // staging for the host-pointer API
    int32_t *d_actions = nullptr;
    double *d_obs = nullptr;
    char *d_small = nullptr, *h_small = nullptr;   // rewards | terminations | truncations | marks: one device block, one pinned block
    uint32_t *h_marks = nullptr, *d_marks_mapped = nullptr;   // pinned, device-mapped marks buffer of the direct path
    uint32_t *marks_out_next = nullptr;            // (cz_step hands the kernel a marks buffer for one launch)
    std::vector<uint32_t> last_marks;          // recipe marks after the most recent cz_step (cz_last_marks)
    char *h_stage = nullptr, *d_stage = nullptr;   // small batches: pinned, device-mapped staging block of cz_step
    // cz_step_device_ring: which ring / output buffers / tables the cached graphs were captured for
    struct RingKey {
        const int32_t *ring = nullptr; int64_t stride = 0; int32_t period = 0;
        double *obs = nullptr, *rew = nullptr; uint8_t *term = nullptr, *trunc = nullptr; hipStream_t stream = nullptr; uint64_t version = 0;
        bool operator==(const RingKey &o) const {
            return ring == o.ring && stride == o.stride && period == o.period && obs == o.obs && rew == o.rew && term == o.term &&
                   trunc == o.trunc && stream == o.stream && version == o.version;
        }
    } ring_key;
    struct RunGraph { int32_t slot, len; hipGraphExec_t ge; uint64_t used; };
    std::vector<RunGraph> ring_graphs;             // replayable runs of this ring, keyed by (first slot, length)
    uint64_t ring_clock = 0;
    int64_t n_graph_kernels = 0, n_direct_kernels = 0;
    uint64_t tables_version = 0;                   // bumped whenever something the launches capture by value changes
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
    int wt_override = -1;          // CZ_WT experiment switch, read once
    bool huge = false;             // the 256-slot / 1024-cell instance (its own LDS image layout)
    unsigned long long *tl_base = nullptr;   // timeline build: stamp buffer, its capacity in launches, launches so far
    int32_t tl_cap = 0;
    int64_t tl_count = 0;
    bool graphs_enabled = true;    // CZ_GRAPHS=0: cz_step_device_ring launches everything directly
    bool ring_fused = false;       // cz_set_ring_fused: runs of cz_step_device_ring / _many go out as fused launches (outputs in place)
    int64_t n_ring_fused_steps = 0;
    int32_t graph_min_run = 48;    // CZ_GRAPH_MIN_RUN: shorter pieces of a ring run are launched directly (see ring_walk)
    int32_t ring_prefix = 0;       // CZ_RING_PREFIX: steps of a cz_step_device_ring call launched directly in front of its first graph
    size_t zero_copy_bytes = (size_t)256 << 10;   // cz_step: batches whose buffers fit use the pinned device-mapped block (CZ_ZERO_COPY_BYTES)
    cz_stats *d_gather = nullptr;
    std::string err;
};

static thread_local std::string g_err;
static int ready(cz_handle h);
static int set_device(cz_handle h);
static int flush_updates(cz_handle h, bool force);

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

// Is the stream this handle's work goes to (a stream of the caller, cz_set_stream) being captured - hipStreamBeginCapture,
// torch.cuda.graph - right now?  The device-pointer steps (cz_step_device, _compact, _many, _ring, cz_rollout*) are then pure
// kernel launches: nothing that queries or synchronises (a staged layout update stays staged until the first call outside the
// capture; no graphs of the library's own inside the caller's), so the capture stays valid and replays do what the launches did.
static bool caller_capturing(cz_handle h) {
    if (h->stream == h->own_stream) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

extern "C" int32_t cz_stream_capturing(cz_handle h) { return h && caller_capturing(h) ? 1 : 0; }

extern "C" int32_t cz_abi_version(void) { return CZ_ABI_VERSION; }
// diagnostic builds only (tools/phase_profile.py): where the kernels write their s_memtime stamps
extern "C" int cz_debug_set_stamps(cz_handle h, void *d_buf) {
    if (!h) return 1;
#ifdef CZ_PROFILE
    h->P.stamps = (unsigned long long *)d_buf;
#else
    (void)d_buf;
    return fail(h, "cz_debug_set_stamps: this is not the diagnostic build (make prof)");
#endif
    return 0;
}
// timeline builds only (tools/timeline.py): launch k of the step kernel writes its waves' entry / exit stamps to
// d_buf + (k % cap_launches) * num_envs * 2 (uint64); cap_launches = 0 switches it off
extern "C" int cz_debug_set_timeline(cz_handle h, void *d_buf, int32_t cap_launches) {
    if (!h) return 1;
#ifdef CZ_TIMELINE
    h->tl_base = (unsigned long long *)d_buf;
    h->tl_cap = d_buf ? cap_launches : 0;
    h->tl_count = 0;
    return 0;
#else
    (void)d_buf; (void)cap_launches;
    return fail(h, "cz_debug_set_timeline: this is not the timeline build (make timeline)");
#endif
}
extern "C" int32_t cz_sizeof_config(void) { return (int32_t)sizeof(cz_config); }
extern "C" int32_t cz_sizeof_stats(void) { return (int32_t)sizeof(cz_stats); }
extern "C" uint32_t cz_action(uint64_t seed, int64_t env_global, int32_t agent, uint32_t step, uint32_t n_actions) {
    return action_hash(seed, env_global, agent, step, n_actions);
}
extern "C" uint32_t cz_next_layout(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t n_layouts) {
    return next_layout(env_global, episode, pool_word, n_layouts);
}
extern "C" uint32_t cz_next_layout_group(int64_t env_global, uint32_t episode, uint32_t pool_word, uint32_t n_layouts, uint32_t groups,
                                         uint32_t active) {
    return next_layout(env_global, episode, pool_word, n_layouts, groups ? groups : 1u, active);
}

extern "C" int cz_create(const cz_config *cfg, cz_handle *out) {
    if (!cfg || !out) return fail(nullptr, "cz_create: null argument");
    *out = nullptr;
    const int C = cfg->width * cfg->height;
    if (cfg->num_envs < 1) return fail(nullptr, "cz_create: num_envs must be >= 1");
    if (cfg->num_agents < 1 || cfg->num_agents > MAX_AGENTS) return fail(nullptr, "cz_create: num_agents must be 1..4");
    if (cfg->num_recipes < cfg->num_agents || cfg->num_recipes > MAX_AGENTS)
        return fail(nullptr, "cz_create: need num_agents <= num_recipes <= 4 (one recipe per agent, cooking_env.py:329)");
    // (the quotient table holds 2W-1 x-entries from index 0 and 2H-1 y-entries from index 64, below the constants at 126/127)
    if (cfg->width < 1 || cfg->height < 1 || cfg->width > 32 || cfg->height > 32)
        return fail(nullptr, "cz_create: grid %dx%d unsupported (W <= 32, H <= 32)", cfg->width, cfg->height);
    static_assert(2 * 32 - 1 <= LUT_Y0 && LUT_Y0 + 2 * 32 - 1 <= LUT_ZERO, "quotient table layout");
    if (cfg->max_dyn < 1 || cfg->max_dyn > 255) return fail(nullptr, "cz_create: max_dyn must be 1..255 (slot + 1 is an 8-bit field)");
    if (cfg->action_scheme != 1 && cfg->action_scheme != 3)
        return fail(nullptr, "cz_create: action_scheme must be 1 or 3 (scheme2 is unusable in the reference)");
    if (cfg->max_steps < 1 || cfg->feat_len < 1) return fail(nullptr, "cz_create: max_steps and feat_len must be positive");
    {   // the kernels address a handle's records and one-step outputs with 32-bit offsets
        const uint64_t rw = ((uint64_t)(20 + (C + 3) / 4 + 2 * cfg->max_dyn) + 15) / 16 * 16;
        if ((uint64_t)cfg->num_envs * rw * 4 >= (1ull << 32) || (uint64_t)cfg->num_envs * cfg->num_agents * 8 >= (1ull << 32))
            return fail(nullptr, "cz_create: num_envs too large for one handle (records must stay below 4 GiB): use several handles");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, "cz_create: no HIP device available (this library has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, "cz_create: device_id %d out of range", cfg->device_id);
    cz_handle h = new cz_handle_s();
    h->cfg = *cfg;
    // from here on a failure must release what was already created
#define CREATE_CHK(call)                                                                          \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            fail(nullptr, "cz_create: %s failed: %s", #call, hipGetErrorString(_e));             \
            cz_destroy(h);                                                                        \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
    CREATE_CHK(hipSetDevice(cfg->device_id));
    CREATE_CHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    CREATE_CHK(hipEventCreate(&h->ev0));
    CREATE_CHK(hipEventCreate(&h->ev1));
    Params &P = h->P;
    memset(&P, 0, sizeof P);
    P.N = cfg->num_envs; P.A = cfg->num_agents; P.W = cfg->width; P.H = cfg->height; P.D = cfg->max_dyn; P.F = cfg->feat_len;
    const int CW = (C + 3) / 4;
    P.dyn0_off = CELL_WORD0 + CW;   // CELL_WORD0 = 8 header + 4 agents + 8 words of running returns
    P.dyn1_off = P.dyn0_off + P.D;
    P.RW = (P.dyn1_off + P.D + 15) / 16 * 16;
    P.scheme = cfg->action_scheme; P.max_steps = cfg->max_steps; P.end_all = cfg->end_condition_all ? 1 : 0;
    P.R = cfg->num_recipes; P.auto_reset = cfg->auto_reset ? 1 : 0; P.env_id_base = cfg->env_id_base;
    P.recipe_reward = cfg->recipe_reward; P.recipe_penalty = cfg->recipe_penalty; P.node_reward = cfg->recipe_node_reward;
    P.time_penalty_step = cfg->max_time_penalty / (double)cfg->max_steps;       // cooking_env.py:307
    P.T = 1;
    if (const char *s = getenv("CZ_WT")) h->wt_override = atoi(s);
    if (const char *s = getenv("CZ_GRAPHS")) h->graphs_enabled = atoi(s) != 0;
    if (const char *s = getenv("CZ_GRAPH_MIN_RUN")) h->graph_min_run = atoi(s) < 4 ? 4 : (atoi(s) > 256 ? 256 : atoi(s));   // 4..RING_MAX_GRAPH
    if (const char *s = getenv("CZ_RING_PREFIX")) h->ring_prefix = atoi(s) < 0 ? 0 : (atoi(s) > 16 ? 16 : atoi(s));
    if (const char *s = getenv("CZ_ZERO_COPY_BYTES")) h->zero_copy_bytes = (size_t)atoll(s);
#ifdef CZ_SMALL_ONLY       // diagnostic libraries that carry the small instance only
    if (!(P.D <= 64 && C <= 64)) { fail(nullptr, "cz_create: this diagnostic library holds the small kernel instance only"); cz_destroy(h); return 1; }
    h->kl = launchers_small();
#else
    h->kl = (P.D <= 64 && C <= 64) ? launchers_small() : (P.D <= 128 && C <= 256) ? launchers_large() : launchers_huge();
#endif
    h->huge = P.D > 128 || C > 256;
    {   // the reward of a step on which no recipe node changed: cooking_env.py:304-307 with zero deltas, same op order
        double x = 0.0;
        x += (double)0 * P.node_reward;
        x += 0.0 * P.recipe_reward;
        x += 0.0 * P.recipe_penalty;
        x += P.time_penalty_step;
        P.reward_idle = x;
    }
    const size_t N = (size_t)P.N;
    const size_t state_bytes = N * P.RW * 4;
    CREATE_CHK(hipMalloc(&h->d_state, state_bytes));
    CREATE_CHK(hipMemsetAsync(h->d_state, 0, state_bytes, h->stream));
    CREATE_CHK(hipMalloc(&h->d_stat_u, N * SU_WORDS * 4));
    CREATE_CHK(hipMalloc(&h->d_stat_f, N * SF_WORDS * 8));
    CREATE_CHK(hipMemsetAsync(h->d_stat_u, 0, N * SU_WORDS * 4, h->stream));
    CREATE_CHK(hipMemsetAsync(h->d_stat_f, 0, N * SF_WORDS * 8, h->stream));
    CREATE_CHK(hipMalloc(&h->d_stats_out, sizeof(cz_stats)));
    CREATE_CHK(hipMalloc(&h->d_dump, N * MAX_AGENTS * sizeof(double)));
    CREATE_CHK(hipMalloc(&h->d_reset_words, N * 3 * sizeof(int32_t)));
    CREATE_CHK(hipMalloc(&h->d_stats_part, (size_t)STAT_CHAINS * 16 * sizeof(unsigned long long)));
    CREATE_CHK(hipStreamSynchronize(h->stream));
    P.state = h->d_state; P.stat_u = h->d_stat_u; P.stat_f = h->d_stat_f;
    {   // observation quotients: (x - ax) / W and (y - ay) / H for every possible difference, computed here with the same
        // IEEE-754 double division Python's int / int true division performs (cooking_env.py:364-368)
        double lut[LUT_SIZE];
        for (int i = 0; i < LUT_SIZE; ++i) lut[i] = 0.0;
        for (int i = 0; i < 2 * P.W - 1; ++i) lut[i] = (double)(i - (P.W - 1)) / (double)P.W;
        for (int i = 0; i < 2 * P.H - 1; ++i) lut[LUT_Y0 + i] = (double)(i - (P.H - 1)) / (double)P.H;
        lut[LUT_ZERO] = 0.0; lut[LUT_ONE] = 1.0;
        // flag truth tables (cz_device.h LUT_*): world_objects.py feature vectors of ChopFood / BlenderFood (int(not done()),
        // chop_state, blend_state), Switch / Block (switch_active / int(walkable)), Agent (orientation one-hot)
        for (int st = 0; st < 4; ++st) {
            lut[LUT_NDONE0 + st] = st == 0 ? 1.0 : 0.0;
            lut[LUT_CH0 + st] = (st & 1) ? 1.0 : 0.0;
            lut[LUT_MA0 + st] = (st & 2) ? 1.0 : 0.0;
            lut[LUT_CF0 + st] = st != 0 ? 1.0 : 0.0;
        }
        for (int k = 1; k <= 4; ++k) lut[LUT_OR0 + 8 * (k - 1) + k] = 1.0;
        // ... followed by one word per lane for the observer-relative features (cz_kernels.h observe): lane 16 a + code of the
        // subtrahend table holds observer a's x (mask in the low half), y (high half) or nothing.  Axis codes (soa.py):
        // 1 x, 2 y, 4 + 2 j x unless j == a, 5 + 2 j y unless j == a (an agent's own position is absolute, cooking_env.py:366-368)
        uint32_t submask[64];
        for (int l = 0; l < 64; ++l) {
            const int a = l >> 4, code = l & 15;
            const bool selx = code == 1 || (code >= 4 && !(code & 1) && (code - 4) / 2 != a);
            const bool sely = code == 2 || (code >= 5 && (code & 1) && (code - 5) / 2 != a);
            submask[l] = (selx ? 0x7F8u : 0u) | (sely ? 0x7F8u << 16 : 0u);
        }
        // ... followed by the despawn / respawn parameters (SpawnCfg, cz_set_spawn; all zero: switched off)
        static_assert(SPAWN_CFG_OFFSET == sizeof lut + sizeof submask, "layout of the block behind Params::lut");
        // ... the image words of the cells' coordinates (init_lds: x + W - 1 and 63 + y + H - 1 as byte offsets into this table,
        // one word per cell) and, per lane, which word of which of the env's recipe rows it holds (load_recipe_rows)
        uint32_t coords[1024], rowsel[64];
        const uint32_t c01 = (uint32_t)((P.W - 1) * 8) | ((uint32_t)((LUT_Y0 + P.H - 1) * 8) << 16);
        for (uint32_t c = 0; c < 1024; ++c) {
            const uint32_t y = c / (uint32_t)P.W, x = c - y * (uint32_t)P.W;
            coords[c] = (y < 64u) ? ((x << 3) | (y << 19)) + c01 : 0u;
        }
        for (uint32_t l = 0; l < 64; ++l) {
            const uint32_t lc = l < 9u * (uint32_t)P.R ? l : 9u * (uint32_t)P.R - 1u;       // lanes past the rows repeat the last word
            rowsel[l] = (8u * (lc / 9u)) | ((4u * (lc % 9u)) << 8);
        }
        static_assert(LUT_BLOCK_BYTES == sizeof lut + sizeof submask + sizeof(SpawnCfg) + sizeof coords + sizeof rowsel, "layout of the block behind Params::lut");
        CREATE_CHK(hipMalloc(&h->d_lut, LUT_BLOCK_BYTES));
        CREATE_CHK(hipMemset(h->d_lut, 0, LUT_BLOCK_BYTES));
        CREATE_CHK(hipMemcpy((char *)h->d_lut + COORD_TABLE_OFFSET, coords, sizeof coords, hipMemcpyHostToDevice));
        CREATE_CHK(hipMemcpy((char *)h->d_lut + ROWSEL_TABLE_OFFSET, rowsel, sizeof rowsel, hipMemcpyHostToDevice));
        CREATE_CHK(hipMemcpy(h->d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
        CREATE_CHK(hipMemcpy((char *)h->d_lut + sizeof lut, submask, sizeof submask, hipMemcpyHostToDevice));
        P.lut = h->d_lut;
        memcpy(h->obs_table, lut, sizeof lut);
    }
#undef CREATE_CHK
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
    void *ptrs[] = {h->d_codes_stage, h->d_spawn_tables, h->d_reset_words, h->d_dump, h->d_lut, h->d_state, h->d_lay_block, h->d_lay_desc, h->d_recipes, h->d_stat_u, h->d_stat_f, h->d_stats_out, h->d_stats_part,
                    h->d_actions, h->d_obs, h->d_small, h->d_gather};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &r : h->ring_graphs)
        if (r.ge) (void)hipGraphExecDestroy(r.ge);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->ev_copy_done) (void)hipEventDestroy(h->ev_copy_done);
    if (h->ev_steps_issued) (void)hipEventDestroy(h->ev_steps_issued);
    if (h->ev_main_copy_done) (void)hipEventDestroy(h->ev_main_copy_done);
    if (h->h_lay_stage) (void)hipHostFree(h->h_lay_stage);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_small) (void)hipHostFree(h->h_small);
    if (h->h_marks) (void)hipHostFree(h->h_marks);
    for (hipEvent_t e : h->kev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return 0;
}

// Order this handle's work on a stream of the caller (e.g. the current stream of the framework that produces the
// actions and consumes the observations), so that no host synchronisation is needed in between.  NULL = back to the
// handle's own stream.  Work already queued on the previous stream is waited for first.
extern "C" int cz_set_stream(cz_handle h, void *hip_stream) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->tables_version++;
    return 0;
}
// Agent despawn / respawn for every world of the batch, on the device (cooking_world.py:267-290).  rates 0 / 0 switch it off.
// The spawn areas come from the level files (their AGENTS entries, parsing.py:118-151), so they are per LEVEL: level_of_layout
// says which level every layout of the resident pool instantiates (NULL: all of them level 0), spawn_x / spawn_y hold per
// (level, agent) `stride` candidate coordinates of which the first n_x / n_y count.
extern "C" int cz_set_spawn(cz_handle h, double despawn_rate, double respawn_rate, int32_t grace_period, uint64_t seed,
                            int32_t n_levels, const uint8_t *level_of_layout, int32_t stride, const uint8_t *spawn_x, const int32_t *n_x,
                            const uint8_t *spawn_y, const int32_t *n_y) {
    if (!h) return fail(nullptr, "null handle");
    const bool on = despawn_rate > 0.0 || respawn_rate > 0.0;
    if (despawn_rate < 0.0 || respawn_rate < 0.0 || grace_period < 0 || (uint32_t)grace_period > spawn_max_grace(h->P.A))
        return fail(h, "cz_set_spawn: rates must be >= 0 and 0 <= grace_period <= %u for %d agent(s) (20 bits of the record's status word hold the "
                       "countdowns of all agents)", spawn_max_grace(h->P.A), h->P.A);
    SpawnCfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.seed = seed; cfg.despawn_rate = despawn_rate; cfg.respawn_rate = respawn_rate; cfg.grace_period = (uint32_t)grace_period | (spawn_grace_bits((uint32_t)grace_period, h->P.A) << 24);
    std::vector<uint8_t> areas, levels;
    if (on) {
        if (!h->P.lay_init) return fail(h, "cz_set_spawn: load the layout pool first (the spawn areas are looked up by layout)");
        if (!spawn_x || !n_x || !spawn_y || !n_y || n_levels < 1 || n_levels > 255 || stride < 1 || stride > 1024)
            return fail(h, "cz_set_spawn: spawn areas missing, or not 1..255 levels with 1..1024 candidates per list");
        const size_t blk = 4 + 2 * (size_t)stride;
        areas.assign((size_t)n_levels * MAX_AGENTS * blk, 0);
        for (int l = 0; l < n_levels; ++l)
            for (int a = 0; a < h->P.A; ++a) {
                const int nx = n_x[l * h->P.A + a], ny = n_y[l * h->P.A + a];
                if (nx < 1 || nx > stride || ny < 1 || ny > stride)
                    return fail(h, "cz_set_spawn: level %d, agent %d: 1..%d x and y candidates", l, a, stride);
                uint8_t *b = areas.data() + ((size_t)l * MAX_AGENTS + a) * blk;
                b[0] = (uint8_t)(nx & 255); b[1] = (uint8_t)(nx >> 8); b[2] = (uint8_t)(ny & 255); b[3] = (uint8_t)(ny >> 8);
                memcpy(b + 4, spawn_x + ((size_t)l * h->P.A + a) * stride, (size_t)stride);
                memcpy(b + 4 + stride, spawn_y + ((size_t)l * h->P.A + a) * stride, (size_t)stride);
            }
        levels.assign((size_t)h->n_layouts, 0);
        for (int i = 0; i < h->n_layouts; ++i) {
            const int lv = level_of_layout ? level_of_layout[i] : 0;
            if (lv >= n_levels) return fail(h, "cz_set_spawn: layout %d is of level %d, but only %d level(s) have spawn areas", i, lv, n_levels);
            levels[(size_t)i] = (uint8_t)lv;
        }
    }
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_spawn_tables) { HIPCHK(h, hipFree(h->d_spawn_tables)); h->d_spawn_tables = nullptr; }
    h->spawn_layouts = 0;
    const uint32_t new_bits = spawn_grace_bits((uint32_t)grace_period, h->P.A);
    if (new_bits != h->spawn_bits) {       // episodes in flight keep their countdowns: re-pack them to the new field width
        hipLaunchKernelGGL(k_repack_grace, dim3((unsigned)((h->P.N + 255) / 256)), dim3(256), 0, h->stream, h->d_state, h->P.RW, (int)h->P.N,
                           h->P.A, h->spawn_bits, new_bits);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->spawn_bits = new_bits;
    }
    if (on) {
        // one allocation: the counter word, the areas, the level of every layout
        const size_t off_areas = 16, off_levels = off_areas + ((areas.size() + 15) & ~(size_t)15);
        HIPCHK(h, hipMalloc(&h->d_spawn_tables, off_levels + levels.size() + 16));
        HIPCHK(h, hipMemset(h->d_spawn_tables, 0, off_levels + levels.size() + 16));
        HIPCHK(h, hipMemcpy((char *)h->d_spawn_tables + off_areas, areas.data(), areas.size(), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy((char *)h->d_spawn_tables + off_levels, levels.data(), levels.size(), hipMemcpyHostToDevice));
        cfg.stride = (uint32_t)stride;
        cfg.exhausted = (uint32_t *)h->d_spawn_tables;
        cfg.areas = (const uint8_t *)h->d_spawn_tables + off_areas;
        cfg.level_of_layout = (const uint8_t *)h->d_spawn_tables + off_levels;
        h->spawn_layouts = h->n_layouts;
    }
    HIPCHK(h, hipMemcpy((char *)h->d_lut + SPAWN_CFG_OFFSET, &cfg, sizeof cfg, hipMemcpyHostToDevice));
    h->P.auto_reset = (h->P.auto_reset & 1) | (on ? 2 : 0);
    h->tables_version++;
    return 0;
}
// respawns that found no free cell of their spawn area within generate_location's 1001 tries (parsing.py:154-167 raises
// ValueError there; the device leaves the agent where it stood) since cz_set_spawn; -1 on error
extern "C" int64_t cz_spawn_exhausted(cz_handle h) {
    if (!h) return -1;
    if (!h->d_spawn_tables) return 0;
    uint32_t n = 0;
    if (hipSetDevice(h->cfg.device_id) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(&n, h->d_spawn_tables, 4, hipMemcpyDeviceToHost) != hipSuccess) { fail(h, "cz_spawn_exhausted: copy failed"); return -1; }
    return (int64_t)n;
}
// the draw the device takes (host mirror, for parity tests): uniform in [0, 1), keyed by (seed, global env id, episode << 32 | t,
// agent, draw index)
extern "C" double cz_spawn_uniform(uint64_t seed, int64_t env_global, uint32_t episode, uint32_t t, int32_t agent, uint32_t draw) {
    return spawn_uniform(seed, (uint64_t)env_global, ((uint64_t)episode << 32) | (uint64_t)t, (uint32_t)agent, draw);
}
extern "C" int32_t cz_record_words(cz_handle h) { return h ? h->P.RW : 0; }
extern "C" int cz_sync(cz_handle h) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// device encoding of one node (layout documented at Ops::recipe_marks in cz_device.h); `children` = 0 for wide rows
static int encode_node(cz_handle h, int recipe, uint32_t j, uint32_t hw, uint32_t children, uint32_t *out) {
    const uint32_t cls = hw & 0xFF, cond = (hw >> 8) & 0xFF, counts = (hw >> 24) & 1;
    uint32_t w = children | (counts << 8);
    if (cls < 16) {
        w |= 0x200u | ((cls <= BLENDER ? cls : 7u) << 10);                  // unknown static class: matches nothing
    } else if (cls < 32) {
        // states: 0 fresh, 1 chopped, 2 mashed, 3 both.  Only Carrot / Banana have a blend_state (the reference
        // raises AttributeError when a blend condition meets another class; here such a node matches nothing)
        const bool blender_food = (cls - 16) == CARROT || (cls - 16) == BANANA;
        uint32_t acc;
        if (cond & 0x10u) acc = cond & 0xFu;                               // explicit accept mask (several conditions)
        else if (cond == COND_NONE) acc = 0xFu;
        else if (cond == COND_CHOPPED) acc = 0xAu;
        else if (cond == COND_NOT_CHOPPED) acc = 0x5u;
        else if (cond == COND_MASHED) acc = blender_food ? 0xCu : 0u;
        else if (cond == COND_NOT_MASHED) acc = blender_food ? 0x3u : 0u;
        else return fail(h, "cz_load_recipes: recipe %d node %u: unknown condition code %u", recipe, j, cond);
        w |= 0x2000u | ((cls - 16) << 16) | (acc << 24);
    }
    // (any other class id, e.g. 255 for a name without objects such as "Agent": neither flag, matches nothing)
    *out = w;
    return 0;
}

extern "C" int cz_load_recipes(cz_handle h, const uint32_t *table, int32_t n, int32_t max_nodes) {
    if (!h || !table || n < 1 || n > 255 || (max_nodes != MAX_NODES && max_nodes != WIDE_NODES)) return fail(h, "cz_load_recipes: bad arguments");
    const bool wide = max_nodes == WIDE_NODES;
    const size_t row_words = wide ? 1 + 2 * WIDE_NODES : 1 + MAX_NODES;
    for (int i = 0; i < n; ++i)
        if (table[(size_t)i * row_words] > (uint32_t)max_nodes) return fail(h, "cz_load_recipes: recipe %d has more than %d nodes", i, max_nodes);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (h->d_recipes) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(h->d_recipes)); h->d_recipes = nullptr; h->P.recipes = nullptr; }
    size_t bytes = (size_t)n * row_words * 4;
    HIPCHK(h, hipMalloc(&h->d_recipes, bytes));
    // device copy: word 0 = node count; node words are re-encoded for the kernels: conditions become an accept mask over
    // the state index chopped | mashed << 1
    std::vector<uint32_t> dev(table, table + (size_t)n * row_words);
    for (int i = 0; i < n; ++i) {
        uint32_t *row = dev.data() + (size_t)i * row_words;
        const uint32_t nn = row[0];
        for (uint32_t j = 0; j < (uint32_t)max_nodes; ++j) {
            if (wide) {
                if (j >= nn) { row[1 + 2 * j] = row[2 + 2 * j] = 0; continue; }
                if (encode_node(h, i, j, row[1 + 2 * j], 0, &row[1 + 2 * j])) return 1;
                row[2 + 2 * j] &= 0xFFFFu;
            } else {
                if (j >= nn) { row[1 + j] = 0; continue; }
                if (encode_node(h, i, j, row[1 + j], (row[1 + j] >> 16) & 0xFF, &row[1 + j])) return 1;
            }
        }
        row[0] = nn & 0xFFu;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_recipes, dev.data(), bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_recipes = n;
    h->P.recipes = h->d_recipes;
    h->tables_version++;
    // Carrying an object from cell to cell changes no co-location (hence no recipe mark) when at most two agents
    // exist (they can never share a cell, cooking_world.py:206-221) and no recipe relates a dynamic-class node to a
    // node of a walkable static class (Floor, Switch, Block) -- the only statics a carried object can be "at".
    int sensitive = h->P.A > 2;
    for (int i = 0; i < n && !sensitive; ++i) {
        const uint32_t *row = table + (size_t)i * row_words;
        auto node = [&](uint32_t j) { return wide ? row[1 + 2 * j] : row[1 + j]; };
        auto kids = [&](uint32_t j) { return wide ? (row[2 + 2 * j] & 0xFFFFu) : ((row[1 + j] >> 16) & 0xFFu); };
        for (uint32_t j = 0; j < row[0] && !sensitive; ++j) {
            uint32_t cls = node(j) & 0xFF, children = kids(j);
            for (uint32_t c = 0; c < row[0]; ++c) {
                if (!((children >> c) & 1)) continue;
                uint32_t ccls = node(c) & 0xFF;
                bool pw = cls == FLOOR || cls == SWITCH || cls == BLOCK, cw = ccls == FLOOR || ccls == SWITCH || ccls == BLOCK;
                bool pd = cls >= 16 && cls < 32, cd = ccls >= 16 && ccls < 32;
                if ((pw && cd) || (pd && cw)) sensitive = 1;
            }
        }
    }
    h->P.walk_touches = sensitive;
    h->P.wide = wide ? 1 : 0;
    return 0;
}

// what the kernels rely on in a layout's tables: descriptors whose halfword indices address a slot / cell / agent of this batch
// and whose axis codes exist; records in which a slot that is not alive carries no container tag (Ops::content_of)
static int validate_layouts(cz_handle h, const char *who, const uint32_t *init_records, const uint32_t *obs_desc, int32_t n) {
    for (size_t i = 0; i < (size_t)n * h->P.F; ++i) {
        uint32_t off = obs_desc[i] & 0xFFFFu, code4 = obs_desc[i] >> 16;
        uint32_t hw = off >> 1, code = code4 >> 2;
        bool ok = !(off & 1u) && !(code4 & 3u) && (code <= 2 || (code >= 4 && code < 4 + 2 * (uint32_t)h->P.A));
        const uint32_t cell0 = h->huge ? Img<16>::CELL0 : Img<1>::CELL0, ag0 = h->huge ? Img<16>::AG0 : Img<1>::AG0;
        const uint32_t zero = h->huge ? Img<16>::ZERO : Img<1>::ZERO;
        if (hw < cell0) ok = ok && (int)(hw / 6) < h->P.D;
        else if (hw < ag0) ok = ok && (int)((hw - cell0) / 4) < h->P.W * h->P.H;
        else if (hw < zero) ok = ok && (int)((hw - ag0) / 8) < h->P.A && ((hw - ag0) & 7u) < 7u;
        else ok = ok && hw == zero;
        if (!ok) return fail(h, "%s: bad observation descriptor %#x at %zu", who, obs_desc[i], i);
    }
    for (int32_t l = 0; l < n; ++l) {
        const uint32_t *r = init_records + (size_t)l * h->P.RW;
        for (int s2 = 0; s2 < h->P.D; ++s2)
            if (!(r[h->P.dyn0_off + s2] & D_ALIVE) && (r[h->P.dyn1_off + s2] & 0xFFu))
                return fail(h, "%s: layout %d: slot %d is not alive but carries a container tag", who, l, s2);
    }
    return 0;
}

extern "C" int cz_load_layouts(cz_handle h, const uint32_t *init_records, const uint32_t *obs_desc, int32_t n) {
    if (!h || !init_records || !obs_desc || n < 1 || n > 65535) return fail(h, "cz_load_layouts: bad arguments");
    if (validate_layouts(h, "cz_load_layouts", init_records, obs_desc, n)) return 1;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->copy_stream) HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    if (h->n_layouts > 0 && n < h->n_layouts) {
        // a smaller pool: the kernels index lay_desc / lay_init with the layout id and the redraw slice of every resident
        // record, so records that point past the new pool would read out of bounds on the next step / observe / auto-reset
        unsigned long long misfits = 0;
        HIPCHK(h, hipMemsetAsync(h->d_stats_part, 0, sizeof misfits, h->stream));
        hipLaunchKernelGGL(k_layout_misfits, dim3((unsigned)((h->P.N + 255) / 256)), dim3(256), 0, h->stream, h->d_state, h->P.RW, h->P.N,
                           (uint32_t)n, h->d_stats_part);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(&misfits, h->d_stats_part, sizeof misfits, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (misfits)
            return fail(h, "cz_load_layouts: %llu resident env record(s) refer to layouts >= %d of the pool being replaced (%d layouts); "
                           "cz_reset / cz_set_state them into the new range first, or load a pool that is not smaller",
                        misfits, n, h->n_layouts);
    }
    if (h->d_lay_block) { HIPCHK(h, hipFree(h->d_lay_block)); h->d_lay_block = nullptr; h->d_lay_init = nullptr; h->P.lay_init = nullptr; }
    if (h->d_lay_desc) { HIPCHK(h, hipFree(h->d_lay_desc)); h->d_lay_desc = nullptr; h->P.lay_desc = nullptr; }
    if (h->h_lay_stage) { HIPCHK(h, hipHostFree(h->h_lay_stage)); h->h_lay_stage = nullptr; }
    size_t b0 = (size_t)n * h->P.RW * 4, b1 = (size_t)n * h->P.F * 4;
    // the records, behind LAY_CTL_WORDS control words (which part of their pool slices the envs draw from: all of it)
    HIPCHK(h, hipMalloc(&h->d_lay_block, LAY_CTL_WORDS * 4 + b0));
    h->d_lay_init = h->d_lay_block + LAY_CTL_WORDS;
    HIPCHK(h, hipMalloc(&h->d_lay_desc, b1 + 16));                 // + padding: descriptors are fetched in pairs
    HIPCHK(h, hipMemsetAsync(h->d_lay_block, 0, LAY_CTL_WORDS * 4, h->stream));
    HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)(h->d_lay_block + LC_GROUPS), 1, 1, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_lay_desc, 0, b1 + 16, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_lay_init, init_records, b0, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_lay_desc, obs_desc, b1, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_layouts = n;
    // (the spawn tables map every layout of the pool to its level: a new pool - even one of the same size - needs cz_set_spawn again)
    if (h->P.auto_reset & 2) h->spawn_layouts = -1;
    h->lay_groups = 1; h->lay_active = 0; h->upd_ranges.clear(); h->staged_on_copy.clear(); h->staged_on_main.clear(); h->copy_pending = false;
    h->tables_version++;
    h->P.lay_init = h->d_lay_init; h->P.lay_desc = h->d_lay_desc; h->P.L = n;
    return 0;
}

// the staged slot ranges of cz_update_layouts that still wait for their copy.  force = false: copy them (on the copy stream)
// if the steps that were issued before the update have completed, else leave them for a later call; force = true: they have
// to be in place before what the caller issues next - if those steps are still running, copy in order on the handle's stream.
static int flush_updates(cz_handle h, bool force) {
    if (h->upd_ranges.empty()) return 0;
    const hipError_t q = hipEventQuery(h->ev_steps_issued);
    if (q != hipSuccess) (void)hipGetLastError();
    if (q != hipSuccess && !force) return 0;
    const hipStream_t st = q == hipSuccess ? h->copy_stream : h->stream;
    const size_t RWb = (size_t)h->P.RW * 4, Fb = (size_t)h->P.F * 4, L = (size_t)h->n_layouts;
    for (const auto &r : h->upd_ranges) {
        HIPCHK(h, hipMemcpyAsync(h->d_lay_init + (size_t)r.first * h->P.RW, h->h_lay_stage + (size_t)r.first * RWb, (size_t)r.count * RWb,
                                 hipMemcpyHostToDevice, st));
        HIPCHK(h, hipMemcpyAsync(h->d_lay_desc + (size_t)r.first * h->P.F, h->h_lay_stage + L * RWb + (size_t)r.first * Fb, (size_t)r.count * Fb,
                                 hipMemcpyHostToDevice, st));
    }
    // (who reads which part of the staging block until when: cz_update_layouts waits before it rewrites the same slots)
    // (the events are re-recorded behind ALL copies issued so far on their stream, so the lists only grow until somebody has
    // waited for the event - cz_update_layouts before it rewrites staged slots - or found it complete)
    if (st == h->copy_stream) {
        if (!h->staged_on_copy.empty() && hipEventQuery(h->ev_copy_done) == hipSuccess) h->staged_on_copy.clear();
        (void)hipGetLastError();
        HIPCHK(h, hipEventRecord(h->ev_copy_done, h->copy_stream));
        h->copy_pending = true;
        h->staged_on_copy.insert(h->staged_on_copy.end(), h->upd_ranges.begin(), h->upd_ranges.end());
    } else {
        if (!h->staged_on_main.empty() && hipEventQuery(h->ev_main_copy_done) == hipSuccess) h->staged_on_main.clear();
        (void)hipGetLastError();
        HIPCHK(h, hipEventRecord(h->ev_main_copy_done, h->stream));
        h->staged_on_main.insert(h->staged_on_main.end(), h->upd_ranges.begin(), h->upd_ranges.end());
    }
    h->upd_ranges.clear();
    return 0;
}

// Fresh layouts UNDER A STEPPING BATCH (the reference instantiates a new level at every reset, cooking_env.py:191-195 /
// parsing.py:21-151; the device redraws from a resident pool): replaces pool slots [first, first + count) without touching
// the handle's stream.  The tables stay where they are (nothing the launches captured changes), the new content goes
// through a pinned staging block of the library (the caller's arrays are free again when this returns) and is copied by a
// stream of its own, AFTER every step issued so far has completed (their envs may still read the old content) and not
// waited for by later steps.  The caller must not replace slots that envs can still draw or are still playing on - that is what
// cz_set_layout_group is for: cut every env's pool slice into groups, let the envs draw from one, refresh another once the
// episodes that started on it are over (at most max_steps + 1 steps after the switch), then switch.  The switch waits for
// the copies (on the device, not on the host), so an env only ever draws a slot whose update has completed.
extern "C" int cz_update_layouts(cz_handle h, int32_t first, int32_t count, const uint32_t *init_records, const uint32_t *obs_desc) {
    if (ready(h)) return 1;
    if (first < 0 || count < 0 || first + count > h->n_layouts || (count > 0 && (!init_records || !obs_desc)))
        return fail(h, "cz_update_layouts: slots [%d, %d) outside the resident pool of %d layouts", first, first + count, h ? h->n_layouts : 0);
    if (count > 0 && validate_layouts(h, "cz_update_layouts", init_records, obs_desc, count)) return 1;
    if (set_device(h)) return 1;
    if (caller_capturing(h)) return fail(h, "cz_update_layouts: not inside a stream capture of the caller (it records an event on the captured stream)");
    const size_t RWb = (size_t)h->P.RW * 4, Fb = (size_t)h->P.F * 4, L = (size_t)h->n_layouts;
    if (!h->copy_stream) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_copy_done, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_steps_issued, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_main_copy_done, hipEventDisableTiming));
    }
    if (!h->h_lay_stage) HIPCHK(h, hipHostMalloc((void **)&h->h_lay_stage, L * (RWb + Fb), hipHostMallocDefault));
    if (count == 0) return 0;               // (count 0: only sets up the copy stream and the staging block, ahead of time)
    // The staging block has a place per slot.  Only an earlier copy of THESE slots that is still running, or not even issued,
    // is in the way (with the rotation of cz_set_layout_group consecutive updates go to different parts of the pool).
    auto overlaps = [&](const std::vector<cz_handle_s::SlotRange> &v) {
        for (const auto &r : v)
            if (r.first < first + count && first < r.first + r.count) return true;
        return false;
    };
    if (overlaps(h->upd_ranges) && flush_updates(h, true)) return 1;
    if (overlaps(h->staged_on_copy)) { HIPCHK(h, hipEventSynchronize(h->ev_copy_done)); h->staged_on_copy.clear(); }
    if (overlaps(h->staged_on_main)) { HIPCHK(h, hipEventSynchronize(h->ev_main_copy_done)); h->staged_on_main.clear(); }
    char *const st_rec = h->h_lay_stage + (size_t)first * RWb, *const st_desc = h->h_lay_stage + L * RWb + (size_t)first * Fb;
    memcpy(st_rec, init_records, (size_t)count * RWb);
    memcpy(st_desc, obs_desc, (size_t)count * Fb);
    // The copy must come after every step issued so far (their envs may still read the old content).  A device-side wait of the
    // copy stream for the handle's stream would do - and slow every step kernel down by a microsecond for as long as it is
    // pending (a second hardware queue with an outstanding barrier: measured in round 3, profiles/r03/rot_probe.txt).  So the copy is issued
    // LATER, by whichever call of this handle first finds that those steps have completed (flush_updates), with nothing to
    // wait for on the device.
    HIPCHK(h, hipEventRecord(h->ev_steps_issued, h->stream));
    h->upd_ranges.push_back({first, count});
    h->n_layout_updates += count;
    return flush_updates(h, false);
}

// From the next step on (stream order) every env draws its next episode's layout from part `active` of its pool slice cut
// into `groups` equal parts (1, 0: the whole slice, the default).  Every slice's length must be a multiple of `groups`.
// Waits - on the device - for the cz_update_layouts copies issued so far.
extern "C" int cz_set_layout_group(cz_handle h, int32_t groups, int32_t active) {
    if (ready(h)) return 1;
    if (groups < 1 || groups > h->n_layouts || active < 0 || active >= groups || h->n_layouts % groups)
        return fail(h, "cz_set_layout_group: need 1 <= groups, 0 <= active < groups, and pool slices that are multiples of groups");
    if (set_device(h)) return 1;
    if (caller_capturing(h)) return fail(h, "cz_set_layout_group: not inside a stream capture of the caller (it orders copies and waits)");
    if (flush_updates(h, true)) return 1;
    if (h->copy_pending) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_copy_done, 0)); h->copy_pending = false; }
    HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)(h->d_lay_block + LC_GROUPS), groups, 1, h->stream));
    HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)(h->d_lay_block + LC_ACTIVE), active, 1, h->stream));
    h->lay_groups = groups; h->lay_active = active;
    return 0;
}
// how many pool slots cz_update_layouts has replaced on this handle so far
extern "C" int64_t cz_layout_updates(cz_handle h) { return h ? h->n_layout_updates : 0; }

static int check_range(cz_handle h, int64_t b, int64_t c) {
    if (!h) return fail(nullptr, "null handle");
    if (b < 0 || c < 0 || b + c > h->P.N) return fail(h, "env range [%lld, %lld) outside [0, %d)", (long long)b, (long long)(b + c), h->P.N);
    return 0;
}

extern "C" int cz_set_state(cz_handle h, int64_t b, int64_t c, const uint32_t *records) {
    if (check_range(h, b, c)) return 1;
    if (c == 0) return 0;
    if (!records) return fail(h, "cz_set_state: null buffer");
    // the words the kernels use as table indices must be in range (everything else lives in registers / LDS)
    for (int64_t i = 0; i < c; ++i) {
        const uint32_t *r = records + (size_t)i * h->P.RW;
        if (h->n_layouts > 0 && r[W_LAYOUT] >= (uint32_t)h->n_layouts) return fail(h, "cz_set_state: record %lld: layout id %u out of range", (long long)i, r[W_LAYOUT]);
        for (int k = 0; k < h->P.R; ++k)
            if (h->n_recipes > 0 && ((r[W_RECIPES] >> (8 * k)) & 0xFFu) >= (uint32_t)h->n_recipes)
                return fail(h, "cz_set_state: record %lld: recipe id %u out of range", (long long)i, (r[W_RECIPES] >> (8 * k)) & 0xFFu);
        const uint32_t base = r[W_POOL] & 0xFFFFu, count = r[W_POOL] >> 16;
        if (count && h->n_layouts > 0 && (int)(base + count) > h->n_layouts) return fail(h, "cz_set_state: record %lld: layout pool slice out of range", (long long)i);
        // a slot that is not alive carries no container tag: the kernels take "tagged" to imply "alive" (Ops::content_of)
        for (int s2 = 0; s2 < h->P.D; ++s2)
            if (!(r[h->P.dyn0_off + s2] & D_ALIVE) && (r[h->P.dyn1_off + s2] & 0xFFu))
                return fail(h, "cz_set_state: record %lld: slot %d is not alive but carries a container tag", (long long)i, s2);
    }
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMemcpyAsync(h->d_state + (size_t)b * h->P.RW, records, (size_t)c * h->P.RW * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cz_get_state(cz_handle h, int64_t b, int64_t c, uint32_t *records) {
    if (check_range(h, b, c)) return 1;
    if (c == 0) return 0;
    if (!records) return fail(h, "cz_get_state: null buffer");
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

static int launch_step(cz_handle h, Params &P, hipStream_t stream = nullptr, bool fused = false) {
    if ((P.auto_reset & 2) && h->spawn_layouts != h->n_layouts)
        return fail(h, "despawn / respawn is on, but the layout pool was reloaded since cz_set_spawn (now %d layouts): call cz_set_spawn "
                       "again (it maps every layout to the level whose spawn areas it uses)", h->n_layouts);
    if (!stream) stream = h->stream;
    // write-through observation stores pay when the launch is short enough for the end-of-kernel L2 write-back to be
    // exposed: one step of a batch of up to ~10 000 envs, whatever its rows weigh (round 6, profiles/r06/wt_sizes.txt: the 2.2 KB
    // rows of the 7x7 levels cross over between 10 240 and 12 288 envs - 43 / 52 MiB -, the 6.7 KB rows of large_16x16 are still
    // ahead at 8192 envs = 210 MiB; until round 5 the rule was "up to 128 MiB", which cost the 7x7 levels 5-9 % from 12 288 to
    // 24 576 envs and large_16x16 2-7 % from 6144 to 8192); streaming stores when one launch's observations do not fit the
    // memory-side cache (256 MiB) any more.  (CZ_WT=0/1/2 overrides, for experiments.)
    const size_t obs_bytes = (size_t)P.N * P.A * P.F * 8;
    if (!fused) P.wt = obs_bytes > ((size_t)224 << 20) ? 2 : P.N <= 10240 ? 1 : 0;
    // (fused: streaming stores only when an agent's row fills whole DRAM pages - 4 KiB and more, config 5: +8 %; with the
    // 2.2 KB rows of the 7x7 levels they lose 15 % against the cache's own write-back order, profiles/r03/wt_ab2.txt)
    else P.wt = (obs_bytes > ((size_t)224 << 20) && (size_t)P.F * 8 >= 4096) ? 2 : 0;
    if (h->wt_override >= 0) P.wt = h->wt_override;
    if (!fused) {             // the one-step kernels store rewards and flags unconditionally
        if (!P.rewards) P.rewards = (double *)h->d_dump;
        if (!P.term) P.term = (uint8_t *)h->d_dump;
        if (!P.trunc) P.trunc = (uint8_t *)h->d_dump;
    }
#ifdef CZ_TIMELINE
    P.timeline = (h->tl_base && h->tl_cap > 0) ? h->tl_base + (size_t)(h->tl_count++ % h->tl_cap) * (size_t)P.N * 2 : nullptr;
#endif
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->ktime && caller_capturing(h)) return fail(h, "kernel timing (cz_kernel_time_reset) cannot run inside a stream capture of the caller");
    if (h->ktime) {
        while (h->kev.size() < h->kev_used + 2) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e));
            h->kev.push_back(e);
        }
        e0 = h->kev[h->kev_used]; e1 = h->kev[h->kev_used + 1];
        h->kev_used += 2;
        HIPCHK(h, hipEventRecord(e0, stream));
    }
    HIPCHK(h, h->kl.step(P, stream, fused ? LAUNCH_FUSED : LAUNCH_ONE));
    if (h->ktime) HIPCHK(h, hipEventRecord(e1, stream));
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
    // persistent scratch of the handle (cz_create: three words per env; the observation staging is shared with cz_step / cz_observe)
    int32_t *const d_lay = h->d_reset_words;
    uint32_t *const d_rec = (uint32_t *)h->d_reset_words + h->P.N, *const d_pool = (uint32_t *)h->d_reset_words + 2 * (size_t)h->P.N;
    HIPCHK(h, hipMemcpyAsync(d_lay, layout_ids, (size_t)c * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_rec, recipe_ids, (size_t)c * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_pool, pools.data(), (size_t)c * 4, hipMemcpyHostToDevice, h->stream));
    size_t ob = (size_t)c * h->P.A * h->P.F * 8;
    if (obs && !h->d_obs) HIPCHK(h, hipMalloc(&h->d_obs, (size_t)h->P.N * h->P.A * h->P.F * 8));
    Params P = h->P;
    hipLaunchKernelGGL(k_count_aborted, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, h->stream, h->d_stat_u, h->d_state, h->P.RW,
                       (long long)b, (int)c);
    hipError_t rc = h->kl.reset(P, h->stream, b, (int)c, d_lay, d_rec, d_pool, obs ? h->d_obs : nullptr);
    if (rc == hipSuccess && obs) rc = hipMemcpyAsync(obs, h->d_obs, ob, hipMemcpyDeviceToHost, h->stream);
    const hipError_t rs = hipStreamSynchronize(h->stream);            // (pools, a host vector, must outlive its copy)
    HIPCHK(h, rc);
    HIPCHK(h, rs);
    return 0;
}

// observe() of the current state (after cz_set_state), host buffer [count][A][F]
extern "C" int cz_observe(cz_handle h, int64_t b, int64_t c, double *obs) {
    if (ready(h) || check_range(h, b, c)) return 1;
    if (c == 0) return 0;
    if (!obs) return fail(h, "cz_observe: null buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    size_t ob = (size_t)c * h->P.A * h->P.F * 8;
    if (!h->d_obs) HIPCHK(h, hipMalloc(&h->d_obs, (size_t)h->P.N * h->P.A * h->P.F * 8));
    Params P = h->P;
    hipError_t rc = h->kl.observe(P, h->stream, b, (int)c, h->d_obs, nullptr);
    if (rc == hipSuccess) rc = hipMemcpyAsync(obs, h->d_obs, ob, hipMemcpyDeviceToHost, h->stream);
    const hipError_t rs = hipStreamSynchronize(h->stream);
    HIPCHK(h, rc);
    HIPCHK(h, rs);
    return 0;
}

// observe() of the current state into DEVICE buffers, float64 rows and / or the compact form (either may be NULL): what a consumer
// that stays on the device needs after cz_reset / cz_set_state, before the first step has written anything.  Stream-ordered,
// no synchronisation (legal inside a stream capture of the caller).
extern "C" int cz_observe_device(cz_handle h, int64_t b, int64_t c, double *d_obs, uint8_t *d_codes) {
    if (ready(h) || check_range(h, b, c)) return 1;
    if (c == 0) return 0;
    if (!d_obs && !d_codes) return fail(h, "cz_observe_device: both output pointers are null");
    if (set_device(h)) return 1;
    Params P = h->P;
    P.wt = 0;
    HIPCHK(h, h->kl.observe(P, h->stream, b, (int)c, d_obs, d_codes));
    return 0;
}
// ... the compact form into a host buffer: codes uint8 [count][A][cz_codes_pitch] (cz_obs_table()[code] is the float64 feature)
extern "C" int cz_observe_compact(cz_handle h, int64_t b, int64_t c, uint8_t *codes) {
    if (ready(h) || check_range(h, b, c)) return 1;
    if (c == 0) return 0;
    if (!codes) return fail(h, "cz_observe_compact: null buffer");
    if (set_device(h)) return 1;
    const size_t row = (size_t)h->P.A * codes_pitch(h->P.F);
    if (!h->d_codes_stage) HIPCHK(h, hipMalloc(&h->d_codes_stage, (size_t)h->P.N * row));
    Params P = h->P;
    P.wt = 0;
    hipError_t rc = h->kl.observe(P, h->stream, b, (int)c, nullptr, (uint8_t *)h->d_codes_stage);
    if (rc == hipSuccess) rc = hipMemcpyAsync(codes, h->d_codes_stage, (size_t)c * row, hipMemcpyDeviceToHost, h->stream);
    const hipError_t rs = hipStreamSynchronize(h->stream);
    HIPCHK(h, rc);
    HIPCHK(h, rs);
    return 0;
}

static int set_device(cz_handle h) {
    // hipSetDevice is not free, hipGetDevice is: switch only when the calling thread is on another GPU (the caller, or
    // another handle, may have changed the thread's current device since the last call)
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->cfg.device_id) HIPCHK(h, hipSetDevice(h->cfg.device_id));
    return 0;
}

extern "C" int cz_step_device(cz_handle h, const int32_t *d_actions, double *d_obs, double *d_rewards, uint8_t *d_term,
                              uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (!d_actions) return fail(h, "cz_step_device: actions pointer is null");
    if (set_device(h)) return 1;
    if (!h->upd_ranges.empty() && !caller_capturing(h) && flush_updates(h, false)) return 1;      // a staged layout update whose time may have come
    Params P = h->P;
    P.actions = d_actions; P.obs = d_obs; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc; P.T = 1;
    P.marks_out = h->marks_out_next; h->marks_out_next = nullptr;
    return launch_step(h, P);
}

// The step with the observation as one byte per feature (and, optionally, the float64 one beside it): see observe() in cz_kernels.h.
extern "C" int cz_step_device_compact(cz_handle h, const int32_t *d_actions, uint8_t *d_codes, double *d_obs, double *d_rewards,
                                      uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (!d_actions || !d_codes) return fail(h, "cz_step_device_compact: actions and codes pointers must not be null");
    if (set_device(h)) return 1;
    if (!h->upd_ranges.empty() && !caller_capturing(h) && flush_updates(h, false)) return 1;
    Params P = h->P;
    P.actions = d_actions; P.obs = d_obs; P.codes = d_codes; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc; P.T = 1;
    return launch_step(h, P);
}
// From now on EVERY one-step launch of the handle (cz_step_device, _many, _ring, cz_step) also writes the compact observation
// to d_codes (uint8 [N][A][pitch]); NULL switches it off.  With d_obs = NULL in those calls the launches write codes only.
extern "C" int cz_set_compact_output(cz_handle h, uint8_t *d_codes) {
    if (!h) return fail(nullptr, "null handle");
    h->P.codes = d_codes;
    h->tables_version++;                  // (graphs captured for the ring carry the pointer)
    return 0;
}
extern "C" int32_t cz_codes_pitch(cz_handle h) { return h ? codes_pitch(h->P.F) : 0; }
extern "C" int cz_obs_table(cz_handle h, double *table) {
    if (!h || !table) return fail(h, "cz_obs_table: null argument");
    memcpy(table, h->obs_table, sizeof h->obs_table);
    return 0;
}
extern "C" const void *cz_obs_table_device(cz_handle h) { return h ? (const void *)h->d_lut : nullptr; }

static bool ring_fusable(cz_handle h, const Params &P, int32_t K, int64_t stride);
static int launch_ring_fused(cz_handle h, Params &P, int32_t K, const int32_t *d_ring, int64_t stride, int32_t period, int32_t first_slot);
// K consecutive steps, one launch each, issued from C: step k reads actions d_actions + k * action_stride (int32 units,
// wrapping every `action_period` steps) and overwrites the same output buffers.  Same work as K cz_step_device calls
// without K trips through the host language.
extern "C" int cz_step_device_many(cz_handle h, int32_t K, const int32_t *d_actions, int64_t action_stride, int32_t action_period,
                                   double *d_obs, double *d_rewards, uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (!d_actions || K < 1 || action_period < 1) return fail(h, "cz_step_device_many: bad arguments");
    if (set_device(h)) return 1;
    if (!h->upd_ranges.empty() && !caller_capturing(h) && flush_updates(h, false)) return 1;      // a staged layout update whose time may have come
    Params P = h->P;
    P.obs = d_obs; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc; P.T = 1;
    P.actions = d_actions;
    if (ring_fusable(h, P, K, action_stride)) return launch_ring_fused(h, P, K, d_actions, action_stride, action_period, 0);   // cz_set_ring_fused
    for (int32_t k = 0; k < K; ++k) {
        P.actions = d_actions + (int64_t)(k % action_period) * action_stride;
        if (launch_step(h, P)) return 1;
    }
    return 0;
}

// The same K launches for callers that keep their actions in a ring of `period` slots (slot s at d_ring + s * stride):
// step k reads slot (first_slot + k) % period.  A run of consecutive slots is captured once into a HIP graph, keyed by
// (first slot, length), and replayed afterwards: that takes the host out of the loop (0.02 us instead of ~2.5 us of CPU
// per launch) and the kernels read their arguments from memory that is not rewritten before every launch.  A run is cut
// where the ring wraps and at RING_MAX_GRAPH launches; pieces shorter than the handle's graph_min_run (48; CZ_GRAPH_MIN_RUN, clamped to
// 4..RING_MAX_GRAPH) are launched directly.  At
// most RING_CACHE graphs are kept (least recently used goes first), so a caller that keeps asking for new (slot, length)
// pairs pays a capture each time -- keep the runs of a loop aligned.  Results are identical to cz_step_device_many.
constexpr int RING_MAX_GRAPH = 256, RING_CACHE = 64;   // (rocprofv3 --kernel-trace aborts on replays of ~1000 kernel nodes: malformed AQL packet)
// captures the launches of slots [slot, slot + len) (nothing executes) and instantiates them
static int ring_capture(cz_handle h, Params &P, const int32_t *d_ring, int64_t stride, int32_t slot, int32_t len, hipGraphExec_t &ge) {
    hipGraph_t g = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int bad = 0;
    for (int s = 0; s < len && !bad; ++s) {
        P.actions = d_ring + (int64_t)(slot + s) * stride;
        bad = launch_step(h, P);
    }
    const hipError_t ec = hipStreamEndCapture(h->stream, &g);
    if (bad) { if (g) (void)hipGraphDestroy(g); return 1; }
    HIPCHK(h, ec);
    const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) { ge = nullptr; HIPCHK(h, ei); }
    return 0;
}
static bool ring_select(cz_handle h, const int32_t *d_ring, int64_t stride, int32_t period, double *d_obs, double *d_rewards,
                        uint8_t *d_term, uint8_t *d_trunc) {
    if (h->ktime || !h->graphs_enabled) return false;
    cz_handle_s::RingKey key;
    key.ring = d_ring; key.stride = stride; key.period = period; key.obs = d_obs; key.rew = d_rewards; key.term = d_term;
    key.trunc = d_trunc; key.stream = h->stream; key.version = h->tables_version;
    if (!(key == h->ring_key)) {                   // another ring, other output buffers or new tables: every graph is stale
        if (!h->ring_graphs.empty()) (void)hipStreamSynchronize(h->stream);
        for (auto &r : h->ring_graphs)
            if (r.ge) (void)hipGraphExecDestroy(r.ge);
        h->ring_graphs.clear();
        h->ring_key = key;
    }
    return true;
}
// the graph of run [slot, slot + len), captured on first use
static int ring_graph(cz_handle h, Params &P, const int32_t *d_ring, int64_t stride, int32_t slot, int32_t len, hipGraphExec_t &out) {
    h->ring_clock++;
    for (auto &r : h->ring_graphs)
        if (r.slot == slot && r.len == len) { r.used = h->ring_clock; out = r.ge; return 0; }
    if ((int)h->ring_graphs.size() >= RING_CACHE) {
        size_t victim = 0;
        for (size_t i = 1; i < h->ring_graphs.size(); ++i)
            if (h->ring_graphs[i].used < h->ring_graphs[victim].used) victim = i;
        HIPCHK(h, hipStreamSynchronize(h->stream));            // it may still be executing
        (void)hipGraphExecDestroy(h->ring_graphs[victim].ge);
        h->ring_graphs.erase(h->ring_graphs.begin() + (long)victim);
    }
    hipGraphExec_t ge = nullptr;
    if (ring_capture(h, P, d_ring, stride, slot, len, ge)) return 1;
    h->ring_graphs.push_back({slot, len, ge, h->ring_clock});
    out = ge;
    return 0;
}
// cz_set_ring_fused: the run as fused launches over the ring's own action rows (cz_rollout_actions' kernel, every step's outputs
// written in place), one launch per stretch of consecutive slots.  Needs densely packed slots (stride = num_envs * num_agents).
static bool ring_fusable(cz_handle h, const Params &P, int32_t K, int64_t stride) {
    return h->ring_fused && K >= 2 && !h->ktime && !P.codes && stride == (int64_t)P.N * P.A;
}
static int launch_ring_fused(cz_handle h, Params &P, int32_t K, const int32_t *d_ring, int64_t stride, int32_t period, int32_t first_slot) {
    // (the kernel addresses the action rows with 32-bit byte offsets from the first row of the launch)
    const int64_t max_T = (int64_t)(0xFFFFFFFFull / ((uint64_t)P.N * (uint64_t)P.A * 4ull));
    if (max_T < 1) return fail(h, "fused ring run: one action row of %d envs x %d agents exceeds the 4 GiB the kernel addresses", P.N, P.A);
    int32_t k = 0;
    while (k < K) {
        const int32_t slot = (int32_t)(((int64_t)first_slot + k) % period);
        int64_t run = K - k;
        if (run > period - slot) run = period - slot;
        if (run > max_T) run = max_T;
        Params Q = P;
        Q.actions = d_ring + (int64_t)slot * stride;
        Q.T = (int32_t)run; Q.seed = 0; Q.step0 = 1u;          // bit 0: outputs in place
        if (launch_step(h, Q, nullptr, true)) return 1;
        k += (int32_t)run;
    }
    h->n_ring_fused_steps += K;
    return 0;
}
// walks the run [first_slot, first_slot + K) piece by piece; launch = false only builds the graphs
static int ring_walk(cz_handle h, int32_t K, const int32_t *d_ring, int64_t stride, int32_t period, int32_t first_slot, double *d_obs,
                     double *d_rewards, uint8_t *d_term, uint8_t *d_trunc, bool launch) {
    Params P = h->P;
    P.obs = d_obs; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc; P.T = 1;
    P.actions = d_ring;
    if (ring_fusable(h, P, K, stride)) return launch ? launch_ring_fused(h, P, K, d_ring, stride, period, first_slot) : 0;
    // (inside a capture of the caller: plain launches - a capture of the library's own cannot nest in it)
    const bool graphs = !caller_capturing(h) && ring_select(h, d_ring, stride, period, d_obs, d_rewards, d_term, d_trunc);
    int32_t k = 0;
    while (k < K) {
        const int32_t slot = (int32_t)(((int64_t)first_slot + k) % period);
        int32_t run = K - k;
        if (run > period - slot) run = period - slot;
        if (run > RING_MAX_GRAPH) run = RING_MAX_GRAPH;
        // Short pieces go out as plain launches: a graph's first kernel starts ~10-16 us after hipGraphLaunch, a directly launched one
        // after ~3-5 us, and the host enqueues a launch (2.8 us) faster than the device runs it - a 20-step region takes 6.7 us per step
        // launched directly against 6.85 us replayed (profiles/r05/k20_modes.txt); long runs are replayed (no host work per launch).
        // Replaying a graph costs the host ~10-16 us before its first kernel starts; a directly launched kernel starts
        // after ~3-5 us.  So the first `ring_prefix` steps of a call's first piece go out as plain launches and keep the GPU busy
        // while the host submits the graph of the rest behind them.
        const int32_t pre = k == 0 ? h->ring_prefix : 0;
        if (graphs && run >= h->graph_min_run + pre) {
            hipGraphExec_t ge = nullptr;
            if (ring_graph(h, P, d_ring, stride, slot + pre, run - pre, ge)) return 1;
            if (launch) {
                for (int32_t j = 0; j < pre; ++j) {
                    P.actions = d_ring + (int64_t)(slot + j) * stride;
                    if (launch_step(h, P)) return 1;
                }
                HIPCHK(h, hipGraphLaunch(ge, h->stream));
                h->n_graph_kernels += run - pre;
                h->n_direct_kernels += pre;
            }
        } else if (launch) {
            for (int32_t j = 0; j < run; ++j) {
                P.actions = d_ring + (int64_t)(slot + j) * stride;
                if (launch_step(h, P)) return 1;
            }
            h->n_direct_kernels += run;
        }
        k += run;
    }
    return 0;
}
// Builds every graph cz_step_device_ring(K, ..., first_slot, ...) would build lazily, without stepping anything (so that a
// measurement does not pay the one-off capture inside its timed region).
extern "C" int cz_ring_prepare(cz_handle h, int32_t K, const int32_t *d_ring, int64_t stride, int32_t period, int32_t first_slot,
                               double *d_obs, double *d_rewards, uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (!d_ring || K < 1 || period < 1 || first_slot < 0 || first_slot >= period) return fail(h, "cz_ring_prepare: bad arguments");
    if (set_device(h)) return 1;
    return ring_walk(h, K, d_ring, stride, period, first_slot, d_obs, d_rewards, d_term, d_trunc, false);
}
extern "C" int cz_step_device_ring(cz_handle h, int32_t K, const int32_t *d_ring, int64_t stride, int32_t period, int32_t first_slot,
                                   double *d_obs, double *d_rewards, uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (!d_ring || K < 1 || period < 1 || first_slot < 0 || first_slot >= period) return fail(h, "cz_step_device_ring: bad arguments");
    if (set_device(h)) return 1;
    if (!h->upd_ranges.empty() && !caller_capturing(h) && flush_updates(h, false)) return 1;      // a staged layout update whose time may have come
    return ring_walk(h, K, d_ring, stride, period, first_slot, d_obs, d_rewards, d_term, d_trunc, true);
}
// Runs of cz_step_device_ring / cz_step_device_many as FUSED launches (opt-in): the K steps of a run whose action slots are densely
// packed go out as one launch per stretch of consecutive slots, the env state staying in registers across the steps and every
// step's outputs written to the caller's [N][A] buffers in place - the same final state, outputs and statistics as K launches,
// without launch boundaries.  Returns the previous setting.
extern "C" int cz_set_ring_fused(cz_handle h, int32_t enabled) {
    if (!h) { fail(nullptr, "null handle"); return -1; }
    const int was = h->ring_fused ? 1 : 0;
    h->ring_fused = enabled != 0;
    return was;
}
extern "C" int64_t cz_ring_fused_steps(cz_handle h, int32_t reset) {
    if (!h) return 0;
    const int64_t n = h->n_ring_fused_steps;
    if (reset) h->n_ring_fused_steps = 0;
    return n;
}
// how many step kernels of this handle were replayed from graphs / launched directly (cz_step_device_ring only);
// reset != 0 zeroes the counters after reading
extern "C" int cz_launch_counts(cz_handle h, int64_t *graph_kernels, int64_t *direct_kernels, int32_t reset) {
    if (!h) return fail(nullptr, "null handle");
    if (graph_kernels) *graph_kernels = h->n_graph_kernels;
    if (direct_kernels) *direct_kernels = h->n_direct_kernels;
    if (reset) h->n_graph_kernels = h->n_direct_kernels = 0;
    return 0;
}

extern "C" int cz_rollout(cz_handle h, int32_t T, uint64_t seed, uint32_t step0, double *d_obs, double *d_rewards,
                          uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (T < 1) return fail(h, "cz_rollout: T must be >= 1");
    // (the kernel addresses rewards / flags with 32-bit offsets from the array base)
    if ((uint64_t)T * (uint64_t)h->P.N * (uint64_t)h->P.A * 8ull > 0xFFFFFFFFull)
        return fail(h, "cz_rollout: T * num_envs * num_agents * 8 must stay below 4 GiB (T <= %llu here): split the rollout",
                    (unsigned long long)(0xFFFFFFFFull / ((uint64_t)h->P.N * h->P.A * 8ull)));
    if (set_device(h)) return 1;
    if (!h->upd_ranges.empty() && !caller_capturing(h) && flush_updates(h, false)) return 1;      // a staged layout update whose time may have come
    Params P = h->P;
    P.actions = nullptr; P.obs = d_obs; P.codes = nullptr; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc;
    P.T = T; P.seed = seed; P.step0 = step0;
    return launch_step(h, P, nullptr, true);
}

// ... with a COMPACT trajectory: d_codes uint8 [T][N][A][cz_codes_pitch] (one byte per feature, see cz_step_device_compact) instead
// of - or, with d_obs, next to - the float64 one: an eighth of the bytes a learner has to read back
extern "C" int cz_rollout_compact(cz_handle h, int32_t T, uint64_t seed, uint32_t step0, uint8_t *d_codes, double *d_obs, double *d_rewards,
                                  uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (T < 1 || !d_codes) return fail(h, "cz_rollout_compact: T must be >= 1 and the codes pointer non-null");
    if ((uint64_t)T * (uint64_t)h->P.N * (uint64_t)h->P.A * 8ull > 0xFFFFFFFFull)
        return fail(h, "cz_rollout_compact: T * num_envs * num_agents * 8 must stay below 4 GiB (T <= %llu here): split the rollout",
                    (unsigned long long)(0xFFFFFFFFull / ((uint64_t)h->P.N * h->P.A * 8ull)));
    if (set_device(h)) return 1;
    if (!h->upd_ranges.empty() && !caller_capturing(h) && flush_updates(h, false)) return 1;
    Params P = h->P;
    P.actions = nullptr; P.obs = d_obs; P.codes = d_codes; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc;
    P.T = T; P.seed = seed; P.step0 = step0;
    return launch_step(h, P, nullptr, true);
}

// The same fused launch over actions of the caller: d_actions int32 [T][N][A], step t of env e reads row t.  What replay and
// open-loop search use instead of T one-step launches: the record stays in registers, every step's outputs land in the
// trajectory buffers, and nothing has to be ordered between launches .
extern "C" int cz_rollout_actions(cz_handle h, int32_t T, const int32_t *d_actions, double *d_obs, double *d_rewards,
                                  uint8_t *d_term, uint8_t *d_trunc) {
    if (ready(h)) return 1;
    if (T < 1 || !d_actions) return fail(h, "cz_rollout_actions: T must be >= 1 and the actions pointer non-null");
    // (the kernel addresses rewards / flags / actions with 32-bit offsets from the array base)
    if ((uint64_t)T * (uint64_t)h->P.N * (uint64_t)h->P.A * 8ull > 0xFFFFFFFFull)
        return fail(h, "cz_rollout_actions: T * num_envs * num_agents * 8 must stay below 4 GiB (T <= %llu here): split the rollout",
                    (unsigned long long)(0xFFFFFFFFull / ((uint64_t)h->P.N * h->P.A * 8ull)));
    if (set_device(h)) return 1;
    if (!h->upd_ranges.empty() && !caller_capturing(h) && flush_updates(h, false)) return 1;      // a staged layout update whose time may have come
    Params P = h->P;
    P.actions = d_actions; P.obs = d_obs; P.codes = nullptr; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc;
    P.T = T; P.seed = 0; P.step0 = 0;
    return launch_step(h, P, nullptr, true);
}

// the device address of pinned, device-mapped host memory (cz_host_alloc, hipHostMalloc, hipHostRegister); nullptr for
// pageable memory
static void *mapped_device_pointer(const void *host) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host) != hipSuccess) {
        (void)hipGetLastError();                      // pageable memory: "invalid value", not an error of ours
        return nullptr;
    }
    return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
}

extern "C" int cz_step(cz_handle h, const int32_t *actions, double *obs, double *rewards, uint8_t *term, uint8_t *trunc) {
    if (ready(h)) return 1;
    if (!actions || !rewards || !term || !trunc) return fail(h, "cz_step: null buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t NA = (size_t)h->P.N * h->P.A;
    {   // host actions are checked here (the device-pointer entry points cannot look: their kernel reads action & 7)
        const int32_t n_actions = h->P.scheme == 3 ? 5 : 8;              // actions.py:17,50
        for (size_t i = 0; i < NA; ++i)
            if (actions[i] >= n_actions)
                return fail(h, "cz_step: action %d of env %zu, agent %zu is outside [0, %d) (negative = despawned)", actions[i],
                            i / (size_t)h->P.A, i % (size_t)h->P.A, n_actions);
    }
    const size_t ob = NA * h->P.F * 8;
    // Small batches (the single-env facade): the step is pure latency, so the kernel reads the actions from and writes
    // its outputs to one pinned, device-mapped host block -- one launch and one synchronisation, no copy commands.
    const size_t o_act = 0, o_rew = (NA * 4 + 15) & ~(size_t)15, o_term = o_rew + NA * 8, o_trunc = o_term + ((NA + 15) & ~(size_t)15),
                 o_marks = o_trunc + ((NA + 15) & ~(size_t)15), o_obs = o_marks + (((size_t)h->P.N * 8 + 15) & ~(size_t)15),
                 total = o_obs + ob;
    if (total <= h->zero_copy_bytes) {
        if (!h->h_stage) {
            HIPCHK(h, hipHostMalloc((void **)&h->h_stage, total, hipHostMallocMapped));
            HIPCHK(h, hipHostGetDevicePointer((void **)&h->d_stage, h->h_stage, 0));
        }
        memcpy(h->h_stage + o_act, actions, NA * 4);
        h->marks_out_next = (uint32_t *)(h->d_stage + o_marks);
        if (cz_step_device(h, (const int32_t *)(h->d_stage + o_act), obs ? (double *)(h->d_stage + o_obs) : nullptr,
                           (double *)(h->d_stage + o_rew), (uint8_t *)(h->d_stage + o_term), (uint8_t *)(h->d_stage + o_trunc))) {
            h->marks_out_next = nullptr;
            return 1;
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (obs) memcpy(obs, h->h_stage + o_obs, ob);
        memcpy(rewards, h->h_stage + o_rew, NA * 8);
        memcpy(term, h->h_stage + o_term, NA);
        memcpy(trunc, h->h_stage + o_trunc, NA);
        h->last_marks.assign((const uint32_t *)(h->h_stage + o_marks), (const uint32_t *)(h->h_stage + o_marks) + 2 * (size_t)h->P.N);
        return 0;
    }
    // Buffers from cz_host_alloc (or any pinned, device-mapped host memory): the kernel reads the actions from and writes
    // every output to them directly -- the observation bytes cross PCIe while the launch is still computing, and no copy
    // command is queued at all.
    {
        void *d_act = mapped_device_pointer(actions), *d_rew = mapped_device_pointer(rewards), *d_term = mapped_device_pointer(term),
             *d_trunc = mapped_device_pointer(trunc), *d_obs = obs ? mapped_device_pointer(obs) : nullptr;
        if (d_act && d_rew && d_term && d_trunc && (!obs || d_obs)) {
            if (!h->h_marks) {
                HIPCHK(h, hipHostMalloc((void **)&h->h_marks, (size_t)h->P.N * 8, hipHostMallocMapped));
                HIPCHK(h, hipHostGetDevicePointer((void **)&h->d_marks_mapped, h->h_marks, 0));
            }
            h->marks_out_next = h->d_marks_mapped;
            if (cz_step_device(h, (const int32_t *)d_act, (double *)d_obs, (double *)d_rew, (uint8_t *)d_term, (uint8_t *)d_trunc)) {
                h->marks_out_next = nullptr;
                return 1;
            }
            HIPCHK(h, hipStreamSynchronize(h->stream));
            h->last_marks.assign(h->h_marks, h->h_marks + 2 * (size_t)h->P.N);
            return 0;
        }
    }
    // Pageable buffers: staged copies.  The small outputs (rewards, flags, marks) live in one device block and come back
    // with one copy command through one pinned block; the observation goes straight into the caller's array.
    const size_t s_rew = 0, s_term = NA * 8, s_trunc = s_term + ((NA + 15) & ~(size_t)15), s_marks = s_trunc + ((NA + 15) & ~(size_t)15),
                 s_total = s_marks + (size_t)h->P.N * 8;
    if (!h->d_actions) {
        HIPCHK(h, hipMalloc(&h->d_actions, NA * 4));
        HIPCHK(h, hipMalloc(&h->d_small, s_total));
        HIPCHK(h, hipHostMalloc((void **)&h->h_small, s_total, hipHostMallocDefault));
    }
    if (obs && !h->d_obs) HIPCHK(h, hipMalloc(&h->d_obs, ob));
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, NA * 4, hipMemcpyHostToDevice, h->stream));
    h->marks_out_next = (uint32_t *)(h->d_small + s_marks);
    if (cz_step_device(h, h->d_actions, obs ? h->d_obs : nullptr, (double *)(h->d_small + s_rew), (uint8_t *)(h->d_small + s_term),
                       (uint8_t *)(h->d_small + s_trunc))) {
        h->marks_out_next = nullptr;
        return 1;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small, s_total, hipMemcpyDeviceToHost, h->stream));
    if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, ob, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(rewards, h->h_small + s_rew, NA * 8);
    memcpy(term, h->h_small + s_term, NA);
    memcpy(trunc, h->h_small + s_trunc, NA);
    h->last_marks.assign((const uint32_t *)(h->h_small + s_marks), (const uint32_t *)(h->h_small + s_marks) + 2 * (size_t)h->P.N);
    return 0;
}

// The host-pointer step with the COMPACT observation instead of the float64 rows: codes uint8 [N][A][cz_codes_pitch] (see
// cz_step_device_compact) - 1/8 of the bytes that have to cross PCIe, which is what a host-array step of thousands of envs
// spends its time on.  Buffers from cz_host_alloc are written by the kernel directly; others through a device staging block.
extern "C" int cz_step_compact(cz_handle h, const int32_t *actions, uint8_t *codes, double *rewards, uint8_t *term, uint8_t *trunc) {
    if (ready(h)) return 1;
    if (!codes) return fail(h, "cz_step_compact: null codes buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const size_t bytes = (size_t)h->P.N * h->P.A * (size_t)codes_pitch(h->P.F);
    uint8_t *const saved = h->P.codes;
    void *direct = mapped_device_pointer(codes);
    if (!direct && !h->d_codes_stage) HIPCHK(h, hipMalloc(&h->d_codes_stage, bytes));
    h->P.codes = direct ? (uint8_t *)direct : (uint8_t *)h->d_codes_stage;
    const int rc = cz_step(h, actions, nullptr, rewards, term, trunc);
    h->P.codes = saved;
    if (rc) return rc;
    if (!direct) {
        HIPCHK(h, hipMemcpyAsync(codes, h->d_codes_stage, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" int cz_last_marks(cz_handle h, uint32_t *out) {
    if (ready(h)) return 1;
    if (!out) return fail(h, "cz_last_marks: null buffer");
    if (h->last_marks.size() != 2 * (size_t)h->P.N) return fail(h, "cz_last_marks: no cz_step has run on this handle yet");
    memcpy(out, h->last_marks.data(), (size_t)h->P.N * 8);
    return 0;
}

// Measurement aid for bench.py: `reps` back-to-back launches of a kernel with the step kernel's grid shape that only
// writes `bytes` (<= 2 GiB) to d_dst; returns the average launch duration.  d_dst is overwritten with zeros.
extern "C" int cz_probe_output_only(cz_handle h, void *d_dst, size_t bytes, int32_t reps, float *us_per_launch) {
    if (ready(h)) return 1;
    if (!d_dst || !us_per_launch || reps < 1 || bytes < 16 || bytes > ((size_t)1 << 31)) return fail(h, "cz_probe_output_only: bad arguments");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const dim3 grid((unsigned)((h->P.N + ENVS_PER_WG - 1) / ENVS_PER_WG)), block(64 * ENVS_PER_WG);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_probe_fill, grid, block, 0, h->stream, d_dst, (uint32_t)bytes);
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_probe_fill, grid, block, 0, h->stream, d_dst, (uint32_t)bytes);
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *us_per_launch = ms * 1e3f / (float)reps;
    return 0;
}

// Measurement aid for bench.py: a CLOSED loop of K steps - cz_step_device, then a policy kernel that derives every agent's
// next action from the observation it has just been given, on the same stream - captured into one HIP graph and replayed
// `reps` times; returns the average time per step (step + policy).  d_actions (int32 [N][A]) is read and rewritten in place.
extern "C" int cz_probe_closed_loop(cz_handle h, int32_t K, int32_t reps, int32_t *d_actions, double *d_obs, double *d_rewards,
                                    uint8_t *d_term, uint8_t *d_trunc, float *us_per_step) {
    if (ready(h)) return 1;
    if (K < 1 || K > 1024 || reps < 1 || !d_actions || !d_obs || !us_per_step) return fail(h, "cz_probe_closed_loop: bad arguments");
    if (set_device(h)) return 1;
    Params P = h->P;
    P.actions = d_actions; P.obs = d_obs; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc; P.T = 1;
    const int rows = h->P.N * h->P.A;
    const uint32_t n_actions = h->P.scheme == 3 ? 5u : 8u;
    const bool was_timing = h->ktime;
    h->ktime = false;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int bad = 0;
    for (int k = 0; k < K && !bad; ++k) {
        bad = launch_step(h, P);
        if (!bad && !getenv("CZ_PROBE_NO_POLICY")) hipLaunchKernelGGL(k_probe_policy, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, d_obs, d_actions, rows, h->P.F, n_actions);
    }
    const hipError_t ec = hipStreamEndCapture(h->stream, &g);
    h->ktime = was_timing;
    if (bad) { if (g) (void)hipGraphDestroy(g); return 1; }
    HIPCHK(h, ec);
    const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIPCHK(h, ei);
    hipError_t rc = hipGraphLaunch(ge, h->stream);                         // warm
    if (rc == hipSuccess) rc = hipEventRecord(h->ev0, h->stream);
    for (int r = 0; r < reps && rc == hipSuccess; ++r) rc = hipGraphLaunch(ge, h->stream);
    if (rc == hipSuccess) rc = hipEventRecord(h->ev1, h->stream);
    if (rc == hipSuccess) rc = hipEventSynchronize(h->ev1);
    float ms = 0.f;
    if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, h->ev0, h->ev1);
    (void)hipGraphExecDestroy(ge);
    HIPCHK(h, rc);
    *us_per_step = ms * 1e3f / (float)((int64_t)reps * K);
    return 0;
}

// Test / measurement aid: ONE launch of that stand-in policy on the handle's stream - d_actions[env][agent] = hash of four features of
// the observation in d_obs (float64 rows) or, with d_obs NULL, in d_codes (compact rows; same actions).  What a caller's policy
// kernel is in tests/test_gpu_capture.py: [cz_probe_policy, cz_step_device] captured into a graph of the CALLER.
extern "C" int cz_probe_policy(cz_handle h, const double *d_obs, const uint8_t *d_codes, int32_t *d_actions) {
    if (ready(h)) return 1;
    if (!d_actions || (!d_obs && !d_codes)) return fail(h, "cz_probe_policy: bad arguments");
    if (set_device(h)) return 1;
    const int rows = h->P.N * h->P.A;
    const uint32_t n_actions = h->P.scheme == 3 ? 5u : 8u;
    if (d_obs) hipLaunchKernelGGL(k_probe_policy, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, d_obs, d_actions, rows, h->P.F, n_actions);
    else hipLaunchKernelGGL(k_probe_policy_codes, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, d_codes, h->d_lut, d_actions, rows,
                            h->P.F, codes_pitch(h->P.F), n_actions);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ... the same closed loop for a consumer of the compact observation: cz_step_device_compact (codes only, no float64 rows), then
// the policy kernel that reads the codes
extern "C" int cz_probe_closed_loop_compact(cz_handle h, int32_t K, int32_t reps, int32_t *d_actions, uint8_t *d_codes, double *d_rewards,
                                            uint8_t *d_term, uint8_t *d_trunc, float *us_per_step) {
    if (ready(h)) return 1;
    if (K < 1 || K > 1024 || reps < 1 || !d_actions || !d_codes || !us_per_step) return fail(h, "cz_probe_closed_loop_compact: bad arguments");
    if (set_device(h)) return 1;
    Params P = h->P;
    P.actions = d_actions; P.obs = nullptr; P.codes = d_codes; P.rewards = d_rewards; P.term = d_term; P.trunc = d_trunc; P.T = 1;
    const int rows = h->P.N * h->P.A;
    const uint32_t n_actions = h->P.scheme == 3 ? 5u : 8u;
    const bool was_timing = h->ktime;
    h->ktime = false;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int bad = 0;
    for (int k = 0; k < K && !bad; ++k) {
        bad = launch_step(h, P);
        if (!bad && !getenv("CZ_PROBE_NO_POLICY")) hipLaunchKernelGGL(k_probe_policy_codes, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, d_codes, h->d_lut, d_actions, rows,
                                     h->P.F, codes_pitch(h->P.F), n_actions);
    }
    const hipError_t ec = hipStreamEndCapture(h->stream, &g);
    h->ktime = was_timing;
    if (bad) { if (g) (void)hipGraphDestroy(g); return 1; }
    HIPCHK(h, ec);
    const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIPCHK(h, ei);
    hipError_t rc = hipGraphLaunch(ge, h->stream);                         // warm
    if (rc == hipSuccess) rc = hipEventRecord(h->ev0, h->stream);
    for (int r = 0; r < reps && rc == hipSuccess; ++r) rc = hipGraphLaunch(ge, h->stream);
    if (rc == hipSuccess) rc = hipEventRecord(h->ev1, h->stream);
    if (rc == hipSuccess) rc = hipEventSynchronize(h->ev1);
    float ms = 0.f;
    if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, h->ev0, h->ev1);
    (void)hipGraphExecDestroy(ge);
    HIPCHK(h, rc);
    *us_per_step = ms * 1e3f / (float)((int64_t)reps * K);
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
// Pinned (page-locked) host memory: buffers handed to cz_step / cz_reset / cz_observe from here are reached by the copy
// engines directly, without the runtime pinning or staging pageable pages on every call.
extern "C" void *cz_host_alloc(cz_handle h, size_t bytes) {
    if (!h) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(h->cfg.device_id) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        fail(h, "cz_host_alloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
extern "C" int cz_host_free(cz_handle h, void *p) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipHostFree(p));
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
static int stats_reduce(cz_handle h) {
    hipLaunchKernelGGL(k_stats_chains, dim3(STAT_CHAINS / 4), dim3(64), 0, h->stream, h->d_stat_u, h->d_stat_f, h->d_state, h->P.RW, h->P.N,
                       h->d_stats_part);
    hipLaunchKernelGGL(k_stats_tree, dim3(1), dim3(256), 0, h->stream, h->d_stats_part, h->d_stats_out);
    HIPCHK(h, hipGetLastError());
    return 0;
}
extern "C" int cz_get_stats(cz_handle h, cz_stats *out) {
    if (!h || !out) return fail(h, "cz_get_stats: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (stats_reduce(h)) return 1;
    HIPCHK(h, hipMemcpyAsync(out, h->d_stats_out, sizeof(cz_stats), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int cz_reset_stats(cz_handle h) {
    if (!h) return fail(nullptr, "null handle");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    hipLaunchKernelGGL(k_stats_clear, dim3((h->P.N + 255) / 256), dim3(256), 0, h->stream, h->d_stat_u, h->d_stat_f, h->d_state,
                       h->P.RW, h->P.N);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// RCCL is loaded lazily so that the library also loads on machines without it (CPU build check)
typedef struct { char internal[128]; } cz_nccl_id;
static void *rccl_lib(cz_handle h) {
    // One process holds one RCCL.  A copy that is already loaded (e.g. the librccl.so.1 a PyTorch wheel bundles, loaded
    // together with the HIP runtime that copy was built against) must be the one that is used: glibc matches loaded
    // objects by SONAME, so ask for the SONAME first and without loading anything (RTLD_NOLOAD), then load by SONAME,
    // then by the development name.
    static void *lib = nullptr;
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) fail(h, "cannot load librccl.so.1 / librccl.so: %s", dlerror());
    return lib;
}
// which files serve this process: the RCCL the communicator binds and the HIP runtime the library runs on (diagnostics
// for mixed-stack problems; bench.py prints both)
extern "C" int cz_runtime_paths(char *rccl_path, char *hip_path, size_t cap) {
    if (rccl_path && cap) {
        rccl_path[0] = 0;
        if (void *lib = rccl_lib(nullptr)) {
            Dl_info info;
            void *sym = dlsym(lib, "ncclGetUniqueId");
            if (sym && dladdr(sym, &info) && info.dli_fname) snprintf(rccl_path, cap, "%s", info.dli_fname);
        }
    }
    if (hip_path && cap) {
        hip_path[0] = 0;
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) snprintf(hip_path, cap, "%s", info.dli_fname);
    }
    return 0;
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
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(h, "cz_comm_init: bad arguments");
    if (h->comm) return fail(h, "cz_comm_init: this handle already has a communicator");
    h->rccl = rccl_lib(h);                       // (process-wide handle, never unloaded)
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
    if (stats_reduce(h)) return 1;
    typedef int (*fn_t)(const void *, void *, size_t, int, void *, hipStream_t);
    fn_t f = (fn_t)dlsym(h->rccl, "ncclAllGather");
    if (!f) return fail(h, "ncclAllGather not found");
    int r = f(h->d_stats_out, h->d_gather, sizeof(cz_stats), /*ncclInt8*/ 0, h->comm, h->stream);
    if (r) return fail(h, "ncclAllGather failed: %d", r);
    HIPCHK(h, hipMemcpyAsync(out, h->d_gather, sizeof(cz_stats) * (size_t)h->n_ranks, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// Barrier over the communicator: a 4-byte RCCL all-reduce on the handle's stream, then a stream synchronisation.  Every
// rank returns only after every rank's earlier work on its handle stream has finished (bench.py brackets its timed
// region with it).
extern "C" int cz_comm_barrier(cz_handle h) {
    if (!h) return fail(nullptr, "null handle");
    if (!h->comm) return fail(h, "cz_comm_barrier: communicator not initialised (cz_comm_init)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    typedef int (*fn_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
    fn_t f = (fn_t)dlsym(h->rccl, "ncclAllReduce");
    if (!f) return fail(h, "ncclAllReduce not found");
    // d_gather holds n_ranks cz_stats: its first word doubles as the barrier token (overwritten by the next all-gather)
    HIPCHK(h, hipMemsetAsync(h->d_gather, 0, 4, h->stream));
    int r = f(h->d_gather, h->d_gather, 1, /*ncclInt32*/ 2, /*ncclSum*/ 0, h->comm, h->stream);
    if (r) return fail(h, "ncclAllReduce failed: %d", r);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
