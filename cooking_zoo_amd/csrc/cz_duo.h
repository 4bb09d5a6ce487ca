// cz_duo.h -- the one-step kernel with TWO wavefronts per env (small instance: one object slot and one grid cell per lane).
//
// Why: at 4096 envs a boundary-ordered launch of k_step is "everybody fetches and computes for ~2 us, then everybody stores
// 18 MB at once" - compute and the store drain do not overlap inside a launch.  Here a HELPER wave per
// env takes the observation encode off the main wave's critical path and starts storing early:
//   * main wave M (waves 0..7 of a workgroup): record fetch, dynamics, rewards, flags, record write-back - k_step's path -
//     and the LATE feature pairs (those that depend on an object or on a mutable cell flag) of observers [0, ceil(NA / 2));
//   * helper wave H (waves 8..15): fetches descriptors / tables while M fetches the record, sleeps on an LDS flag until the
//     agents' positions are final (right after the walking half of the step), builds the agent part of the LDS image and the
//     per-observer subtrahend table, and encodes + stores every EARLY pair (static objects, agents: functions of the agents'
//     final positions only) of every observer while M still resolves interactions, recipes and rewards; after a second flag
//     it encodes the late pairs of the remaining observers.
// Which pairs are late is a mark in the descriptor words (DESC_LATE, set by the library when layouts are loaded).  Every pair
// is stored exactly once.  Hand-off: two LDS words per env, written by M after the data they publish (the DS operations of a
// wave execute in order, the LDS serves the waves of a workgroup one operation at a time), reset by H before the workgroup's
// table barrier.  Semantics: cooking_env.py:352-373 (get_feature_vector), identical outputs to k_step.
#pragma once

namespace cz {

struct DuoSync {
    uint32_t agw[MAX_AGENTS];      // the agents' final words of this pass (x | y << 8 | orientation << 16 | ...)
    uint32_t go1, go2;             // agents final / objects and cell flags final
    uint32_t layout, pad;          // the layout the pass ends on (a reset pass changes it: H then fetches the new descriptors)
};

constexpr int DUO_EPW = 8;         // envs per workgroup: 16 waves

// wait until *flag becomes non-zero (H only; M always gets there: both waves of an env are resident in the same workgroup)
__device__ __forceinline__ void duo_wait(volatile uint32_t *flag) {
    while (rfl(*flag) == 0u) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
}

// the pairs of this lane for observers [A0, A1): LATE selects which pairs are stored (the others leave the range of their
// buffer resource: dropped by the memory pipeline)
template <int NA, int A0, int A1, int MODE /* 0: early pairs, 1: late pairs, 2: all */>
__device__ __forceinline__ void duo_encode(const Params &P, Lds<1> &s, const double *lut, uint32_t layout, uint32_t (&dsc)[OBS_CHUNK], int lane,
                                           double *__restrict__ out /* [A][F] of this env */) {
    if constexpr (A0 < A1) {
        decltype(__builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0)) rs[NA];
#pragma unroll
        for (int a = A0; a < A1; ++a) rs[a] = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)a * (uint32_t)P.F, 0, P.F * 8, 0x00020000);
        const uint32_t wt = (uint32_t)P.wt;
        const char *lutb = reinterpret_cast<const char *>(lut);
        const char *imgb = reinterpret_cast<const char *>(s.img);
        const char *subb = reinterpret_cast<const char *>(s.sub);
        for (int chunk = 0; chunk * 128 * OBS_PAIRS < P.F; ++chunk) {
            if (chunk > 0) load_desc(P, layout, chunk, lane, dsc);
            uint32_t b[OBS_CHUNK];
#pragma unroll
            for (int j = 0; j < OBS_CHUNK; ++j) b[j] = *reinterpret_cast<const uint16_t *>(imgb + (dsc[j] & 0xFFFFu));
            int sb[NA][OBS_CHUNK];
#pragma unroll
            for (int a = A0; a < A1; ++a)
#pragma unroll
                for (int j = 0; j < OBS_CHUNK; ++j) sb[a][j] = *reinterpret_cast<const int32_t *>(subb + 64 * a + desc_code(dsc[j]));
            double2_t v[NA][OBS_PAIRS];
#pragma unroll
            for (int a = A0; a < A1; ++a)
#pragma unroll
                for (int i = 0; i < OBS_PAIRS; ++i) {
                    v[a][i].x = *reinterpret_cast<const double *>(lutb + ((int)b[2 * i] - sb[a][2 * i]));
                    v[a][i].y = *reinterpret_cast<const double *>(lutb + ((int)b[2 * i + 1] - sb[a][2 * i + 1]));
                }
            const uint32_t f0b = (uint32_t)(chunk * OBS_PAIRS * 128 + 2 * lane) * 8u;
            uint32_t off[OBS_PAIRS];
#pragma unroll
            for (int i = 0; i < OBS_PAIRS; ++i) off[i] = (MODE == 2 || ((int32_t)dsc[2 * i] < 0) == (MODE == 1)) ? f0b + (uint32_t)i * 1024u : 0xFFFFFFF0u;
#define CZ_DUO_STORES(AUX)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < OBS_PAIRS; ++i)                                                                    \
        _Pragma("unroll") for (int a = A0; a < A1; ++a)                                                                      \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, v[a][i]), rs[a], off[i], 0, AUX)
            if (wt == 1u) { CZ_DUO_STORES(16); }
            else if (wt == 2u) { CZ_DUO_STORES(2); }
            else { CZ_DUO_STORES(0); }
#undef CZ_DUO_STORES
        }
        if (P.F > 128 * OBS_PAIRS) load_desc(P, layout, 0, lane, dsc);
    }
}

// the per-observer subtrahend table from the agents' words (lane 16 a + code; see observe()): written by H for its early pass
// and, with the same values, by M for its late one - neither has to wait for the other
__device__ __forceinline__ void duo_sub_table(Lds<1> &s, uint32_t Ao, uint32_t submask, int lane) {
    constexpr uint32_t PICK_XY = 0x0C010C00u;
    const uint32_t q = ((__builtin_amdgcn_perm(0u, Ao, PICK_XY) << 3) & submask);
    (&s.sub[0][0])[lane] = (int)((q & 0xFFFFu) + (q >> 16));
}

template <int NA, int SCHEME>
__global__ __launch_bounds__(64 * 2 * DUO_EPW) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_step_duo(uint32_t *e_state, const int32_t *e_actions, const double *e_lut, int32_t e_N, int32_t e_RW, int32_t e_W, int32_t e_H,
                int32_t e_D, int32_t e_dyn0, int32_t e_dyn1, const Params P0) {
    constexpr int OPL = 1, CPL = 1, EPW = DUO_EPW;
    constexpr int NA_M = (NA + 1) / 2;                       // observers whose late pairs M encodes; H takes the rest
    Params P = P0;
    P.state = e_state; P.actions = e_actions; P.lut = e_lut; P.N = e_N; P.RW = e_RW; P.W = e_W; P.H = e_H; P.D = e_D;
    P.dyn0_off = e_dyn0; P.dyn1_off = e_dyn1;
    __shared__ Lds<CPL> lds_all[EPW];
    __shared__ double lut[LUT_SIZE];
    __shared__ DuoSync sync_all[EPW];
    int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)rfl(threadIdx.x >> 6);
    const int slot = wave & (EPW - 1);
    const bool helper = wave >= EPW;
    const int env_raw = (int)blockIdx.x * EPW + slot;
    const int env = min(env_raw, P.N - 1);
    Lds<CPL> &lds = lds_all[slot];
    DuoSync &sy = sync_all[slot];
    Ctx cx{P.W, P.H, P.D, P.W * P.H, lane};
    // experiment switch (CZ_STOP): 0 the design above; 1 H leaves after the barrier, M encodes everything; 2 H only waits for
    // both flags, M encodes everything; 3 no early pass: H encodes all pairs of its observers after the second flag
    const int variant = P.stop < 0 ? 0 : P.stop;
    uint32_t *rec = P.state + (uint32_t)env * (uint32_t)P.RW;
    uint32_t *img32 = reinterpret_cast<uint32_t *>(lds.img);
    constexpr uint32_t PICK_XY = 0x0C010C00u;
    const uint32_t c01 = (uint32_t)((P.W - 1) * 8) | ((uint32_t)((LUT_Y0 + P.H - 1) * 8) << 16);

#ifdef CZ_TIMELINE
    // timeline build: word 0 = M's entry time, word 1 = the later of the two waves' exit times | XCC_ID << 32 (64-bit maximum;
    // cz_debug_set_timeline zeroes the buffer)
    const uint64_t tl_in = wall_clock64();
    const auto tl_exit = [&]() {
        unsigned long long *const tlb = late_params((unsigned)offsetof(StepArgsMirror, p))->timeline;
        if (tlb && lane == 0 && env_raw < P.N) {
            const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
            atomicMax(tlb + 2 * (size_t)env + 1, (wall_clock64() & 0xFFFFFFFFull) | ((unsigned long long)xcc << 32));
            if (!helper) tlb[2 * (size_t)env] = (tl_in & 0xFFFFFFFFull) | ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32);
        }
    };
#define CZ_DUO_TL_EXIT() tl_exit()
#else
#define CZ_DUO_TL_EXIT() do { } while (0)
#endif
    if (helper && variant == 4) return;      // (experiment: the helper waves leave at once; M does everything incl. the tables)
    if (helper) {
        // ================================================================ H
        const unsigned tl = threadIdx.x - 64u * EPW;                     // 0..511: the first 256 carry the quotient table
        const double lutv = ldg<double>(P.lut, min(tl, (unsigned)LUT_SIZE - 1u) * 8u);
        typedef const __attribute__((address_space(4))) uint32_t *kconst_u32;
        const uint32_t layout0 = ((kconst_u32)rec)[W_LAYOUT];
        uint32_t dsc[OBS_CHUNK];
        if (P.obs) load_desc(P, layout0, 0, lane, dsc);
        const uint32_t submask = load_submask(P, lane);
        init_lds<CPL>(P, cx, lds);                                       // the "absent" halfword, the cells' coordinates
        img32[(Img<CPL>::CELL0 >> 1) + 2 * lane + 1] = (uint32_t)(LUT_CF0 * 8) | ((uint32_t)(LUT_ONE * 8) << 16);   // flag (M rewrites it) | 1
        if (lane == 0) { sy.go1 = 0u; sy.go2 = 0u; }
        if (tl < (unsigned)LUT_SIZE) lut[tl] = lutv;
        __syncthreads();
        if (env_raw >= P.N || !P.obs || variant == 1) { CZ_DUO_TL_EXIT(); return; }
        double *const out = P.obs + (uint64_t)(uint32_t)env * (uint64_t)(uint32_t)(NA * P.F);
        duo_wait(&sy.go1);
        const uint32_t lay = rfl(*(volatile uint32_t *)&sy.layout);
        if (lay != layout0) load_desc(P, lay, 0, lane, dsc);            // (a reset pass moved the env to another layout)
        if (variant == 0) {
            const uint32_t A = *(volatile uint32_t *)&sy.agw[lane & 3], Ao = *(volatile uint32_t *)&sy.agw[lane >> 4];
            const uint32_t o8 = __umul24((A >> 16) & 7u, 0x80008u);
            if (lane < NA) {
                uint4_t agv;
                agv.x = (__builtin_amdgcn_perm(0u, A, PICK_XY) << 3) + c01;
                agv.y = o8 + ((uint32_t)((LUT_OR0 + 0) * 8) | ((uint32_t)((LUT_OR0 + 8) * 8) << 16));
                agv.z = o8 + ((uint32_t)((LUT_OR0 + 16) * 8) | ((uint32_t)((LUT_OR0 + 24) * 8) << 16));
                agv.w = (uint32_t)(LUT_ONE * 8);
                *reinterpret_cast<uint4_t *>(img32 + (Img<CPL>::AG0 >> 1) + 4 * lane) = agv;
            }
            duo_sub_table(lds, Ao, submask, lane);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (variant == 0) duo_encode<NA, 0, NA, 0>(P, lds, lut, lay, dsc, lane, out);
        if (NA_M < NA) {
            duo_wait(&sy.go2);
            if (variant == 0) duo_encode<NA, NA_M, NA, 1>(P, lds, lut, lay, dsc, lane, out);
            else if (variant == 3) duo_encode<NA, NA_M, NA, 2>(P, lds, lut, lay, dsc, lane, out);
        }
        CZ_DUO_TL_EXIT();
        return;
    }

    // ==================================================================== M
    const unsigned kp_off = (unsigned)offsetof(StepArgsMirror, p);
    double *retp = reinterpret_cast<double *>(rec + RET_WORD0);
    const int av = ldg<int>(P.actions, ((uint32_t)env * (uint32_t)NA + (uint32_t)min(lane, NA - 1)) * 4u);
    double ret = ldg<double>(retp, ((uint32_t)lane & 3u) * 8u);
    Env<OPL, CPL, NA> e;
    load_env(P, e, cx, rec, false);
    uint32_t rowv = load_recipe_rows(P, e.recipes, lane);
    uint32_t dsc[OBS_CHUNK];
    if (P.obs) load_desc(P, e.layout, 0, lane, dsc);
    const uint32_t submask = load_submask(P, lane);
    const int64_t env_global = P.env_id_base + env;
    if (variant == 4) {
        const double lutv = ldg<double>(P.lut, min(threadIdx.x, (unsigned)LUT_SIZE - 1u) * 8u);
        init_lds<CPL>(P, cx, lds);
        if (threadIdx.x < (unsigned)LUT_SIZE) lut[threadIdx.x] = lutv;
    }
    __syncthreads();
    if (env_raw >= P.N) return;
    CZ_SETPRIO(1);                                                       // M is the critical path: H fills in behind it

    Dirty dt{};
    StepOut o;
    step_env<OPL, CPL, NA, SCHEME>(P, kp_off, e, cx, (uint32_t)av, env_global, rowv, lds, dsc, dt, o, [&]() {
        if (lane < MAX_AGENTS) *(volatile uint32_t *)&sy.agw[lane] = e.agw;
        if (lane == 0) *(volatile uint32_t *)&sy.layout = e.layout;
        asm volatile("" ::: "memory");
        if (lane == 0) *(volatile uint32_t *)&sy.go1 = 1u;
    });
    const bool cells_dirty = dt.cells != 0, objs_dirty = (dt.touched | dt.interacted | dt.moved) != 0, header_dirty = o.header;
    const double myrew = o.myrew;
    if (o.stepped) ret += myrew;
    const KParams kp = late_params(kp_off);
    uint32_t su_old = 0u;
    double sf_old = 0.0, ret_done = 0.0;
    if (o.finished) {
        const uint32_t *su = kp->stat_u + (size_t)env * SU_WORDS;
        const double *sf = kp->stat_f + (size_t)env * SF_WORDS;
        su_old = ldg<uint32_t>(su, ((uint32_t)lane & 15u) * 4u);
        sf_old = ldg<double>(sf, (uint32_t)(SF_SUM0 + ((uint32_t)lane & 3u)) * 8u);
        ret_done = ret;
        ret = 0.0;
    }
    {
        const uint32_t oidx = (uint32_t)env * (uint32_t)NA + (uint32_t)lane;
        double *const rewards = kp->rewards;
        uint8_t *const term = kp->term, *const trunc = kp->trunc;
        if (lane < NA) {
            stg<double>(rewards, oidx * 8u, myrew);
            stg<uint8_t>(term, oidx, (uint8_t)o.term);
            stg<uint8_t>(trunc, oidx, (uint8_t)o.trunc);
            if (o.gone && ((o.gone >> lane) & 1u)) stg<uint8_t>(trunc, oidx, (uint8_t)1);
        }
        uint32_t *const marks_out = kp->marks_out;
        if (marks_out && lane < 2) stg<uint32_t>(marks_out, (2u * (uint32_t)env + (uint32_t)lane) * 4u, lane == 0 ? e.marks : e.marks_hi);
    }
    if (P.obs) {
        // ---- the late part of the image: objects (3 dwords per slot), the cells' mutable flag, and the subtrahend table
        const uint32_t dead = (uint32_t)(LUT_ABSENT * 8) * 0x10001u;
        {
            const uint32_t w = e.d0[0];
            const bool alive = (w & D_ALIVE) != 0u;
            const uint32_t st = (w >> 25) & 3u;
            uint32_t q0 = (__builtin_amdgcn_perm(0u, w, PICK_XY) << 3) + c01;
            uint32_t q1 = __umul24(st, 0x80008u) + ((uint32_t)(LUT_NDONE0 * 8) | ((uint32_t)(LUT_CH0 * 8) << 16));
            uint32_t q2 = (st << 3) + ((uint32_t)(LUT_MA0 * 8) | ((uint32_t)(LUT_ONE * 8) << 16));
            q0 = alive ? q0 : dead; q1 = alive ? q1 : dead; q2 = alive ? q2 : dead;
            img32[(Img<CPL>::OBJ0 >> 1) + 3 * lane] = q0;
            img32[(Img<CPL>::OBJ0 >> 1) + 3 * lane + 1] = q1;
            img32[(Img<CPL>::OBJ0 >> 1) + 3 * lane + 2] = q2;
        }
        img32[(Img<CPL>::CELL0 >> 1) + 2 * lane + 1] = ((e.cell[0] >> 2) & 0x18u) | ((uint32_t)(LUT_CF0 * 8) | ((uint32_t)(LUT_ONE * 8) << 16));
        {
            const uint32_t Ao = (uint32_t)__builtin_amdgcn_ds_bpermute((lane >> 4) << 2, (int)e.agw);
            duo_sub_table(lds, Ao, submask, lane);
            if (variant != 0 && lane < NA) {
                const uint32_t A = e.agw, o8 = __umul24((A >> 16) & 7u, 0x80008u);
                uint4_t agv;
                agv.x = (__builtin_amdgcn_perm(0u, A, PICK_XY) << 3) + c01;
                agv.y = o8 + ((uint32_t)((LUT_OR0 + 0) * 8) | ((uint32_t)((LUT_OR0 + 8) * 8) << 16));
                agv.z = o8 + ((uint32_t)((LUT_OR0 + 16) * 8) | ((uint32_t)((LUT_OR0 + 24) * 8) << 16));
                agv.w = (uint32_t)(LUT_ONE * 8);
                *reinterpret_cast<uint4_t *>(img32 + (Img<CPL>::AG0 >> 1) + 4 * lane) = agv;
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (NA_M < NA && lane == 0) *(volatile uint32_t *)&sy.go2 = 1u;
        asm volatile("" ::: "memory");
        double *const out = P.obs + (uint64_t)(uint32_t)env * (uint64_t)(uint32_t)(NA * P.F);
        if (variant == 0) duo_encode<NA, 0, NA_M, 1>(P, lds, lut, e.layout, dsc, lane, out);
        else if (variant == 3) duo_encode<NA, 0, NA_M, 2>(P, lds, lut, e.layout, dsc, lane, out);
        else duo_encode<NA, 0, NA, 2>(P, lds, lut, e.layout, dsc, lane, out);
    }
    if (o.finished) {
        uint32_t *su = kp->stat_u + (size_t)env * SU_WORDS;
        double *sf = kp->stat_f + (size_t)env * SF_WORDS;
        const uint32_t a_of = (uint32_t)lane - SU_COMPLETED0;
        const uint32_t root = P.wide ? (((a_of < 2u ? e.marks : e.marks_hi) >> (16u * (a_of & 1u))) & 1u) : ((e.marks >> (8u * (a_of & 3u))) & 1u);
        const uint32_t inc = lane == (int)SU_EPISODES ? 1u : lane == (int)SU_LENSUM ? e.t : lane == (int)SU_TRUNC ? (uint32_t)o.trunc
                             : lane == (int)SU_TERM ? (uint32_t)o.term : a_of < (uint32_t)NA ? root : 0u;
        if (lane < (int)SU_COMPLETED0 + NA && lane != (int)SU_STEPS) stg<uint32_t>(su, (uint32_t)lane * 4u, su_old + inc);
        if (lane < NA) stg<double>(sf, (uint32_t)(SF_SUM0 + lane) * 8u, sf_old + ret_done);
    }
    store_env(P, e, cx, rec, cells_dirty, objs_dirty, header_dirty, false);
    if (lane < NA) stg<double>(retp, (uint32_t)lane * 8u, ret);
    CZ_DUO_TL_EXIT();
#undef CZ_DUO_TL_EXIT
}

}  // namespace cz
