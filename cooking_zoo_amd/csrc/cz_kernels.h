// cz_kernels.h -- kernel templates of the step path (included by the per-size instantiation units).
#pragma once
#include "cz_device.h"

namespace cz {

// Diagnostic build only (make prof -> libcookingzoo_hip_prof.so, -DCZ_PROFILE): s_memrealtime stamps (100 MHz, one clock for
// the whole device, so that first start / last end of a launch can be read across XCDs) at phase boundaries,
// written to a buffer of their own (Params::stamps, 16 slots per env: 0..7 the phases, 8..10 inside the reward phase); the
// shipped library contains none of this.
#ifdef CZ_PROFILE
#define CZ_STAMP(i)                                                                                    \
    do {                                                                                               \
        unsigned long long _t;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (lane == 0 && P.stamps) P.stamps[(size_t)env * 16 + (i)] = _t;                               \
    } while (0)
#elif defined(CZ_MARKERS)
// marker build (tools/phase_cut.py, never loaded): an assembler comment at every phase boundary and no instruction - if the
// instruction stream equals the shipped one (tools/isa_diff.py), the comments are positions in the shipped code
#define CZ_STAMP(i) asm volatile("; CZ_MARK " #i)
#else
#define CZ_STAMP(i) do { } while (0)
#endif

// Parameters that are needed late (output pointers, statistics, the reward constants of the rare "a recipe node changed"
// path, the layout pool of the reset path) are read from the argument segment where they are used, through a pointer the
// optimiser cannot see through -- otherwise every one of them is fetched at kernel entry and held in SGPRs for the whole
// kernel (~30 SGPRs: spills in the non-fused kernel, ~90 spilled SGPRs in the fused one).
typedef const __attribute__((address_space(4))) Params *KParams;
struct StepArgsMirror { uint32_t *a; const int32_t *b; const double *c; int32_t i[8]; Params p; };   // (k_step: 7 ints + padding)   // k_step's argument list
__device__ __forceinline__ KParams late_params(unsigned offset) {
    const __attribute__((address_space(4))) char *k =
        (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));                      // loads through the result stay behind this point
    return reinterpret_cast<KParams>(k + offset);
}
#define CZ_LATE_STEP() late_params((unsigned)offsetof(StepArgsMirror, p))

// LDS image of one env (halfwords, cooking_zoo_amd/soa.py IMG_*): every halfword is a BYTE offset into `lut`
// Two layouts: up to 128 slots / 256 cells (CPL <= 4), and the "huge" one for up to 256 slots / 1024 cells (CPL = 16).
template <int CPL>
struct Img {
    static constexpr bool HUGE = CPL > 4;
    static constexpr int OBJ0 = 0, CELL0 = HUGE ? 1536 : 768, AG0 = CELL0 + (HUGE ? 4096 : 1024), ZERO = AG0 + 32, HALFWORDS = ZERO + 8;
};
constexpr int LUT_ABSENT = 255, LUT_SIZE = 256;
#ifndef CZ_ENVS_PER_WG
#define CZ_ENVS_PER_WG 8
#endif
constexpr int ENVS_PER_WG = CZ_ENVS_PER_WG;   // one wavefront per env, this many per workgroup (no cross-wave communication)
// the huge instance keeps 13.6 KB of LDS per env: four of them (the minimum that still stages the 256-entry table with one
// entry per thread) fit the 64 KB a workgroup may declare
template <int CPL> constexpr int envs_per_wg() { return CPL > 4 ? 4 : ENVS_PER_WG; }
constexpr int OBS_PAIRS = 3;       // feature pairs per lane and chunk: 3 x 128 = 384 features per chunk
constexpr int OBS_CHUNK = 2 * OBS_PAIRS;   // descriptor words held per lane

// (the quotient table `lut` itself is shared by the waves of a workgroup: [0..2W-2] (i-(W-1))/W | [63..63+2H-2] (i-(H-1))/H |
// [126] 0.0 | [127] 1.0 | [128..255] 0.0)
template <int CPL>
struct Lds {
    int32_t sub[MAX_AGENTS][16];         // per observer: what to subtract (x8) for each axis code (first: two of its rows are read
                                         // with one ds_read2_b32, whose offsets reach 1 KB)
    uint16_t img[Img<CPL>::HALFWORDS];   // objects 6 hw each | cells 4 hw each | agents 8 hw each | the "absent" halfword
    // recipe evaluation scratch: node bits per cell, per object kind and per cell type (recipe_marks_cells; instances with up
    // to 4 cells per lane) or matched-location bit sets per node, CPL words each (recipe_marks of the 32x32 instance,
    // recipe_marks_wide)
    uint64_t locs[CPL <= 4 ? 64 * CPL + 8 : WIDE_NODES * CPL];
};

// Memory access helpers: a wave-uniform base pointer plus a 32-bit unsigned per-lane byte offset, which the backend
// turns into the `global_load/store v, v_off, s[base:base+1]` form (no 64-bit per-lane address arithmetic).
template <class T>
__device__ __forceinline__ T ldg(const void *sbase, uint32_t voff) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(sbase) + voff);
}
template <class T>
__device__ __forceinline__ void stg(void *sbase, uint32_t voff, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(sbase) + voff) = v;
}
// write-through store (`global_store ... sc1`): the bytes leave the XCD's L2 right away instead of staying dirty until
// the end-of-kernel write-back, which would otherwise serialise ~B / 6 TB/s behind the kernel (MI355X_MICROARCH.md,
// "boundary" and "publish-large" rows).  Used for the streaming outputs (observations), which nobody re-reads on the GPU.
template <class T>
__device__ __forceinline__ void stg_wt(void *sbase, uint32_t voff, T v) {
    __hip_atomic_store(reinterpret_cast<T *>(reinterpret_cast<char *>(sbase) + voff), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

// Loads are clamped instead of exec-masked (no branches): every address stays inside the record.
template <int OPL, int CPL, int NA>
__device__ __forceinline__ void load_env(const Params &P, Env<OPL, CPL, NA> &e, const Ctx &cx, const uint32_t *__restrict__ rec) {
    const uint32_t lane = (uint32_t)cx.lane;
    // header words: wave-uniform, one scalar load (nothing in this kernel writes a record before its loads are done, and the
    // scalar cache is cold at kernel start like every other cache)
    uint32_t hw[HDR_WORDS];
    {
        typedef const __attribute__((address_space(4))) uint32_t *kconst_u32;
        const kconst_u32 hp = (kconst_u32)rec;
#pragma unroll
        for (int i = 0; i < HDR_WORDS; ++i) hw[i] = hp[i];
    }
    const uint32_t aw = ldg<uint32_t>(rec, (AGENT_WORD0 + (lane & 3u)) * 4u);        // agent words, lane a = agent a
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const uint32_t c = lane + 64u * k;
        const uint32_t v = ldg<uint8_t>(rec, CELL_WORD0 * 4u + min(c, (uint32_t)cx.C - 1u));
        e.cell[k] = (c < (uint32_t)cx.C) ? v : 0u;
    }
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
        const uint32_t s = lane + 64u * k, sc = min(s, (uint32_t)cx.D - 1u);
        const uint32_t a = ldg<uint32_t>(rec, ((uint32_t)P.dyn0_off + sc) * 4u), b = ldg<uint32_t>(rec, ((uint32_t)P.dyn1_off + sc) * 4u);
        e.d0[k] = (s < (uint32_t)cx.D) ? a : 0u;
        e.d1[k] = (s < (uint32_t)cx.D) ? b : 0u;
    }
    // every load of the record is issued before anything looks at a header word (whose first use waits for the scalar load:
    // without the fence the scheduler puts that wait, a memory round trip, in front of the remaining vector loads)
    __builtin_amdgcn_sched_barrier(0);
    e.t = hw[W_T]; e.marks = hw[W_MARKS]; e.layout = hw[W_LAYOUT]; e.status = hw[W_STATUS];
    e.episode = hw[W_EPISODE]; e.recipes = hw[W_RECIPES]; e.pool = hw[W_POOL];
    // (this test is also where the wave waits for its late scalar arguments -- before any LDS traffic is in flight, which
    // shares the wait counter with scalar loads: measured 0.1 us better than letting the first use wait further down)
    e.marks_hi = P.wide ? hw[W_MARKS_HI] : 0u;
    e.agw = (lane < (uint32_t)NA) ? aw : 0u;
}

// header + agents in one 12-lane store (v_writelane assembles the words), cells / objects only when they changed;
template <int OPL, int CPL, int NA>
__device__ __forceinline__ void store_env(const Params &P, const Env<OPL, CPL, NA> &e, const Ctx &cx, uint32_t *__restrict__ rec,
                                          bool cells_dirty, bool objs_dirty, bool header_dirty = true) {
    const uint32_t lane = (uint32_t)cx.lane;
    if (header_dirty) {
        uint32_t h = 0;
        h = wrl(e.t, W_T, h);
        h = wrl(e.marks, W_MARKS, h);
        h = wrl(e.layout, W_LAYOUT, h);
        h = wrl(e.status, W_STATUS, h);
        h = wrl(e.episode, W_EPISODE, h);
        h = wrl(e.recipes, W_RECIPES, h);
        h = wrl(e.pool, W_POOL, h);
        h = wrl(e.marks_hi, W_MARKS_HI, h);
        if (lane < (uint32_t)HDR_WORDS) stg<uint32_t>(rec, lane * 4u, h);
    } else if (lane == 0u) {
        stg<uint32_t>(rec, W_T * 4u, e.t);                          // the step counter is all that changed (the usual case)
    }
    if (lane < (uint32_t)MAX_AGENTS) stg<uint32_t>(rec, (AGENT_WORD0 + lane) * 4u, e.agw);
    if (cells_dirty) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const uint32_t c = lane + 64u * k;
            if (c < (uint32_t)cx.C) stg<uint8_t>(rec, CELL_WORD0 * 4u + c, (uint8_t)e.cell[k]);
        }
    }
    if (objs_dirty) {
#pragma unroll
        for (int k = 0; k < OPL; ++k) {
            const uint32_t s = lane + 64u * k;
            if (s < (uint32_t)cx.D) {
                stg<uint32_t>(rec, ((uint32_t)P.dyn0_off + s) * 4u, e.d0[k]);
                stg<uint32_t>(rec, ((uint32_t)P.dyn1_off + s) * 4u, e.d1[k]);
            }
        }
    }
}

// the recipe rows of this env, one word per lane: lane 9r + i = word i of the row of recipe r (which r, i a lane stands for is
// a constant of the lane, made by the host: 8 r | 4 i << 8; lanes past the rows repeat the last word)
__device__ __forceinline__ uint32_t load_recipe_rows(const Params &P, uint32_t recipes, int lane) {
    const uint32_t sel = ldg<uint32_t>(P.lut, ROWSEL_TABLE_OFFSET + (uint32_t)lane * 4u);
    const uint32_t id = (recipes >> (sel & 0xFFu)) & 0xFFu;
    return ldg<uint32_t>(P.recipes, __umul24(id, (1u + MAX_NODES) * 4u) + (sel >> 8));
}

// every recipe of the env from scratch (reset): sets e.marks (and e.marks_hi for wide tables)
template <int OPL, int CPL, int NA>
__device__ __forceinline__ void all_marks(const Params &P, Env<OPL, CPL, NA> &e, const Ctx &cx, uint32_t rowv, Lds<CPL> &s) {
    uint32_t lo = 0, hi = 0;
    if (__builtin_expect(P.wide != 0, 0)) {
#pragma nounroll
        for (int r = 0; r < P.R; ++r) {
            const uint32_t id = (e.recipes >> (8 * r)) & 0xFFu;
            const uint32_t m = Ops<OPL, CPL, NA, 3>::recipe_marks_wide(e, cx, P.recipes + (size_t)id * (1 + 2 * WIDE_NODES), s.locs);
            if (r < 2) lo |= m << (16 * r); else hi |= m << (16 * (r - 2));
        }
    } else {
        if constexpr (CPL <= 4) {
            lo = Ops<OPL, CPL, NA, 3>::recipe_marks_cells(e, cx, rowv, 0xFu, P.R, s.locs);
        } else {
#pragma nounroll
            for (int r = 0; r < P.R; ++r) lo |= Ops<OPL, CPL, NA, 3>::recipe_marks(e, cx, rowv, 9 * r, s.locs) << (8 * r);
        }
    }
    e.marks = lo; e.marks_hi = hi;
}

// the per-lane constant of the subtrahend table (observe): word `lane` behind the 256 doubles of the quotient table
__device__ __forceinline__ uint32_t load_submask(const Params &P, int lane) { return ldg<uint32_t>(P.lut, (uint32_t)(LUT_SIZE * 8 + lane * 4)); }

// once per kernel and workgroup: the quotient table (thread `tid` of `nthreads`), followed by a workgroup barrier
__device__ __forceinline__ void init_lut(const Params &P, double *lut, int tid, int nthreads) {
    for (int i = tid; i < LUT_SIZE; i += nthreads) lut[i] = ldg<double>(P.lut, (uint32_t)i * 8u);
}

// once per kernel and env: the constant part of the image (cell coordinates: a host-made table, one word per cell)
template <int CPL>
__device__ __forceinline__ void init_lds(const Params &P, const Ctx &cx, Lds<CPL> &s) {
    const int lane = cx.lane;
    s.img[Img<CPL>::ZERO] = (uint16_t)(LUT_ABSENT * 8);           // (every lane, same value: cheaper than switching lanes off)
    uint32_t *img32 = reinterpret_cast<uint32_t *>(s.img);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const uint32_t c = (uint32_t)(lane + 64 * k);
        img32[(Img<CPL>::CELL0 >> 1) + 2 * c] = ldg<uint32_t>(P.lut, COORD_TABLE_OFFSET + c * 4u);
    }
}

// the descriptor words of this lane for one chunk: pair i of a chunk covers features [(chunk*OBS_PAIRS + i)*128 + 2*lane, +1].
// Buffer loads over one resource that spans the whole descriptor table (the layout's row is the scalar offset): no clamps, no
// masks, no branches on F.  Words past the row's F descriptors are the next layout's - valid image offsets like any other,
// read for nothing since the stores of features past F are out of their row's range - or, behind the table's end, 0.
__device__ __forceinline__ void load_desc(const Params &P, uint32_t layout, int chunk, int lane, uint32_t (&dsc)[OBS_CHUNK]) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(P.lay_desc), 0, P.L * P.F * 4, 0x00020000);
    const uint32_t row = layout * (uint32_t)P.F * 4u;                                       // uniform
#pragma unroll
    for (int i = 0; i < OBS_CHUNK / 2; ++i) {
        const uint32_t f = (uint32_t)(chunk * (OBS_CHUNK / 2) + i) * 128u + 2u * (uint32_t)lane;
        const uint2_t d = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, f * 4u, row, 0));
        dsc[2 * i] = d.x; dsc[2 * i + 1] = d.y;
    }
}

// cooking_env.py:352-373 get_feature_vector for every agent of the env: out[a][f] = lut[img[desc.hw] - sub[a][desc.code]]
// P.wt selects the cache policy of the observation stores (wave-uniform; policy in cz_api.hip launch_step):
//   1 = write-through (`buffer_store_dwordx4 ... sc1`): the bytes leave the XCD's L2 while the kernel still computes instead
//       of staying dirty until the end-of-kernel write-back (which serialises ~B / 6 TB/s behind every launch:
//       MI355X_MICROARCH.md "boundary" / "publish-large").  Pays for one launch per step of a moderate batch.
//   2 = streaming (`... nt`): for output buffers larger than the memory-side cache.  With plain stores a launch's 278+ MiB of
//       observations sweep the 256 MiB Infinity Cache, so the records, tables and descriptors the NEXT waves read come
//       from HBM, queued behind the writes: a one-step launch of 65 536 envs takes 80 us with plain stores and 56 us with
//       streaming ones (profiles/r03/aux_sweep.txt).
//   0 = plain: the fused kernel (its launch boundary is amortised over T steps) and batches in between.
// `codes` (optional, cz_step_device_compact): the same observation as ONE BYTE per feature - the index of the table entry
// the feature's value is, out[a][f] == lut[codes[a][f]] bit for bit - in rows of codes_pitch(F) bytes (padding: 255, an entry
// that is 0.0): 1/8 of the bytes for consumers that stay on the device.  Lanes take four adjacent features each, one
// descriptor b128 load and one 4-byte store per observer and 256 features; no table reads at all.  `out` may be null then.
__host__ __device__ inline int codes_pitch(int F) { return (F + 15) & ~15; }
// the descriptor words of this lane's first four features, fetched with the prologue's loads (layout: which row they are of)
constexpr int CODES_PREFETCH = 2;               // rounds of 256 features (F = 278: both)
struct CodesPrefetch { uint4_t d[CODES_PREFETCH]; uint32_t layout; };
__device__ __forceinline__ uint4_t load_desc4(const Params &P, uint32_t layout, uint32_t f) {
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(P.lay_desc), 0, P.L * P.F * 4, 0x00020000);
    return __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(rd, f * 4u, layout * (uint32_t)P.F * 4u, 0));
}
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) {                       // v_pk_sub_u16: two 16-bit lanes, no borrow across
    return __builtin_bit_cast(uint32_t, (ushort2_t)(__builtin_bit_cast(ushort2_t, a) - __builtin_bit_cast(ushort2_t, b)));
}
template <int OPL, int CPL, int NA, bool F64 = true>
__device__ __forceinline__ void observe(const Params &P, const Env<OPL, CPL, NA> &e, const Ctx &cx, Lds<CPL> &s,
                                        const double *lut, uint32_t (&dsc)[OBS_CHUNK], uint32_t submask, double *__restrict__ out /* [A][F] of this env */,
                                        bool objs_changed = true, bool cells_changed = true, uint8_t *__restrict__ codes = nullptr /* [A][Fp] */,
                                        CodesPrefetch *pre = nullptr) {
    uint32_t *img32 = reinterpret_cast<uint32_t *>(s.img);
    const uint32_t dead = (uint32_t)(LUT_ABSENT * 8) * 0x10001u;
    const uint32_t c01 = (uint32_t)((P.W - 1) * 8) | ((uint32_t)((LUT_Y0 + P.H - 1) * 8) << 16);
    // Flags are not computed: the halfword of a flag is the entry of a small truth table, indexed by the raw state bits
    // (LUT_NDONE0 / LUT_CH0 / LUT_MA0 by chopped | mashed << 1, LUT_CF0 by the cell's ACTIVE | WALK << 1 bits, LUT_OR0 + 8 k
    // by the orientation), so an image word is one multiply-add of the extracted field.
    constexpr uint32_t PICK_XY = 0x0C010C00u;                       // v_perm_b32: bytes x, 0, y, 0 of a word x | y << 8 | ...
    // ---- objects: 3 dwords per slot (in a fused rollout only when an object moved or changed state since the last encode)
    if (objs_changed)
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
        const uint32_t w = e.d0[k];
        const bool alive = (w & D_ALIVE) != 0u;
        const uint32_t st = (w >> 25) & 3u;
        uint32_t q0 = (__builtin_amdgcn_perm(0u, w, PICK_XY) << 3) + c01;
        uint32_t q1 = __umul24(st, 0x80008u) + ((uint32_t)(LUT_NDONE0 * 8) | ((uint32_t)(LUT_CH0 * 8) << 16));
        uint32_t q2 = (st << 3) + ((uint32_t)(LUT_MA0 * 8) | ((uint32_t)(LUT_ONE * 8) << 16));
        q0 = alive ? q0 : dead; q1 = alive ? q1 : dead; q2 = alive ? q2 : dead;
        const int slot = cx.lane + 64 * k;
        img32[(Img<CPL>::OBJ0 >> 1) + 3 * slot] = q0;
        img32[(Img<CPL>::OBJ0 >> 1) + 3 * slot + 1] = q1;
        img32[(Img<CPL>::OBJ0 >> 1) + 3 * slot + 2] = q2;
    }
    // ---- cells: the mutable flag (switch_active / block walkable) + the constant 1
    if (cells_changed)
#pragma unroll
    for (int k = 0; k < CPL; ++k)
        img32[(Img<CPL>::CELL0 >> 1) + 2 * (cx.lane + 64 * k) + 1] =
            ((e.cell[k] >> 2) & 0x18u) | ((uint32_t)(LUT_CF0 * 8) | ((uint32_t)(LUT_ONE * 8) << 16));
    // ---- agents: 4 dwords each (lane a builds agent a's), and the per-observer subtrahend table
    {
        const uint32_t A = e.agw;
        const uint32_t o8 = __umul24((A >> 16) & 7u, 0x80008u);
        if (cx.lane < NA) {
            uint4_t agv;
            agv.x = (__builtin_amdgcn_perm(0u, A, PICK_XY) << 3) + c01;
            agv.y = o8 + ((uint32_t)((LUT_OR0 + 0) * 8) | ((uint32_t)((LUT_OR0 + 8) * 8) << 16));
            agv.z = o8 + ((uint32_t)((LUT_OR0 + 16) * 8) | ((uint32_t)((LUT_OR0 + 24) * 8) << 16));
            agv.w = (uint32_t)(LUT_ONE * 8);
            *reinterpret_cast<uint4_t *>(img32 + (Img<CPL>::AG0 >> 1) + 4 * cx.lane) = agv;
        }
        // sub[a][code]: lane 16a + code.  axis codes: 1 -> x, 2 -> y, 4+2j -> x unless j == a, 5+2j -> y unless j == a.  Which
        // coordinate a lane's entry takes is a constant of the lane: `submask` (0x7F8 in the low half: x, in the high half:
        // y), made by the host next to the quotient table and loaded with the prologue's loads.
        const uint32_t Ao = (uint32_t)__builtin_amdgcn_ds_bpermute((cx.lane >> 4) << 2, (int)A);     // the observer's word
        const uint32_t q = ((__builtin_amdgcn_perm(0u, Ao, PICK_XY) << 3) & submask);               // x * 8 | y * 8 << 16, masked
        (&s.sub[0][0])[cx.lane] = (int)((q & 0xFFFFu) + (q >> 16));         // (rows of observers >= NA: never read)
    }
    __builtin_amdgcn_wave_barrier();             // one wave owns this LDS region: DS ops of a wave execute in order
    // One buffer resource per observer whose range is exactly that observer's row (F * 8 bytes): the per-lane offset
    // (feature index * 8) is range-checked by the hardware, dword by dword (the scalar offset would be part of that check,
    // so the row's base goes into the resource).  Features past F - whole pairs, or the second half of the last pair when
    // F is odd - are dropped by the memory pipeline and the store sequence needs no lane masks or branches at all.
    const char *lutb = reinterpret_cast<const char *>(lut);
    const char *imgb = reinterpret_cast<const char *>(s.img);
    const char *subb = reinterpret_cast<const char *>(s.sub);
    if (codes) {
        const int Fp = codes_pitch(P.F);
        decltype(__builtin_amdgcn_make_buffer_rsrc(codes, 0, 0, 0)) rc[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) rc[a] = __builtin_amdgcn_make_buffer_rsrc(codes + (size_t)a * (uint32_t)Fp, 0, Fp, 0x00020000);
        // (the first rounds' descriptor words come in registers: fetched behind the prologue or at the top of a fused step, and
        // fetched again by the caller when a reset pass has moved the env to another layout - `pre` is of e.layout here)
        for (int f0 = 0; f0 < P.F; f0 += 256) {
            const uint32_t f = (uint32_t)f0 + 4u * (uint32_t)cx.lane;                       // this lane's first feature
            uint4_t d;
            if (pre && f0 < 256 * CODES_PREFETCH) d = f0 == 0 ? pre->d[0] : pre->d[1];
            else d = load_desc4(P, e.layout, f);
            const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
            uint32_t b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] = *reinterpret_cast<const uint16_t *>(imgb + (dw[k] & 0xFFFFu));
            int sb[NA][4];
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) sb[a][k] = *reinterpret_cast<const int32_t *>(subb + 64 * a + (dw[k] >> 16));
            // A code is (image halfword - subtrahend) >> 3, both multiples of 8 below 2048: two features share a register as 16-bit
            // halves (v_pk_sub_u16), one 32-bit shift by 3 leaves each code in the low byte of its half, v_perm_b32 gathers the
            // four bytes of a lane; bytes past F are forced to the padding value by an OR (vector-only: no scalar work per round)
            const uint32_t p01 = b[0] | (b[1] << 16), p23 = b[2] | (b[3] << 16);
            const int left = P.F - (int)f;                                                  // features of this lane that exist
            const uint32_t pad = left >= 4 ? 0u : left <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << (8u * (uint32_t)left));
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const uint32_t s01 = (uint32_t)sb[a][0] | ((uint32_t)sb[a][1] << 16), s23 = (uint32_t)sb[a][2] | ((uint32_t)sb[a][3] << 16);
                const uint32_t c01 = pk_sub_u16(p01, s01) >> 3, c23 = pk_sub_u16(p23, s23) >> 3;
                // (write-through like the float64 rows of a one-step launch: nothing stays dirty until the end-of-kernel write-back)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(c23, c01, 0x06040200u) | pad, rc[a], f, 0, 16);
            }
        }
    }
    if (!F64 || !out) { __builtin_amdgcn_wave_barrier(); return; }      // (F64 = false: the codes-only instance carries no float64 path)
    decltype(__builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0)) rs[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) rs[a] = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)a * (uint32_t)P.F, 0, P.F * 8, 0x00020000);
    const uint32_t wt = (uint32_t)P.wt;                                                 // wave-uniform
    for (int chunk = 0; chunk * 128 * OBS_PAIRS < P.F; ++chunk) {
        if (chunk > 0) load_desc(P, e.layout, chunk, cx.lane, dsc);
        // branch-free stages so that the LDS reads of all pairs and observers are in flight together
        // (descriptor words past F are 0: they read image halfword 0 and their stores are out of range)
        uint32_t b[OBS_CHUNK];
        int sb[NA][OBS_CHUNK];
        double2_t v[NA][OBS_PAIRS];
#pragma unroll
        for (int j = 0; j < OBS_CHUNK; ++j) b[j] = *reinterpret_cast<const uint16_t *>(imgb + (dsc[j] & 0xFFFFu));
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int j = 0; j < OBS_CHUNK; ++j) sb[a][j] = *reinterpret_cast<const int32_t *>(subb + 64 * a + (dsc[j] >> 16));
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int i = 0; i < OBS_PAIRS; ++i) {
                v[a][i].x = *reinterpret_cast<const double *>(lutb + ((int)b[2 * i] - sb[a][2 * i]));
                v[a][i].y = *reinterpret_cast<const double *>(lutb + ((int)b[2 * i + 1] - sb[a][2 * i + 1]));
            }
        const uint32_t f0b = (uint32_t)(chunk * OBS_PAIRS * 128 + 2 * cx.lane) * 8u;      // byte offset of this lane's first pair
#define CZ_OBS_STORES(AUX)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < OBS_PAIRS; ++i)                                                                    \
        _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                                       \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, v[a][i]), rs[a], f0b + (uint32_t)i * 1024u, 0, AUX)
        if (wt == 1u) { CZ_OBS_STORES(16); }            // sc1: write-through
        else if (wt == 2u) { CZ_OBS_STORES(2); }        // nt: streaming
        else { CZ_OBS_STORES(0); }
#undef CZ_OBS_STORES
    }
    // the caller keeps the descriptors of chunk 0 across steps (fused rollout): restore them if later chunks replaced them
    if (P.F > 128 * OBS_PAIRS) load_desc(P, e.layout, 0, cx.lane, dsc);
    __builtin_amdgcn_wave_barrier();
}

struct StepOut {
    double myrew;                  // lane a < NA: the reward of agent a (cooking_env.py:255-261); 0.0 on the other lanes
    uint32_t term, trunc;
    bool stepped, finished;        // a world step was executed / it ended the episode
    bool header;                   // a header word other than `t` changed (marks, status, episode, layout)
    uint32_t gone;                 // despawn / respawn: bit a = agent a left in this step (reported truncated once)
#ifdef CZ_TIMELINE
    uint32_t dbg;                  // timeline build: 1 an object moved or changed, 2 recipe graphs re-evaluated, 4 marks changed, 8 somebody interacted
#endif
};

// One accumulated_step (cooking_env.py:243-269) of one env held in registers.
// HINT (the fused instances): the reset pass and the end of an episode are laid out off the fall-through path - one step in max_steps + 1
// takes them; in the one-step kernels the same hint cost the launches under a cooking policy 1.5 % (profiles/r05/ab_experiments.txt)
#define CZ_RARE(hint, x) ((hint) ? __builtin_expect(!!(x), 0) : !!(x))
template <int OPL, int CPL, int NA, int SCHEME, bool HINT = false>
__device__ __forceinline__ void step_env(const Params &P, unsigned kp_off, Env<OPL, CPL, NA> &e, const Ctx &cx, uint32_t acts,
                                         int64_t env_global, uint32_t &rowv, Lds<CPL> &lds, uint32_t (&dsc)[OBS_CHUNK], Dirty &dt, StepOut &o) {
    using O = Ops<OPL, CPL, NA, SCHEME>;
    o.myrew = 0.0;
    o.term = 0; o.trunc = 0; o.stepped = false; o.finished = false; o.header = false; o.gone = 0u;
#ifdef CZ_TIMELINE
    o.dbg = 0u;
#endif
    // P.auto_reset: bit 0 = next-step auto-reset, bit 1 = agent despawn / respawn is on (cz_set_spawn)
    const bool spawning = (P.auto_reset & 2) != 0;
    const SpawnCfg *const spawn_cfg = reinterpret_cast<const SpawnCfg *>(reinterpret_cast<const char *>(P.lut) + SPAWN_CFG_OFFSET);
    if (CZ_RARE(HINT, (e.status & ST_DONE) != 0u)) {
        o.header = true;
        if (P.auto_reset & 1) {
            // next-step autoreset: reset() of cooking_env.py:178-210 from the layout pool
            e.episode += 1;
            const uint32_t *const lay0 = late_params(kp_off)->lay_init;
            // (the control words are constant while a kernel runs: the host changes them between launches, in stream order)
            typedef const __attribute__((address_space(4))) uint32_t *kconst_u32;
            const kconst_u32 ctl = (kconst_u32)(lay0 - LAY_CTL_WORDS);
            uint32_t lay = next_layout(env_global, e.episode, e.pool, (uint32_t)P.L, ctl[LC_GROUPS], ctl[LC_ACTIVE]);
            uint32_t recipes = e.recipes, episode = e.episode, pool = e.pool;
            load_env(P, e, cx, lay0 + (size_t)lay * P.RW);
            e.t = 0; e.layout = lay; e.status = 0; e.episode = episode; e.recipes = recipes; e.pool = pool;
            if (spawning) {       // a fresh world: everybody present, grace periods running (load_level.py:67-68, parsing.py:142)
                typedef const __attribute__((address_space(4))) SpawnCfg *kcfg;
                e.status = spawn_initial_status(((kcfg)spawn_cfg)->grace_period, NA);
            }
            all_marks(P, e, cx, rowv, lds);
            if (P.obs) load_desc(P, lay, 0, cx.lane, dsc);
            dt.cells = 1; dt.touched = 1; dt.interacted = 1;          // everything must be written back
        } else {
            o.term = (e.status & ST_TERM) ? 1u : 0u;
            o.trunc = (e.status & ST_TRUNC) ? 1u : 0u;
        }
        return;
    }
    o.stepped = true;
    e.t += 1;                                                        // cooking_env.py:244
    // (a despawned agent is not in the list world_step acts on, cooking_world.py:105-108: agents_walk takes it out like an
    // agent with action -1)
#if defined(CZ_PROFILE)
    const int lane = cx.lane; const long long env = env_global - P.env_id_base;
#endif
    O::perform_agent_actions(e, cx, acts, dt);                      // cooking_world.py:104-108
    CZ_STAMP(2);
    O::progress_and_link(e, cx, dt);
    if (spawning) {                             // :109-110 handle_agent_spawn
        o.gone = O::handle_agent_spawn(e, cx, spawn_cfg, env_global);
        o.header = true;                        // (the grace counters live in the status word)
    }
    CZ_STAMP(3);
    // compute_rewards cooking_env.py:290-315
    const bool truncated = (int)e.t >= P.max_steps;                 // compute_truncated :333-350
    const uint32_t before = e.marks;
    uint32_t after = before;
    o.myrew = cx.lane < NA ? P.reward_idle : 0.0;
    if (dt.moved & (uint32_t)P.walk_touches) { dt.touched = 1; dt.kinds = ~0ull; }
    if (dt.kinds & 0xFull) dt.kinds = ~0ull;                        // a Plate moved: it drags its content along -> anything
    bool done = false;
    if (__builtin_expect(P.wide != 0, 0)) {
        // Wide recipe tables (a graph with more than 8 nodes in the book): no filters, every recipe of the env is
        // re-evaluated whenever an object moved or changed; marks are 16 bits per recipe (recipes 0, 1 in `marks`, 2, 3 in
        // `marks_hi`).  Without a change the marks, hence `done` (false, or the env would not be stepping), stay.
        if (dt.touched) {
            uint32_t lo = 0, hi = 0;
#pragma nounroll
            for (int r = 0; r < P.R; ++r) {
                const uint32_t id = (e.recipes >> (8 * r)) & 0xFFu;
                const uint32_t *row = P.recipes + (size_t)id * (1 + 2 * WIDE_NODES);
                const uint32_t mb_r = ((r < 2 ? e.marks : e.marks_hi) >> (16 * (r & 1))) & 0xFFFFu;
                const uint32_t ma = O::recipe_marks_wide(e, cx, row, lds.locs);
                if (r < 2) lo |= ma << (16 * r); else hi |= ma << (16 * (r - 2));
                if (ma != mb_r) {
                    uint32_t countmask = 0;
                    const int n = (int)(rfl(row[0]) & 0xFFu);
#pragma nounroll
                    for (int j = 0; j < n; ++j) countmask |= ((rfl(row[1 + 2 * j]) >> 8) & 1u) << j;
                    const int goals_before = __popc(~mb_r & countmask), goals_after = __popc(~ma & countmask);
                    const bool completed = ma & 1, completion_before = mb_r & 1;
                    const bool malus = !completed && completion_before, bonus = completed && !completion_before;
                    const KParams kp = late_params(kp_off);
                    double x = 0.0;
                    x += (double)(goals_before - goals_after) * kp->node_reward;
                    x += (bonus ? 1.0 : 0.0) * kp->recipe_reward;
                    x += (malus ? 1.0 : 0.0) * kp->recipe_penalty;
                    x += kp->time_penalty_step;
                    if (cx.lane == r && r < NA) o.myrew = x;
                }
            }
            o.header |= lo != e.marks || hi != e.marks_hi;
            e.marks = lo; e.marks_hi = hi;
        }
        // recipe roots are bit 0 of each 16-bit field
        const uint32_t rlo = e.marks & (P.R >= 2 ? 0x00010001u : 0x00000001u);
        const uint32_t rhi = P.R >= 3 ? (e.marks_hi & (P.R >= 4 ? 0x00010001u : 0x00000001u)) : 0u;
        const int n_roots = __popc(rlo) + __popc(rhi);
        done = P.end_all ? (n_roots == P.R) : (n_roots != 0);
    } else {
        if (dt.touched) {
            // Which recipes must be re-evaluated?  Marks are a pure function of object state, and (recipe.py:77-104)
            //  * an object that matches no node of a recipe (class + state conditions) neither before nor after its change is
            //    in no node's matched list either way, and nothing else asks where it is: the recipe cannot see it;
            //  * a mere MOVE leaves every leaf's mark alone (leaves have no location constraint), so it can only matter to a
            //    node that has children, and such a node is only looked at when all its children are marked: if no node of
            //    the recipe is in that position now, no mark can change (induction from the leaves up).
            // Both tests run for all nodes of all recipes at once: lane 9r + 1 + j holds node j of recipe r (rowv).
            const uint32_t nw = rowv, lane_u = (uint32_t)cx.lane;
            const uint32_t r_of = (lane_u * 57u) >> 9;                                   // lane / 9
            const uint32_t seen = (uint32_t)(dt.kinds >> ((nw >> 14) & 0x3Cu)) & ((nw >> 24) & 0xFu);
            const uint32_t kids = nw & 0xFFu, mb = (before >> (8u * (r_of & 3u))) & 0xFFu;
            constexpr uint64_t NODE_LANES = 0x1FEull | (0x1FEull << 9) | (0x1FEull << 18) | (0x1FEull << 27);
            const bool rel = (nw & 0x2000u) != 0u && seen != 0u, sens = kids != 0u && (mb & kids) == kids;
            // the rewards of recipe r whose marks went from mb_r to ma (cooking_env.py:255-261, recipe.py:36-40)
            const auto reward_of = [&](int r, uint32_t mb_r, uint32_t ma) {
                // goals_completed sums (recipe.py:36-40): open goal slots before / after
                const uint32_t countmask = (uint32_t)((ballot((nw & 0x100u) != 0u) >> (9 * r + 1)) & 0xFFull);
                const int goals_before = __popc(~mb_r & countmask), goals_after = __popc(~ma & countmask);
                const bool completed = ma & 1, completion_before = mb_r & 1;
                const bool malus = !completed && completion_before, bonus = completed && !completion_before;
                const KParams kp = late_params(kp_off);
                double x = 0.0;
                x += (double)(goals_before - goals_after) * kp->node_reward;
                x += (bonus ? 1.0 : 0.0) * kp->recipe_reward;
                x += (malus ? 1.0 : 0.0) * kp->recipe_penalty;
                x += kp->time_penalty_step;
                if (cx.lane == r && r < NA) o.myrew = x;                       // recipe r is agent r's (cooking_env.py:255-261)
            };
            CZ_STAMP(8);
            if constexpr (CPL <= 4) {
                // "some node of recipe r" for both tests in one OR over the wave: bit r = visible, bit 4 + r = sensitive
                const uint32_t flags = (rel ? 1u : 0u) | (sens ? 16u : 0u);
                const uint32_t any = O::wave_or(lanes(NODE_LANES) ? (flags << (r_of & 3u)) : 0u);
                const uint32_t which = any & (dt.statechg != 0u ? 0xFu : (any >> 4)) & ((1u << P.R) - 1u);   // the recipes to re-evaluate
                if (which) {
                    CZ_SETPRIO(3);
#ifdef CZ_TIMELINE
                    o.dbg |= 2u;
#endif
                    const uint32_t fresh = O::recipe_marks_cells(e, cx, rowv, which, P.R, lds.locs);
                    CZ_STAMP(9);
                    const uint32_t bytes = ((which * 0x204081u) & 0x01010101u) * 0xFFu;    // bit r -> byte r
                    after = (before & ~bytes) | (fresh & bytes);
                    if (after != before) {
#pragma nounroll
                        for (int r = 0; r < P.R; ++r) {
                            const uint32_t mb_r = (before >> (8 * r)) & 0xFFu, ma = (after >> (8 * r)) & 0xFFu;
                            if (ma != mb_r) reward_of(r, mb_r, ma);
                        }
                    }
                }
            } else {
                const uint64_t relmask = ballot(rel) & NODE_LANES, sensmask = ballot(sens) & NODE_LANES;
                after = 0;
#pragma nounroll
                for (int r = 0; r < P.R; ++r) {
                    const uint32_t mb_r = (before >> (8 * r)) & 0xFF;
                    const bool visible = ((relmask >> (9 * r + 1)) & 0xFFull) != 0ull;
                    const bool sensitive = dt.statechg != 0u || ((sensmask >> (9 * r + 1)) & 0xFFull) != 0ull;
                    uint32_t ma = mb_r;
                    if (visible && sensitive) {
                        CZ_SETPRIO(3);
                        ma = O::recipe_marks(e, cx, rowv, 9 * r, lds.locs);
                    }
                    after |= ma << (8 * r);
                    if (ma != mb_r) reward_of(r, mb_r, ma);
                }
            }
            CZ_STAMP(10);
        }
        e.marks = after;
        o.header |= after != before;
#ifdef CZ_TIMELINE
        o.dbg |= (dt.touched ? 1u : 0u) | (after != before ? 4u : 0u) | (dt.interacted ? 8u : 0u);
#endif
        // recipe roots are bit 0 of each marks byte
        const uint32_t roots = after & 0x01010101u & (P.R >= 4 ? 0xFFFFFFFFu : ((1u << (8 * P.R)) - 1u));
        done = P.end_all ? (__popc(roots) == P.R) : (roots != 0u);
    }
    o.term = done ? 1u : 0u;
    o.trunc = truncated ? 1u : 0u;
    if (CZ_RARE(HINT, done || truncated)) {
        e.status |= ST_DONE | (done ? ST_TERM : 0u) | (truncated ? ST_TRUNC : 0u);
        o.finished = true;
        o.header = true;
    }
}

// ENVS_PER_WG envs per workgroup (one wavefront each, no cross-wave communication).
// FUSED = 0: one step, actions from memory.  FUSED = 1: P.T steps, on-device action stream, outputs [t][env].  FUSED = 2: P.T
// steps over the caller's actions [t][env][agent] (cz_rollout_actions; its own instance so that the on-device stream's loop
// carries none of it: as a run-time branch it cost the random-action rollout 4 %).  FUSED = 3: one step that also (or only)
// writes the compact observation (cz_step_device_compact / cz_set_compact_output; its own instance as well: compiled into the
// ordinary one-step kernel the path cost every launch 0.2 us, register allocation and a longer prologue, even when unused).
// FUSED = 4: P.T steps over the on-device action stream with a compact trajectory [t][env][agent][pitch] (cz_rollout_compact);
// FUSED = 5: the same without a float64 trajectory beside it (codes only).
// What the very first loads of a wave need travels as leading scalar kernel arguments: the build preloads them into
// SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count, gfx940+), so the record / action / table loads are issued
// without waiting for an argument fetch; everything else stays in the by-value block `P0`, fetched meanwhile.
struct Early {
    uint32_t *state;               // Params::state
    const int32_t *actions;        // Params::actions
    const double *lut;             // Params::lut
    int32_t N, RW, W, H, D, dyn0_off, dyn1_off;
};
__host__ __device__ inline Early early_of(const Params &P) {
    return Early{P.state, P.actions, P.lut, P.N, P.RW, P.W, P.H, P.D, P.dyn0_off, P.dyn1_off};
}

template <int OPL, int CPL, int NA, int SCHEME, int FUSED_MODE>
__device__ __forceinline__ void step_kernel(uint32_t *e_state, const int32_t *e_actions, const double *e_lut, int32_t e_N,
                                            int32_t e_RW, int32_t e_W, int32_t e_H, int32_t e_D, int32_t e_dyn0,
                                            int32_t e_dyn1, const Params &P0) {
#ifdef CZ_TIMELINE
    // Timeline build (make timeline -> libcookingzoo_hip_tl.so): the shipped kernel plus two reads of the device-wide 100 MHz
    // clock per wave - at its first instruction and behind its last store - and
    // one 16-byte store of lane 0.  No waits are added in between (unlike the phase stamps of `make prof`).
    const uint64_t tl_in = wall_clock64();
#endif
    Params P = P0;
    P.state = e_state; P.actions = e_actions; P.lut = e_lut; P.N = e_N; P.RW = e_RW; P.W = e_W; P.H = e_H; P.D = e_D;
    P.dyn0_off = e_dyn0; P.dyn1_off = e_dyn1;
    constexpr int EPW = envs_per_wg<CPL>();
    __shared__ Lds<CPL> lds_all[EPW];
    __shared__ double lut[LUT_SIZE];
    int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)rfl(threadIdx.x >> 6);
    // (the last workgroup may be partial: its spare waves shadow the last env up to the barrier, then leave)
    const int env_raw = (int)blockIdx.x * EPW + wave;
    const int env = min(env_raw, P.N - 1);
    static_assert(64 * EPW >= LUT_SIZE, "one table entry per thread");
    // the quotient table is shared by the workgroup: its load joins the other loads of the prologue (no branch here:
    // a branch would split the kernel-argument fetch into several dependent round trips)
    const double lutv = ldg<double>(P.lut, min(threadIdx.x, (unsigned)LUT_SIZE - 1u) * 8u);
    // The encode's per-lane constant, fetched in front of the record: the wait for the record then covers it.  Fetched behind
    // the record (as until round 4) its first use - in the image build, behind the reward / termination stores - needed an
    // `s_waitcnt vmcnt(0)`: gfx950 counts loads and stores in one in-order counter, so that wait also stood for the
    // acknowledgement of every store issued before it (in a fused launch: the previous step's observation rows).
    // (The larger instances keep the late fetch, i.e. their round-3 code: their launches are bound by HBM writes, and on the
    // one box where the two forms differed the wait in front of the encode helped - config 5: 318-323 us per launch with it,
    // 338-342 us without.  Config 5 varies by +-4 % between boxes and allocations anyway, profiles/r04/cfg5_boxes.txt.)
    constexpr bool SUBMASK_EARLY = CPL == 1;
    uint32_t submask = 0;
    if (SUBMASK_EARLY) submask = load_submask(P, (int)(threadIdx.x & 63u));
    Lds<CPL> &lds = lds_all[wave];
    Ctx cx{P.W, P.H, P.D, P.W * P.H, lane};
    CZ_STAMP(0);
    uint32_t *rec = P.state + (uint32_t)env * (uint32_t)P.RW;        // (cz_create: the records of a handle stay below 4 GiB)
    double *retp = reinterpret_cast<double *>(rec + RET_WORD0);
    constexpr bool FUSED = FUSED_MODE == 1 || FUSED_MODE == 2 || FUSED_MODE == 4 || FUSED_MODE == 5, EXT = FUSED_MODE == 2;
    constexpr bool CODES = FUSED_MODE == 3 || FUSED_MODE == 4 || FUSED_MODE == 5;
    // mode 5: a fused rollout that writes codes ONLY (cz_rollout_compact without a float64 trajectory).  Without the float64 path -
    // its six descriptor registers, its encode - the codes' own descriptor words fit the registers for the whole launch (mode 4 reloads
    // them every step: held there they cost 136 vector registers, three waves per SIMD)
    constexpr bool CODES_ONLY = FUSED_MODE == 5;
    int av = 0;
#ifdef CZ_TIMELINE
    const uint64_t tl_seen = wall_clock64();
#endif
    // ---- every load of the step is issued here, before anything waits
    if (!FUSED) av = ldg<int>(P.actions, ((uint32_t)env * (uint32_t)NA + (uint32_t)min(lane, NA - 1)) * 4u);
    double ret = ldg<double>(retp, ((uint32_t)lane & 3u) * 8u);                                       // running episode return, lane a = agent a
    Env<OPL, CPL, NA> e;
    load_env(P, e, cx, rec);
    init_lds<CPL>(P, cx, lds);
    uint32_t rowv = load_recipe_rows(P, e.recipes, lane);
    uint32_t dsc[OBS_CHUNK];
    if (!CODES_ONLY && P.obs) load_desc(P, e.layout, 0, lane, dsc);
    if (!SUBMASK_EARLY) submask = load_submask(P, lane);
    const int64_t env_global = P.env_id_base + env;
    bool cells_dirty = false, objs_dirty = false, header_dirty = FUSED;
    bool img_objs = true, img_cells = true;                 // which parts of the LDS image the next encode must rebuild
    if (threadIdx.x < (unsigned)LUT_SIZE) lut[threadIdx.x] = lutv;
    __syncthreads();
    if (env_raw >= P.N) return;
    CZ_STAMP(1);
    // the compact observation's descriptor words (cz_step_device_compact): fetched here, behind the prologue's waits - a branch
    // on a late argument in front of the prologue's loads costs every launch 0.4 us - and hidden by the dynamics
    CodesPrefetch cpre;
    cpre.layout = e.layout;
    if (CODES && (!FUSED || CODES_ONLY)) {
#pragma unroll
        for (int r = 0; r < CODES_PREFETCH; ++r) cpre.d[r] = load_desc4(P, e.layout, 256u * r + 4u * (uint32_t)lane);
    }

    const int T = FUSED ? P.T : 1;
#ifdef CZ_TIMELINE
    uint32_t tl_dbg = 0u;
#endif
    // a fused rollout over caller-supplied actions (cz_rollout_actions: [T][N][A] int32): step t's action words are loaded one
    // step ahead, so the round trip hides behind the previous step's work
    const uint32_t act_lane = ((uint32_t)env * (uint32_t)NA + (uint32_t)min(lane, NA - 1)) * 4u;
    if (EXT) av = ldg<int>(e_actions, act_lane);
#pragma nounroll
    for (int t = 0; t < T; ++t) {
        // The fused kernel re-reads its argument block every step (scalar loads that hit the constant cache): nothing of
        // it then stays live across the loop's back edge, which is what used to spill ~90 SGPRs.
        Params Pt;
        if (FUSED) {
            static_assert(sizeof(Params) % 8 == 0, "copied as 64-bit words");
            typedef uint64_t __attribute__((may_alias)) word_t;
            const __attribute__((address_space(4))) word_t *src = (const __attribute__((address_space(4))) word_t *)CZ_LATE_STEP();
            word_t *dst = reinterpret_cast<word_t *>(&Pt);
#pragma unroll
            for (unsigned i = 0; i < sizeof(Params) / 8; ++i) dst[i] = src[i];
            Pt.state = e_state; Pt.actions = e_actions; Pt.lut = e_lut; Pt.N = e_N; Pt.RW = e_RW; Pt.W = e_W; Pt.H = e_H; Pt.D = e_D;
            Pt.dyn0_off = e_dyn0; Pt.dyn1_off = e_dyn1;
        } else {
            Pt = P;
        }
        if (CODES && FUSED && !CODES_ONLY) {
            // (a fused rollout fetches the code descriptors again at the top of every step - hidden by the step's dynamics - instead
            // of holding eight more registers across the loop: with them the kernel no longer fits four waves per SIMD)
            cpre.layout = e.layout;
#pragma unroll
            for (int r = 0; r < CODES_PREFETCH; ++r) cpre.d[r] = load_desc4(Pt, e.layout, 256u * r + 4u * (uint32_t)lane);
        }
        uint32_t acts;                                             // lane a = action of agent a
        if (!FUSED) acts = (uint32_t)av;
        else if (EXT) {
            acts = (uint32_t)av;
            // (cz_rollout_actions checks that T * N * A * 4 fits 32 bits; the last step re-reads its own row)
            av = ldg<int>(e_actions, act_lane + (uint32_t)min(t + 1, T - 1) * ((uint32_t)Pt.N * (uint32_t)NA * 4u));
        } else acts = action_hash(Pt.seed, env_global, lane & 3, Pt.step0 + (uint32_t)t, SCHEME == 3 ? 5u : 8u);
        Dirty dt{};
        StepOut o;
        step_env<OPL, CPL, NA, SCHEME, FUSED>(Pt, (unsigned)offsetof(StepArgsMirror, p), e, cx, acts, env_global, rowv, lds, dsc, dt, o);
#ifdef CZ_TIMELINE
        tl_dbg |= o.dbg;
#endif
        CZ_STAMP(4);
        cells_dirty |= dt.cells != 0;
        objs_dirty |= (dt.touched | dt.interacted | dt.moved) != 0;
        header_dirty |= o.header;
        // ---- running return (lane a = agent a) and, at episode end only, the per-env statistics
        const double myrew = o.myrew;
        if (o.stepped) ret += myrew;
        const KParams kp = CZ_LATE_STEP();
        // The statistics are a read-modify-write of words in memory (across launches: device-scope loads, write-through stores,
        // like the record).  A wave that ends an episode issues the loads here and adds / stores behind the encode: the round
        // trip (~1 us after a launch boundary) used to make exactly these waves the last ones of a launch in which nobody acts.
        uint32_t su_old = 0u;
        double sf_old = 0.0, ret_done = 0.0;
        if (CZ_RARE(FUSED, o.finished)) {
            const uint32_t *su = kp->stat_u + (size_t)env * SU_WORDS;
            const double *sf = kp->stat_f + (size_t)env * SF_WORDS;
            su_old = ldg<uint32_t>(su, ((uint32_t)lane & 15u) * 4u);
            sf_old = ldg<double>(sf, (uint32_t)(SF_SUM0 + ((uint32_t)lane & 3u)) * 8u);
            ret_done = ret;
            ret = 0.0;
        }
        // ---- outputs of this step.  One wave-uniform base pointer per array and a 32-bit per-lane offset (cz_rollout checks
        // that T * N * A * 8 fits): the `global_store v, v_off, s[base]` form, no 64-bit per-lane address arithmetic.  The
        // one-step kernels store unconditionally - the host hands them a scratch row for an array the caller does not want.
        // (mode 2 with P.step0 & 1 - cz_set_ring_fused: the steps of an action ring fused into one launch - writes every step's
        // outputs to row `env`, like the one-step launches it stands in for; a wave's stores to one address keep their order)
        const bool in_place = EXT && (Pt.step0 & 1u) != 0u;
        const size_t row = FUSED ? ((size_t)(in_place ? 0 : t) * Pt.N + env) : (size_t)env;
        // (the compact path's descriptor words were fetched long ago; waiting for them HERE costs nothing, while behind the
        // stores below the same wait would also stand for those stores' acknowledgement - one counter for loads and stores)
        if (CODES) {
            if (CZ_RARE(FUSED, cpre.layout != e.layout)) {                 // a reset pass has moved the env to another layout
                cpre.layout = e.layout;
#pragma unroll
                for (int r = 0; r < CODES_PREFETCH; ++r) cpre.d[r] = load_desc4(Pt, e.layout, 256u * r + 4u * (uint32_t)lane);
            }
#pragma unroll
            for (int r = 0; r < CODES_PREFETCH; ++r) asm volatile("" :: "v"(cpre.d[r]));
        }
        {
            const uint32_t oidx = (uint32_t)row * (uint32_t)NA + (uint32_t)lane;          // [row][agent]
            double *const rewards = kp->rewards;
            uint8_t *const term = kp->term, *const trunc = kp->trunc;
            if (lane < NA) {
                if (!FUSED || rewards) stg<double>(rewards, oidx * 8u, myrew);
                if (!FUSED || term) stg<uint8_t>(term, oidx, (uint8_t)o.term);
                if (!FUSED || trunc) {
                    stg<uint8_t>(trunc, oidx, (uint8_t)o.trunc);
                    // whoever was despawned in this step is reported truncated once (cooking_env.py:344-349): a second store by
                    // those lanes, on the rare steps on which somebody leaves
                    if (o.gone && ((o.gone >> lane) & 1u)) stg<uint8_t>(trunc, oidx, (uint8_t)1);
                }
            }
        }
        if (!FUSED) {
            uint32_t *const marks_out = kp->marks_out;
            if (marks_out && lane < 2) stg<uint32_t>(marks_out, (2u * (uint32_t)env + (uint32_t)lane) * 4u, lane == 0 ? e.marks : e.marks_hi);   // infos["recipe_done"] of the host API
        }
        CZ_STAMP(5);
        img_objs |= (dt.touched | dt.moved) != 0;
        img_cells |= dt.cells != 0;
        if (Pt.obs || CODES) {
            // (env row x row length: a 32 x 32 -> 64-bit product, two scalar multiplies)
            uint8_t *const codes = CODES ? Pt.codes + (uint64_t)(uint32_t)row * (uint64_t)(uint32_t)(NA * codes_pitch(Pt.F)) : nullptr;
            double *const obs_row = !CODES_ONLY && Pt.obs ? Pt.obs + (uint64_t)(uint32_t)row * (uint64_t)(uint32_t)(NA * Pt.F) : nullptr;
            observe<OPL, CPL, NA, !CODES_ONLY>(Pt, e, cx, lds, lut, dsc, submask, obs_row, img_objs, img_cells, codes, CODES ? &cpre : nullptr);
            img_objs = false; img_cells = false;
        }
        if (CZ_RARE(FUSED, o.finished)) {      // (the state is still that of the finished episode: the reset is the next pass)
            uint32_t *su = kp->stat_u + (size_t)env * SU_WORDS;
            double *sf = kp->stat_f + (size_t)env * SF_WORDS;
            const uint32_t a_of = (uint32_t)lane - SU_COMPLETED0;               // lane SU_COMPLETED0 + a: recipe a completed?
            const uint32_t root = Pt.wide ? (((a_of < 2u ? e.marks : e.marks_hi) >> (16u * (a_of & 1u))) & 1u) : ((e.marks >> (8u * (a_of & 3u))) & 1u);
            const uint32_t inc = lane == (int)SU_EPISODES ? 1u : lane == (int)SU_LENSUM ? e.t : lane == (int)SU_TRUNC ? (uint32_t)o.trunc
                                 : lane == (int)SU_TERM ? (uint32_t)o.term : a_of < (uint32_t)NA ? root : 0u;
            if (lane < (int)SU_COMPLETED0 + NA && lane != (int)SU_STEPS) stg<uint32_t>(su, (uint32_t)lane * 4u, su_old + inc);
            if (lane < NA) stg<double>(sf, (uint32_t)(SF_SUM0 + lane) * 8u, sf_old + ret_done);
        }
        CZ_STAMP(6);
    }
    store_env(P, e, cx, rec, cells_dirty, objs_dirty, header_dirty);
    if (lane < NA) stg<double>(retp, (uint32_t)lane * 8u, ret);
    CZ_STAMP(7);
#ifdef CZ_TIMELINE
    {
        // word 0: entry time (low 32 bits) | HW_ID << 32 (wave, SIMD, CU, SH, SE);  word 1: exit time | XCC_ID << 32
        const uint64_t tl_out = wall_clock64();
        const uint32_t hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
        unsigned long long *const tl = CZ_LATE_STEP()->timeline;
        if (tl && lane == 0) {
            typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
            ull2 v;
            v.x = (tl_in & 0xFFFFFFFFull) | ((unsigned long long)(hw_id & 0xFFFFu) << 32) | ((unsigned long long)tl_dbg << 48);
            v.y = (tl_out & 0xFFFFFFFFull) | ((unsigned long long)(xcc & 15u) << 32) | (((tl_seen - tl_in) & 0xFFFFFFFull) << 36);
            *reinterpret_cast<ull2 *>(tl + 2 * (size_t)env) = v;
        }
    }
#endif
}

// (experiment switch: occupancy target of the one-step kernels of the small and the middle instance.  All of them reach six
// waves per SIMD by themselves except the middle instance with four agents - config 5's kernel: 82 vector registers, four
// waves - and forcing that one to six (80 registers, 3 spilled) changed nothing: 195 against 197 M env-steps/s,
// profiles/r03/wpe_ab.txt)
#ifdef CZ_STEP_MIN_WPE
#define CZ_STEP_ATTR __attribute__((amdgpu_waves_per_eu(((FUSED == 0 || FUSED == 3) && OPL <= 2) ? CZ_STEP_MIN_WPE : 1)))
#else
#define CZ_STEP_ATTR
#endif
template <int OPL, int CPL, int NA, int SCHEME, int FUSED>
__global__ __launch_bounds__(64 * envs_per_wg<CPL>()) CZ_STEP_ATTR void k_step(uint32_t *e_state, const int32_t *e_actions, const double *e_lut, int32_t e_N,
                                                          int32_t e_RW, int32_t e_W, int32_t e_H, int32_t e_D, int32_t e_dyn0,
                                                          int32_t e_dyn1, const Params P0) {
    step_kernel<OPL, CPL, NA, SCHEME, FUSED>(e_state, e_actions, e_lut, e_N, e_RW, e_W, e_H, e_D, e_dyn0, e_dyn1, P0);
}
// reset(): cooking_env.py:178-210 for envs [env_begin, env_begin + count)
template <int OPL, int CPL, int NA>
__global__ __launch_bounds__(64) void k_reset(const Params P, int64_t env_begin, const int32_t *__restrict__ layout_ids,
                                              const uint32_t *__restrict__ recipe_words, const uint32_t *__restrict__ pool_words,
                                              double *obs_out) {
    __shared__ Lds<CPL> lds;
    __shared__ double lut[LUT_SIZE];
    init_lut(P, lut, (int)threadIdx.x, 64);
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
    if (P.auto_reset & 2)
        e.status = spawn_initial_status(rfl(reinterpret_cast<const SpawnCfg *>(reinterpret_cast<const char *>(P.lut) + SPAWN_CFG_OFFSET)->grace_period), NA);
    e.pool = rfl(pool_words[i]);
    uint32_t rowv = load_recipe_rows(P, e.recipes, lane);
    all_marks(P, e, cx, rowv, lds);
    store_env(P, e, cx, rec, true, true);
    if (lane < MAX_AGENTS) reinterpret_cast<double *>(rec + RET_WORD0)[lane] = 0.0;
    if (obs_out) {
        uint32_t dsc[OBS_CHUNK];
        load_desc(P, e.layout, 0, lane, dsc);
        observe(P, e, cx, lds, lut, dsc, load_submask(P, lane), obs_out + (size_t)i * NA * P.F);
    }
}

// observe() only (after cz_set_state)
template <int OPL, int CPL, int NA>
__global__ __launch_bounds__(64) void k_observe(const Params P, int64_t env_begin, double *obs_out, uint8_t *codes_out) {
    __shared__ Lds<CPL> lds;
    __shared__ double lut[LUT_SIZE];
    init_lut(P, lut, (int)threadIdx.x, 64);
    const int i = blockIdx.x;
    const int lane = (int)threadIdx.x;
    Ctx cx{P.W, P.H, P.D, P.W * P.H, lane};
    init_lds<CPL>(P, cx, lds);
    Env<OPL, CPL, NA> e;
    load_env(P, e, cx, P.state + (size_t)(env_begin + i) * P.RW);
    uint32_t dsc[OBS_CHUNK];
    load_desc(P, e.layout, 0, lane, dsc);
    observe(P, e, cx, lds, lut, dsc, load_submask(P, lane), obs_out ? obs_out + (size_t)i * NA * P.F : nullptr, true, true,
            codes_out ? codes_out + (size_t)i * NA * codes_pitch(P.F) : nullptr);
}

// launchers exported by each instantiation unit
enum : int { LAUNCH_ONE = 0, LAUNCH_FUSED = 1 };
struct Launchers {
    // mode: LAUNCH_ONE = one step; LAUNCH_FUSED = P.T steps per launch (actions: P.actions, or the on-device stream when null)
    hipError_t (*step)(const Params &, hipStream_t, int mode);
    hipError_t (*reset)(const Params &, hipStream_t, int64_t, int, const int32_t *, const uint32_t *, const uint32_t *, double *);
    hipError_t (*observe)(const Params &, hipStream_t, int64_t, int, double *, uint8_t *);
};

template <int OPL, int CPL>
struct Inst {
    template <int NA>
    static hipError_t step_na(const Params &P, hipStream_t st, int mode) {
        const bool fused = mode == LAUNCH_FUSED;
        constexpr int EPW = envs_per_wg<CPL>();
        const dim3 grid((unsigned)((P.N + EPW - 1) / EPW)), block(64 * EPW);
        const Early E = early_of(P);
#define CZ_LAUNCH_STEP(S, F) \
    hipLaunchKernelGGL((k_step<OPL, CPL, NA, S, F>), grid, block, 0, st, E.state, E.actions, E.lut, E.N, E.RW, E.W, E.H, E.D, E.dyn0_off, E.dyn1_off, P)
        if (fused && P.actions) {
            if (P.scheme == 3) CZ_LAUNCH_STEP(3, 2);
            else CZ_LAUNCH_STEP(1, 2);
        } else if (fused && P.codes && !P.obs) {
            if (P.scheme == 3) CZ_LAUNCH_STEP(3, 5);
            else CZ_LAUNCH_STEP(1, 5);
        } else if (fused && P.codes) {
            if (P.scheme == 3) CZ_LAUNCH_STEP(3, 4);
            else CZ_LAUNCH_STEP(1, 4);
        } else if (fused) {
            if (P.scheme == 3) CZ_LAUNCH_STEP(3, 1);
            else CZ_LAUNCH_STEP(1, 1);
        } else if (!P.actions) {
            return hipErrorInvalidValue;
        } else if (P.codes) {
            if (P.scheme == 3) CZ_LAUNCH_STEP(3, 3);
            else CZ_LAUNCH_STEP(1, 3);
        } else {
            if (P.scheme == 3) CZ_LAUNCH_STEP(3, 0);
            else CZ_LAUNCH_STEP(1, 0);
        }
#undef CZ_LAUNCH_STEP
        return hipGetLastError();
    }
    static hipError_t step(const Params &P, hipStream_t st, int mode) {
        switch (P.A) {
        case 1: return step_na<1>(P, st, mode);
        case 2: return step_na<2>(P, st, mode);
        case 3: return step_na<3>(P, st, mode);
        default: return step_na<4>(P, st, mode);
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
    static hipError_t observe(const Params &P, hipStream_t st, int64_t b, int n, double *obs, uint8_t *codes) {
        switch (P.A) {
        case 1: hipLaunchKernelGGL((k_observe<OPL, CPL, 1>), dim3(n), dim3(64), 0, st, P, b, obs, codes); break;
        case 2: hipLaunchKernelGGL((k_observe<OPL, CPL, 2>), dim3(n), dim3(64), 0, st, P, b, obs, codes); break;
        case 3: hipLaunchKernelGGL((k_observe<OPL, CPL, 3>), dim3(n), dim3(64), 0, st, P, b, obs, codes); break;
        default: hipLaunchKernelGGL((k_observe<OPL, CPL, 4>), dim3(n), dim3(64), 0, st, P, b, obs, codes); break;
        }
        return hipGetLastError();
    }
};

Launchers launchers_small();   // D <= 64 slots, W*H <= 64 cells
Launchers launchers_large();   // D <= 128 slots, W*H <= 256 cells
Launchers launchers_huge();    // D <= 255 slots, W*H <= 1024 cells

}  // namespace cz
