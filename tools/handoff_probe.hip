// Experiment: can the launch boundary between dependent step kernels be replaced by a PER-ENV hand-off?
//
// Step k+1 of env e only depends on step k of env e.  If consecutive step kernels are allowed to overlap (two streams
// used alternately, or hipExtAnyOrderLaunch on one stream) and every wave waits for its own env's sequence word instead
// of the whole previous kernel, the ~1.6 us launch boundary and the start skew could hide behind the other kernel's work.
// The hand-off itself costs a device-scope store that has to become visible to a polling wave on possibly another XCD.
//
// This probe runs a stand-in for the step kernel (512 workgroups x 8 waves, one "env" per wave, a dependent ALU chain of
// a calibrated length, 4 KB of output per env, a sequence word per env 64 B apart) in three launch patterns and prints the
// wall time per step:
//   P1  one stream, ordinary launches (the kernel boundary orders the steps)                  -- what the product does
//   P2  two streams used alternately, every wave spins on seq[env] == k before it works
//   P3  one stream, hipExtAnyOrderLaunch, same spin
//   P4  as P2, but the state words are read by the poll itself (they share the sequence word's 64-byte line)
// Every spin has a wall-clock deadline.  build: hipcc --offload-arch=gfx950 -O2 -o /tmp/handoff_probe tools/handoff_probe.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                                   \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                             \
        }                                                                                        \
    } while (0)

constexpr int NWG = 512, WG_THREADS = 512, NENV = NWG * 8;
constexpr uint64_t DEADLINE_TICKS = 100000000ull / 5;        // 0.2 s of the 100 MHz wall clock

typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(WG_THREADS) void k_work(uint32_t *seq, uint32_t expect, int spin, int iters, uint4_t *out, uint32_t *err) {
    const int lane = (int)(threadIdx.x & 63u);
    const int env = (int)blockIdx.x * 8 + (int)(threadIdx.x >> 6);
    uint32_t *my = seq + (size_t)env * 16;
    uint32_t merged = 0;
    if (spin == 2) {
        // mailbox style: the state words share the sequence word's 64-byte line and come back with the poll itself (one
        // load per lane covers the line; a line read that shows the new number was served after the data had landed)
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const uint64_t t0 = wall_clock64();
        for (;;) {
            merged = __hip_atomic_load(my + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_readfirstlane(merged) == expect) break;
            if (wall_clock64() - t0 > DEADLINE_TICKS) { if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            __builtin_amdgcn_s_sleep(2);
        }
    } else if (spin) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // somebody gave up: drain quickly
        const uint64_t t0 = wall_clock64();
        while (__hip_atomic_load(my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != expect) {
            if (wall_clock64() - t0 > DEADLINE_TICKS) { if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            __builtin_amdgcn_s_sleep(2);
        }
        // no acquire fence: on this multi-XCD part it would invalidate the whole L2; the state below is read with
        // device-scope (cache-bypassing) loads instead
        asm volatile("" ::: "memory");
    }
    // the "state" of the env: read what the previous step left, work on it, write it back (write-through when overlapping)
    uint32_t x = spin == 2 ? (uint32_t)__builtin_amdgcn_ds_bpermute((1 + (lane & 7)) << 2, (int)merged) + (uint32_t)lane
                           : __hip_atomic_load(my + 1 + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (uint32_t)lane;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u + (x >> 7);
    uint4_t v = {x, x ^ 1u, x ^ 2u, x ^ 3u};
    uint4_t *o = out + (size_t)env * 256 + lane;               // 4 KB per env
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v, o + 64 * j);
    if (lane < 8) __hip_atomic_store(my + 1 + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // no release fence either (an agent-scope release writes back the whole L2: 120 us per launch when every wave does it):
    // the state went out with device-scope write-through stores; wait for their acknowledgement, then publish
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(my, expect + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 0;
    CHK(hipSetDevice(0));
    hipStream_t s[2];
    CHK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    uint32_t *seq, *err;
    uint4_t *out;
    CHK(hipMalloc(&seq, (size_t)NENV * 64));
    CHK(hipMalloc(&err, 256));
    CHK(hipMalloc(&out, (size_t)NENV * 4096));
    const int K = 1000;
    auto reset = [&]() { CHK(hipMemset(seq, 0, (size_t)NENV * 64)); CHK(hipMemset(err, 0, 256)); CHK(hipDeviceSynchronize()); };
    auto p1 = [&](int it) {
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_work, dim3(NWG), dim3(WG_THREADS), 0, s[0], seq, (uint32_t)k, 0, it, out, err);
        CHK(hipStreamSynchronize(s[0]));
    };
    auto p2 = [&](int it) {
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_work, dim3(NWG), dim3(WG_THREADS), 0, s[k & 1], seq, (uint32_t)k, 1, it, out, err);
        CHK(hipStreamSynchronize(s[0])); CHK(hipStreamSynchronize(s[1]));
    };
    auto p4 = [&](int it) {
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_work, dim3(NWG), dim3(WG_THREADS), 0, s[k & 1], seq, (uint32_t)k, k == 0 ? 1 : 2, it, out, err);
        CHK(hipStreamSynchronize(s[0])); CHK(hipStreamSynchronize(s[1]));
    };
    auto p3 = [&](int it) {
        for (int k = 0; k < K; ++k)
            hipExtLaunchKernelGGL(k_work, dim3(NWG), dim3(WG_THREADS), 0, s[0], nullptr, nullptr, k == 0 ? 0 : hipExtAnyOrderLaunch, seq, (uint32_t)k, 1, it, out, err);
        CHK(hipStreamSynchronize(s[0]));
    };
    if (iters == 0) {                                          // calibrate the ALU chain: P1 at ~6.2 us per launch
        for (int it : {200, 400, 600, 800, 1000, 1200}) {
            reset(); p1(it); reset();
            double t0 = now(); p1(it); double t1 = now();
            printf("calibration: iters %4d -> %.3f us per launch (P1)\n", it, (t1 - t0) / K * 1e6);
            if ((t1 - t0) / K * 1e6 >= 6.0 && iters == 0) iters = it;
        }
        if (iters == 0) iters = 1200;
        printf("using iters = %d\n", iters);
    }
    const char *names[4] = {"P1 one stream, kernel boundaries", "P2 two streams alternately + per-env spin", "P3 one stream, any-order launches + per-env spin",
                            "P4 as P2, state in the sequence word's line (one round trip less)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int p = 0; p < 4; ++p) {
            reset();
            if (p == 0) p1(iters); else if (p == 1) p2(iters); else if (p == 2) p3(iters); else p4(iters);
            reset();
            double t0 = now();
            if (p == 0) p1(iters); else if (p == 1) p2(iters); else if (p == 2) p3(iters); else p4(iters);
            double t1 = now();
            uint32_t herr = 0, last = 0;
            CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(&last, seq + (size_t)(NENV - 1) * 16, 4, hipMemcpyDeviceToHost));
            printf("%-52s %.3f us per step  (err %u, last seq %u of %d)\n", names[p], (t1 - t0) / K * 1e6, herr, last, K);
        }
    return 0;
}
