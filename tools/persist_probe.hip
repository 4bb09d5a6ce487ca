// Experiment: what would a PERSISTENT step kernel pay per step for its handshake?
//
// A persistent kernel keeps the env records in registers and, per step, (1) waits until the step's actions are there,
// (2) steps, (3) tells the consumer that every env of the batch is done.  (1) and (3) replace the launch boundary of the
// one-kernel-per-step design (1.8 us in graph replay on this part).  This probe measures them with NO work in between:
//
//   M1  device-only handshake: a one-wave "driver" kernel publishes step k in a flag, a consumer grid shaped like the step
//       kernel (512 workgroups x 8 waves, every wave polls on its own = one env per wave) sees it and reports completion;
//       the driver polls the completion counters.  Three completion schemes: one counter, one counter per XCD-sized group
//       of workgroups (8), one counter per 8 workgroups (64).
//   M2  the same handshake driven from a HIP stream, which is what a caller with its own kernels (a policy network) has:
//       hipStreamWriteValue32(flag, k) ; hipStreamWaitValue32(done >= k) per step, between two empty kernels.
//   M0  the baseline: an empty kernel of the step kernel's grid shape, K launches in one HIP graph (the launch boundary).
//
// Every spin loop has a wall-clock deadline, so a scheduling surprise ends the probe instead of hanging the GPU.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/persist_probe tools/persist_probe.hip && /tmp/persist_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>

#define CHK(x)                                                                                   \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                             \
        }                                                                                        \
    } while (0)

constexpr int NWG = 512, WG_THREADS = 512;
constexpr uint64_t DEADLINE_TICKS = 100000000ull / 4;        // wall_clock64 runs at 100 MHz: 0.25 s

// polls are relaxed device-scope loads (an acquire load would invalidate the caches on every poll); one fence after success
__device__ __forceinline__ uint32_t ld_dev(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void acquire_dev() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
__device__ __forceinline__ void st_dev(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// consumer: every wave waits for step k on its own, the workgroup reports once all its waves have seen it
// pollers = 8: every wave polls (one env per wave, waves independent); pollers = 1: wave 0 polls, the barrier releases the rest
__global__ __launch_bounds__(WG_THREADS) void k_consumer(const uint32_t *flag, uint32_t *done, uint32_t *err, int K, int groups, int pollers) {
    const uint64_t t0 = wall_clock64();
    uint32_t *my = done + ((int)blockIdx.x % groups) * 32;       // counters 128 B apart
    const bool poll = (int)(threadIdx.x >> 6) < pollers;
    for (int k = 1; k <= K; ++k) {
        if (poll) {
            while (ld_dev(flag) < (uint32_t)k) {
                if (wall_clock64() - t0 > DEADLINE_TICKS) { if (threadIdx.x == 0) st_dev(err, 1u); break; }
            }
            acquire_dev();
        }
        __syncthreads();
        if (ld_dev(err)) return;
        if (threadIdx.x == 0) __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// driver (one wave): publish step k, wait until all workgroups reported it; lane g polls counter g
__global__ __launch_bounds__(64) void k_driver(uint32_t *flag, const uint32_t *done, uint32_t *err, int K, int groups, uint64_t *stamps) {
    const uint64_t t0 = wall_clock64();
    const int lane = (int)threadIdx.x;
    const uint32_t per_group = (uint32_t)(NWG / groups);
    for (int k = 1; k <= K; ++k) {
        if (lane == 0) { st_dev(flag, (uint32_t)k); stamps[k] = wall_clock64(); }
        for (;;) {
            bool ok = lane >= groups || ld_dev(done + lane * 32) >= per_group * (uint32_t)k;
            if (__all(ok)) { acquire_dev(); break; }
            if (wall_clock64() - t0 > DEADLINE_TICKS) { if (lane == 0) st_dev(err, 2u); return; }
        }
    }
    if (lane == 0) stamps[K + 1] = wall_clock64();
}

// responder for M2 (one wave): answers flag >= k with done = k
__global__ __launch_bounds__(64) void k_responder(const uint32_t *flag, uint32_t *done, uint32_t *err, int K) {
    const uint64_t t0 = wall_clock64();
    if (threadIdx.x != 0) return;
    for (int k = 1; k <= K; ++k) {
        while (ld_dev(flag) < (uint32_t)k) {
            if (wall_clock64() - t0 > 8 * DEADLINE_TICKS) { st_dev(err, 3u); return; }
        }
        __hip_atomic_store(done, (uint32_t)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(WG_THREADS) void k_empty(uint32_t *sink) {
    if (sink && threadIdx.x == 0 && blockIdx.x == 0xFFFFFFu) *sink = 1;
}

int main() {
    CHK(hipSetDevice(0));
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    uint32_t *flag, *done, *err;
    uint64_t *stamps;
    const int K = 2000;
    CHK(hipMalloc(&flag, 256));
    CHK(hipMalloc(&done, 64 * 128));
    CHK(hipMalloc(&err, 256));
    CHK(hipMalloc(&stamps, (K + 2) * 8));
    std::vector<uint64_t> h(K + 2);

    // ---- M0: empty kernel, K launches in graphs of 200
    {
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(NWG), dim3(WG_THREADS), 0, s1, (uint32_t *)nullptr);
        CHK(hipStreamEndCapture(s1, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipGraphLaunch(ge, s1));
        CHK(hipStreamSynchronize(s1));
        CHK(hipEventRecord(e0, s1));
        for (int i = 0; i < 10; ++i) CHK(hipGraphLaunch(ge, s1));
        CHK(hipEventRecord(e1, s1));
        CHK(hipEventSynchronize(e1));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("M0 empty kernel (512 x 512 threads), graph replay: %.3f us per launch\n", ms * 1e3 / 2000.0);
    }

    // ---- M1: device-only handshake
    for (int pollers : {8, 1})
    for (int groups : {1, 8, 64}) {
        CHK(hipMemset(flag, 0, 256)); CHK(hipMemset(done, 0, 64 * 128)); CHK(hipMemset(err, 0, 256));
        CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_consumer, dim3(NWG), dim3(WG_THREADS), 0, s1, flag, done, err, K, groups, pollers);
        hipLaunchKernelGGL(k_driver, dim3(1), dim3(64), 0, s2, flag, done, err, K, groups, stamps);
        CHK(hipStreamSynchronize(s2));
        CHK(hipStreamSynchronize(s1));
        uint32_t herr = 0;
        CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(h.data(), stamps, (K + 2) * 8, hipMemcpyDeviceToHost));
        if (herr) { printf("M1 groups=%d: deadline hit (err %u)\n", groups, herr); continue; }
        // skip the first 100 rounds (the consumer grid may still be arriving)
        double us = (double)(h[K + 1] - h[101]) / 100.0 / (double)(K - 100);
        printf("M1 device-only handshake, %d polling wave(s) per workgroup, %2d completion counter(s): %.3f us per round (no work)\n", pollers, groups, us);
    }

    // ---- M2: stream-driven handshake
    {
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
        printf("M2 hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
        uint32_t *sig = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
        if (!can || e != hipSuccess) {
            printf("M2 skipped: stream wait-value not available (%s)\n", hipGetErrorString(e));
        } else {
            const int K2 = 500;
            CHK(hipMemset(flag, 0, 256)); CHK(hipMemset(sig, 0, 8)); CHK(hipMemset(err, 0, 256));
            CHK(hipDeviceSynchronize());
            hipLaunchKernelGGL(k_responder, dim3(1), dim3(64), 0, s2, flag, sig, err, K2);
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            CHK(hipEventRecord(e0, s1));
            for (int k = 1; k <= K2; ++k) {
                hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s1, (uint32_t *)nullptr);          // "policy" kernel
                CHK(hipStreamWriteValue32(s1, flag, (uint32_t)k, 0));
                CHK(hipStreamWaitValue32(s1, sig, (uint32_t)k, hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
            CHK(hipEventRecord(e1, s1));
            {   // watchdog: if the chain does not drain within 3 s, satisfy every wait from a third stream
                hipStream_t s3;
                CHK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
                int spins = 0;
                while (hipEventQuery(e1) == hipErrorNotReady && spins < 3000) { usleep(1000); ++spins; }
                if (spins >= 3000) {
                    printf("M2 watchdog fired: releasing the waits\n");
                    CHK(hipMemsetD32Async((hipDeviceptr_t)sig, 0x7FFFFFFF, 1, s3));
                    CHK(hipMemsetD32Async((hipDeviceptr_t)flag, 0x7FFFFFFF, 1, s3));
                    CHK(hipStreamSynchronize(s3));
                }
            }
            CHK(hipEventSynchronize(e1));
            CHK(hipStreamSynchronize(s2));
            float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t herr = 0;
            CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            printf("M2 stream-driven: tiny kernel + write-value + wait-value: %.3f us per round (err %u)\n", ms * 1e3 / K2, herr);
            // reference chain on one stream: tiny kernel + empty step-shaped kernel
            CHK(hipEventRecord(e0, s1));
            for (int k = 1; k <= K2; ++k) {
                hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s1, (uint32_t *)nullptr);
                hipLaunchKernelGGL(k_empty, dim3(NWG), dim3(WG_THREADS), 0, s1, (uint32_t *)nullptr);
            }
            CHK(hipEventRecord(e1, s1));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("M2 reference: tiny kernel + empty step-shaped kernel, direct launches: %.3f us per round\n", ms * 1e3 / K2);
        }
    }
    return 0;
}
