// How many workgroups of the step kernel's shape (512 threads, 34 432 B of LDS, ~106 SGPRs) are resident at once?
// Every workgroup bumps a counter, keeps the maximum it has seen, holds its slot for ~100 us, and leaves.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/residency_probe tools/residency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int LDS_BYTES>
__global__ __launch_bounds__(512) void k_hold(unsigned *now, unsigned *peak, unsigned long long hold_ticks) {
    __shared__ char pad[LDS_BYTES];
    pad[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned v = atomicAdd(now, 1u) + 1u;
        atomicMax(peak, v);
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(now, 1u);
    if (pad[(threadIdx.x * 7) & 511] == 123 && hold_ticks == 1) now[1] = 1;
}

int main() {
    unsigned *d;
    CHK(hipMalloc(&d, 64));
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d, LDS per CU (maxSharedMemoryPerMultiProcessor) %zu, per block %zu\n", p.multiProcessorCount, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock);
    auto run = [&](auto kern, const char *name, int blocks) {
        CHK(hipMemset(d, 0, 64));
        int occ = 0;
        CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 512, 0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, d + 2, 10000ull);     // 100 us at 100 MHz
        CHK(hipDeviceSynchronize());
        unsigned h[4];
        CHK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("%s: %d workgroups launched, peak resident %u (occupancy query: %d per CU = %d)\n", name, blocks, h[2], occ, occ * p.multiProcessorCount);
    };
    run(k_hold<34432>, "512 threads, 34432 B LDS", 2048);
    run(k_hold<16384>, "512 threads, 16384 B LDS", 4096);
    run(k_hold<60000>, "512 threads, 60000 B LDS", 2048);
    return 0;
}
