// How fast does s_memtime tick?  (calibrates tools/phase_profile.py)   hipcc --offload-arch=gfx950 -O2 -o memtime_calib memtime_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long ticks, unsigned long long *out) {
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - m0 < ticks) {}
    out[0] = __builtin_readcyclecounter() - t0;
    out[1] = wall_clock64() - w0;
    out[2] = __builtin_amdgcn_s_memtime() - m0;
}
__global__ void empty() {}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned long long ticks : {100000ull, 1000000ull, 10000000ull}) {
        spin<<<1, 64>>>(ticks, d);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spin<<<1, 64>>>(ticks, d);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("memtime ticks %llu: event %.1f us  -> %.1f ticks/us | readcyclecounter %llu | wall_clock64 %llu (%.1f MHz)\n", ticks, ms * 1e3, h[2] / (ms * 1e3), h[0], h[1], h[1] / (ms * 1e3));
    }
    // back-to-back empty kernels: the floor of one launch
    for (int i = 0; i < 100; ++i) empty<<<512, 512>>>();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 1000; ++i) empty<<<512, 512>>>();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("empty kernel <<<512,512>>> back to back: %.2f us per launch\n", ms);
    hipEventRecord(a);
    for (int i = 0; i < 1000; ++i) empty<<<1, 64>>>();
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("empty kernel <<<1,64>>> back to back: %.2f us per launch\n", ms);
    return 0;
}
