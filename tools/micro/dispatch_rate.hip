// How fast can the chip start (and retire) wavefronts?   hipcc --offload-arch=gfx950 -O2 -o dispatch_rate dispatch_rate.hip
// empty kernels of G workgroups x B threads, with and without a static LDS allocation, back to back on one stream
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty() {}
template <int BYTES>
__global__ void with_lds(int *out) {
    __shared__ char buf[BYTES];
    if (out) out[0] = buf[threadIdx.x];          // never true: keeps the allocation alive
}
__global__ void spin(unsigned long long ticks, int *out) {
    __shared__ char buf[35 * 1024];
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - m0 < ticks) {}
    if (out) out[0] = buf[threadIdx.x];
}
template <class F>
static float time_us(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
int main() {
    const int grids[] = {512, 2048, 8192, 32768};
    for (int g : grids) {
        float t = time_us([&] { empty<<<g, 512>>>(); }, 200);
        printf("empty            <<<%6d, 512>>>: %8.2f us  -> %7.1f waves/us\n", g, t, g * 8 / t);
    }
    for (int g : grids) {
        float t = time_us([&] { with_lds<35 * 1024><<<g, 512>>>(nullptr); }, 200);
        printf("35 KB LDS        <<<%6d, 512>>>: %8.2f us  -> %7.1f waves/us\n", g, t, g * 8 / t);
    }
    for (int g : {4096, 65536}) {
        float t = time_us([&] { empty<<<g, 64>>>(); }, 200);
        printf("empty            <<<%6d,  64>>>: %8.2f us  -> %7.1f waves/us\n", g, t, g / t);
    }
    for (int g : {512, 2048, 8192}) {
        float t = time_us([&] { spin<<<g, 512>>>(12000ull, nullptr); }, 50);   // 12000 ticks = 5 us of wave lifetime
        printf("5 us spin, 35 KB <<<%6d, 512>>>: %8.2f us  (ideal with 4 workgroups per CU: %.1f us)\n", g, t, 5.0 * ((g + 1023) / 1024));
    }
    return 0;
}
