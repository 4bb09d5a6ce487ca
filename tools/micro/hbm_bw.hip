// What does this box's HBM actually deliver?   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o hbm_bw hbm_bw.hip
// streaming fill (write only, the step path's pattern), copy (read + write) and read-sum over 2 GiB buffers, 16 bytes per lane
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void fill(u4 *__restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = u4{1, 2, 3, 4};
}
__global__ void fill_nt(u4 *__restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(u4{1, 2, 3, 4}, &d[i]);
}
// write-through (sc1) stores, the form the step kernel uses for its observations
__global__ void fill_wt(u4 *__restrict__ d, size_t n) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(d, 0, 0x7FFFFFFF, 0x00020000);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n && i < (0x7FFFFFFFull >> 4); i += (size_t)gridDim.x * blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(u4{1, 2, 3, 4}, rs, (unsigned)(i * 16), 0, 16);
}
__global__ void copy(u4 *__restrict__ d, const u4 *__restrict__ s, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void rsum(const u4 *__restrict__ s, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { u4 v = s[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    u4 *a, *b; unsigned *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    const int grid = 256 * 16, block = 256;
    float tf = time_ms([&] { fill<<<grid, block>>>(a, n); }, 20);
    float tn = time_ms([&] { fill_nt<<<grid, block>>>(a, n); }, 20);
    const size_t nw = ((size_t)1 << 30) / 16;                       // (32-bit buffer offsets: 1 GiB)
    float tw = time_ms([&] { fill_wt<<<grid, block>>>(a, nw); }, 20);
    printf("fill variants: plain %.0f GB/s | nontemporal %.0f GB/s | write-through sc1 (1 GiB) %.0f GB/s\n", bytes / tf / 1e6, bytes / tn / 1e6, nw * 16 / tw / 1e6);
    for (int g : {512, 1024, 2048, 8192, 32768}) {
        float t = time_ms([&] { fill<<<g, 512>>>(a, n); }, 10);
        printf("  plain fill, grid %5d x 512: %.0f GB/s\n", g, bytes / t / 1e6);
    }
    float tc = time_ms([&] { copy<<<grid, block>>>(b, a, n); }, 20);
    float tr = time_ms([&] { rsum<<<grid, block>>>(a, n, o); }, 20);
    printf("HBM streaming over 2 GiB: fill %.0f GB/s | copy %.0f GB/s (read + write bytes) | read %.0f GB/s\n",
           bytes / tf / 1e6, 2.0 * bytes / tc / 1e6, bytes / tr / 1e6);
    // the step path's burst size: 18.9 MB of write-only output per launch
    const size_t small = 19 << 20;
    float ts = time_ms([&] { fill<<<grid, block>>>(a, small / 16); }, 200);
    printf("fill of 19 MiB (one launch of the 4096-env step writes this much): %.2f us per launch -> %.0f GB/s\n", ts * 1e3, small / ts / 1e6);
    float tsw = time_ms([&] { fill_wt<<<512, 512>>>(a, small / 16); }, 200);
    printf("same with write-through stores, 512 x 512 threads (the step kernel's shape): %.2f us per launch -> %.0f GB/s\n", tsw * 1e3, small / tsw / 1e6);
    float tsp = time_ms([&] { fill<<<512, 512>>>(a, small / 16); }, 200);
    printf("same with plain stores, 512 x 512 threads: %.2f us per launch -> %.0f GB/s\n", tsp * 1e3, small / tsp / 1e6);
    return 0;
}
