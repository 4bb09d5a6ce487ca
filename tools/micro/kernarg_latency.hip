// What does the kernel-argument fetch cost per launch?   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value
// Back-to-back launches of kernels that must read their LAST argument word before they can exit, for by-value structs of
// different sizes, against an empty kernel and a kernel that chases one pointer held in its arguments.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int WORDS> struct Arg { int *p; int w[WORDS - 2]; };
__global__ void empty() {}
template <int WORDS> __global__ void uses(Arg<WORDS> a) { if (a.w[WORDS - 3] == 12345) a.p[0] = 1; }
// second-level: a 256-byte block behind a pointer in the arguments
__global__ void chase(const int *block, int *out) { if (block[63] == 12345) out[0] = 1; }
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
    int *blk; hipMalloc(&blk, 256); hipMemset(blk, 0, 256);
    for (int g : {512}) {
        printf("grid %d x 512: empty %.2f us | by-value struct of 16 B %.2f | 64 B %.2f | 128 B %.2f | 256 B %.2f | 384 B %.2f | pointer to a 256 B block %.2f\n", g,
               time_us([&] { empty<<<g, 512>>>(); }, 1000),
               time_us([&] { uses<4><<<g, 512>>>(Arg<4>{}); }, 1000), time_us([&] { uses<16><<<g, 512>>>(Arg<16>{}); }, 1000),
               time_us([&] { uses<32><<<g, 512>>>(Arg<32>{}); }, 1000), time_us([&] { uses<64><<<g, 512>>>(Arg<64>{}); }, 1000),
               time_us([&] { uses<96><<<g, 512>>>(Arg<96>{}); }, 1000), time_us([&] { chase<<<g, 512>>>(blk, nullptr); }, 1000));
    }
    return 0;
}
