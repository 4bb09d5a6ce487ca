// Does a hipGraph shorten a chain of dependent launches?   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o graph_launch graph_launch.hip
// 1000 dependent launches of (a) an empty kernel, (b) a kernel that reads one argument and writes 19 MiB (the step's shape):
// issued one by one on a stream vs. captured once into a graph and replayed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void empty() {}
__global__ void fill(u4 *d, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = u4{1, 2, 3, 4};
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    u4 *buf; hipMalloc(&buf, 19 << 20);
    const unsigned n = (19u << 20) / 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int K = 1000;
    for (int which = 0; which < 2; ++which) {
        auto launch = [&] { if (which == 0) empty<<<512, 512, 0, st>>>(); else fill<<<512, 512, 0, st>>>(buf, n); };
        for (int i = 0; i < 50; ++i) launch();
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(a, st);
        for (int i = 0; i < K; ++i) launch();
        hipEventRecord(b, st);
        auto t1 = std::chrono::steady_clock::now();
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / K;
        // capture the same chain
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < K; ++i) launch();
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        auto t2 = std::chrono::steady_clock::now();
        hipEventRecord(a, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(b, st);
        auto t3 = std::chrono::steady_clock::now();
        hipEventSynchronize(b);
        float msg; hipEventElapsedTime(&msg, a, b);
        double host_g = std::chrono::duration<double, std::micro>(t3 - t2).count() / K;
        printf("%s: stream %.2f us per launch on the GPU (host %.2f us per launch) | graph of %d nodes %.2f us per node (host %.3f us per node)\n",
               which == 0 ? "empty <<<512,512>>>        " : "fill 19 MiB <<<512,512>>>  ", ms * 1e3 / K, host_us, K, msg * 1e3 / K, host_g);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
