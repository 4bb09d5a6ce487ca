// Does kernel-argument PRELOAD (gfx940+: the first dwords of the argument segment arrive in user SGPRs at wave launch,
// `-mllvm -amdgpu-kernarg-preload-count=N`) remove the cold argument fetch from a short launch's critical path?
//   hipcc --offload-arch=gfx950 -O2 -o kernarg_preload_off kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-kernarg-preload-count=14 -o kernarg_preload_on kernarg_preload.hip
// Each kernel must read its arguments before it can exit; `chain` additionally chases a pointer held in the arguments
// (the step kernel's "arguments -> env record" dependency).  Back-to-back launches on one stream, and the same
// launches replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty() {}
__global__ void uses4(int *p, int a, int b) { if (b == 12345) p[0] = a; }
__global__ void uses14(int *p, int *q, int *r, int *s, int *t, int *u, int a, int b) { if (b == 12345) p[0] = a + (q != r) + (s != t) + (u != nullptr); }
__global__ void chain(const int *rec, int *out, int stride, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = rec[(size_t)min(i, n - 1) * stride];
    if (v == 12345) out[0] = 1;
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
template <class F>
static float graph_us(F launch, int per_graph, int reps) {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < per_graph; ++i) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / (reps * per_graph);
}
int main() {
    int *blk; hipMalloc(&blk, 4096 * 64 * 64 * 4); hipMemset(blk, 0, 4096 * 64 * 64 * 4);
    const int g = 512;
    printf("stream  <<<%d,512>>>: empty %.2f | 16 B args %.2f | 56 B args %.2f | args -> record (256 B stride) %.2f us per launch\n", g,
           time_us([&] { empty<<<g, 512>>>(); }, 2000), time_us([&] { uses4<<<g, 512>>>(blk, 1, 2); }, 2000),
           time_us([&] { uses14<<<g, 512>>>(blk, blk, blk, blk, blk, blk, 1, 2); }, 2000),
           time_us([&] { chain<<<g, 512>>>(blk, blk, 64, 4096 * 64); }, 2000));
    printf("graph   <<<%d,512>>>: empty %.2f | 16 B args %.2f | 56 B args %.2f | args -> record (256 B stride) %.2f us per launch\n", g,
           graph_us([&](hipStream_t s) { empty<<<g, 512, 0, s>>>(); }, 32, 100), graph_us([&](hipStream_t s) { uses4<<<g, 512, 0, s>>>(blk, 1, 2); }, 32, 100),
           graph_us([&](hipStream_t s) { uses14<<<g, 512, 0, s>>>(blk, blk, blk, blk, blk, blk, 1, 2); }, 32, 100),
           graph_us([&](hipStream_t s) { chain<<<g, 512, 0, s>>>(blk, blk, 64, 4096 * 64); }, 32, 100));
    return 0;
}
