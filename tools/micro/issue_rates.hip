// What a CU issues per cycle, by instruction class and by the number of resident waves - the units the step kernel's waves share.
//   hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip && ./issue_rates
// Every kernel runs ITER iterations of a 32-instruction body per wave; grids of 256 x W single-wave workgroups put W waves on every CU
// (W = 1 ... 32).  Reported: cycles per instruction per WAVE (the latency a wave sees) and instructions per cycle per CU (the throughput
// all its waves share), at the 2.4 GHz shader clock; timed with s_memtime inside one wave per CU is not needed - HIP events around
// launches that last ~100 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))
constexpr int ITER = 4000;

// 32 INDEPENDENT scalar adds per iteration (eight registers, four each)
__global__ __launch_bounds__(64) void k_salu_indep(unsigned *out) {
    unsigned a = threadIdx.x;
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP4("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                          "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n")
                     ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
    if (out) out[0] = a;
}
// 32 DEPENDENT scalar adds per iteration (one register)
__global__ __launch_bounds__(64) void k_salu_dep(unsigned *out) {
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP32("s_add_u32 s20, s20, 1\n") ::: "s20", "scc");
    if (out) out[0] = threadIdx.x;
}
// 32 dependent vector adds
__global__ __launch_bounds__(64) void k_valu_dep(unsigned *out) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP32("v_add_u32 %0, %0, 1\n") : "+v"(v));
    if (out) out[threadIdx.x] = v;
}
// 16 x (vector compare -> scalar use of the mask): the VALU -> SALU hand-off of every ballot / uniform branch condition
__global__ __launch_bounds__(64) void k_handoff_cmp(unsigned *out) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP4(REP4("v_cmp_gt_u32 vcc, %0, 3\n s_and_b64 s[20:21], vcc, exec\n")) : "+v"(v) :: "vcc", "s20", "s21", "scc");
    if (out) out[threadIdx.x] = v;
}
// 16 x (v_readlane -> scalar add -> vector use): reading one object / cell out of the wave and acting on it
__global__ __launch_bounds__(64) void k_handoff_readlane(unsigned *out) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP4(REP4("v_readlane_b32 s20, %0, 5\n s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, s20\n")) : "+v"(v) :: "s20", "scc");   // 48 instructions
    if (out) out[threadIdx.x] = v;
}
// 32 x (compare + TAKEN branch to the next instruction): the cost of a wave-uniform branch
__global__ __launch_bounds__(64) void k_branch_taken(unsigned *out) {
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP4(REP4("s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n s_nop 0\n1:\n s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 2f\n s_nop 0\n2:\n"))
                     ::: "s20", "scc");                                     // 32 compares + 32 taken branches per iteration
    if (out) out[0] = threadIdx.x;
}
// 32 x (compare + NOT-taken branch)
__global__ __launch_bounds__(64) void k_branch_not_taken(unsigned *out) {
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP4(REP4("s_cmp_lg_u32 s20, s20\n s_cbranch_scc1 1f\n1:\n s_cmp_lg_u32 s20, s20\n s_cbranch_scc1 2f\n2:\n")) ::: "s20", "scc");
    if (out) out[0] = threadIdx.x;
}
// the step kernel's mix in miniature: per iteration 16 scalar + 14 vector + 2 LDS + 3 taken branches, all on one dependent chain through
// a vector register, a scalar register and the LDS
__global__ __launch_bounds__(64) void k_mix(unsigned *out) {
    __shared__ unsigned lds[64];
    unsigned v = threadIdx.x;
    lds[threadIdx.x] = v;
    for (int i = 0; i < ITER; ++i)
        asm volatile(REP4("v_cmp_gt_u32 vcc, %0, 3\n s_and_b64 s[20:21], vcc, exec\n s_ff1_i32_b64 s22, s[20:21]\n v_readlane_b32 s23, %0, s22\n"
                          "s_add_u32 s23, s23, 1\n s_and_b32 s23, s23, 63\n v_add_u32 %0, %0, s23\n")
                     "s_cmp_lg_u32 s23, 99\n s_cbranch_scc1 1f\n s_nop 0\n1:\n"
                     "v_lshlrev_b32 %0, 2, %0\n v_and_b32 %0, 252, %0\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n"
                     "s_cmp_lg_u32 s23, 98\n s_cbranch_scc1 2f\n s_nop 0\n2:\n"
                     : "+v"(v) :: "vcc", "s20", "s21", "s22", "s23", "scc", "memory");
    if (out) out[threadIdx.x] = v;
}

template <class K>
static void run(const char *name, K kern, int instr_per_iter) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    printf("%-44s", name);
    for (int W : {1, 2, 4, 8, 16, 32}) {
        dim3 grid(256 * W), block(64);
        hipLaunchKernelGGL(kern, grid, block, 0, 0, (unsigned *)nullptr);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, grid, block, 0, 0, (unsigned *)nullptr);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double cycles = ms * 1e-3 * 2.4e9, per_wave = cycles / ((double)ITER * instr_per_iter);
        printf("  W=%-2d %5.2f cyc/instr/wave %5.2f instr/cyc/CU |", W, per_wave, W / per_wave);
    }
    printf("\n");
}
int main() {
    printf("%d iterations per wave; W waves per CU (256 x W one-wave workgroups); cycles at 2.4 GHz\n", ITER);
    run("32 independent s_add_u32", k_salu_indep, 32);
    run("32 dependent s_add_u32", k_salu_dep, 32);
    run("32 dependent v_add_u32", k_valu_dep, 32);
    run("16 x (v_cmp -> s_and of the mask)", k_handoff_cmp, 32);
    run("16 x (v_readlane -> s_add -> v_add)", k_handoff_readlane, 48);
    run("32 x (s_cmp + taken s_cbranch)", k_branch_taken, 64);
    run("32 x (s_cmp + not-taken s_cbranch)", k_branch_not_taken, 64);
    run("step-kernel-like mix (31 instr + 2 waits)", k_mix, 4 * 7 + 3 + 3 + 4 + 3);
    return 0;
}
