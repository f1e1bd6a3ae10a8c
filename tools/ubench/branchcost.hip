// Microbenchmark: what a branch costs a wave that has a SIMD to itself (and one that shares it): cycles per
// not-taken branch, per taken forward branch over one or over sixteen instructions, and per taken backward branch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
#define NOP16 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"

template <int OP>
__global__ void k(unsigned long long* out, unsigned* sink)
{
    unsigned a = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 16; it++) {
        if (OP == 0) { REP64(asm volatile("s_cmp_eq_u32 s10, s10\n s_cbranch_scc0 1f\n v_add_u32 %0, 1, %0\n1:" : "+v"(a) : : "scc", "s10");) }          // not taken
        if (OP == 1) { REP64(asm volatile("s_cmp_eq_u32 s10, s10\n s_cbranch_scc1 1f\n v_add_u32 %0, 1, %0\n1:" : "+v"(a) : : "scc", "s10");) }          // taken, over 1
        if (OP == 2) { REP64(asm volatile("s_cmp_eq_u32 s10, s10\n s_cbranch_scc1 1f\n" NOP16 "1: v_add_u32 %0, 1, %0" : "+v"(a) : : "scc", "s10");) }   // taken, over 16
        if (OP == 3) { REP64(asm volatile("s_cmp_eq_u32 s10, s10\n s_nop 0\n v_add_u32 %0, 1, %0" : "+v"(a) : : "scc", "s10");) }                         // no branch
        if (OP == 4) { REP64(asm volatile("s_mov_b32 s11, 2\n 1: s_sub_u32 s11, s11, 1\n s_cmp_lg_u32 s11, 0\n s_cbranch_scc1 1b\n v_add_u32 %0, 1, %0" : "+v"(a) : : "scc", "s11");) }  // one taken backward branch, one not taken
        if (OP == 5) { REP64(asm volatile("s_cmp_eq_u32 s10, s10\n s_cbranch_scc1 1f\n" NOP16 NOP16 NOP16 NOP16 "1: v_add_u32 %0, 1, %0" : "+v"(a) : : "scc", "s10");) }   // taken, over 64
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a == 0x1234567u) sink[0] = 1;
}

template <int OP>
void run(const char* name, unsigned long long* d_out, unsigned* d_sink, double per)
{
    for (int threads : {64, 256, 512}) {
        k<OP><<<1, threads>>>(d_out, d_sink);
        k<OP><<<1, threads>>>(d_out, d_sink);
        CK(hipDeviceSynchronize());
        unsigned long long h[16];
        CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? h[w] : mx;
        printf("%-44s %4d threads on one CU: %.1f cycles per group (%g instructions)\n", name, threads, mx / (16.0 * 64), per);
    }
}

int main()
{
    unsigned long long* d_out; unsigned* d_sink;
    CK(hipMalloc(&d_out, 16 * 8)); CK(hipMalloc(&d_sink, 4));
    run<3>("cmp, nop, add (no branch)", d_out, d_sink, 3);
    run<0>("cmp, branch not taken, add", d_out, d_sink, 3);
    run<1>("cmp, branch taken over 1 instruction", d_out, d_sink, 2);
    run<2>("cmp, branch taken over 16, add", d_out, d_sink, 3);
    run<5>("cmp, branch taken over 64, add", d_out, d_sink, 3);
    run<4>("mov, 2x(sub, cmp, branch back once), add", d_out, d_sink, 8);
    return 0;
}
