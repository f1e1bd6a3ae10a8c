// Microbenchmark: issue cost (cycles per wave-instruction) of the VALU ops the codec leans on,
// for 1 and 2 waves per SIMD (blocks of 256 / 512 threads on one CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP>
__global__ void k(unsigned long long* out, unsigned* sink, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, s = (a0 & 31) + 1;
    unsigned long long b0 = ((unsigned long long)a0 << 32) | a1, b1 = ((unsigned long long)a2 << 32) | a3, b2 = b0 * 3, b3 = b1 * 5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 16; it++) {
        if (OP == 0) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 1) { REP64(asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));) }
        if (OP == 2) { REP64(asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(s));) }
        if (OP == 3) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 4) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 5) { REP64(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %0" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : : "vcc");) }
        if (OP == 6) { REP64(asm volatile("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 7) { REP64(asm volatile("v_alignbit_b32 %0, %0, %1, %4\n v_alignbit_b32 %1, %1, %2, %4\n v_alignbit_b32 %2, %2, %3, %4\n v_alignbit_b32 %3, %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 8) { REP64(asm volatile("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 9) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "s10", "s11");) }
        if (OP == 10) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4" : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3), "v"(s));) }  // dependent chain
        if (OP == 11) { REP64(asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %0, %4, %0" : "+v"(b0) : "v"(b1), "v"(b2), "v"(b3), "v"(s));) }
        if (OP == 12) { REP64(asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 13) { REP64(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)b0 ^ (unsigned)b1 ^ (unsigned)b2 ^ (unsigned)b3) == 0x1234567u) sink[0] = 1;
}

template <int OP>
void run(const char* name, unsigned long long* d_out, unsigned* d_sink)
{
    for (int threads : {64, 256, 512, 1024}) {
        k<OP><<<1, threads>>>(d_out, d_sink, 1);
        k<OP><<<1, threads>>>(d_out, d_sink, 2);
        CK(hipDeviceSynchronize());
        unsigned long long h[16];
        CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        const double ninstr = 16.0 * 64 * 4;
        unsigned long long mn = ~0ull, mx = 0;
        for (int w = 0; w < threads / 64; w++) { mn = h[w] < mn ? h[w] : mn; mx = h[w] > mx ? h[w] : mx; }
        printf("%-22s %4d threads/CU: cycles per wave-instruction min %.2f max %.2f\n", name, threads, mn / ninstr, mx / ninstr);
    }
}

int main()
{
    unsigned long long* d_out; unsigned* d_sink;
    CK(hipMalloc(&d_out, 16 * 8)); CK(hipMalloc(&d_sink, 4));
    run<0>("v_add_u32", d_out, d_sink);

    run<1>("v_lshlrev_b64", d_out, d_sink);
    run<2>("v_lshrrev_b64", d_out, d_sink);

    run<9>("v_cndmask_b32 sgpr", d_out, d_sink);

    run<5>("v_cmp_lt_u64", d_out, d_sink);

    run<10>("v_add_u32 dependent", d_out, d_sink);
    run<11>("v_lshlrev_b64 dependent", d_out, d_sink);
    return 0;
}
