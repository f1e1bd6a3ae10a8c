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
        if (OP == 14) { REP64(asm volatile("v_bfe_u32 %0, %0, %4, 6\n v_bfe_u32 %1, %1, %4, 6\n v_bfe_u32 %2, %2, %4, 6\n v_bfe_u32 %3, %3, %4, 6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 15) { REP64(asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 16) { REP64(asm volatile("v_add_u32_e64 %0, %0, %4\n v_add_u32_e64 %1, %1, %4\n v_add_u32_e64 %2, %2, %4\n v_add_u32_e64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 17) { REP64(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 18) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));) }
        if (OP == 19) { REP64(asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %4\n v_cmp_lt_u32_e64 s[10:11], %1, %4\n v_cmp_lt_u32_e64 s[10:11], %2, %4\n v_cmp_lt_u32_e64 s[10:11], %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "s10", "s11");) }
        if (OP == 20) { REP64(asm volatile("v_xor_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 21) { REP64(asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 22) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 23) { REP64(asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 s10, s10, 1\n v_add_u32 %1, %1, %4\n s_add_u32 s10, s10, 1\n v_add_u32 %2, %2, %4\n s_add_u32 s10, s10, 1\n v_add_u32 %3, %3, %4\n s_add_u32 s10, s10, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "s10", "scc");) }
        if (OP == 24) { REP64(asm volatile("v_ashrrev_i32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_not_b32 %2, %2\n v_max_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 25) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 26) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 27) { REP64(asm volatile("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 28) { REP64(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 29) { REP64(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 30) { REP64(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 31) { REP64(asm volatile("v_and_b32 %0, 63, %0\n v_and_b32 %1, 63, %1\n v_and_b32 %2, 63, %2\n v_and_b32 %3, 63, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 32) { REP64(asm volatile("v_add_u32 %0, 0x3000, %0\n v_add_u32 %1, 0x3000, %1\n v_add_u32 %2, 0x3000, %2\n v_add_u32 %3, 0x3000, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 33) { REP64(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 34) { REP64(asm volatile("v_add_u32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n s_nop 0\n v_add_u32 %2, %2, %4\n s_nop 0\n v_add_u32 %3, %3, %4\n s_nop 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 35) { REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 36) { REP64(asm volatile("v_sub_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_sub_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_sub_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_sub_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 37) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 38) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 39) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 40) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 41) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");) }
        if (OP == 42) { REP64(asm volatile("s_and_b64 vcc, exec, s[10:11]\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 50) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 51) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 52) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 53) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 54) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 55) { REP64(asm volatile("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %0, %4, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 56) { REP64(asm volatile("v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %1, %2, %4, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %0, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 57) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_cmp_lt_u32 vcc, %0, %4\n v_add_u32 %1, %1, %4\n v_cmp_lt_u32 vcc, %1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 58) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 59) { REP64(asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %2, %4\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc", "s10", "s11");) }
        if (OP == 13) { REP64(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)b0 ^ (unsigned)b1 ^ (unsigned)b2 ^ (unsigned)b3) == 0x1234567u) sink[0] = 1;
}

template <int OP>
void run(const char* name, unsigned long long* d_out, unsigned* d_sink)
{
    for (int threads : {256, 1024}) {
        k<OP><<<1, threads>>>(d_out, d_sink, 1);
        k<OP><<<1, threads>>>(d_out, d_sink, 2);
        CK(hipDeviceSynchronize());
        unsigned long long h[16];
        CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        const double ninstr = 16.0 * 64 * 4;
        unsigned long long mn = ~0ull, mx = 0;
        for (int w = 0; w < threads / 64; w++) { mn = h[w] < mn ? h[w] : mn; mx = h[w] > mx ? h[w] : mx; }
        printf("%-40s %4d threads/CU: cycles per wave-instruction min %.2f max %.2f\n", name, threads, mn / ninstr, mx / ninstr);
    }
}

int main()
{
    unsigned long long* d_out; unsigned* d_sink;
    CK(hipMalloc(&d_out, 16 * 8)); CK(hipMalloc(&d_sink, 4));
    // 256 threads = one wave per SIMD, 1024 = four: "max" at 1024 threads / 4 is what the SIMD sustains.
    // Measured on MI355X: add/sub/and/or/xor/not/mov_b32/lshrrev_b32/ashrrev_i32 run at one per 2.6
    // cycles per SIMD, everything else listed here (compares, cndmask, min/max, lshlrev_b32, bfe,
    // ffbh, alignbit, mul24, all three-operand and all 64-bit forms, SDWA) at one per 4.3; one wave
    // issues an instruction of any kind (scalar ones too) every 5.2 cycles at best; and a
    // v_cndmask_b32 on VCC directly behind another one takes 19 cycles instead of 5.
    run<0>("v_add_u32", d_out, d_sink);
    run<30>("v_sub_u32", d_out, d_sink);
    run<12>("v_or_b32", d_out, d_sink);
    run<29>("v_xor_b32", d_out, d_sink);
    run<28>("v_mov_b32", d_out, d_sink);
    run<27>("v_lshrrev_b32", d_out, d_sink);
    run<8>("v_lshlrev_b32", d_out, d_sink);
    run<33>("v_min/max_u32", d_out, d_sink);
    run<9>("v_cndmask_b32 sgpr", d_out, d_sink);
    run<4>("v_cmp_lt_u32 vcc", d_out, d_sink);
    run<5>("v_cmp_lt_u64", d_out, d_sink);
    run<6>("v_ffbh_u32", d_out, d_sink);
    run<7>("v_alignbit_b32", d_out, d_sink);
    run<14>("v_bfe_u32", d_out, d_sink);
    run<15>("v_lshl_add_u32", d_out, d_sink);
    run<17>("v_add3_u32", d_out, d_sink);
    run<21>("v_and_or_b32", d_out, d_sink);
    run<22>("v_mul_u32_u24", d_out, d_sink);
    run<36>("v_sub_u32_sdwa", d_out, d_sink);
    run<1>("v_lshlrev_b64", d_out, d_sink);
    run<18>("v_lshl_add_u64", d_out, d_sink);
    run<13>("v_mov_b64", d_out, d_sink);
    run<23>("v_add + s_add (x8/4)", d_out, d_sink);
    run<58>("cmp->cnd same vcc back-to-back", d_out, d_sink);
    run<53>("cnd vcc, 1 between", d_out, d_sink);
    run<52>("cnd vcc directly after cnd vcc", d_out, d_sink);
    return 0;
}
