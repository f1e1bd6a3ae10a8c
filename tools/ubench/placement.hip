// Microbenchmark: where do the waves of a grid land?  Every wave records the XCD, compute unit and SIMD it runs on
// (HW_ID register) and spins for a while; the host prints how many waves each SIMD of the machine held at the
// same time.  usage: placement <workgroups> <threads per workgroup> <LDS bytes per workgroup>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k(unsigned* out, unsigned long long* when, unsigned spin)
{
    extern __shared__ unsigned char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = threadIdx.x;
    for (unsigned i = 0; i < spin; i++) { x = x * 1664525u + 1013904223u; lds[(x >> 8) % 64] = (unsigned char)x; }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) { out[2 * w] = hw; out[2 * w + 1] = xcc + (lds[0] & 0u); when[2 * w] = t0; when[2 * w + 1] = t1; }
}

int main(int argc, char** argv)
{
    const unsigned wgs = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 64, ldsb = argc > 3 ? atoi(argv[3]) : 20000;
    const unsigned waves = wgs * (threads / 64);
    unsigned* d; unsigned long long* dw;
    CK(hipMalloc(&d, waves * 8)); CK(hipMalloc(&dw, waves * 16));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    k<<<wgs, threads, ldsb>>>(d, dw, 20000);
    CK(hipDeviceSynchronize());
    k<<<wgs, threads, ldsb>>>(d, dw, 20000);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * waves); std::vector<unsigned long long> t(2 * waves);
    CK(hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(t.data(), dw, waves * 16, hipMemcpyDeviceToHost));
    // waves that were running when the first wave to finish finished
    unsigned long long first_end = ~0ull;
    for (unsigned w = 0; w < waves; w++) first_end = t[2 * w + 1] < first_end ? t[2 * w + 1] : first_end;
    std::map<unsigned, unsigned> per_simd, per_cu;
    unsigned live = 0;
    for (unsigned w = 0; w < waves; w++) {
        if (argc > 4 && t[2 * w] > first_end) continue;   // (the clocks of the XCDs are not one clock: only with a fifth argument)
        live++;
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 15u;
        const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        const unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_simd[(cuid << 2) | simd]++; per_cu[cuid]++;
    }
    std::map<unsigned, unsigned> hist_simd, hist_cu;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    if (argc > 5) for (unsigned w = 0; w < 12 && w < waves; w++) printf("  wave %u: hw_id %08x xcc %u\n", w, h[2 * w], h[2 * w + 1]);
    printf("%u workgroups x %u threads, %u B LDS: %u waves in the first round on %zu SIMDs of %zu CUs\n", wgs, threads, ldsb, live, per_simd.size(), per_cu.size());
    printf("  waves per SIMD in use:");
    for (auto& kv : hist_simd) printf("  %u waves: %u SIMDs", kv.first, kv.second);
    printf("\n  waves per CU in use:  ");
    for (auto& kv : hist_cu) printf("  %u waves: %u CUs", kv.first, kv.second);
    printf("\n");
    return 0;
}
