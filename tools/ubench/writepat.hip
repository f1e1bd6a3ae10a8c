// Microbenchmark: cost of scattered small stores.  Each lane owns a slot of `stride` bytes and
// writes NB bytes at its start (one slot per lane, 1M slots), in pieces of PIECE bytes per store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned U4 __attribute__((ext_vector_type(4)));
typedef unsigned U2 __attribute__((ext_vector_type(2)));

template <int NB, int PIECE>
__global__ __launch_bounds__(64) void lane_writes(char* out, size_t stride, size_t n)
{
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    char* p = out + s * stride;
#pragma unroll
    for (int o = 0; o < NB; o += PIECE) {
        if (PIECE == 8) *reinterpret_cast<U2*>(p + o) = U2{(unsigned)s, (unsigned)o};
        else *reinterpret_cast<U4*>(p + o) = U4{(unsigned)s, (unsigned)o, 1u, 2u};
        if (PIECE == 8 && NB > 8) asm volatile("s_nop 0");  // keep the stores separate
    }
}

// cooperative: a wave writes NB bytes to each of its 64 slots, LPR lanes x 16 B per slot per instruction
template <int NB>
__global__ __launch_bounds__(64) void coop_writes(char* out, size_t stride, size_t n)
{
    constexpr int LPR = NB / 16, RPI = 64 / LPR;
    const unsigned lane = threadIdx.x;
    const size_t s0 = (size_t)blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 64 / RPI; i++) {
        const size_t s = s0 + i * RPI + lane / LPR;
        if (s < n) *reinterpret_cast<U4*>(out + s * stride + (lane % LPR) * 16) = U4{(unsigned)s, lane, 1u, 2u};
    }
}

template <class F>
float timeit(F f, int reps = 10)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t n = 1000000, stride = 1728;
    char* out;
    CK(hipMalloc(&out, n * stride));
    CK(hipMemset(out, 0, n * stride));
    const unsigned blocks = (unsigned)((n + 63) / 64);
    auto rep = [&](const char* name, float ms, int nb) { printf("%-34s %.4f ms  (%.1f GB/s useful)\n", name, ms, n * (double)nb / ms / 1e6); };
    rep("lane 8 B x1", timeit([&] { lane_writes<8, 8><<<blocks, 64>>>(out, stride, n); }), 8);
    rep("lane 8 B x4 (32 B)", timeit([&] { lane_writes<32, 8><<<blocks, 64>>>(out, stride, n); }), 32);
    rep("lane 8 B x8 (64 B)", timeit([&] { lane_writes<64, 8><<<blocks, 64>>>(out, stride, n); }), 64);
    rep("lane 8 B x16 (128 B)", timeit([&] { lane_writes<128, 8><<<blocks, 64>>>(out, stride, n); }), 128);
    rep("lane 16 B x1", timeit([&] { lane_writes<16, 16><<<blocks, 64>>>(out, stride, n); }), 16);
    rep("lane 16 B x2 (32 B)", timeit([&] { lane_writes<32, 16><<<blocks, 64>>>(out, stride, n); }), 32);
    rep("lane 16 B x4 (64 B)", timeit([&] { lane_writes<64, 16><<<blocks, 64>>>(out, stride, n); }), 64);
    rep("lane 16 B x8 (128 B)", timeit([&] { lane_writes<128, 16><<<blocks, 64>>>(out, stride, n); }), 128);
    rep("coop 32 B per slot", timeit([&] { coop_writes<32><<<blocks, 64>>>(out, stride, n); }), 32);
    rep("coop 64 B per slot", timeit([&] { coop_writes<64><<<blocks, 64>>>(out, stride, n); }), 64);
    rep("coop 128 B per slot", timeit([&] { coop_writes<128><<<blocks, 64>>>(out, stride, n); }), 128);
    rep("coop 256 B per slot", timeit([&] { coop_writes<256><<<blocks, 64>>>(out, stride, n); }), 256);
    CK(hipGetLastError());
    return 0;
}
