// Microbenchmark: the encoder's input pattern without the encoding.  A wave owns 64 rows (series) of P
// points in two arrays; step after step it reads SEG bytes of every row from both, 8 lanes x 16 bytes per
// 128-byte line, as the compressor's chunk fetch does (DEPTH steps in flight).  Beside it: the same bytes
// read as contiguous blocks per wave, and sequentially.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned U4 __attribute__((ext_vector_type(4)));

template <int SEG, bool NT, int DEPTH>
__global__ __launch_bounds__(64) void rows_kernel(const char* ts, const char* val, size_t row_bytes, size_t nrows, unsigned* sink)
{
    const unsigned lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * 64;
    constexpr int LPR = SEG / 16, RPI = 64 / LPR, N = 64 / RPI;
    const unsigned piece = lane % LPR, lrow = lane / LPR;
    const size_t steps = row_bytes / SEG;
    U4 acc = {0, 0, 0, 0};
    U4 q[DEPTH][2 * N];
    auto fetch = [&](size_t c, U4* d) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            size_t row = row0 + lrow + i * RPI;
            if (row >= nrows) row = nrows - 1;
            const size_t o = row * row_bytes + c * SEG + piece * 16;
            d[2 * i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const U4*>(ts + o)) : *reinterpret_cast<const U4*>(ts + o);
            d[2 * i + 1] = NT ? __builtin_nontemporal_load(reinterpret_cast<const U4*>(val + o)) : *reinterpret_cast<const U4*>(val + o);
        }
    };
#pragma unroll
    for (int k = 0; k < DEPTH; k++) fetch(k < (int)steps ? k : 0, q[k]);
    for (size_t c = 0; c < steps; c += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
#pragma unroll
            for (int i = 0; i < 2 * N; i++) acc ^= q[k][i];
            const size_t nc = c + k + DEPTH;
            fetch(nc < steps ? nc : 0, q[k]);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

// the compressor's own geometry: rows of any length on a grid of 128-byte lines, the 64 rows of a wave 2^shift apart
template <bool NT>
__global__ __launch_bounds__(64) void rows_aligned_kernel(const char* ts, const char* val, size_t row_bytes, size_t nrows, unsigned shift, unsigned* sink)
{
    const unsigned lane = threadIdx.x, block = blockIdx.x;
    const unsigned r = block & ((8u << shift) - 1u);
    const size_t group = ((size_t)(block >> (3 + shift)) << 3) + (r & 7u);
    const size_t row0 = (group << (6 + shift)) + (r >> 3);
    const unsigned piece = lane % 8, lrow = lane / 8;
    const size_t steps = (row_bytes + 127 + 64) / 128;
    U4 acc = {0, 0, 0, 0};
    U4 q[16];
    size_t base[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        size_t row = row0 + ((size_t)(lrow + i * 8) << shift);
        if (row >= nrows) row = nrows - 1;
        base[i] = ((row * row_bytes) & ~(size_t)127) + piece * 16;
    }
    auto fetch = [&](size_t c) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const size_t o = base[i] + c * 128;
            q[2 * i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const U4*>(ts + o)) : *reinterpret_cast<const U4*>(ts + o);
            q[2 * i + 1] = NT ? __builtin_nontemporal_load(reinterpret_cast<const U4*>(val + o)) : *reinterpret_cast<const U4*>(val + o);
        }
    };
    fetch(0);
    for (size_t c = 0; c < steps; c++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc ^= q[i];
        fetch(c + 1 < steps ? c + 1 : 0);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

template <int BLK, bool NT>
__global__ __launch_bounds__(64) void blocks_kernel(const char* ts, const char* val, size_t bytes, unsigned* sink)
{
    const size_t nblk = bytes / BLK;
    U4 acc = {0, 0, 0, 0};
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        U4 q[16];
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const char* p = ts + b * BLK + o * 1024 + threadIdx.x * 16;
            const char* r = val + b * BLK + o * 1024 + threadIdx.x * 16;
            q[2 * o] = NT ? __builtin_nontemporal_load(reinterpret_cast<const U4*>(p)) : *reinterpret_cast<const U4*>(p);
            q[2 * o + 1] = NT ? __builtin_nontemporal_load(reinterpret_cast<const U4*>(r)) : *reinterpret_cast<const U4*>(r);
        }
#pragma unroll
        for (int o = 0; o < 16; o++) acc ^= q[o];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

template <class F>
float timeit(F f, int reps = 10)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const size_t nrows = 1000000, P = argc > 1 ? (size_t)atoi(argv[1]) : 128, row_bytes = P * 8;
    const size_t bytes = nrows * row_bytes;
    char *ts, *val; unsigned* sink;
    CK(hipMalloc(&ts, bytes + 4096)); CK(hipMalloc(&val, bytes + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(ts, 1, bytes + 4096)); CK(hipMemset(val, 2, bytes + 4096));
    const unsigned blocks = (unsigned)((nrows + 63) / 64);
    auto rep = [&](const char* name, float ms) { printf("%-52s %.4f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); };
    printf("rows of %zu bytes, %zu rows, two arrays\n", row_bytes, nrows);
    rep("blocks of 8 KB per wave, streaming loads", timeit([&] { blocks_kernel<8192, true><<<256 * 32, 64>>>(ts, val, bytes, sink); }));
    rep("blocks of 8 KB per wave, plain loads", timeit([&] { blocks_kernel<8192, false><<<256 * 32, 64>>>(ts, val, bytes, sink); }));
    rep("rows, 128 B per step, 1 step in flight, 7 waves/CU", timeit([&] { rows_kernel<128, true, 1><<<blocks, 64, 22272>>>(ts, val, row_bytes, nrows, sink); }));
    rep("rows, 128 B per step, 2 steps in flight, 7 waves/CU", timeit([&] { rows_kernel<128, true, 2><<<blocks, 64, 22272>>>(ts, val, row_bytes, nrows, sink); }));
    rep("rows, 128 B per step, 1 step in flight, 16 waves/CU", timeit([&] { rows_kernel<128, true, 1><<<blocks, 64, 10000>>>(ts, val, row_bytes, nrows, sink); }));
    rep("rows, 128 B per step, 2 steps in flight, 16 waves/CU", timeit([&] { rows_kernel<128, true, 2><<<blocks, 64, 10000>>>(ts, val, row_bytes, nrows, sink); }));
    rep("rows, 128 B, plain loads, 1 step, 7 waves/CU", timeit([&] { rows_kernel<128, false, 1><<<blocks, 64, 22272>>>(ts, val, row_bytes, nrows, sink); }));
    rep("rows, 256 B per step, 1 step in flight, 7 waves/CU", timeit([&] { rows_kernel<256, true, 1><<<blocks, 64, 22272>>>(ts, val, row_bytes, nrows, sink); }));
    {
        const unsigned shift = (P % 16 == 0) ? 0 : 1;  // P = 120: rows of equal phase are two apart
        const size_t per = 64ull << shift, groups = (nrows + per - 1) / per;
        const unsigned gblocks = (unsigned)((shift ? ((groups + 7) & ~7ull) : groups) << shift);
        rep("rows on the line grid, as the compressor, 7 waves/CU", timeit([&] { rows_aligned_kernel<true><<<gblocks, 64, 22272>>>(ts, val, row_bytes, nrows, shift, sink); }));
        rep("the same, plain loads", timeit([&] { rows_aligned_kernel<false><<<gblocks, 64, 22272>>>(ts, val, row_bytes, nrows, shift, sink); }));
        rep("the same, consecutive rows per wave", timeit([&] { rows_aligned_kernel<true><<<blocks, 64, 22272>>>(ts, val, row_bytes, nrows, 0, sink); }));
    }
    CK(hipGetLastError());
    return 0;
}
