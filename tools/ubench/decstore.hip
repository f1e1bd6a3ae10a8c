// Microbenchmark: the decoder's output pattern without the decoding.  A wave owns 64 rows (series) of P
// points in two arrays (timestamps, values); step after step it writes SEG bytes of every row to both,
// 8 lanes x 16 bytes per 128-byte line, as the decoder's chunk flush does.  Variants: bytes per row and
// step (128, 256, 512), streaming or plain stores, and a fully sequential write of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned U4 __attribute__((ext_vector_type(4)));

// SEG bytes per row and step; GAP: s_sleep units between steps (stands for the decoding of a chunk)
template <int SEG, bool NT, int GAP>
__global__ __launch_bounds__(64) void rows_kernel(char* ts, char* val, size_t row_bytes, size_t nrows, unsigned shift)
{
    const unsigned lane = threadIdx.x, block = blockIdx.x;
    // rows of equal line phase per wave, the waves of a group 8 workgroups apart (one XCD)
    const unsigned r = block & ((8u << shift) - 1u);
    const size_t group = ((size_t)(block >> (3 + shift)) << 3) + (r & 7u);
    const size_t row0 = (group << (6 + shift)) + (r >> 3);
    constexpr int LPR = SEG / 16;      // lanes per row and instruction
    constexpr int RPI = 64 / LPR;      // rows per instruction
    const unsigned piece = lane % LPR, lrow = lane / LPR;
    const size_t steps = row_bytes / SEG;  // whole segments only
    for (size_t c = 0; c < steps; c++) {
#pragma unroll
        for (int i = 0; i < 64 / RPI; i++) {
            const size_t row = row0 + ((size_t)(lrow + i * RPI) << shift);
            if (row < nrows) {
                const size_t o = row * row_bytes + c * SEG + piece * 16;
                const U4 v = {(unsigned)row, (unsigned)c, lane, 1u};
                if (NT) {
                    __builtin_nontemporal_store(v, reinterpret_cast<U4*>(ts + o));
                    __builtin_nontemporal_store(v, reinterpret_cast<U4*>(val + o));
                } else {
                    *reinterpret_cast<U4*>(ts + o) = v;
                    *reinterpret_cast<U4*>(val + o) = v;
                }
            }
        }
        if (GAP) __builtin_amdgcn_s_sleep(GAP);
    }
}

// a wave writes BLK contiguous bytes of one array, then of the other; waves take blocks in turn
template <int BLK, bool NT, bool TWO>
__global__ __launch_bounds__(64) void blocks_kernel(char* ts, char* val, size_t bytes)
{
    const size_t nblk = bytes / BLK;
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
#pragma unroll 8
        for (int o = 0; o < BLK; o += 1024) {
            const U4 v = {(unsigned)b, (unsigned)o, 1u, 2u};
            char* p = ts + b * BLK + o + threadIdx.x * 16;
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<U4*>(p)); else *reinterpret_cast<U4*>(p) = v;
        }
        if (TWO) {
#pragma unroll 8
            for (int o = 0; o < BLK; o += 1024) {
                const U4 v = {(unsigned)b, (unsigned)o, 1u, 2u};
                char* p = val + b * BLK + o + threadIdx.x * 16;
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<U4*>(p)); else *reinterpret_cast<U4*>(p) = v;
            }
        }
    }
}

// rows kernel with all timestamp stores of a step before its value stores
template <int SEG, bool NT>
__global__ __launch_bounds__(64) void rows_split_kernel(char* ts, char* val, size_t row_bytes, size_t nrows)
{
    const unsigned lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * 64;
    constexpr int LPR = SEG / 16, RPI = 64 / LPR;
    const unsigned piece = lane % LPR, lrow = lane / LPR;
    const size_t steps = row_bytes / SEG;
    for (size_t c = 0; c < steps; c++) {
        for (int w = 0; w < 2; w++) {
            char* dst = w ? val : ts;
#pragma unroll
            for (int i = 0; i < 64 / RPI; i++) {
                const size_t row = row0 + lrow + i * RPI;
                if (row < nrows) {
                    const size_t o = row * row_bytes + c * SEG + piece * 16;
                    const U4 v = {(unsigned)row, (unsigned)c, lane, 1u};
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<U4*>(dst + o)); else *reinterpret_cast<U4*>(dst + o) = v;
                }
            }
        }
    }
}

// the decompressor's own geometry: rows of any length on a grid of 128-byte lines (a row's first and last
// line may be shared with its neighbours: only the row's own bytes are written), the 64 rows of a wave
// 2^shift apart; GAP: s_sleep units between steps
template <bool NT, int GAP>
__global__ __launch_bounds__(64) void rows_aligned_kernel(char* ts, char* val, size_t row_bytes, size_t nrows, unsigned shift)
{
    const unsigned lane = threadIdx.x, block = blockIdx.x;
    const unsigned r = block & ((8u << shift) - 1u);
    const size_t group = ((size_t)(block >> (3 + shift)) << 3) + (r & 7u);
    const size_t row0 = (group << (6 + shift)) + (r >> 3);
    const unsigned piece = lane % 8, lrow = lane / 8;
    const size_t steps = (row_bytes + 127 + 64) / 128;
    for (size_t c = 0; c < steps; c++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const size_t row = row0 + ((size_t)(lrow + i * 8) << shift);
            const size_t lo = row * row_bytes, hi = lo + row_bytes;
            const size_t o = (lo & ~(size_t)127) + c * 128 + piece * 16;
            if (row < nrows && o >= lo && o + 16 <= hi) {
                const U4 v = {(unsigned)row, (unsigned)c, lane, 1u};
                if (NT) {
                    __builtin_nontemporal_store(v, reinterpret_cast<U4*>(ts + o));
                    __builtin_nontemporal_store(v, reinterpret_cast<U4*>(val + o));
                } else {
                    *reinterpret_cast<U4*>(ts + o) = v;
                    *reinterpret_cast<U4*>(val + o) = v;
                }
            }
        }
        if (GAP) __builtin_amdgcn_s_sleep(GAP);
    }
}

// the same geometry with the store's cache-policy bits spelled out (MODE: 0 none, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 nt,
// 6 sc1 nt, 7 sc0 sc1 nt)
template <int MODE>
__device__ __forceinline__ void store_mode(char* p, U4 v)
{
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(64) void rows_mode_kernel(char* ts, char* val, size_t row_bytes, size_t nrows, unsigned shift)
{
    const unsigned lane = threadIdx.x, block = blockIdx.x;
    const unsigned r = block & ((8u << shift) - 1u);
    const size_t group = ((size_t)(block >> (3 + shift)) << 3) + (r & 7u);
    const size_t row0 = (group << (6 + shift)) + (r >> 3);
    const unsigned piece = lane % 8, lrow = lane / 8;
    const size_t steps = (row_bytes + 127 + 64) / 128;
    for (size_t c = 0; c < steps; c++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const size_t row = row0 + ((size_t)(lrow + i * 8) << shift);
            const size_t lo = row * row_bytes, hi = lo + row_bytes;
            const size_t o = (lo & ~(size_t)127) + c * 128 + piece * 16;
            if (row < nrows && o >= lo && o + 16 <= hi) {
                const U4 v = {(unsigned)row, (unsigned)c, lane, 1u};
                store_mode<MODE>(ts + o, v);
                store_mode<MODE>(val + o, v);
            }
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void seq_kernel(char* ts, char* val, size_t bytes)
{
    const size_t n = bytes / 16, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const U4 v = {(unsigned)i, 0u, 1u, 2u};
        if (NT) {
            __builtin_nontemporal_store(v, reinterpret_cast<U4*>(ts) + i);
            __builtin_nontemporal_store(v, reinterpret_cast<U4*>(val) + i);
        } else {
            reinterpret_cast<U4*>(ts)[i] = v;
            reinterpret_cast<U4*>(val)[i] = v;
        }
    }
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
    char *ts, *val;
    CK(hipMalloc(&ts, bytes + 4096)); CK(hipMalloc(&val, bytes + 4096));
    CK(hipMemset(ts, 0, bytes)); CK(hipMemset(val, 0, bytes));
    const unsigned blocks = (unsigned)((nrows + 63) / 64);
    auto rep = [&](const char* name, float ms) { printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); };
    printf("rows of %zu bytes, %zu rows, two arrays\n", row_bytes, nrows);
    rep("hipMemsetAsync of both arrays", timeit([&] { CK(hipMemsetAsync(ts, 1, bytes, 0)); CK(hipMemsetAsync(val, 2, bytes, 0)); }));
    rep("sequential, streaming, 256 x 32 blocks", timeit([&] { seq_kernel<true><<<256 * 32, 256>>>(ts, val, bytes); }));
    rep("sequential, plain, 256 x 32 blocks", timeit([&] { seq_kernel<false><<<256 * 32, 256>>>(ts, val, bytes); }));
    rep("sequential, plain, 256 x 4 blocks", timeit([&] { seq_kernel<false><<<256 * 4, 256>>>(ts, val, bytes); }));
    rep("blocks of 8 KB per wave, plain, both", timeit([&] { blocks_kernel<8192, false, true><<<256 * 32, 64>>>(ts, val, bytes); }));
    rep("blocks of 8 KB per wave, streaming, both", timeit([&] { blocks_kernel<8192, true, true><<<256 * 32, 64>>>(ts, val, bytes); }));
    rep("blocks of 64 KB per wave, plain, both", timeit([&] { blocks_kernel<65536, false, true><<<256 * 32, 64>>>(ts, val, bytes); }));
    rep("blocks of 64 KB per wave, streaming, both", timeit([&] { blocks_kernel<65536, true, true><<<256 * 32, 64>>>(ts, val, bytes); }));
    rep("blocks of 64 KB, plain, both, 2048 waves", timeit([&] { blocks_kernel<65536, false, true><<<256 * 8, 64>>>(ts, val, bytes); }));
    rep("blocks of 8 KB, plain, ts only (x2 bytes)", 2 * timeit([&] { blocks_kernel<8192, false, false><<<256 * 32, 64>>>(ts, val, bytes); }));
    rep("rows, 128 B, plain, ts then val", timeit([&] { rows_split_kernel<128, false><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows); }));
    rep("rows, 128 B, streaming, ts then val", timeit([&] { rows_split_kernel<128, true><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows); }));
    rep("rows, 128 B, plain, ts then val, 32 waves/CU", timeit([&] { rows_split_kernel<128, false><<<blocks, 64>>>(ts, val, row_bytes, nrows); }));
    rep("sequential, streaming", timeit([&] { seq_kernel<true><<<256 * 8, 256>>>(ts, val, bytes); }));
    rep("sequential, plain", timeit([&] { seq_kernel<false><<<256 * 8, 256>>>(ts, val, bytes); }));
    rep("rows, 128 B per step, streaming", timeit([&] { rows_kernel<128, true, 0><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 128 B per step, plain", timeit([&] { rows_kernel<128, false, 0><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 256 B per step, streaming", timeit([&] { rows_kernel<256, true, 0><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 512 B per step, streaming", timeit([&] { rows_kernel<512, true, 0><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 128 B per step, streaming, gap 8", timeit([&] { rows_kernel<128, true, 8><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 128 B per step, streaming, gap 32", timeit([&] { rows_kernel<128, true, 32><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 128 B per step, streaming, gap 64", timeit([&] { rows_kernel<128, true, 64><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 256 B per step, streaming, gap 64", timeit([&] { rows_kernel<256, true, 64><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 256 B per step, streaming, gap 127", timeit([&] { rows_kernel<256, true, 127><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
    rep("rows, 128 B, streaming, phase pairs (shift 1)", timeit([&] { rows_kernel<128, true, 32><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 1); }));
    {
        const unsigned shift = (P % 16 == 0) ? 0 : 1;  // P = 120: rows of equal phase are two apart
        const size_t per = 64ull << shift, groups = (nrows + per - 1) / per;
        const unsigned gblocks = (unsigned)((shift ? ((groups + 7) & ~7ull) : groups) << shift);
        rep("rows on the line grid, as the decompressor, 8 waves/CU", timeit([&] { rows_aligned_kernel<true, 0><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, gap 64", timeit([&] { rows_aligned_kernel<true, 64><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, plain stores", timeit([&] { rows_aligned_kernel<false, 0><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, consecutive rows per wave", timeit([&] { rows_aligned_kernel<true, 0><<<blocks, 64, 19968>>>(ts, val, row_bytes, nrows, 0); }));
        rep("the same, store policy: none", timeit([&] { rows_mode_kernel<0><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, store policy: nt", timeit([&] { rows_mode_kernel<1><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, store policy: sc0", timeit([&] { rows_mode_kernel<2><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, store policy: sc1", timeit([&] { rows_mode_kernel<3><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, store policy: sc0 sc1", timeit([&] { rows_mode_kernel<4><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, store policy: sc0 nt", timeit([&] { rows_mode_kernel<5><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, store policy: sc1 nt", timeit([&] { rows_mode_kernel<6><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
        rep("the same, store policy: sc0 sc1 nt", timeit([&] { rows_mode_kernel<7><<<gblocks, 64, 19968>>>(ts, val, row_bytes, nrows, shift); }));
    }
    CK(hipGetLastError());
    return 0;
}
