// Microbenchmark: HBM read bandwidth of the access shapes the tiled codec kernels use.
// Each wave owns 64 rows of ROWB bytes (a "tile"); per step it loads SEG bytes from each row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned U4 __attribute__((ext_vector_type(4)));

// SEG bytes per row per step (16..1024), rows of rowb bytes, 64 rows per wave
template <int SEG, int DEPTH, int LDSB = 0>
__global__ __launch_bounds__(64) void read_rows(const char* a, const char* b, size_t rowb, unsigned* sink)
{
    __shared__ unsigned char occ[LDSB ? LDSB : 1];
    if (LDSB) occ[threadIdx.x] = 1;
    const unsigned lane = threadIdx.x;
    constexpr int LPR = SEG / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;          // rows per instruction
    constexpr int NI = 64 / RPI;           // instructions per step per array
    const size_t tile = (size_t)blockIdx.x * 64 * rowb;
    unsigned acc = 0;
    const int steps = (int)(rowb / SEG);
    for (int c = 0; c < steps; c += DEPTH) {
        U4 ra[DEPTH][NI], rb[DEPTH][NI];
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const size_t row = (size_t)i * RPI + lane / LPR;
                const size_t off = tile + row * rowb + (size_t)(c + d) * SEG + (lane % LPR) * 16;
                ra[d][i] = *reinterpret_cast<const U4*>(a + off);
                rb[d][i] = *reinterpret_cast<const U4*>(b + off);
            }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int i = 0; i < NI; i++) acc ^= ra[d][i].x ^ ra[d][i].w ^ rb[d][i].y ^ rb[d][i].z;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// seg64 pattern with the LDS round trip of the codec kernels: MODE 0 = occupancy only (LDS
// declared, unused), 1 = stage to LDS then read own row, 2 = same with the next chunk's
// loads issued before the LDS reads (software prefetch)
template <int MODE, int LDSB>
__global__ __launch_bounds__(64) void read_rows_lds(const char* a, const char* b, size_t rowb, unsigned* sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
    const unsigned lane = threadIdx.x;
    const size_t tile = (size_t)blockIdx.x * 64 * rowb;
    unsigned acc = 0;
    const int steps = (int)(rowb / 64);
    auto taddr = [](unsigned row, unsigned piece) { return row * 64 + ((piece ^ ((row >> 2) & 3)) << 4); };
    U4 ra[4], rb[4];
#define FETCH(c)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; i++) {                                      \
        const size_t row = (size_t)i * 16 + lane / 4;                                    \
        const size_t off = tile + row * rowb + (size_t)(c) * 64 + (lane % 4) * 16;       \
        ra[i] = *reinterpret_cast<const U4*>(a + off);                                   \
        rb[i] = *reinterpret_cast<const U4*>(b + off);                                   \
    }
    if (MODE == 0) {
        lds[lane] = 1;
        for (int c = 0; c < steps; c++) {
            FETCH(c)
#pragma unroll
            for (int i = 0; i < 4; i++) acc ^= ra[i].x ^ ra[i].w ^ rb[i].y ^ rb[i].z;
        }
        acc ^= lds[lane ^ 1];
    } else {
        FETCH(0)
        for (int c = 0; c < steps; c++) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned r = i * 16 + lane / 4;
                *reinterpret_cast<U4*>(lds + taddr(r, lane % 4)) = ra[i];
                *reinterpret_cast<U4*>(lds + 4096 + taddr(r, lane % 4)) = rb[i];
            }
            __syncthreads();
            if (MODE == 2 && c + 1 < steps) { FETCH(c + 1) }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const U4 t = *reinterpret_cast<const U4*>(lds + taddr(lane, q));
                const U4 v = *reinterpret_cast<const U4*>(lds + 4096 + taddr(lane, q));
                acc ^= t.x ^ t.w ^ v.y ^ v.z;
            }
            if (MODE == 1 && c + 1 < steps) { FETCH(c + 1) }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

struct alignas(16) RowInfo { long long base; unsigned lo, hi; };
template <int MODE>
__global__ __launch_bounds__(64) void read_rows128_lds(const char* a, const char* b, size_t rowb, unsigned* sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384 + 1024];
    RowInfo* rows = reinterpret_cast<RowInfo*>(lds + 16384);
    const unsigned lane = threadIdx.x;
    const size_t tile = (size_t)blockIdx.x * 64 * rowb;
    rows[lane] = RowInfo{(long long)(tile + lane * rowb), 0u, (unsigned)rowb};
    __syncthreads();
    unsigned acc = 0;
    const int steps = (int)(rowb / 128);
    auto taddr = [](unsigned row, unsigned piece) { return row * 128 + ((piece ^ ((row >> 1) & 7)) << 4); };
    U4 ra[8], rb[8];
#define FETCH128(c)                                                                      \
    _Pragma("unroll") for (int i = 0; i < 8; i++) {                                      \
        const RowInfo ri = rows[i * 8 + lane / 8];                                       \
        const size_t off = (size_t)ri.base + (size_t)(c) * 128 + (lane % 8) * 16;        \
        ra[i] = *reinterpret_cast<const U4*>(a + off);                                   \
        rb[i] = *reinterpret_cast<const U4*>(b + off);                                   \
    }
    FETCH128(0)
    for (int c = 0; c < steps; c++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned r = i * 8 + lane / 8;
            *reinterpret_cast<U4*>(lds + taddr(r, lane % 8)) = ra[i];
            *reinterpret_cast<U4*>(lds + 8192 + taddr(r, lane % 8)) = rb[i];
        }
        __syncthreads();
        if (MODE == 2 && c + 1 < steps) { FETCH128(c + 1) }
#pragma unroll 1
        for (int q = 0; q < 8; q++) {
            const U4 t = *reinterpret_cast<const U4*>(lds + taddr(lane, q));
            const U4 v = *reinterpret_cast<const U4*>(lds + 8192 + taddr(lane, q));
            acc ^= t.x ^ t.w ^ v.y ^ v.z;
        }
        if (MODE == 1 && c + 1 < steps) { FETCH128(c + 1) }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// stores: each wave writes SEG bytes per row per step to its 64 rows of two arrays
template <int SEG>
__global__ __launch_bounds__(64) void write_rows(char* a, char* b, size_t rowb)
{
    const unsigned lane = threadIdx.x;
    constexpr int LPR = SEG / 16, RPI = 64 / LPR, NI = 64 / RPI;
    const size_t tile = (size_t)blockIdx.x * 64 * rowb;
    const int steps = (int)(rowb / SEG);
    for (int c = 0; c < steps; c++) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const size_t row = (size_t)i * RPI + lane / LPR;
            const size_t off = tile + row * rowb + (size_t)c * SEG + (lane % LPR) * 16;
            const U4 v = {lane, (unsigned)c, (unsigned)i, 7u};
            *reinterpret_cast<U4*>(a + off) = v;
            *reinterpret_cast<U4*>(b + off) = v;
        }
        __builtin_amdgcn_s_sleep(8);  // spread the steps in time a little, like a kernel that computes in between
    }
}

// contiguous: wave streams its 64*rowb bytes in 1 KiB instructions
template <int DEPTH>
__global__ __launch_bounds__(64) void read_contig(const char* a, const char* b, size_t rowb, unsigned* sink)
{
    const unsigned lane = threadIdx.x;
    const size_t tile = (size_t)blockIdx.x * 64 * rowb;
    const int steps = (int)(64 * rowb / 1024);
    unsigned acc = 0;
    for (int c = 0; c < steps; c += DEPTH) {
        U4 ra[DEPTH], rb[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const size_t off = tile + (size_t)(c + d) * 1024 + lane * 16;
            ra[d] = *reinterpret_cast<const U4*>(a + off);
            rb[d] = *reinterpret_cast<const U4*>(b + off);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc ^= ra[d].x ^ ra[d].w ^ rb[d].y ^ rb[d].z;
    }
    if (acc == 0x12345678u) sink[0] = acc;
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
    const size_t rowb = 1024, nrows = 1000000 / 64 * 64;
    const size_t bytes = rowb * nrows;
    char *a, *b; unsigned* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const unsigned blocks = (unsigned)(nrows / 64);
    auto rep = [&](const char* name, float ms) { printf("%-28s %.4f ms  %.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); };
    rep("contig depth1", timeit([&] { read_contig<1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("contig depth4", timeit([&] { read_contig<4><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg64 depth1", timeit([&] { read_rows<64, 1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg64 depth4", timeit([&] { read_rows<64, 4><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg32 depth1", timeit([&] { read_rows<32, 1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg64 depth2", timeit([&] { read_rows<64, 2><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg128 depth1", timeit([&] { read_rows<128, 1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg128 depth2", timeit([&] { read_rows<128, 2><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg256 depth1", timeit([&] { read_rows<256, 1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg512 depth1", timeit([&] { read_rows<512, 1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("rows seg1024 depth1", timeit([&] { read_rows<1024, 1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg128 d1 lds17K (9 w/CU)", timeit([&] { read_rows<128, 1, 17408><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg128 d1 lds32K (5 w/CU)", timeit([&] { read_rows<128, 1, 32768><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg128 d2 lds32K (5 w/CU)", timeit([&] { read_rows<128, 2, 32768><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg64 d1 lds17K", timeit([&] { read_rows<64, 1, 17408><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg128 lds round trip", timeit([&] { read_rows128_lds<1><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg128 lds + prefetch", timeit([&] { read_rows128_lds<2><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("WRITE rows seg64", timeit([&] { write_rows<64><<<blocks, 64>>>(a, b, rowb); }));
    rep("WRITE rows seg128", timeit([&] { write_rows<128><<<blocks, 64>>>(a, b, rowb); }));
    rep("WRITE rows seg256", timeit([&] { write_rows<256><<<blocks, 64>>>(a, b, rowb); }));
    rep("WRITE rows seg1024", timeit([&] { write_rows<1024><<<blocks, 64>>>(a, b, rowb); }));
    rep("seg64 occupancy 8K lds", timeit([&] { read_rows_lds<0, 8192><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg64 occupancy 16K lds", timeit([&] { read_rows_lds<0, 16384><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg64 lds round trip", timeit([&] { read_rows_lds<1, 8192><<<blocks, 64>>>(a, b, rowb, sink); }));
    rep("seg64 lds + prefetch", timeit([&] { read_rows_lds<2, 8192><<<blocks, 64>>>(a, b, rowb, sink); }));
    CK(hipGetLastError());
    return 0;
}
