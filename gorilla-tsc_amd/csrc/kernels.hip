// HIP kernels for the Gorilla codec on gfx950 (MI355X).
//
// Parallelisation: one lane per series block, 64 blocks per wavefront.  The
// codec is a serial bit-level state machine per block (the window state and the
// bit position of point i depend on every earlier point), so lanes never talk
// to each other; what the wavefront shares is the memory traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "codec_device.hpp"
#include "kernels.hpp"

namespace gtsc {

// ------------------------------------------------------------------------
// Direct form: each lane reads its own points and writes its own words with
// plain global accesses.  Correct for every layout; used for tiny batches and
// as the fallback when a series is too ragged for the tiled form.
// ------------------------------------------------------------------------
template <int FMT>
__global__ __launch_bounds__(256) void encode_direct_kernel(EncodeArgs a)
{
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.nseries) return;
    const uint64_t p0 = a.pts.begin(s), np = a.pts.size(s);
    uint64_t* out = a.out_words + a.slots.begin(s);
    const uint64_t cap = a.slots.size(s);
    const int64_t bts = a.block_ts[s];
    uint32_t st = 0;
    auto sink = [&](uint32_t i, uint64_t w) {
        if (i < cap) out[i] = w; else st = ST_OVERFLOW;
    };
    Encoder<FMT> e;
    e.begin(bts, sink);
    for (uint64_t i = 0; i < np; i++) e.add(bts, a.ts[p0 + i], a.val[p0 + i], sink);
    uint32_t n = e.finish(sink);
    a.out_nwords[s] = n;
    if (a.status) a.status[s] = st;
    if (st) atomicOr(a.flags, st);
}

template <int FMT>
__global__ __launch_bounds__(256) void decode_direct_kernel(DecodeArgs a)
{
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.nseries) return;
    const uint64_t* in = a.words + a.slots.begin(s);
    uint64_t nw64 = a.in_nwords ? a.in_nwords[s] : a.slots.size(s);
    const uint32_t nw = nw64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nw64;
    const uint64_t p0 = a.pts.begin(s), cap = a.pts.size(s);
    auto src = [&](uint32_t i) -> uint64_t { return i < nw ? in[i] : 0ull; };
    Decoder<FMT> d;
    d.begin(nw, src);
    uint64_t n = 0;
    uint32_t st = 0;
    int64_t t; uint64_t v;
    while (d.next(t, v, src)) {
        if (n < cap) { a.ts[p0 + n] = t; a.val[p0 + n] = v; } else st = ST_OVERFLOW;
        n++;
    }
    st |= d.st;
    a.out_npoints[s] = n;
    if (a.status) a.status[s] = st;
    if (st) atomicOr(a.flags, st);
}

// ------------------------------------------------------------------------
// Tiled form.  A wavefront owns 64 consecutive series.  Points travel between
// HBM and the lanes through an LDS tile of 64 rows x CH points so that every
// global access is a coalesced run of 16-byte pieces inside one series' row,
// while each lane still consumes / produces its own series serially.
//
//   HBM  --(lane l: row 16i + l/4, piece l%4, 16 B)-->  registers  --> LDS tile
//   LDS tile --(lane j: its own row j, piece q)--> the lane's encoder
//
// The tile is XOR-swizzled: piece q of row r sits at r*64 + ((q ^ ((r>>2)&3))<<4).
// That keeps all three access shapes free of bank conflicts: the loader's
// ds_write_b128 (8 consecutive lanes = 2 rows x 4 pieces cover 128 B), the
// per-lane ds_read_b128 (16 lanes with distinct row%16 cover 256 B) and the
// decoder's flush read (rows 16i + l/4).
// ------------------------------------------------------------------------
constexpr uint32_t CH = 8;                    // points per row per chunk
constexpr uint32_t ROW_BYTES = CH * 8;        // 64
constexpr uint32_t PIECES = ROW_BYTES / 16;   // 16-byte pieces per row
constexpr uint32_t ROWS_PER_PASS = 64 / PIECES;
constexpr uint32_t PASSES = 64 / ROWS_PER_PASS;
constexpr uint32_t TILE_BYTES = 64 * ROW_BYTES;

__device__ __forceinline__ uint32_t tile_addr(uint32_t row, uint32_t piece)
{
    return row * ROW_BYTES + ((piece ^ ((row >> 2) & (PIECES - 1))) << 4);
}

struct alignas(16) U4 { uint32_t x, y, z, w; };
struct alignas(8) U2 { uint32_t x, y; };

// 16-byte global access at 8-byte alignment (series rows start at any point index)
__device__ __forceinline__ U4 load16(const void* p)
{
    U4 r;
    __builtin_memcpy(&r, __builtin_assume_aligned(p, 8), 16);
    return r;
}
__device__ __forceinline__ void store16(void* p, const U4& v)
{
    __builtin_memcpy(__builtin_assume_aligned(p, 8), &v, 16);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t y = (uint32_t)__shfl_xor((int)x, d, 64);
        x = x > y ? x : y;
    }
    return x;
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, uint32_t src)
{
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)src, 64);
    uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src, 64);
    return ((uint64_t)hi << 32) | lo;
}

template <int FMT>
__global__ __launch_bounds__(64) void encode_tiled_kernel(EncodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * TILE_BYTES];
    uint8_t* const tile_ts = lds;
    uint8_t* const tile_val = lds + TILE_BYTES;

    const uint32_t lane = threadIdx.x;
    const uint64_t s = (uint64_t)blockIdx.x * 64 + lane;
    const bool valid = s < a.nseries;
    const uint64_t p0 = valid ? a.pts.begin(s) : 0;
    const uint64_t np64 = valid ? a.pts.size(s) : 0;
    const uint32_t np = np64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)np64;
    const uint32_t nchunks = (wave_max_u32(np) + CH - 1) / CH;

    // loader role: rows this lane fetches pieces of, fixed for the whole tile
    const uint32_t piece = lane % PIECES;
    uint64_t row_p0[PASSES];
    uint32_t row_np[PASSES];
#pragma unroll
    for (uint32_t i = 0; i < PASSES; i++) {
        const uint32_t r = i * ROWS_PER_PASS + lane / PIECES;
        row_p0[i] = shfl_u64(p0, r);
        row_np[i] = (uint32_t)__shfl((int)np, (int)r, 64);
    }
    U4 rt[PASSES], rv[PASSES];
    auto fetch = [&](uint32_t c) {
        const uint32_t first = c * CH + 2 * piece;  // first of the two points of this piece
#pragma unroll
        for (uint32_t i = 0; i < PASSES; i++) {
            const int64_t rem = (int64_t)row_np[i] - (int64_t)first;
            const int64_t* pt = a.ts + row_p0[i] + first;
            const uint64_t* pv = a.val + row_p0[i] + first;
            U4 t = {0, 0, 0, 0}, v = {0, 0, 0, 0};
            if (rem >= 2) {
                t = load16(pt);
                v = load16(pv);
            } else if (rem == 1) {
                const uint64_t t1 = (uint64_t)pt[0], v1 = pv[0];
                t.x = (uint32_t)t1; t.y = (uint32_t)(t1 >> 32);
                v.x = (uint32_t)v1; v.y = (uint32_t)(v1 >> 32);
            }
            rt[i] = t;
            rv[i] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (uint32_t i = 0; i < PASSES; i++) {
            const uint32_t r = i * ROWS_PER_PASS + lane / PIECES;
            *reinterpret_cast<U4*>(tile_ts + tile_addr(r, piece)) = rt[i];
            *reinterpret_cast<U4*>(tile_val + tile_addr(r, piece)) = rv[i];
        }
    };

    uint64_t* const out = a.out_words + (valid ? a.slots.begin(s) : 0);
    const uint64_t cap = valid ? a.slots.size(s) : 0;
    const int64_t bts = valid ? a.block_ts[s] : 0;
    uint32_t st = 0;
    auto sink = [&](uint32_t i, uint64_t w) {
        if (i < cap) out[i] = w; else st = ST_OVERFLOW;
    };
    Encoder<FMT> e;
    if (nchunks) fetch(0);
    if (valid) e.begin(bts, sink);
    for (uint32_t c = 0; c < nchunks; c++) {
        __syncthreads();  // every lane has read the previous chunk
        stage();
        __syncthreads();
        if (c + 1 < nchunks) fetch(c + 1);  // in flight while this chunk is encoded
#pragma unroll 1
        for (uint32_t q = 0; q < PIECES; q++) {
            const U4 t = *reinterpret_cast<const U4*>(tile_ts + tile_addr(lane, q));
            const U4 v = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q));
            const uint32_t k = c * CH + 2 * q;
            if (k < np) e.add(bts, (int64_t)(((uint64_t)t.y << 32) | t.x), ((uint64_t)v.y << 32) | v.x, sink);
            if (k + 1 < np) e.add(bts, (int64_t)(((uint64_t)t.w << 32) | t.z), ((uint64_t)v.w << 32) | v.z, sink);
        }
    }
    if (valid) {
        const uint32_t n = e.finish(sink);
        a.out_nwords[s] = n;
        if (a.status) a.status[s] = st;
        if (st) atomicOr(a.flags, st);
    }
}

template <int FMT>
__global__ __launch_bounds__(64) void decode_tiled_kernel(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * TILE_BYTES + 64 * 4];
    uint8_t* const tile_ts = lds;
    uint8_t* const tile_val = lds + TILE_BYTES;
    uint32_t* const row_count = reinterpret_cast<uint32_t*>(lds + 2 * TILE_BYTES);

    const uint32_t lane = threadIdx.x;
    const uint64_t s = (uint64_t)blockIdx.x * 64 + lane;
    const bool valid = s < a.nseries;
    const uint64_t* in = a.words + (valid ? a.slots.begin(s) : 0);
    uint64_t nw64 = valid ? (a.in_nwords ? a.in_nwords[s] : a.slots.size(s)) : 0;
    const uint32_t nw = nw64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nw64;
    const uint64_t p0 = valid ? a.pts.begin(s) : 0;
    const uint64_t cap64 = valid ? a.pts.size(s) : 0;
    const uint32_t cap = cap64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap64;

    // flusher role: rows this lane stores pieces of
    const uint32_t piece = lane % PIECES;
    uint64_t row_p0[PASSES];
#pragma unroll
    for (uint32_t i = 0; i < PASSES; i++) row_p0[i] = shfl_u64(p0, i * ROWS_PER_PASS + lane / PIECES);

    auto src = [&](uint32_t i) -> uint64_t { return i < nw ? in[i] : 0ull; };
    Decoder<FMT> d;
    bool active = valid;
    if (valid) d.begin(nw, src);
    uint32_t n = 0;       // pairs decoded
    uint32_t st = 0;
    for (uint32_t c = 0; __any(active); c++) {
#pragma unroll 1
        for (uint32_t k = 0; k < CH; k++) {
            if (active) {
                int64_t t; uint64_t v;
                if (d.next(t, v, src)) {
                    const uint32_t ad = tile_addr(lane, k >> 1) + ((k & 1) << 3);
                    *reinterpret_cast<uint64_t*>(tile_ts + ad) = (uint64_t)t;
                    *reinterpret_cast<uint64_t*>(tile_val + ad) = v;
                    n++;
                } else {
                    active = false;
                }
            }
        }
        row_count[lane] = n < cap ? n : cap;  // points of this row that may be stored
        __syncthreads();
        const uint32_t first = c * CH + 2 * piece;
#pragma unroll
        for (uint32_t i = 0; i < PASSES; i++) {
            const uint32_t r = i * ROWS_PER_PASS + lane / PIECES;
            const int64_t rem = (int64_t)row_count[r] - (int64_t)first;
            if (rem >= 1) {
                const U4 t = *reinterpret_cast<const U4*>(tile_ts + tile_addr(r, piece));
                const U4 v = *reinterpret_cast<const U4*>(tile_val + tile_addr(r, piece));
                int64_t* pt = a.ts + row_p0[i] + first;
                uint64_t* pv = a.val + row_p0[i] + first;
                if (rem >= 2) {
                    store16(pt, t);
                    store16(pv, v);
                } else {
                    pt[0] = (int64_t)(((uint64_t)t.y << 32) | t.x);
                    pv[0] = ((uint64_t)v.y << 32) | v.x;
                }
            }
        }
        __syncthreads();
    }
    if (valid) {
        if (n > cap) st |= ST_OVERFLOW;
        st |= d.st;
        a.out_npoints[s] = n;
        if (a.status) a.status[s] = st;
        if (st) atomicOr(a.flags, st);
    }
}

// ------------------------------------------------------------------------
// Compaction: exclusive scan of image lengths, then a wave-per-series copy.
// ------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint64_t wave_inclusive_scan(uint64_t x)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// exclusive scan of the block's SCAN_TILE items; returns the block total in `total`
__device__ __forceinline__ void block_exclusive_scan(uint64_t (&item)[SCAN_ITEMS], uint64_t& total)
{
    __shared__ uint64_t wave_sum[SCAN_THREADS / 64];
    uint64_t local = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { uint64_t t = item[k]; item[k] = local; local += t; }
    uint64_t incl = wave_inclusive_scan(local);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        uint64_t ws = wave_sum[w];
        if (w < wave) base += ws;
        tot += ws;
    }
    __syncthreads();
    uint64_t excl = base + incl - local;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) item[k] += excl;
    total = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums_kernel(const uint64_t* in, uint64_t n, uint64_t* tile_sum)
{
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t item[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) item[k] = base + k < n ? in[base + k] : 0;
    uint64_t total;
    block_exclusive_scan(item, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// single block: in-place exclusive scan of tile sums (any count, looped)
__global__ __launch_bounds__(SCAN_THREADS) void scan_spine_kernel(uint64_t* tile_sum, uint64_t ntiles)
{
    uint64_t carry = 0;
    for (uint64_t start = 0; start < ntiles; start += SCAN_TILE) {
        uint64_t base = start + (uint64_t)threadIdx.x * SCAN_ITEMS;
        uint64_t item[SCAN_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) item[k] = base + k < ntiles ? tile_sum[base + k] : 0;
        uint64_t total;
        block_exclusive_scan(item, total);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < ntiles) tile_sum[base + k] = item[k] + carry;
        carry += total;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_finish_kernel(const uint64_t* in, uint64_t n, const uint64_t* tile_excl, uint64_t* out /* n+1 */)
{
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t item[SCAN_ITEMS], orig[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { orig[k] = base + k < n ? in[base + k] : 0; item[k] = orig[k]; }
    uint64_t total;
    block_exclusive_scan(item, total);
    const uint64_t off = tile_excl[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) {
            out[base + k] = item[k] + off;
            if (base + k == n - 1) out[n] = item[k] + off + orig[k];
        }
    }
}

__global__ __launch_bounds__(256) void compact_copy_kernel(uint64_t nseries, const uint64_t* slot_words, Layout slots,
                                                           const uint64_t* nwords, uint64_t* dst, const uint64_t* dst_off)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t s = wave; s < nseries; s += nwaves) {
        const uint64_t* src = slot_words + slots.begin(s);
        uint64_t* d = dst + dst_off[s];
        const uint64_t n = nwords[s];
        for (uint64_t i = lane; i < n; i += 64) d[i] = src[i];
    }
}

// ------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------
static bool use_direct_kernels()
{
    static const bool v = [] { const char* e = getenv("GTSC_KERNEL"); return e && !strcmp(e, "direct"); }();
    return v;
}

hipError_t launch_encode(const EncodeArgs& a, int format, hipStream_t stream)
{
    if (a.nseries == 0) return hipSuccess;
    if (use_direct_kernels()) {
        const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
        if (format == FMT_V2) hipLaunchKernelGGL(encode_direct_kernel<FMT_V2>, dim3(blocks), dim3(256), 0, stream, a);
        else                  hipLaunchKernelGGL(encode_direct_kernel<FMT_V1>, dim3(blocks), dim3(256), 0, stream, a);
    } else {
        const uint32_t blocks = (uint32_t)((a.nseries + 63) / 64);
        if (format == FMT_V2) hipLaunchKernelGGL(encode_tiled_kernel<FMT_V2>, dim3(blocks), dim3(64), 0, stream, a);
        else                  hipLaunchKernelGGL(encode_tiled_kernel<FMT_V1>, dim3(blocks), dim3(64), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_decode(const DecodeArgs& a, int format, hipStream_t stream)
{
    if (a.nseries == 0) return hipSuccess;
    if (use_direct_kernels()) {
        const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
        if (format == FMT_V2) hipLaunchKernelGGL(decode_direct_kernel<FMT_V2>, dim3(blocks), dim3(256), 0, stream, a);
        else                  hipLaunchKernelGGL(decode_direct_kernel<FMT_V1>, dim3(blocks), dim3(256), 0, stream, a);
    } else {
        const uint32_t blocks = (uint32_t)((a.nseries + 63) / 64);
        if (format == FMT_V2) hipLaunchKernelGGL(decode_tiled_kernel<FMT_V2>, dim3(blocks), dim3(64), 0, stream, a);
        else                  hipLaunchKernelGGL(decode_tiled_kernel<FMT_V1>, dim3(blocks), dim3(64), 0, stream, a);
    }
    return hipGetLastError();
}

uint64_t scan_scratch_words(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

hipError_t launch_exclusive_scan(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* scratch, hipStream_t stream)
{
    if (n == 0) return hipMemsetAsync(out, 0, sizeof(uint64_t), stream);
    const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((uint32_t)ntiles), dim3(SCAN_THREADS), 0, stream, in, n, scratch);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, scratch, ntiles);
    hipLaunchKernelGGL(scan_finish_kernel, dim3((uint32_t)ntiles), dim3(SCAN_THREADS), 0, stream, in, n, scratch, out);
    return hipGetLastError();
}

hipError_t launch_compact_copy(uint64_t nseries, const uint64_t* slot_words, Layout slots, const uint64_t* nwords,
                               uint64_t* dst, const uint64_t* dst_off, hipStream_t stream)
{
    if (nseries == 0) return hipSuccess;
    uint64_t blocks = (nseries + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(compact_copy_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, nseries, slot_words, slots,
                       nwords, dst, dst_off);
    return hipGetLastError();
}

}  // namespace gtsc
