// HIP kernels for the Gorilla codec on gfx950 (MI355X).
//
// Parallelisation: one lane per series block, 64 blocks per wavefront.  The
// codec is a serial bit-level state machine per block (the window state and the
// bit position of point i depend on every earlier point), so lanes never talk
// to each other; what the wavefront shares is the memory traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
hipError_t launch_encode(const EncodeArgs& a, int format, hipStream_t stream)
{
    if (a.nseries == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
    if (format == FMT_V2) hipLaunchKernelGGL(encode_direct_kernel<FMT_V2>, dim3(blocks), dim3(256), 0, stream, a);
    else                  hipLaunchKernelGGL(encode_direct_kernel<FMT_V1>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_decode(const DecodeArgs& a, int format, hipStream_t stream)
{
    if (a.nseries == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
    if (format == FMT_V2) hipLaunchKernelGGL(decode_direct_kernel<FMT_V2>, dim3(blocks), dim3(256), 0, stream, a);
    else                  hipLaunchKernelGGL(decode_direct_kernel<FMT_V1>, dim3(blocks), dim3(256), 0, stream, a);
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
