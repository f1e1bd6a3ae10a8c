// Kernel argument blocks and host-side launchers (internal to the library).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec_device.hpp"

namespace gtsc {

// What a context lets its caller tune (gtsc_set_tunable; the environment variable named with each, if set when
// the context is created, replaces the default).  Nothing here changes results, only which kernel produces them.
struct Tunables {
    uint64_t long_points = 1024;           // blocks of at least this many points are "long": a wave or workgroup per block may take them
    uint64_t vlong_points = 16384;         // ... and from this many on "very long"
    uint64_t wave_decode_max = 12288;      // batches of up to this many long blocks: the wave-per-block decompressor takes them
    uint64_t wave_decode_max_dense = 2048; // the same when the long blocks' images come to more than dense_bits per point
    uint64_t wg_decode_max = 512;          // batches of up to this many long blocks: the sparse ones get a workgroup each
    uint64_t wg_decode_vlong_max = 640;    // very long blocks with sparse images get a workgroup each while together they come to no more than this many times the longest of them
    uint32_t wg_dense_bits = 40;           // a block counts as sparse below this many compressed bits per point of capacity
    uint32_t dense_bits = 64;              // see wave_decode_max_dense
    uint64_t sparse_encode_min = 57344;    // batches of at least this many long blocks of sparse streams: the lane-per-series compressor
    uint64_t coop_encode_max = 384;        // batches of up to this many long blocks: the compressor gives each a workgroup
    uint32_t sparse_hint = 0;              // 1: the streams to compress are sparse (a few bits per point)
    uint32_t decode_ring = 0;              // input rings of the lane-per-series decompressor: 8 or 16 words; 0 = by the batch's density, on the device
    uint32_t run_skip = 1;                 // 0: the lane-per-series kernels keep to their lock-step loops
    uint32_t direct = 0;                   // 1: the plain lane-per-series kernels only
};

struct EncodeArgs {
    uint64_t nseries;
    const int64_t* block_ts;
    const int64_t* ts;
    const uint64_t* val;
    Layout pts;          // points of series s
    uint64_t* out_words;
    Layout slots;        // output slot of series s
    uint64_t* out_nwords;
    uint32_t* status;    // may be null
    uint32_t* flags;     // one word: OR of all status bits raised
    uint64_t* out_nbits; // may be null: length of each stream in bits
    uint32_t no_close;   // nonzero: leave the blocks open (image = snapshot of getLongArray())
    int64_t* dfcm_tables;  // DFCM predictor: nseries zero-filled tables of dfcm_mask + 1 entries; null = last-value predictor
    uint32_t dfcm_mask;
    uint32_t group_shift;  // set by the launcher (tiled kernels): see series_of_lane()
    // set by the launcher: series with at least this many points are coded by the wave-per-block
    // kernel and left alone by the lane-per-series kernels (~0: no such series)
    uint64_t long_threshold;
    uint32_t run_skip;     // set by the launcher: the tiled kernel may code chunks by run skipping (see encode_tiled_kernel)
    Tunables tun;          // from the caller
    uint32_t* dispatch;    // from the caller, host memory, may be null: receives the DISPATCH_* bits of the kernels enqueued
};

// What a batch of images looks like, worked out on the device before the decompressors start (plan_decode_kernel):
// every kernel of the batch reads it and they agree on who takes which block.
struct DecodePlan {
    uint64_t nlong, nvlong;          // series with a capacity of at least long_points / vlong_points points
    uint64_t words_long, cap_long;   // the long ones' image words and capacities, summed
    uint64_t words_all, cap_all;     // the same over every 64th series of the batch
    uint64_t cap_vlong_sparse;       // capacities of the very long series with sparse images, summed ...
    uint64_t cap_vlong_longest;      // ... and the largest of them
    uint64_t words_longest;          // the largest image among the long ones
    // The workgroup-per-block decompressor's blocks, largest images first (wg_queue_kernel fills it in where a batch
    // has more series than the device has compute units): sixteen classes by image size relative to the largest,
    // each a list of series numbers, and the ticket the workgroups draw their next block with.  A block takes as
    // long as its chain of points, whatever else runs: the long ones must start first, or the last of them to be
    // drawn is what the whole batch waits for.
    static constexpr uint32_t QUEUE_CLASSES = 16, QUEUE_CAP = 2048;
    uint32_t queue_count[QUEUE_CLASSES];
    uint32_t queue_ticket;
    uint32_t queue_overflow;         // a class had more blocks than its list holds: the workgroups go by series number instead
    uint32_t queue_list[QUEUE_CLASSES][QUEUE_CAP];
    static constexpr size_t HEADER_BYTES = 9 * 8 + (QUEUE_CLASSES + 2) * 4;   // what a new plan zeroes
};

struct DecodeArgs {
    uint64_t nseries;
    const uint64_t* words;
    Layout slots;             // image slot of series s
    const uint64_t* in_nwords;  // may be null
    int64_t* ts;
    uint64_t* val;
    Layout pts;               // capacity slot of series s
    uint64_t* out_npoints;
    uint32_t* status;         // may be null
    uint32_t* flags;
    int64_t* dfcm_tables;     // as in EncodeArgs
    uint32_t dfcm_mask;
    uint32_t group_shift;     // as in EncodeArgs
    // Series whose capacity is at least long_threshold points may be decoded a wave or a workgroup per block
    // (~0: none; set by the launcher); block_owner() says which kernel takes one, from the plan.
    uint64_t long_threshold;
    const DecodePlan* plan;   // set by the launcher (null: no long blocks, and the ring is not chosen by density)
    DecodePlan* plan_buf;     // from the caller: device memory for the plan
    uint32_t run_skip;        // set by the launcher: the 8-word form may decode chunks by run skipping (see decode_tiled_kernel)
    uint32_t ring_by_plan;    // set by the launcher: the lane-per-series kernel runs in both ring forms and the plan says which one works
    Tunables tun;             // from the caller
    uint32_t* dispatch;       // as in EncodeArgs
};

// What the first points of a dense image say about which decompressor suits it: true where its timestamps stray from
// their cadence early on (two of the delta-of-deltas of points 3 to 6 not zero: such streams have points of
// ever-changing length, which is what the wave-per-block decompressor is slowest at), and where points 7 to 10 are of
// one length on a regular cadence (values that reuse one window, once the first few have settled it: the
// workgroup-per-block decompressor takes such a stream 1,024 points a round).
template <int FMT>
__device__ inline bool workgroup_suits_early(const uint64_t* image, uint64_t nw64)
{
    const uint32_t nw = nw64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nw64;
    auto src = [image, nw](uint32_t i) -> uint64_t { return i < nw ? image[i] : 0ull; };
    Decoder<FMT> d;
    d.begin(nw, src);
    int64_t t = 0, before = 0, delta = 0;
    uint64_t v;
    uint32_t strays = 0, same_length = 0, length = 0;
    for (int i = 0; i < 10; i++) {
        const uint64_t at = (uint64_t)d.widx * 64 + d.off;
        if (!d.next(t, v, src)) return false;
        const uint32_t bits = (uint32_t)((uint64_t)d.widx * 64 + d.off - at);
        if (i >= 2 && t - before != delta) strays++;
        if (i == 5 && strays >= 2) return true;
        if (i >= 6 && (i == 6 || bits == length)) same_length++;
        length = bits;
        delta = t - before;
        before = t;
    }
    return strays == 0 && same_length == 4 && length > 3;
}

// Which decompressor takes series s: the lane-per-series kernel (a lane), the wave-per-block kernel or the
// workgroup-per-block kernel.  Sparse streams of few or very long blocks get a workgroup each (it follows the chain
// of point starts several times faster where points are short or irregular), and so do dense ones of few blocks
// whose timestamps stray from their cadence or whose points are of one length; other long blocks get a wave each while the batch holds few enough of
// them to leave the lanes of the lane-per-series kernel mostly idle -- fewer when the images are dense, because
// irregular long points are what the wave-per-block kernel is slowest at.
constexpr int OWNER_LANE = 0, OWNER_WAVE = 1, OWNER_WG = 2;
template <int FMT>
__device__ __forceinline__ int block_owner(const DecodeArgs& a, uint64_t s)
{
    const uint64_t cap = a.pts.size(s);
    if (cap < a.long_threshold) return OWNER_LANE;
    const DecodePlan& pl = *a.plan;
    const uint64_t nw = a.in_nwords ? a.in_nwords[s] : a.slots.size(s);
    const bool sparse = nw * 64 < cap * a.tun.wg_dense_bits;
    // (workgroups take their blocks a thousand or so at a time, waves all at once but four times slower: what counts is
    // how many times the longest block the sparse very long ones come to together)
    if (sparse && (pl.nlong <= a.tun.wg_decode_max ||
                   (cap >= a.tun.vlong_points && pl.cap_vlong_sparse <= a.tun.wg_decode_vlong_max * pl.cap_vlong_longest))) return OWNER_WG;
    if (!sparse && pl.nlong <= a.tun.wg_decode_max && a.tun.wg_decode_max != 0) {
        const uint64_t ssz = a.slots.size(s);
        if (workgroup_suits_early<FMT>(a.words + a.slots.begin(s), nw < ssz ? nw : ssz)) return OWNER_WG;
    }
    const bool dense_batch = pl.words_long * 64 > pl.cap_long * a.tun.dense_bits;
    return pl.nlong <= (dense_batch ? a.tun.wave_decode_max_dense : a.tun.wave_decode_max) ? OWNER_WAVE : OWNER_LANE;
}

hipError_t launch_encode(const EncodeArgs& a, int format, hipStream_t stream);
// wave-per-block compressor over the series of `a` with at least a.long_threshold points (block_codec.hip)
hipError_t launch_encode_blocks(const EncodeArgs& a, int format, hipStream_t stream);
// A second stream for launch_decode: the workgroup-per-block decompressor goes there, beside the other kernels of
// the batch instead of before them (each of the three is as slow as its longest block and leaves most of the
// device idle towards its end).  `planned` and `joined` are events of the caller's, used for nothing else.
struct DecodeFork { hipStream_t side; hipEvent_t planned, joined; };
hipError_t launch_decode(const DecodeArgs& a, int format, hipStream_t stream, const DecodeFork* fork = nullptr);
// wave-per-block decompressor (block_codec.hip) and workgroup-per-block decompressor (block_decode_wg.hip): each
// takes the series of `a` that block_owner() gives it
hipError_t launch_decode_blocks(const DecodeArgs& a, int format, hipStream_t stream);
hipError_t launch_decode_blocks_wg(const DecodeArgs& a, int format, hipStream_t stream);
// fills a.plan_buf (block_codec.hip)
hipError_t launch_decode_plan(const DecodeArgs& a, hipStream_t stream);
// Which kernels a launcher enqueued, for gtsc_last_dispatch(): the values of GTSC_DISPATCH_* in include/gtsc.h
constexpr uint32_t DISPATCH_LANE_DIRECT = 1u, DISPATCH_LANE_TILED = 2u, DISPATCH_LANE_RING16 = 4u, DISPATCH_LANE_BOTH_RINGS = 8u,
                   DISPATCH_WAVE = 16u, DISPATCH_WG = 32u, DISPATCH_PLAN = 64u, DISPATCH_COOP = 128u, DISPATCH_SPLIT = 256u;

// out[0..n] = exclusive prefix sums of in[0..n) (out[n] = total); scratch holds
// scan_scratch_words(n) words.
uint64_t scan_scratch_words(uint64_t n);
// The library's static device scratch array (see gtsc_compact_device_noalloc): its size in words, and
// its address on the current device (looked up once per device, then remembered).
constexpr uint64_t NOALLOC_SCRATCH_WORDS = 8192;
hipError_t noalloc_scratch_of_current_device(uint64_t** out);
hipError_t launch_exclusive_scan(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* scratch, hipStream_t stream);
// word images <-> ByteBuffer byte images (big-endian bytes of the same bit stream)
hipError_t launch_words_to_bytes(uint64_t nseries, const uint64_t* words, Layout wslots, const uint64_t* nbits,
                                 uint8_t* bytes, Layout bslots, uint64_t* out_nbytes, uint32_t* flags, hipStream_t stream);
hipError_t launch_bytes_to_words(uint64_t nseries, const uint8_t* bytes, Layout bslots, const uint64_t* nbytes,
                                 uint64_t* words, Layout wslots, uint64_t* out_nwords, uint32_t* flags, hipStream_t stream);
hipError_t launch_compact_copy(uint64_t nseries, const uint64_t* slot_words, Layout slots, const uint64_t* nwords,
                               uint64_t* dst, const uint64_t* dst_off, hipStream_t stream);

}  // namespace gtsc
