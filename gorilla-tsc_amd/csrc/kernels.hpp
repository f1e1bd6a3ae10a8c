// Kernel argument blocks and host-side launchers (internal to the library).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec_device.hpp"

namespace gtsc {

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
    // from the caller: what the streams compressed before looked like -- 1 = a few bits per point (long blocks of such
    // streams are then left to the lane-per-series kernel when the batch is large), 0 = dense or unknown
    uint32_t sparse;
    uint64_t* stats;       // may be null: [3] += words written, [4] += points coded (a sample of the blocks)
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
    uint32_t deep_ring;       // the tiled kernel's input rings: 0 = 8 words; 1 = the streams are dense, 16 words unless the batch's waves fill the units better with 8; 2 = 16 words
    uint64_t* stats;          // may be null: [0] += words read, [1] += points stored (a sample of the waves); [2] += steps | points << 32 of a sample of the wave-per-block kernel's blocks
    // Series whose capacity is at least long_threshold points are decoded by the wave-per-block kernel and left
    // alone by the lane-per-series kernels (~0: none) -- provided the batch holds at most long_max_series of
    // them, which both kernels read from *long_count (null: no such condition).  Set by the launcher.
    uint64_t long_threshold;
    const uint64_t* long_count;
    uint64_t long_max_series;
    uint64_t* long_count_buf; // from the caller: one device word the launcher may use for the count (may be null)
    uint64_t long_blocks_max; // from the caller: up to how many long blocks a batch goes to the wave-per-block kernel (0: the built-in figure)
    uint32_t run_skip;        // set by the launcher: the 8-word form may decode chunks by run skipping (see decode_tiled_kernel)
};

hipError_t launch_encode(const EncodeArgs& a, int format, hipStream_t stream);
// wave-per-block compressor over the series of `a` with at least a.long_threshold points (block_codec.hip)
hipError_t launch_encode_blocks(const EncodeArgs& a, int format, hipStream_t stream);
hipError_t launch_decode(const DecodeArgs& a, int format, hipStream_t stream);
// wave-per-block decompressor and the count of long series it shares with the lane-per-series kernels (block_codec.hip)
hipError_t launch_decode_blocks(const DecodeArgs& a, int format, hipStream_t stream);
hipError_t launch_count_long(uint64_t nseries, Layout pts, uint64_t long_threshold, uint64_t* count, hipStream_t stream);

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
