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

#include <mutex>

#include "codec_device.hpp"
#include "diag.hpp"   // measurement hooks: empty in the shipped library
#include "kernels.hpp"
#include "wave_util.hpp"

namespace gtsc {

// ------------------------------------------------------------------------
// Direct form: each lane reads its own points and writes its own words with
// plain global accesses.  Correct for every layout; used for tiny batches and
// as the fallback when a series is too ragged for the tiled form.
// ------------------------------------------------------------------------
// word sink of the direct form: the lane's own slot in global memory
struct GlobalSink {
    static constexpr bool ALWAYS = false;
    uint64_t* out; uint64_t cap; uint32_t* st;
    __device__ __forceinline__ void operator()(uint32_t i, uint64_t w) const { if (i < cap) out[i] = w; else *st = ST_OVERFLOW; }
};

template <int FMT, int PRED = PRED_LAST>
__global__ __launch_bounds__(256) void encode_direct_kernel(EncodeArgs a)
{
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.nseries) return;
    const uint64_t p0 = a.pts.begin(s), np = a.pts.size(s);
    if (np >= a.long_threshold) return;  // the wave-per-block kernel's
    uint64_t* out = a.out_words + a.slots.begin(s);
    const uint64_t cap = a.slots.size(s);
    const int64_t bts = a.block_ts[s];
    uint32_t st = 0;
    GlobalSink sink{out, cap, &st};
    Encoder<FMT, PRED> e;
    e.begin(bts, sink);
    if (PRED == PRED_DFCM) e.dfcm = Dfcm{a.dfcm_tables + s * ((uint64_t)a.dfcm_mask + 1), 0u, a.dfcm_mask};
    for (uint64_t i = 0; i < np; i++) e.add(bts, a.ts[p0 + i], a.val[p0 + i], sink);
    if (a.out_nbits) a.out_nbits[s] = e.bits() + (a.no_close ? 0u : 37u);
    uint32_t n = a.no_close ? e.snapshot(sink) : e.finish(sink);
    a.out_nwords[s] = n;
    if (a.status) a.status[s] = st;
    if (st) atomicOr(a.flags, st);
}

// true when series s is a wave's or a workgroup's: long, and the batch holds few enough long ones
template <int FMT>
__device__ __forceinline__ bool left_to_block_kernel(const DecodeArgs& a, uint64_t s)
{
    return block_owner<FMT>(a, s) != OWNER_LANE;
}

template <int FMT, int PRED = PRED_LAST>
__global__ __launch_bounds__(256) void decode_direct_kernel(DecodeArgs a)
{
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.nseries) return;
    if (left_to_block_kernel<FMT>(a, s)) return;
    const uint64_t* in = a.words + a.slots.begin(s);
    uint64_t nw64 = a.in_nwords ? a.in_nwords[s] : a.slots.size(s);
    const uint32_t nw = nw64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nw64;
    const uint64_t p0 = a.pts.begin(s), cap = a.pts.size(s);
    auto src = [&](uint32_t i) -> uint64_t { return i < nw ? in[i] : 0ull; };
    Decoder<FMT, PRED> d;
    d.begin(nw, src);
    if (PRED == PRED_DFCM) d.dfcm = Dfcm{a.dfcm_tables + s * ((uint64_t)a.dfcm_mask + 1), 0u, a.dfcm_mask};
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
// Tiled form.  A wavefront owns 64 consecutive series ("rows").  Points travel
// between HBM and the lanes through LDS tiles of 64 rows x 16 points so that
// every global access is a full, aligned 128-byte line of one row (8 lanes x
// 16 B), while each lane still consumes / produces its own series serially.
//
// Rows start at arbitrary point offsets, so each row gets a phase
//     phi = (index of its first point in the array + alignment of the array) mod 16
// and lives in "virtual slots" v = phi + point index.  Chunk c of the tile holds
// the slots [16c, 16c+16) of every row: a chunk boundary is then a 128-byte line
// boundary in memory for every row, whatever its length or start.  Slots outside
// [phi, phi + npoints) are empty; only a row's first and last chunk have any.
//
//   HBM --(lane l: row 8i + l/8, piece l%8, 16 B)--> registers --> LDS tile
//   LDS tile --(lane j: its own row j, piece q)--> the lane's encoder
//   (decode runs the same path backwards)
//
// The tile is XOR-swizzled: piece q of row r sits at r*128 + ((q ^ ((r>>1)&7))<<4),
// which keeps the loader's ds_write_b128 (8 lanes = one row), the per-lane
// ds_read_b128 (16 lanes with distinct row%16 cover 256 B) and the decoder's
// flush reads free of bank conflicts.
// ------------------------------------------------------------------------
constexpr uint32_t CH = 16;                   // slots per row per chunk
constexpr uint32_t ROW_BYTES = CH * 8;        // 128
constexpr uint32_t PIECES = ROW_BYTES / 16;   // 16-byte pieces per row: 8
constexpr uint32_t ROWS_PER_PASS = 64 / PIECES;  // 8
constexpr uint32_t PASSES = 64 / ROWS_PER_PASS;  // 8
constexpr uint32_t TILE_BYTES = 64 * ROW_BYTES;  // 8 KiB

__device__ __forceinline__ uint32_t tile_addr(uint32_t row, uint32_t piece)
{
    return row * ROW_BYTES + ((piece ^ ((row >> 1) & (PIECES - 1))) << 4);
}

typedef uint32_t U4 __attribute__((ext_vector_type(4)));
// the same 16 bytes where only 8-byte alignment is known (value arrays on the other 16-byte phase
// than the timestamps the chunk grid is laid out on)
typedef U4 U4a8 __attribute__((aligned(8)));

// 16-byte global access at 8-byte alignment
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
__device__ __forceinline__ uint64_t lo64(const U4& v) { return ((uint64_t)v.y << 32) | v.x; }
__device__ __forceinline__ uint64_t hi64(const U4& v) { return ((uint64_t)v.w << 32) | v.z; }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t y = (uint32_t)__shfl_xor((int)x, d, 64);
        x = x > y ? x : y;
    }
    return x;
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t x, uint32_t src_lane)
{
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)x, (int)src_lane, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(x >> 32), (int)src_lane, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Which series a lane of a tiled kernel codes.  With a uniform number of points per series the
// position of a series inside its 128-byte lines (its phase) repeats every T = 2^shift series;
// a wave then takes every T-th series, so that all its rows share one phase and start and end
// their chunks together (T consecutive workgroups cover 64*T consecutive series).  shift = 0:
// consecutive series.
__device__ __forceinline__ uint64_t series_of_lane(uint32_t block, uint32_t lane, uint32_t shift)
{
    // The 2^shift waves of a group share the 128-byte lines where their series meet.  Workgroups go
    // to the 8 XCDs in turn, each with an L2 of its own: the waves of a group are 8 workgroups
    // apart, so they run on the same XCD and a shared line is fetched (or completed) once.
    const uint32_t r = block & ((8u << shift) - 1u);
    const uint64_t group = ((uint64_t)(block >> (3 + shift)) << 3) + (r & 7u);
    return (group << (6 + shift)) + (r >> 3) + ((uint64_t)lane << shift);
}
// shift for a layout: uniform point counts only; phase period = 16 / gcd(points mod 16, 16) series
static uint32_t group_shift_of(const Layout& pts)
{
    if (DIAG_STAMPED) return 0;  // (a stamped build: the stamps of a wave overwrite the outputs of the series after its first)
    if (pts.off) return 0;
    const uint32_t m = (uint32_t)(pts.uniform & 15);
    if (m == 0) return 0;
    return 4u - (uint32_t)__builtin_ctz(m);  // m = 8 -> 1, 4 or 12 -> 2, even -> 3, odd -> 4
}
static uint32_t grouped_blocks(uint64_t nseries, uint32_t shift)
{
    const uint64_t per = 64ull << shift;
    const uint64_t groups = (nseries + per - 1) / per;
    return (uint32_t)((shift ? ((groups + 7) & ~7ull) : groups) << shift);  // whole sets of 8 groups (see series_of_lane)
}

// what a loader lane needs to know about a row
struct alignas(16) RowInfo {
    int64_t base;  // array index of virtual slot 0 (first point index - phi)
    uint32_t lo;   // first occupied slot (phi)
    uint32_t hi;   // one past the last occupied slot
};

constexpr uint32_t MAX_ROW_POINTS = 0x7FFFFF00u;

// Output ring of the encoder: per row RING_WORDS words, copied out in blocks of RING_BLOCK.
// The rows are looked at after every two points, and a copy is made when some row holds
// RING_TRIGGER or more words that are not yet written out: two points add at most 228 bits,
// which moves the word being filled at most 4 slots on, so below the trigger a row stays
// within its 16 slots until the next look.  Copying as late as that lets many rows have a
// whole block by the time one row forces the copy.
constexpr uint32_t RING_WORDS = 16;
constexpr uint32_t RING_BLOCK = 8;
constexpr uint32_t RING_TRIGGER = RING_WORDS - 4;

// ring rows are padded by one word so that the lanes' ds_write_b64 spread over the banks
constexpr uint32_t RING_ROW_BYTES = RING_WORDS * 8 + 8;

// word sink of the tiled form: the lane's row of the LDS ring
struct RingSink {
    static constexpr bool ALWAYS = true;  // a slot is simply rewritten until its word is complete
    uint8_t* row;
    __device__ __forceinline__ void operator()(uint32_t i, uint64_t w) const
    {
        *reinterpret_cast<uint64_t*>(row + ((i & (RING_WORDS - 1)) << 3)) = w;
    }
};

// The v2 format does int arithmetic on timestamps (G/GorillaCompressor.java:74,102-103): a lane
// only needs the low half of each timestamp, plus one bit per point telling whether the whole
// timestamp is zero (the reference's "no point yet" test).  So the v2 timestamp tile holds 4
// bytes per slot; v1 keeps all 8.
template <int FMT> struct TsTile {
    static constexpr uint32_t SLOT_BYTES = (FMT == FMT_V2) ? 4 : 8;
    static constexpr uint32_t ROW = CH * SLOT_BYTES;      // 64 or 128 bytes
    static constexpr uint32_t BYTES = 64 * ROW;
    static constexpr uint32_t ROW_PIECES = ROW / 16;      // 16-byte pieces per row: 4 or 8
    // byte address of 16-byte piece `p16` of `row`, swizzled like tile_addr()
    __device__ static __forceinline__ uint32_t addr(uint32_t row, uint32_t p16)
    {
        constexpr uint32_t SH = ROW_PIECES == 4 ? 2 : 1;
        return row * ROW + ((p16 ^ ((row >> SH) & (ROW_PIECES - 1))) << 4);
    }
};

// SPLIT: the value array sits on another phase of a 128-byte line than the timestamp array.  The pieces
// are laid out on the timestamps' alignment; on the other 16-byte phase a value piece straddles a 16-byte
// boundary, and where only one of its two slots belongs to the row the other one may lie in a page that
// is not there: such pieces are fetched slot by slot.  The value loads of this form are cached ones.
template <int FMT, bool SPLIT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void encode_tiled_kernel(EncodeArgs a)
{
    using TT = TsTile<FMT>;
    constexpr uint32_t OFF_VAL = TT::BYTES;
    constexpr uint32_t OFF_RING = OFF_VAL + TILE_BYTES;
    constexpr uint32_t OFF_ROWS = OFF_RING + 64 * RING_ROW_BYTES;
    constexpr uint32_t OFF_ZERO = OFF_ROWS + 64 * sizeof(RowInfo);
    __shared__ __attribute__((aligned(16))) uint8_t lds[OFF_ZERO + 64 * 4 + GTSC_LDS_PAD];
    uint8_t* const tile_ts = lds;
    uint8_t* const tile_val = lds + OFF_VAL;
    uint8_t* const ring = lds + OFF_RING;  // 64 rows x RING_WORDS compressed words
    RowInfo* const rows = reinterpret_cast<RowInfo*>(lds + OFF_ROWS);
    uint32_t* const rowzero = reinterpret_cast<uint32_t*>(lds + OFF_ZERO);  // v2: bit k = slot k's timestamp is 0

    const uint32_t lane = threadIdx.x;
    const uint64_t s = series_of_lane(blockIdx.x, lane, a.group_shift);
    const bool valid = s < a.nseries && a.pts.size(s) < a.long_threshold;  // long blocks are the wave-per-block kernel's
    const uint64_t p0 = valid ? a.pts.begin(s) : 0;
    const uint64_t np64 = valid ? a.pts.size(s) : 0;
    const uint32_t np = np64 > MAX_ROW_POINTS ? MAX_ROW_POINTS : (uint32_t)np64;
    const uint32_t phi = (uint32_t)((p0 + ((uintptr_t)a.ts >> 3)) & (CH - 1));
    rows[lane] = RowInfo{(int64_t)p0 - (int64_t)phi, phi, phi + np};
    rowzero[lane] = 0;
    const uint32_t nchunks = wave_max_u32(np ? (phi + np + CH - 1) / CH : 0u);

    const uint32_t piece = lane % PIECES;
    const uint32_t lrow = lane / PIECES;  // loader / flusher role: rows lrow + 8i
    U4 rt[PASSES], rv[PASSES] = {};
#pragma unroll
    for (uint32_t i = 0; i < PASSES; i++) rt[i] = U4{~0u, ~0u, ~0u, ~0u};  // a piece never fetched holds no zero timestamp

    const uint64_t cap64 = valid ? a.slots.size(s) : 0;
    const uint32_t cap = cap64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap64;
    const uint64_t sbeg = valid ? a.slots.begin(s) : 0;  // the flusher lanes get it by shuffle
    const int64_t bts = valid ? a.block_ts[s] : 0;
    // loader role: array index of slot 0 of chunk 0 of each of the lane's eight rows
    int64_t rbase[PASSES];
#pragma unroll
    for (uint32_t i = 0; i < PASSES; i++) rbase[i] = (int64_t)shfl_u64(p0 - phi, lrow + i * ROWS_PER_PASS);
    // Completed words go to the lane's row of the ring (word w in slot w % RING_WORDS); the
    // wavefront copies them out in whole 64-byte blocks.
    uint32_t flushed = 0;  // words of this row already written to HBM (a multiple of RING_BLOCK)
    uint32_t trigger = RING_TRIGGER;  // flushed + RING_TRIGGER: a copy is due once a row's widx reaches it
    RingSink sink{ring + lane * RING_ROW_BYTES};
    Encoder<FMT> e;
    e.widx = 0;
    e.started = false;
    if (valid) e.begin(bts, sink);

// Copies ring words [flushed, END) of every row to its slot; END is the row's count rounded
// down to whole blocks (FINAL == false) or the count itself.  Lane l serves row 8i + l/8,
// ring slots 2*(l%8) and the next.  Words beyond a row's slot capacity are dropped.  What a
// flusher lane needs to know about a row comes from the row's own lane by shuffle, and all LDS
// reads are issued before the first store, so the passes do not wait for one another.
#define GTSC_FLUSH(FINAL)                                                                      \
    {                                                                                          \
        wave_sync(); /* the lanes' ring writes */                                              \
        uint64_t fx0[PASSES], fx1[PASSES], fdst[PASSES];                                       \
        uint32_t fmode[PASSES]; /* 0 nothing, 1 first word only, 2 both words */               \
        _Pragma("unroll") for (uint32_t i = 0; i < PASSES; i++) {                              \
            const uint32_t r = lrow + i * ROWS_PER_PASS;                                       \
            const uint32_t rf = (uint32_t)__shfl((int)flushed, (int)r, 64);                    \
            const uint32_t rw = (uint32_t)__shfl((int)e.widx, (int)r, 64);                     \
            const uint32_t rcap = (uint32_t)__shfl((int)cap, (int)r, 64);                      \
            const uint32_t end = (FINAL) ? rw : rf + ((rw - rf) & ~(RING_BLOCK - 1));          \
            const uint32_t w0 = rf + ((2 * piece - rf) & (RING_WORDS - 1));                    \
            const uint64_t* src = reinterpret_cast<const uint64_t*>(ring + r * RING_ROW_BYTES + piece * 16); \
            fx0[i] = src[0]; fx1[i] = src[1];                                                  \
            fdst[i] = shfl_u64(sbeg, r) + w0;                                                  \
            fmode[i] = (w0 < end && w0 < rcap) ? ((w0 + 1 < end && w0 + 1 < rcap) ? 2u : 1u) : 0u; \
        }                                                                                      \
        _Pragma("unroll") for (uint32_t i = 0; i < PASSES; i++) {                              \
            uint64_t* dst = a.out_words + fdst[i];                                             \
            if (DIAG_NO_ENC_STORES && !(FINAL)) {                                              \
                asm volatile("" ::"v"(fx0[i]), "v"(fx1[i]), "v"(dst), "v"(fmode[i]));          \
            } else if (fmode[i] == 2) {                                                        \
                const U4 v = {(uint32_t)fx0[i], (uint32_t)(fx0[i] >> 32), (uint32_t)fx1[i], (uint32_t)(fx1[i] >> 32)}; \
                store16(dst, v);                                                               \
            } else if (fmode[i] == 1) {                                                        \
                dst[0] = fx0[i];                                                               \
            }                                                                                  \
        }                                                                                      \
        if (!(FINAL)) { flushed += (e.widx - flushed) & ~(RING_BLOCK - 1); trigger = flushed + RING_TRIGGER; } \
        wave_sync(); /* the slots may be written again */                                      \
    }

// Loads chunk C of the tile into rt/rv (a macro: the arrays must stay in registers).  A piece
// is fetched whole when either of its two slots is occupied: pieces are 16-byte aligned in
// memory, so the access stays inside a page that holds a valid point, and the lanes never
// look at slots outside [phi, phi + npoints).  There is deliberately one code path with a
// single-armed `if`: diverging paths that both define the quads make the register allocator
// copy them where the paths join, which waits for the loads on the spot.
#define GTSC_FETCH(C)                                                                          \
    {                                                                                          \
        const uint32_t v0 = (C) * CH + 2 * piece; /* first of the two slots of this piece */   \
        if (all_lanes((C) * CH >= phi && (C) * CH + CH <= phi + np)) {                         \
            /* every row has all 16 slots of this chunk: nothing to ask the rows */            \
            _Pragma("unroll") for (uint32_t i = 0; i < PASSES; i++) {                          \
                const int64_t idx = rbase[i] + (int64_t)v0;                                    \
                rt[i] = __builtin_nontemporal_load(reinterpret_cast<const U4*>(a.ts + idx));   \
                rv[i] = SPLIT ? load16(a.val + idx)                                            \
                              : __builtin_nontemporal_load(reinterpret_cast<const U4*>(a.val + idx)); \
            }                                                                                  \
        } else {                                                                               \
            U4 ri_[PASSES]; /* RowInfo as one 16-byte LDS read each, all issued before any is used */ \
            _Pragma("unroll") for (uint32_t i = 0; i < PASSES; i++)                            \
                ri_[i] = *reinterpret_cast<const U4*>(&rows[lrow + i * ROWS_PER_PASS]);        \
            _Pragma("unroll") for (uint32_t i = 0; i < PASSES; i++) {                          \
                const int64_t idx = (int64_t)(((uint64_t)ri_[i].y << 32) | ri_[i].x) + (int64_t)v0; \
                if ((v0 + 1 >= ri_[i].z) & (v0 < ri_[i].w)) {                                  \
                    /* 16-byte aligned pieces of a stream that is read once */                 \
                    rt[i] = __builtin_nontemporal_load(reinterpret_cast<const U4*>(a.ts + idx)); \
                    if (!SPLIT) {                                                              \
                        rv[i] = __builtin_nontemporal_load(reinterpret_cast<const U4*>(a.val + idx)); \
                    } else {                                                                   \
                        const bool in0 = v0 >= ri_[i].z, in1 = v0 + 1 < ri_[i].w;              \
                        if (in0 & in1) {                                                       \
                            rv[i] = load16(a.val + idx);                                       \
                        } else if (in0) {                                                      \
                            const uint64_t x = a.val[idx];                                     \
                            rv[i].x = (uint32_t)x; rv[i].y = (uint32_t)(x >> 32);              \
                        } else if (in1) { /* (neither: an empty row whose place falls into this piece) */ \
                            const uint64_t x = a.val[idx + 1];                                 \
                            rv[i].z = (uint32_t)x; rv[i].w = (uint32_t)(x >> 32);              \
                        }                                                                      \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
    }

    GTSC_ENC_STAMPS_BEGIN
    uint32_t run_wait = 0;  // wave-uniform: chunks to code in lock step before run skipping is tried again
    uint32_t run_backoff = 4;
    wave_sync();  // rows[] visible
    if (nchunks) GTSC_FETCH(0)
    GTSC_TICK(tk_fetch)
    for (uint32_t c = 0; c < nchunks; c++) {
        wave_sync();  // every lane has read the previous chunk
        uint32_t zmin = 0xFFFFFFFFu;  // v2: 0 when one of the lane's pieces holds a zero timestamp
#pragma unroll
        for (uint32_t i = 0; i < PASSES; i++) {
            const uint32_t r = lrow + i * ROWS_PER_PASS;
            // Opaque to the optimiser: otherwise the element picks below are folded into the
            // branches that loaded the quads and the wave waits for each load where it is issued.
            asm volatile("" : "+v"(rt[i]), "+v"(rv[i]));
            if constexpr (FMT == FMT_V2) {
                // low halves of the piece's two timestamps: 8 bytes at slot 2*piece of the row
                // (two dword stores: ds_write2_b32 takes them from wherever the load left them)
                uint32_t* const dst = reinterpret_cast<uint32_t*>(tile_ts + TT::addr(r, piece >> 1) + ((piece & 1) << 3));
                dst[0] = rt[i].x;
                dst[1] = rt[i].z;
                zmin = min(zmin, min(rt[i].x | rt[i].y, rt[i].z | rt[i].w));
            } else {
                *reinterpret_cast<U4*>(tile_ts + TT::addr(r, piece)) = rt[i];
            }
            *reinterpret_cast<U4*>(tile_val + tile_addr(r, piece)) = rv[i];
        }
        if constexpr (FMT == FMT_V2) {
            // a zero timestamp practically never occurs: the pieces are only looked at one by one
            // (to mark the slot for its row) when the cheap test above has seen one
            if (any_lane(zmin == 0)) {
#pragma unroll
                for (uint32_t i = 0; i < PASSES; i++) {
                    const uint32_t zb = ((rt[i].x | rt[i].y) == 0 ? 1u : 0u) | ((rt[i].z | rt[i].w) == 0 ? 2u : 0u);
                    if (zb) atomicOr(&rowzero[lrow + i * ROWS_PER_PASS], zb << (2 * piece));
                }
            }
        }
        wave_sync();
        GTSC_TICK(tk_stage)
        if (c + 1 < nchunks) GTSC_FETCH(c + 1)  // in flight while this chunk is encoded
        GTSC_TICK(tk_fetch)
        const bool interior = c * CH >= phi && c * CH + CH <= phi + np;
        const bool idle = c != 0 && c * CH >= phi + np;  // the row ended in an earlier chunk
        if constexpr (FMT == FMT_V2) {
            const uint32_t zbits = rowzero[lane];
            if (__builtin_expect(zbits != 0, 0)) rowzero[lane] = 0;
            const uint32_t bts_lo = (uint32_t)bts;
            // every lane has all 16 slots occupied, is past its first point and sees no zero
            // timestamp: no guards, no first-point test, four points per step
            // (rows that ended in an earlier chunk sit the chunk out: a batch of very unequal blocks
            // keeps this loop for its long blocks)
            const bool plain_chunk = all_lanes(idle || (interior && e.started && zbits == 0));
            if (plain_chunk && a.run_skip && run_wait == 0) {
                // Run skipping.  Most points of a sparse stream are the two bits '00' (the interval and the value
                // of the point before).  First the lane goes over its 16 slots with two comparisons each and
                // marks the others; then it takes one step per marked point -- the zero bits of the run before it
                // and the point's own codes, in one piece when they fit a word -- so a chunk costs as many
                // full codings as its busiest row has changes (5 or 6 of 16 when one value in ten changes).
                uint32_t marks = 0;  // bit 31 - j: the point in slot j is not '00'
                uint32_t pt = e.pts_lo, pd = (uint32_t)e.stored_delta;
                uint64_t pv = e.prev;
#pragma unroll
                for (uint32_t q = 0; q < PIECES; q += 2) {
                    const U4 tl = *reinterpret_cast<const U4*>(tile_ts + TT::addr(lane, q >> 1));
                    const U4 va = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q));
                    const U4 vb = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q + 1));
                    const uint32_t tj[4] = {tl.x, tl.y, tl.z, tl.w};
                    const uint64_t vj[4] = {lo64(va), hi64(va), lo64(vb), hi64(vb)};
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++) {
                        const uint32_t nd = tj[i] - pt;
                        const bool other = (nd != pd) | (vj[i] != pv);
                        marks |= other ? 0x80000000u >> (2 * q + i) : 0u;
                        pt = tj[i]; pd = nd; pv = vj[i];
                    }
                }
                if (idle) marks = 0;
                uint32_t pos = 0, steps = 0;
                const uint32_t tswz = ((lane >> 2) & 3u) << 2, vswz = ((lane >> 1) & (PIECES - 1)) << 1;
                while (any_lane(marks != 0)) {
                    if (marks != 0) {
                        const uint32_t j = (uint32_t)__clz((int)marks);
                        const uint32_t run = j - pos;
                        // nothing changed over the run: same value, same interval
                        e.pts_lo += run * (uint32_t)e.stored_delta;
                        const uint32_t tsj = *reinterpret_cast<const uint32_t*>(tile_ts + lane * TT::ROW + ((j ^ tswz) << 2));
                        const uint64_t vj = *reinterpret_cast<const uint64_t*>(tile_val + lane * ROW_BYTES + ((j ^ vswz) << 3));
                        const typename Encoder<FMT>::Code cd = e.code_next(tsj, 0u, vj);
                        const uint32_t zl = 2u * run, len = cd.hlen + cd.plen;
                        if (all_lanes(zl + len <= 64u)) {
                            e.put((cd.head << cd.plen) | cd.pay, zl + len, sink);  // plen <= 62 here
                        } else {
                            e.put(0ull, zl, sink);
                            e.emit(cd, sink);
                        }
                        pos = j + 1;
                        marks &= ~(0x80000000u >> j);
                    }
                    steps++;
                    if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                }
                if (!idle) {
                    e.put(0ull, 2u * (CH - pos), sink);  // the run that ends the chunk
                    e.pts_lo = pt;
                    e.stored_delta = (int64_t)(int32_t)pd;
                    e.prev = pv;
                }
                if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                // from eight steps on the lock-step loop is the quicker one: streams whose points differ from the
                // one before that often sit out more and more chunks
                if (steps > 7) {
                    run_wait = run_backoff;
                    run_backoff = run_backoff < 32 ? run_backoff * 2 : 64;
                } else {
                    run_backoff = 4;
                }
            } else if (plain_chunk) {
                if (run_wait) run_wait--;
#pragma unroll 1
                for (uint32_t q = 0; q < PIECES; q += 2) {
                    const U4 tl = *reinterpret_cast<const U4*>(tile_ts + TT::addr(lane, q >> 1));
                    const U4 va = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q));
                    const U4 vb = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q + 1));
                    if (DIAG_NO_ENC_ARITH) {
                        e.acc ^= lo64(va) ^ hi64(va) ^ lo64(vb) ^ hi64(vb) ^ tl.x ^ tl.y ^ tl.z ^ tl.w;
                    } else {
                        if (!idle) e.add_next2(tl.x, 0u, lo64(va), tl.y, 0u, hi64(va), sink);
                        if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                        if (!idle) e.add_next2(tl.z, 0u, lo64(vb), tl.w, 0u, hi64(vb), sink);
                    }
                    if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                }
            } else {
#pragma unroll 1
                for (uint32_t q = 0; q < PIECES; q += 2) {
                    const U4 tl = *reinterpret_cast<const U4*>(tile_ts + TT::addr(lane, q >> 1));
                    const U4 va = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q));
                    const U4 vb = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q + 1));
                    const uint32_t k = c * CH + 2 * q - phi;  // point index of slot 2q (wraps below phi)
                    const uint32_t zq = zbits >> (2 * q);
                    if (k < np) e.add_lo(bts_lo, tl.x, zq & 1u, lo64(va), sink);
                    if (k + 1 < np) e.add_lo(bts_lo, tl.y, (zq >> 1) & 1u, hi64(va), sink);
                    if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                    if (k + 2 < np) e.add_lo(bts_lo, tl.z, (zq >> 2) & 1u, lo64(vb), sink);
                    if (k + 3 < np) e.add_lo(bts_lo, tl.w, (zq >> 3) & 1u, hi64(vb), sink);
                    if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                }
            }
        } else if (all_lanes(interior || idle)) {
            // v1, every lane has all 16 slots occupied.  Run skipping as above, with 64-bit timestamps; a zero
            // timestamp anywhere in the chunk (the "no point yet" test) leaves the chunk to the loop below.
            bool skipped = false;
            if (a.run_skip && run_wait == 0 && all_lanes(idle || e.started)) {
                uint32_t marks = 0, zeros = 0;
                uint64_t pt = ((uint64_t)e.pts_hi << 32) | e.pts_lo, pd = (uint64_t)e.stored_delta, pv = e.prev;
#pragma unroll
                for (uint32_t q = 0; q < PIECES; q++) {
                    const U4 t = *reinterpret_cast<const U4*>(tile_ts + TT::addr(lane, q));
                    const U4 v = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q));
                    const uint64_t tj[2] = {lo64(t), hi64(t)}, vj[2] = {lo64(v), hi64(v)};
#pragma unroll
                    for (uint32_t i = 0; i < 2; i++) {
                        const uint64_t nd = tj[i] - pt;
                        const bool other = (nd != pd) | (vj[i] != pv);
                        marks |= other ? 0x80000000u >> (2 * q + i) : 0u;
                        zeros |= tj[i] == 0 ? 1u : 0u;
                        pt = tj[i]; pd = nd; pv = vj[i];
                    }
                }
                if (!any_lane(zeros != 0 && !idle)) {
                    skipped = true;
                    if (idle) marks = 0;
                    uint32_t pos = 0, steps = 0;
                    const uint32_t vswz = ((lane >> 1) & (PIECES - 1)) << 1;
                    while (any_lane(marks != 0)) {
                        if (marks != 0) {
                            const uint32_t j = (uint32_t)__clz((int)marks);
                            const uint32_t run = j - pos;
                            const uint64_t before = (((uint64_t)e.pts_hi << 32) | e.pts_lo) + (uint64_t)run * (uint64_t)e.stored_delta;
                            e.pts_lo = (uint32_t)before; e.pts_hi = (uint32_t)(before >> 32);
                            const uint64_t tsj = *reinterpret_cast<const uint64_t*>(tile_ts + lane * TT::ROW + ((j ^ vswz) << 3));
                            const uint64_t vj = *reinterpret_cast<const uint64_t*>(tile_val + lane * ROW_BYTES + ((j ^ vswz) << 3));
                            const typename Encoder<FMT>::Code cd = e.code_next((uint32_t)tsj, (uint32_t)(tsj >> 32), vj);
                            const uint32_t zl = 2u * run, len = cd.hlen + cd.plen;
                            if (all_lanes(zl + len <= 64u)) {
                                e.put((cd.head << cd.plen) | cd.pay, zl + len, sink);
                            } else {
                                e.put(0ull, zl, sink);
                                e.emit(cd, sink);
                            }
                            pos = j + 1;
                            marks &= ~(0x80000000u >> j);
                        }
                        steps++;
                        if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                    }
                    if (!idle) {
                        e.put(0ull, 2u * (CH - pos), sink);
                        e.pts_lo = (uint32_t)pt; e.pts_hi = (uint32_t)(pt >> 32);
                        e.stored_delta = (int64_t)pd;
                        e.prev = pv;
                    }
                    if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
                    if (steps > 7) {
                        run_wait = run_backoff;
                        run_backoff = run_backoff < 32 ? run_backoff * 2 : 64;
                    } else {
                        run_backoff = 4;
                    }
                }
            } else if (run_wait) {
                run_wait--;
            }
            // otherwise: no guards, and pairs of ordinary points (past the first one, no zero timestamp
            // before the second) go through together
            if (!skipped) {
#pragma unroll 1
            for (uint32_t q = 0; q < PIECES; q++) {
                const U4 t = *reinterpret_cast<const U4*>(tile_ts + TT::addr(lane, q));
                const U4 v = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q));
                if (!idle) {
                    if (all_lanes(e.started && lo64(t) != 0)) {
                        e.add_next2(t.x, t.y, lo64(v), t.z, t.w, hi64(v), sink);
                        e.started = hi64(t) != 0;
                    } else {
                        e.add(bts, (int64_t)lo64(t), lo64(v), sink);
                        e.add(bts, (int64_t)hi64(t), hi64(v), sink);
                    }
                }
                if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
            }
            }  // !skipped
        } else {
#pragma unroll 1
            for (uint32_t q = 0; q < PIECES; q++) {
                const U4 t = *reinterpret_cast<const U4*>(tile_ts + TT::addr(lane, q));
                const U4 v = *reinterpret_cast<const U4*>(tile_val + tile_addr(lane, q));
                const uint32_t k = c * CH + 2 * q - phi;
                if (k < np) e.add(bts, (int64_t)lo64(t), lo64(v), sink);
                if (k + 1 < np) e.add(bts, (int64_t)hi64(t), hi64(v), sink);
                if (any_lane(e.widx >= trigger)) GTSC_FLUSH(false)
            }
        }
        GTSC_TICK(tk_comp)
    }
#undef GTSC_FETCH
    if (valid) {
        if (a.out_nbits) a.out_nbits[s] = e.bits() + (a.no_close ? 0u : 37u);
        if (a.no_close) e.snapshot(sink); else e.finish(sink);
    }
    GTSC_FLUSH(true)
    GTSC_ENC_STAMPS_END(a, s, lane)
#undef GTSC_FLUSH
    if (valid) {
        // a block beyond the tiled form's point limit was coded up to the limit: reported, not silent
        const uint32_t st = (e.widx > cap || np64 > MAX_ROW_POINTS) ? ST_OVERFLOW : 0u;
        a.out_nwords[s] = e.widx;
        if (a.status) a.status[s] = st;
        if (st) atomicOr(a.flags, st);
    }
}

// Decoder geometry.  Output tiles hold 16 slots per row like the encoder's: stores of whole
// 128-byte lines run about 1.5x faster than 64-byte half lines.
//
// The compressed words reach a lane through its own LDS ring of 8 words, which the lane fills
// itself, 16 bytes at a time: nothing about the input is shared between lanes, so there is no
// wave-level hand-shake.  (Neighbouring 16-byte pieces of an image come from the same cache
// line; the compressed stream is the small side of the traffic.)  A top-up is requested when
// the ring holds <= 6 unread words and is written to the ring at the next check (two points
// later), so its latency hides behind decoding.  A lane that consumes words faster than that
// (a point can take 114 bits) is caught by the same comparison that detects the end of the
// image: the wholesale path only commits a point if the three words of the next window are
// already in the ring (new position < guard = loaded - 2); otherwise the exact path runs and
// waits for the missing words.
constexpr uint32_t DCH = CH;                      // slots per row per chunk
//
// The kernel exists with a ring of 8 and of 16 words.  With 8 it keeps 8 waves per CU, which is
// what sparse streams (a few bits per point) want.  Dense streams (tens of bits per point) want
// the words further ahead of the parser than 8 slots allow: under the decoder's own store traffic
// a top-up takes longer than the two points between request and landing, and every landing is
// a wait.  With 16 slots (7 waves per CU) a landing is put off until a lane is down to five unread
// words or the load is five checks old, which by then has arrived.  Which form runs is the
// caller's choice (Tunables::decode_ring) or, left open, follows from the density of the batch as
// the plan kernel measured it (DecodeArgs::ring_by_plan: both forms are launched, one returns).
constexpr uint32_t IN_NEVER = 0xFFFFFFFFu;        // "limit" of a lane with nothing left to load

// The decoder's timestamp tile holds only the low half of each timestamp (4 bytes per slot):
// within a chunk of 16 points the high half practically never changes, so a row carries one
// high half per chunk (DecRow::ts_hi) and the flush puts the two together.  A point whose high
// half differs from its row's (a 2^32 ms boundary, or wild input) is written to memory by its
// own lane right away and masked out of the flush (DecRow::info, bits 16..31).
constexpr uint32_t DEC_TS_TILE_BYTES = 64 * DCH * 4;
__device__ __forceinline__ uint32_t dec_ts_addr(uint32_t row, uint32_t slot)
{
    return row * (DCH * 4) + ((slot ^ (((row >> 2) & 7) << 1)) << 2);  // even swizzle: a pair of slots stays 8 contiguous bytes
}
struct alignas(16) DecRow {
    int64_t base;    // array index of slot 0 of chunk 0 (first point index - phi)
    uint32_t info;   // this chunk: first occupied slot | (one past the last) << 8 | (slots written directly) << 16
    uint32_t ts_hi;  // high half of the chunk's timestamps
};
constexpr uint32_t DEC_ROW_FULL = 16u << 8;       // info of a row with all 16 slots to flush

template <int FMT, uint32_t RING>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void decode_tiled_kernel(DecodeArgs a)
{
    constexpr uint32_t IN_RING_WORDS = RING;          // ring capacity per row (8 or 16)
    constexpr uint32_t TOPUP = RING == 16 ? 4 : 2;    // words per top-up (one or two 16-byte loads of the lane)
    constexpr uint32_t IN_RING_LOW = RING - TOPUP;    // top up while unread words <= this (room for one top-up)
    constexpr uint32_t IN_RING_ROW_BYTES = (IN_RING_WORDS + 3) * 8;  // + mirror of slots 0 and 1 (a 3-word window never wraps) + bank padding
    constexpr uint32_t OFF_TS = TILE_BYTES;
    constexpr uint32_t OFF_RING = OFF_TS + DEC_TS_TILE_BYTES;
    constexpr uint32_t OFF_ROWS = OFF_RING + 64 * IN_RING_ROW_BYTES;
    constexpr uint32_t OFF_RUNS = OFF_ROWS + 64 * sizeof(DecRow);
    constexpr bool RUN_SKIP = RING == 8;              // the form sparse streams take: see the event loop below
    __shared__ __attribute__((aligned(16))) uint8_t lds[OFF_RUNS + (RUN_SKIP ? 64 * sizeof(U4) : 0) + GTSC_DEC_LDS_PAD];
    uint8_t* const tile_val = lds;
    uint8_t* const tile_ts = lds + OFF_TS;
    uint8_t* const ring = lds + OFF_RING;
    DecRow* const rows = reinterpret_cast<DecRow*>(lds + OFF_ROWS);
    U4* const runs = reinterpret_cast<U4*>(lds + OFF_RUNS);  // RUN_SKIP only: {timestamp before the chunk (low half), delta, slots written, -}

    GTSC_DEC_STAMPS_ENTRY
    const uint32_t lane = threadIdx.x;
    if (a.ring_by_plan) {
        // both forms of the kernel were launched: the one that suits the batch's density works (above 40 compressed bits
        // per point of capacity the 16-word rings)
        const bool dense = a.plan->words_all * 64 > a.plan->cap_all * 40;
        if (dense != (RING == 16)) return;
    }
    const uint64_t s = series_of_lane(blockIdx.x, lane, a.group_shift);
    const bool valid = s < a.nseries && !left_to_block_kernel<FMT>(a, s);
    GTSC_DEC_STAMP_PROBE(valid)
    // The slot bounds are known before the image length is: the first 64 bytes of the slot are
    // requested right away and the length is looked at when they have arrived (one memory
    // latency instead of two).  Ring words past the image's end are never parsed.
    const uint64_t ssz64 = valid ? a.slots.size(s) : 0;
    const uint32_t ssz = ssz64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ssz64;
    const uint64_t* const image = a.words + (valid ? a.slots.begin(s) : 0);
    const uint64_t p0 = valid ? a.pts.begin(s) : 0;
    const uint64_t cap64 = valid ? a.pts.size(s) : 0;
    const uint32_t cap = cap64 > MAX_ROW_POINTS ? MAX_ROW_POINTS : (uint32_t)cap64;
    const uint32_t phi = (uint32_t)((p0 + ((uintptr_t)a.ts >> 3)) & (DCH - 1));
    rows[lane] = DecRow{(int64_t)p0 - (int64_t)phi, 0u, 0u};

    const uint32_t piece = lane % PIECES;
    const uint32_t lrow = lane / PIECES;  // flusher role: rows lrow + 8i
    uint64_t* const my_ring = reinterpret_cast<uint64_t*>(ring + lane * IN_RING_ROW_BYTES);

    WindowDecoder<FMT> d;
    d.widx = 0; d.off = 0; d.st = 0;
    bool active = valid;

    // 16 bytes of the slot at word `W` (even): a slot's last odd word comes as the upper half of
    // the 16 bytes before it, so there is one load instruction whatever the case (two arms would
    // make the compiler wait for each load where they join).  BACK says which half to use.
#define GTSC_RING_LOAD(Q, BACK, W)                                                             \
    {                                                                                          \
        BACK = ((W) + 1 < ssz) ? 0u : 1u;                                                      \
        if ((W) < ssz && ssz >= 2) Q = load16(image + ((W) - BACK));                           \
    }
#define GTSC_RING_STORE(Q, BACK, W)                                                            \
    {                                                                                          \
        asm volatile("" : "+v"(Q));                                                            \
        /* BACK (the last odd word of a slot): the upper half moves down and the upper word is  \
           zero -- by mask, not by select (four v_cndmask in a row cost three times their price) */ \
        const uint32_t bm_ = 0u - (BACK);                                                      \
        const U4 q_ = {Q.x ^ ((Q.x ^ Q.z) & bm_), Q.y ^ ((Q.y ^ Q.w) & bm_), Q.z & ~bm_, Q.w & ~bm_}; \
        const uint64_t x0 = lo64(q_), x1 = hi64(q_);                                           \
        const uint32_t sl = (W) & (IN_RING_WORDS - 1);                                         \
        my_ring[sl] = x0; my_ring[sl + 1] = x1;                                                \
        if (sl == 0) { my_ring[IN_RING_WORDS] = x0; my_ring[IN_RING_WORDS + 1] = x1; }         \
    }
    GTSC_DIAG_POISON_RING(my_ring, IN_RING_WORDS + 3)
    // ---- initial fill: words 0..7 of the lane's own slot (separate variables, not an array: the
    // compiler would copy a whole array of quads, and wait for its loads, after every conditional load)
    uint64_t nw64 = ssz64;
    {
        U4 q0, q1, q2, q3;  // not loaded: left undefined, never parsed
        uint32_t b0, b1, b2, b3;
        GTSC_RING_LOAD(q0, b0, 0u)
        GTSC_RING_LOAD(q1, b1, 2u)
        GTSC_RING_LOAD(q2, b2, 4u)
        GTSC_RING_LOAD(q3, b3, 6u)
        if (valid && a.in_nwords) nw64 = a.in_nwords[s];  // in flight together with the ring's words
        GTSC_DEC_STAMP_P1
        GTSC_RING_STORE(q0, b0, 0u)
        GTSC_RING_STORE(q1, b1, 2u)
        GTSC_RING_STORE(q2, b2, 4u)
        GTSC_RING_STORE(q3, b3, 6u)
        GTSC_DEC_STAMP_P2
    }
    // an image cannot be longer than its slot
    const uint32_t nw = nw64 > ssz ? ssz : (uint32_t)nw64;
    uint32_t filled = 8;  // the ring holds image words [filled - RING, filled) that exist (an even number); 8 are loaded up front
    // wholesale commits need new widx < guard; top-ups are requested once widx >= limit
    uint32_t guard = filled >= nw ? nw : filled - 2;
    uint32_t limit = filled >= nw ? IN_NEVER : (filled > IN_RING_LOW ? filled - IN_RING_LOW : 0u);
    if (valid) d.begin(nw, my_ring[0]);

    // ---- top-up state: the lane's load in flight between two checks
    uint32_t pend = 0;  // 1: a top-up of this lane is in flight (an integer: see lanes())
    uint32_t dense = 0;  // wave-uniform: > 0 after a lane outran its top-ups: the rings are then looked after at every point for a while
    U4 tq, tq2;  // defined whenever pend is set (tq2: the second half of a four-word top-up)
    uint32_t tback = 0, tback2 = 0;
    // The source of the next top-up is worked out when the ring's end moves, not when the load is
    // made: computed on the spot, the address would go through registers that the compiler takes
    // from the previous top-up's, and writing those makes it wait for everything in flight --
    // the chunk's stores included (one counter, in order).
    const uint64_t* tsrc = image;
    const uint64_t* tsrc2 = image;
    // A four-word top-up is two loads made together.  Where the slot ends before the second pair of
    // words, the second load fetches the first pair again (no branch of its own: the compiler would
    // wait for both loads where its arms join); what it puts into the ring lies past the image.
#define GTSC_TOPUP_NEXT()                                                                      \
    {                                                                                          \
        tback = (filled + 1 < ssz) ? 0u : 1u;                                                  \
        tsrc = image + (filled - tback);                                                       \
        if constexpr (TOPUP == 4) {                                                            \
            tback2 = (filled + 3 < ssz) ? 0u : 1u;                                             \
            tsrc2 = (filled + 2 < ssz) ? image + (filled + 2 - tback2) : tsrc;                 \
        }                                                                                      \
    }
    GTSC_TOPUP_NEXT()
#define GTSC_TOPUP_LAND()                                                                      \
    {                                                                                          \
        bool plain_ = false;                                                                   \
        if constexpr (TOPUP == 4) plain_ = !any_lane(filled + 8u > ssz);                       \
        if (plain_) {                                                                          \
            /* no landing lane is within two top-ups of its slot's end: whole pairs now and next time */ \
            asm volatile("" : "+v"(tq), "+v"(tq2));                                            \
            const uint32_t sl = filled & (IN_RING_WORDS - 1);                                  \
            const uint64_t x0 = lo64(tq), x1 = hi64(tq);                                       \
            my_ring[sl] = x0; my_ring[sl + 1] = x1; my_ring[sl + 2] = lo64(tq2); my_ring[sl + 3] = hi64(tq2); \
            if (sl == 0) { my_ring[IN_RING_WORDS] = x0; my_ring[IN_RING_WORDS + 1] = x1; }     \
            filled += 4;                                                                       \
            tsrc += 4; tsrc2 += 4;                                                             \
        } else {                                                                               \
            GTSC_RING_STORE(tq, tback, filled)                                                 \
            if constexpr (TOPUP == 4) GTSC_RING_STORE(tq2, tback2, filled + 2)                 \
            filled += TOPUP;                                                                   \
            GTSC_TOPUP_NEXT()                                                                  \
        }                                                                                      \
        guard = filled >= nw ? nw : filled - 2;                                                \
        limit = (filled >= nw || !active) ? IN_NEVER : (filled > IN_RING_LOW ? filled - IN_RING_LOW : 0u);                   \
        pend = 0;                                                                              \
    }
#define GTSC_TOPUP_ISSUE() /* lanes with widx >= limit: filled < nw <= ssz */                  \
    {                                                                                          \
        tq = load16(tsrc);                                                                     \
        if constexpr (TOPUP == 4) tq2 = load16(tsrc2);                                         \
        pend = 1;                                                                              \
        limit = IN_NEVER; /* nothing more until this one has landed */                         \
    }

    uint32_t n = 0;     // pairs decoded (keeps counting on a finished lane; nfin holds its result)
    uint32_t nfin = 0;
    uint32_t st = 0;
    uint32_t skip = phi;               // slots of the first chunk that lie before the row's first point
    uint64_t slow_m = lanes(valid);    // wave-uniform: lanes whose next point needs the serial path (first point, zero timestamp, skip)
    uint64_t active_m = lanes(valid);  // wave-uniform: the lanes still decoding
    uint32_t ts_ref = 0;               // high half that this chunk's timestamps of the lane share
    uint32_t n0 = 0;                   // n when the chunk began
    uint32_t dmask = 0;                // slots of this chunk whose timestamps the lane stored itself
    GTSC_DEC_STAMPS_BEGIN
    // The three-word window of a lane's next point is read from its ring as soon as the position is
    // known -- at the end of the point before -- so that the LDS round trip runs under the rest of
    // that point (votes, tile writes, top-up checks) instead of at the head of the next one.  A point
    // that commits wholesale has its successor's window inside the ring already (new position <
    // guard); top-ups that land in between write other slots.
    uint64_t w0 = my_ring[d.widx & (IN_RING_WORDS - 1)], w1 = my_ring[(d.widx & (IN_RING_WORDS - 1)) + 1],
             w2 = my_ring[(d.widx & (IN_RING_WORDS - 1)) + 2];
#define GTSC_WINDOW_READ()                                                                     \
    {                                                                                          \
        if (DIAG_NO_RING_READS) {                                                              \
            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));                                   \
        } else {                                                                               \
            const uint64_t* w = reinterpret_cast<const uint64_t*>(my_ring) + (d.widx & (IN_RING_WORDS - 1)); \
            w0 = w[0]; w1 = w[1]; w2 = w[2];                                                   \
        }                                                                                      \
    }
// One point of every lane.  Finished lanes ride along on the wholesale path (their rows are
// bounded at flush time), so a ragged wave keeps using it.
#define GTSC_POINT(KK)                                                                         \
    {                                                                                          \
        bool done = false;                                                                     \
        if (slow_m == 0) {                                                                     \
            const uint64_t hi = WindowDecoder<FMT>::window(w0, w1, d.off);                     \
            const uint32_t off0 = d.off;                                                       \
            auto lo = [w1, w2, off0] { return WindowDecoder<FMT>::window(w1, w2, off0); };     \
            /* regular cadence in every live lane: the timestamp code is the bit '0' */        \
            typename WindowDecoder<FMT>::Parsed p;                                             \
            uint64_t eos_m = 0; /* the vote is taken where the flag is made: as a flag it would \
                                   cross the join through a vector register */                 \
            if ((lanes((int64_t)hi < 0) & active_m) == 0) {                                    \
                p = d.template parse<true>(hi, lo);                                            \
            } else {                                                                           \
                p = d.template parse<false>(hi, lo);                                           \
                eos_m = lanes(p.eos);                                                          \
            }                                                                                  \
            WindowDecoder<FMT> t = d;                                                          \
            t.commit(p);                                                                       \
            /* each vote is one comparison; the masks are combined as scalars */               \
            const uint64_t rare_m = (eos_m | (lanes(p.changed) & lanes(p.nbits == 0)) |        \
                                     lanes(t.widx >= guard) | lanes(t.stored_ts == 0) |        \
                                     lanes((uint32_t)(t.stored_ts >> 32) != ts_ref)) & active_m; \
            if (DIAG_NO_RING_READS || rare_m == 0) {                                           \
                d = t;                                                                         \
                GTSC_WINDOW_READ()                                                             \
                GTSC_TILE_WRITE(uint32_t, tile_ts + adt + (KK) * 4, d.stored_ts)               \
                GTSC_TILE_WRITE(uint64_t, tile_val + ad + (KK) * 8, d.val)                     \
                n++;                                                                           \
                done = true;                                                                   \
            }                                                                                  \
        }                                                                                      \
        if (!done) {                                                                           \
            if (active) {                                                                      \
                if (skip) {                                                                    \
                    skip--;                                                                    \
                } else {                                                                       \
                    int64_t t; uint64_t v;                                                     \
                    if (d.next(w0, w1, w2, t, v)) {                                            \
                        const uint32_t h = (uint32_t)((uint64_t)t >> 32);                      \
                        if (h != ts_ref) {                                                     \
                            if (n == n0) {                                                     \
                                ts_ref = h; /* first point of the chunk sets the row's high half */ \
                            } else {                                                           \
                                dmask |= 1u << (2 * k2 + (KK));                                \
                                if (n < cap) a.ts[p0 + n] = t;                                 \
                            }                                                                  \
                        }                                                                      \
                        *reinterpret_cast<uint32_t*>(tile_ts + adt + (KK) * 4) = (uint32_t)t;  \
                        *reinterpret_cast<uint64_t*>(tile_val + ad + (KK) * 8) = v;            \
                        n++;                                                                   \
                    } else {                                                                   \
                        active = false;                                                        \
                        nfin = n;                                                              \
                        limit = IN_NEVER;                                                      \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
            /* a lane that has run ahead of its top-ups waits for the words of its next window */ \
            for (;;) {                                                                         \
                const bool shortw = active && filled < nw && d.widx + 2 >= filled;             \
                if (!any_lane(shortw)) break;                                                     \
                dense = 3;                                                                     \
                if (shortw) {                                                                  \
                    if (!pend) GTSC_TOPUP_ISSUE()                                              \
                    GTSC_TOPUP_LAND()                                                          \
                }                                                                              \
            }                                                                                  \
            slow_m = lanes(skip != 0 || d.stored_ts == 0) & lanes(active);                     \
            active_m = lanes(active);                                                          \
            GTSC_WINDOW_READ()                                                                 \
        }                                                                                      \
    }
// One step of the run-skipping loop (see below); STAMPS: the form that takes interval changes along.
#define GTSC_RUN_STEP(STAMPS)                                                                  \
    {                                                                                                                                                          \
                        /* the run: each '00' is a point; a lane that is through the chunk has no room and stays put */                                        \
                        const uint64_t hi = WindowDecoder<FMT>::window(w0, w1, d.off);                                                                         \
                        const uint32_t zeros = (uint32_t)__clz((int)(uint32_t)(hi >> 32));  /* 32 when there is no one */                                      \
                        const uint32_t room = DCH - k;                                                                                                         \
                        const uint32_t m = (zeros >> 1) < room ? (zeros >> 1) : room;                                                                          \
                        const uint32_t o = d.off + 2u * m;  /* <= 63 + 32 */                                                                                   \
                        /* the point after the run, parsed from the words at hand: (w0,w1,w2), or (w1,w2) when */                                              \
                        /* the run crossed into w1 -- a payload that needs the word after w2 then drops the attempt */                                         \
                        WindowDecoder<FMT> t = d;                                                                                                              \
                        t.widx = d.widx + (o >> 6);                                                                                                            \
                        t.off = o & 63u;                                                                                                                       \
                        const uint64_t u0 = o >= 64u ? w1 : w0, u1 = o >= 64u ? w2 : w1;                                                                       \
                        const uint64_t hi2 = WindowDecoder<FMT>::window(u0, u1, t.off);                                                                        \
                        const uint64_t u2 = w2;                                                                                                                \
                        const uint32_t off2 = t.off;                                                                                                           \
                        auto lo2 = [u1, u2, off2] { return WindowDecoder<FMT>::window(u1, u2, off2); };                                                        \
                        const uint64_t more_m = lanes(m < room);  /* lanes with a point to parse */                                                            \
                        typename WindowDecoder<FMT>::Parsed p;                                                                                                 \
                        const bool more = m < room;                                                                                                            \
                        const uint32_t cnt = m + (more ? 1u : 0u);                                                                                             \
                        const uint64_t delta = (uint64_t)d.stored_delta;                                                                                       \
                        const uint64_t neg_m = lanes((int64_t)hi2 < 0) & more_m;  /* timestamp codes other than '0' */                                         \
                        if (!(STAMPS)) {                                                                                                                       \
                            p = t.template parse<true>(hi2, lo2);                                                                                              \
                            t.stored_ts = d.stored_ts + (uint64_t)cnt * delta;                                                                                 \
                            irregular_m |= neg_m;                                                                                                              \
                            bad_m |= neg_m;                                                                                                                    \
                        } else {                                                                                                                               \
                            p = t.template parse<false>(hi2, lo2);                                                                                             \
                            bad_m |= lanes(p.eos) & more_m;  /* end of stream is the lock-step loop's */                                                       \
                            /* the run keeps the delta; the point after it changes it (it has to stay within 0..2^32-1, */                                     \
                            /* so that the chunk's timestamps never decrease: see the test of the high half below) */                                          \
                            const uint64_t delta2 = delta + (more ? (uint64_t)p.dd : 0ull);                                                                    \
                            bad_m |= lanes((uint32_t)(delta2 >> 32) != 0u) & active_m;                                                                         \
                            t.stored_delta = (int64_t)delta2;                                                                                                  \
                            t.stored_ts = d.stored_ts + (uint64_t)m * delta + (more ? delta2 : 0ull);                                                          \
                            if (more && delta2 != delta) {                                                                                                     \
                                /* this point's timestamp, and the next one's as the new delta predicts it, go to the */                                       \
                                /* timestamp tile (a next point that changes the interval again writes its own over */                                         \
                                /* it); the flush takes the delta from the two */                                                                              \
                                const uint32_t e = k + m;                                                                                                      \
                                *reinterpret_cast<uint32_t*>(tile_ts + dec_ts_addr(lane, e)) = (uint32_t)t.stored_ts;                                          \
                                if (e + 1 < DCH) *reinterpret_cast<uint32_t*>(tile_ts + dec_ts_addr(lane, e + 1)) = (uint32_t)t.stored_ts + (uint32_t)delta2;  \
                                run_stamps |= (3u << e) & 0xFFFFu;                                                                                             \
                            }                                                                                                                                  \
                        }                                                                                                                                      \
                        bad_m |= more_m & ((lanes(p.changed) & lanes(p.nbits == 0)) | (lanes(o >= 64u) & lanes(p.o3 - off2 > 64u)));                           \
                        if (more) {                                                                                                                            \
                            t.nbits = p.nbits;                                                                                                                 \
                            t.trail = p.trail;                                                                                                                 \
                            t.val = d.val ^ (p.bits << ((uint32_t)p.trail & 63u));                                                                             \
                            t.advance(p.o3);                                                                                                                   \
                            *reinterpret_cast<uint64_t*>(my_vals + (((k + m) ^ swz) << 3)) = t.val;                                                            \
                            run_slots |= 1u << (k + m);                                                                                                        \
                        }                                                                                                                                      \
                        bad_m |= lanes(t.widx >= guard) & active_m;                                                                                            \
                        d = t;                                                                                                                                 \
                        k += cnt;                                                                                                                              \
                        n += cnt;                                                                                                                              \
                        steps++;                                                                                                                               \
                        GTSC_WINDOW_READ()                                                                                                                     \
    }
    // Run skipping (8-word form).  Most points of a sparse stream are the two bits '00': regular cadence,
    // value unchanged.  Walking the 16 slots of a chunk in lock step spends a full parse on each of them in
    // every lane.  Here the lanes go through a chunk at their own pace instead: in one step a lane jumps
    // over the run of '00' points at its position (a count of leading zeros) and parses the one point after
    // it, so a chunk takes as many steps as its busiest row has changed values (5 or 6 of 16 when one value
    // in ten changes).  The lane writes only the values that changed to its tile row, marks their slots, and
    // leaves (timestamp before the chunk, delta) for the flush, which fills in the rest: the latest marked
    // slot at or before each slot gives its value, the slot number times the delta its timestamp.
    // A chunk qualifies when every row of the wave that is still under way is past its first point, has 16 slots
    // to fill and a delta of 0..2^32-1 (lanes without a series, or past the end of theirs, sit it out).  A point that changes the
    // interval (a timestamp code other than '0') is parsed like a changed value; its timestamp and the next
    // one's go to the timestamp tile for the flush.  Whatever is not ordinary in any lane (end of stream, a
    // delta outside 0..2^32-1, a window never defined, words not yet in the ring, a timestamp whose high half moves) drops
    // the attempt: the lanes go back to where the chunk began and take the lock-step loop, which knows all
    // of that; nothing the attempt did needs undoing (it lands no top-ups and the tile is rewritten).
    uint32_t run_wait = 0;     // wave-uniform: chunks to sit out before the next attempt
    bool stamps_mode = false;  // wave-uniform: the run-skipping step that takes interval changes along is in use
    uint32_t calm = 0;         // chunks in a row in that form without an interval change
    if constexpr (RUN_SKIP) runs[lane] = U4{0u, 0u, 1u, 0u};  // a row without a series: never stored, but looked at
    GTSC_RUNSTAT_DECL
    uint32_t run_backoff = 1;  // doubles with every failed attempt
    const uint32_t swz = ((lane >> 1) & (PIECES - 1)) << 1;  // slot j of this lane's tile row sits at (j ^ swz) * 8
    uint8_t* const my_vals = tile_val + lane * ROW_BYTES;
    for (uint32_t c = 0; active_m != 0; c++) {
        ts_ref = (uint32_t)(d.stored_ts >> 32);
        n0 = n;
        dmask = 0;
        if (dense) dense--;
        bool skipped = false;      // wave-uniform: the chunk was decoded by run skipping
        GTSC_RUNSTAT_CHUNK((c != 0 && run_wait == 0) ? ((dense != 0 ? 1u : 0u) | (slow_m != 0 ? 2u : 0u) | (active_m != ~0ull ? 4u : 0u) | (lanes(n + DCH > cap) ? 8u : 0u) | (lanes((uint32_t)((uint64_t)d.stored_delta >> 32) != 0u) ? 16u : 0u)) : 0u)
        uint32_t run_slots = 1u;   // slots of the chunk whose values the lane wrote (slot 0: the value before the chunk)
        uint32_t run_stamps = 0;   // slots of the chunk whose timestamps the lane wrote (see below)
        const uint32_t run_ts = (uint32_t)d.stored_ts, run_delta = (uint32_t)d.stored_delta;
        if constexpr (RUN_SKIP) {
            if (run_wait) {
                run_wait--;
            } else if (a.run_skip && c != 0 && dense == 0 && slow_m == 0 &&
                       ((lanes(n + DCH > cap) | lanes((uint32_t)((uint64_t)d.stored_delta >> 32) != 0u)) & active_m) == 0) {
                const WindowDecoder<FMT> d_at = d;
                const uint64_t w0_at = w0, w1_at = w1, w2_at = w2;
                uint32_t k = 0, steps = 0;
                uint64_t bad_m = 0;
                uint64_t irregular_m = 0;  // plain steps only: lanes that met a timestamp code other than '0'
                // One step of every lane, in one of two forms.  The plain form drops the attempt at an interval change
                // (the following chunks are tried in the other form, which takes such points along): streams of
                // perfectly regular cadence keep the shorter step (a lone wave per SIMD: 12 % per chunk).  The form
                // is the wave's, not the step's: a vote per step on the timestamp codes would cost what it saves.
                *reinterpret_cast<uint64_t*>(my_vals + (swz << 3)) = d.val;
                k = active ? 0u : DCH;  // slots of the chunk the lane has behind it (a lane without a series, or past its end: all)
                if (stamps_mode) { while (any_lane(k < DCH)) { GTSC_RUN_STEP(true) if (bad_m) break; } }
                else             { while (any_lane(k < DCH)) { GTSC_RUN_STEP(false) if (bad_m) break; } }
                if (irregular_m) {
                    // an interval changed: this chunk is the lock-step loop's, the next ones take such points along
                    stamps_mode = true;
                    calm = 0;
                }
                if (stamps_mode && bad_m == 0) {
                    // eight chunks in a row without an interval change: back to the plain step
                    if (any_lane(run_stamps != 0)) calm = 0;
                    else if (++calm >= 8) stamps_mode = false;
                }
                bad_m |= lanes((uint32_t)(d.stored_ts >> 32) != ts_ref) & active_m;
                GTSC_RUNSTAT_ATTEMPT(steps, bad_m == 0)
                if (bad_m == 0) {
                    skipped = true;
                    // a chunk of ten steps or more would have been quicker in lock step (nine is where the two meet):
                    // streams whose values change that often sit out more and more chunks, like failed attempts
                    if (steps > 9) {
                        run_wait = run_backoff < 4 ? 4 : run_backoff;
                        run_backoff = run_wait < 32 ? run_wait * 2 : 64;
                    } else {
                        run_backoff = 1;
                    }
                    if (pend) GTSC_TOPUP_LAND()
                    const bool low = d.widx >= limit;
                    if (low) GTSC_TOPUP_ISSUE()
                } else {
                    d = d_at;
                    n = n0;
                    w0 = w0_at; w1 = w1_at; w2 = w2_at;
                    if (irregular_m == 0) {  // (after an interval change the next chunk is tried right away, in the other form)
                        run_wait = run_backoff;
                        run_backoff = run_backoff < 64 ? run_backoff * 2 : 64;
                    }
                }
            }
        }
        // the eight pairs of a chunk unrolled: tile addresses and the which-check-is-it tests become constants
        // (3 % on gauges, 5 % on long gauge series; the code still fits the instruction cache)
        if (!skipped) {
#pragma unroll
        for (uint32_t k2 = 0; k2 < DCH / 2; k2++) {
            const uint32_t ad = tile_addr(lane, k2);
            const uint32_t adt = dec_ts_addr(lane, 2 * k2);
            GTSC_POINT(0)
            if (dense) {  // a stream this dense needs a top-up per point to keep up (2 words per point at most)
                if (pend) GTSC_TOPUP_LAND()
                const bool low0 = d.widx >= limit;
                if (low0) GTSC_TOPUP_ISSUE()
            }
            GTSC_POINT(1)
            if constexpr (RING == 8) {
                // sparse streams: a look at the rings every second pair is enough (16 bytes last 25 points of a
                // gauge; a lane that does run short is tended to where it happens, see GTSC_POINT)
                if (k2 & 1u) {
                    if (pend) GTSC_TOPUP_LAND()
                    const bool low = d.widx >= limit;
                    if (low) GTSC_TOPUP_ISSUE()
                }
            } else {
                // requests are made at every second check only, and land at the next such check: what the wave
                // waits for then is two checks old in every lane.  In between only a lane about to run out lands.
                if (k2 & 1u) {
                    if (pend) { GTSC_LAND_STAMP() GTSC_TOPUP_LAND() }
                    const bool low = d.widx >= limit;
                    if (low) GTSC_TOPUP_ISSUE()
                } else {
                    const bool due = pend != 0 && filled - d.widx <= GTSC_LAND_LOW;
                    if (any_lane(due)) {
                        GTSC_LAND_STAMP()
                        if (pend) GTSC_TOPUP_LAND()
                        const bool low = d.widx >= limit;
                        if (low) GTSC_TOPUP_ISSUE()
                    }
                }
            }
        }
        }  // !skipped
        if (DIAG_NO_RING_READS) {  // (nothing tells such a build where a stream ends)
            if (active && c * DCH >= cap) { active = false; nfin = n; }
            active_m = lanes(active);
        }
        GTSC_TICK(tk_comp)
        // ---- flush the chunk: what this row has to write out, in slots of this chunk
        const uint32_t nn = active ? n : nfin;
        const uint32_t stored = nn < cap ? nn : cap;  // points of this row that may be written out
        const uint32_t end = phi + stored;            // in slots counted from chunk 0
        const uint32_t lo_rel = c == 0 ? phi : 0u;
        const uint32_t hi_rel = end <= c * DCH ? 0u : (end - c * DCH < DCH ? end - c * DCH : DCH);
        const uint32_t info = lo_rel | (hi_rel << 8) | (dmask << 16);
        *reinterpret_cast<uint2*>(&rows[lane].info) = make_uint2(info, ts_ref);
        if constexpr (RUN_SKIP) {
            if (skipped) runs[lane] = U4{run_ts, run_delta, run_slots, run_stamps};
        }
        wave_sync();
        const uint32_t j0 = 2 * piece;
        const uint32_t v0 = c * DCH + j0;
        U4 ri[PASSES];
#pragma unroll
        for (uint32_t i = 0; i < PASSES; i++) ri[i] = *reinterpret_cast<const U4*>(&rows[lrow + i * ROWS_PER_PASS]);
        if (RUN_SKIP && skipped) {
            // a chunk decoded by run skipping (every row that has a series has all 16 slots): a slot's value is the one written at
            // the latest marked slot at or before it, its timestamp the one before the chunk plus (slot + 1) deltas
            // Rows whose interval changed inside the chunk hold timestamps in the tile at the slots marked in the
            // record's fourth word, always in pairs (the point that changed it and the one after): a slot at a
            // mark has that timestamp, a slot after the last mark the one there plus so many times the
            // difference of the pair; before the first mark the row's own start and delta apply.
            uint64_t q0[PASSES], q1[PASSES];
            uint32_t t0[PASSES], t1[PASSES];
            const uint32_t upto0 = (2u << j0) - 1u, upto1 = (4u << j0) - 1u;  // slots 0..j0, 0..j0+1
            const bool stamped = any_lane(run_stamps != 0);
#pragma unroll
            for (uint32_t i = 0; i < PASSES; i++) {
                const uint32_t r = lrow + i * ROWS_PER_PASS;
                const U4 rr = runs[r];
                const uint32_t rswz = ((r >> 1) & (PIECES - 1)) << 1;
                const uint32_t s0 = 31u - (uint32_t)__clz((int)(rr.z & upto0)), s1 = 31u - (uint32_t)__clz((int)(rr.z & upto1));
                q0[i] = *reinterpret_cast<const uint64_t*>(tile_val + r * ROW_BYTES + ((s0 ^ rswz) << 3));
                q1[i] = *reinterpret_cast<const uint64_t*>(tile_val + r * ROW_BYTES + ((s1 ^ rswz) << 3));
                t0[i] = rr.x + (j0 + 1u) * rr.y;
                t1[i] = t0[i] + rr.y;
            }
            if (stamped) {  // a pass of its own, so that waves without such rows pay one vote for it
#pragma unroll
                for (uint32_t i = 0; i < PASSES; i++) {
                    const uint32_t r = lrow + i * ROWS_PER_PASS;
                    const uint32_t marks = runs[r].w;
                    const uint32_t m0 = marks & upto0, m1 = marks & upto1;
                    const uint32_t e0 = m0 ? 31u - (uint32_t)__clz((int)m0) : 1u, e1 = m1 ? 31u - (uint32_t)__clz((int)m1) : 1u;
                    const uint32_t a0 = *reinterpret_cast<const uint32_t*>(tile_ts + dec_ts_addr(r, e0));
                    const uint32_t b0 = *reinterpret_cast<const uint32_t*>(tile_ts + dec_ts_addr(r, e0 ? e0 - 1u : 0u));
                    const uint32_t a1 = *reinterpret_cast<const uint32_t*>(tile_ts + dec_ts_addr(r, e1));
                    const uint32_t b1 = *reinterpret_cast<const uint32_t*>(tile_ts + dec_ts_addr(r, e1 ? e1 - 1u : 0u));
                    if (m0) t0[i] = a0 + (j0 - e0) * (a0 - b0);
                    if (m1) t1[i] = a1 + (j0 + 1u - e1) * (a1 - b1);
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < PASSES; i++) {
                const int64_t idx = (int64_t)(((uint64_t)ri[i].y << 32) | ri[i].x) + (int64_t)v0;
                const U4 t = {t0[i], ri[i].w, t1[i], ri[i].w};
                const U4 v = {(uint32_t)q0[i], (uint32_t)(q0[i] >> 32), (uint32_t)q1[i], (uint32_t)(q1[i] >> 32)};
                if (ri[i].z != DEC_ROW_FULL) continue;  // a row without a series, or one that has ended
                GTSC_CHUNK_STORE(U4, a.ts + idx, t)
                GTSC_CHUNK_STORE(U4a8, a.val + idx, v)
            }
        } else if (all_lanes(info == DEC_ROW_FULL)) {
            // every row has all 16 slots of this chunk: read everything, then store everything
            uint2 qt[PASSES];
            U4 qv[PASSES];
#pragma unroll
            for (uint32_t i = 0; i < PASSES; i++) {
                const uint32_t r = lrow + i * ROWS_PER_PASS;
                qt[i] = *reinterpret_cast<const uint2*>(tile_ts + dec_ts_addr(r, j0));
                qv[i] = *reinterpret_cast<const U4*>(tile_val + tile_addr(r, piece));
            }
#pragma unroll
            for (uint32_t i = 0; i < PASSES; i++) {
                const int64_t idx = (int64_t)(((uint64_t)ri[i].y << 32) | ri[i].x) + (int64_t)v0;
                const U4 t = {qt[i].x, ri[i].w, qt[i].y, ri[i].w};
                GTSC_CHUNK_STORE(U4, a.ts + idx, t)
                GTSC_CHUNK_STORE(U4a8, a.val + idx, qv[i])
            }
        } else {
            // ragged chunk (a block's first or last one, or directly stored timestamps): the same
            // reads up front, then each piece stores what it has
            uint2 qt[PASSES];
            U4 qv[PASSES];
#pragma unroll
            for (uint32_t i = 0; i < PASSES; i++) {
                const uint32_t r = lrow + i * ROWS_PER_PASS;
                qt[i] = *reinterpret_cast<const uint2*>(tile_ts + dec_ts_addr(r, j0));
                qv[i] = *reinterpret_cast<const U4*>(tile_val + tile_addr(r, piece));
            }
#pragma unroll
            for (uint32_t i = 0; i < PASSES; i++) {
                const int64_t idx = (int64_t)(((uint64_t)ri[i].y << 32) | ri[i].x) + (int64_t)v0;
                const uint32_t rlo = ri[i].z & 255u, rhi = (ri[i].z >> 8) & 255u, direct = ri[i].z >> 16;
                const bool ok0 = j0 >= rlo && j0 < rhi;
                const bool ok1 = j0 + 1 >= rlo && j0 + 1 < rhi;
                const bool t0 = ok0 && !((direct >> j0) & 1u), t1 = ok1 && !((direct >> (j0 + 1)) & 1u);
                if (ok0 & ok1 & t0 & t1) {
                    store16(a.ts + idx, U4{qt[i].x, ri[i].w, qt[i].y, ri[i].w});
                    store16(a.val + idx, qv[i]);
                } else if (ok0 | ok1) {
                    if (t0) a.ts[idx] = (int64_t)(((uint64_t)ri[i].w << 32) | qt[i].x);
                    if (t1) a.ts[idx + 1] = (int64_t)(((uint64_t)ri[i].w << 32) | qt[i].y);
                    if (ok0) a.val[idx] = lo64(qv[i]);
                    if (ok1) a.val[idx + 1] = hi64(qv[i]);
                }
            }
        }
        wave_sync();
        GTSC_TICK(tk_flush)
    }
    GTSC_RUNSTAT_PRINT(lane)
#undef GTSC_TOPUP_LAND
#undef GTSC_TOPUP_ISSUE
#undef GTSC_TOPUP_NEXT
#undef GTSC_RING_LOAD
#undef GTSC_RING_STORE
#undef GTSC_POINT
#undef GTSC_RUN_STEP
    n = active ? n : nfin;
    if (valid) {
        if (n > cap) st |= ST_OVERFLOW;
        st |= d.st;
        a.out_npoints[s] = n;
        if (a.status) a.status[s] = st;
        if (st) atomicOr(a.flags, st);
    }
    GTSC_DEC_STAMPS_END(a, s, lane)
}

// ------------------------------------------------------------------------
// Byte images.  ByteBufferBitOutput holds the same bit stream as LongArrayOutput, byte by
// byte: byte i is bits 8i..8i+7 of the stream, i.e. the words written big-endian
// (G/ByteBufferBitOutput.java:42-96); after close() the buffer holds floor(N/8)+1 bytes.
// One wavefront per series.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void words_to_bytes_kernel(uint64_t nseries, const uint64_t* words, Layout wslots,
                                                             const uint64_t* nbits, uint8_t* bytes, Layout bslots,
                                                             uint64_t* out_nbytes, uint32_t* flags)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t s = wave; s < nseries; s += nwaves) {
        const uint64_t* src = words + wslots.begin(s);
        uint8_t* dst = bytes + bslots.begin(s);
        const uint64_t n = (nbits[s] >> 3) + 1, cap = bslots.size(s), wcap = wslots.size(s);
        const uint64_t m = n < cap ? n : cap;
        for (uint64_t i = lane; i < m; i += 64) {
            const uint64_t w = (i >> 3) < wcap ? src[i >> 3] : 0;
            dst[i] = (uint8_t)(w >> (56 - 8 * (i & 7)));
        }
        if (lane == 0) {
            out_nbytes[s] = n;
            if (n > cap) atomicOr(flags, ST_OVERFLOW);
        }
    }
}

// The word image a LongArrayInput needs for such bytes: the bytes packed big-endian into
// words, zero-padded, plus the one word of look-ahead the reader loads eagerly
// (G/LongArrayInput.java:48-52).
__global__ __launch_bounds__(256) void bytes_to_words_kernel(uint64_t nseries, const uint8_t* bytes, Layout bslots,
                                                             const uint64_t* nbytes, uint64_t* words, Layout wslots,
                                                             uint64_t* out_nwords, uint32_t* flags)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t s = wave; s < nseries; s += nwaves) {
        const uint8_t* src = bytes + bslots.begin(s);
        uint64_t* dst = words + wslots.begin(s);
        const uint64_t bcap = bslots.size(s);
        const uint64_t nb = nbytes ? (nbytes[s] < bcap ? nbytes[s] : bcap) : bcap;
        const uint64_t n = (nb + 7) / 8 + 1, cap = wslots.size(s);
        const uint64_t m = n < cap ? n : cap;
        for (uint64_t i = lane; i < m; i += 64) {
            uint64_t w = 0;
            for (uint32_t k = 0; k < 8; k++) {
                const uint64_t b = 8 * i + k;
                w = (w << 8) | (b < nb ? src[b] : 0u);
            }
            dst[i] = w;
        }
        if (lane == 0) {
            out_nwords[s] = n;
            if (n > cap) atomicOr(flags, ST_OVERFLOW);
        }
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
        // A series that overflowed its slot reports the length it would have had: only what the slot
        // holds is there to copy (the offsets follow the reported lengths, the rest stays unwritten).
        const uint64_t cap = slots.size(s);
        const uint64_t n = nwords[s] < cap ? nwords[s] : cap;
        for (uint64_t i = lane; i < n; i += 64) d[i] = src[i];
    }
}

// ------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------
hipError_t launch_encode(const EncodeArgs& a0, int format, hipStream_t stream)
{
    if (a0.nseries == 0) return hipSuccess;
    EncodeArgs a = a0;
    uint32_t none = 0;
    uint32_t& did = a.dispatch ? *a.dispatch : none;
    a.long_threshold = ~0ull;
    if (a.dfcm_tables) {  // the table look-ups are per-lane global accesses anyway: direct form only
        if (format != FMT_V2) return hipErrorInvalidValue;
        const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
        did |= DISPATCH_LANE_DIRECT;
        hipLaunchKernelGGL((encode_direct_kernel<FMT_V2, PRED_DFCM>), dim3(blocks), dim3(256), 0, stream, a);
    } else if (a.tun.direct) {
        const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
        did |= DISPATCH_LANE_DIRECT;
        if (format == FMT_V2) hipLaunchKernelGGL(encode_direct_kernel<FMT_V2>, dim3(blocks), dim3(256), 0, stream, a);
        else                  hipLaunchKernelGGL(encode_direct_kernel<FMT_V1>, dim3(blocks), dim3(256), 0, stream, a);
    } else if (!a.pts.off && a.pts.uniform >= a.tun.long_points && !(a.tun.sparse_hint && a.nseries >= a.tun.sparse_encode_min)) {
        // a uniform batch of long blocks: all of it is the wave-per-block kernel's -- unless the caller says the
        // streams are sparse and the blocks are many: a lane per series then skips their runs of unchanged points and
        // is the faster one once the lanes fill the machine (65,536 x 7,200 gauges: 1.53 against 1.90 ms)
        a.long_threshold = 0;
        return launch_encode_blocks(a, format, stream);
    } else {
        // Per-series lengths are only known on the device: both kernels run over the batch, each
        // taking the series on its side of the threshold.
        const bool mixed = a.pts.off != nullptr;
        if (mixed) {
            a.long_threshold = a.tun.long_points;
            if (hipError_t e = launch_encode_blocks(a, format, stream)) return e;
            if (a.long_threshold == 0) return hipSuccess;  // every block is "long": nothing is left for the lanes
        }
        EncodeArgs g = a;
        g.run_skip = a.tun.run_skip;
        g.group_shift = group_shift_of(a.pts);
        const uint32_t blocks = grouped_blocks(a.nseries, g.group_shift);
        // Arrays on different phases of a 128-byte line take the SPLIT form: needed when the addresses differ
        // by an odd number of 8-byte words, and the better one for any other difference, because its value loads
        // are cached ones (a row's 128 bytes of values then span two lines, each of which two chunks read).
        const bool split = (((uintptr_t)a.ts ^ (uintptr_t)a.val) & 120u) != 0;
        did |= DISPATCH_LANE_TILED | (split ? DISPATCH_SPLIT : 0u);
        if (format == FMT_V2) {
            if (split) hipLaunchKernelGGL((encode_tiled_kernel<FMT_V2, true>), dim3(blocks), dim3(64), 0, stream, g);
            else       hipLaunchKernelGGL((encode_tiled_kernel<FMT_V2, false>), dim3(blocks), dim3(64), 0, stream, g);
        } else {
            if (split) hipLaunchKernelGGL((encode_tiled_kernel<FMT_V1, true>), dim3(blocks), dim3(64), 0, stream, g);
            else       hipLaunchKernelGGL((encode_tiled_kernel<FMT_V1, false>), dim3(blocks), dim3(64), 0, stream, g);
        }
    }
    return hipGetLastError();
}

// Dense streams decode faster from 16-word rings, but a compute unit holds 7 such waves against 8 with
// 8-word rings.  When a batch is only a few rounds of waves (long series: a wave lives for
// milliseconds), what counts is how the waves fill the rounds: 4,096 waves are two full rounds of 8 per
// unit but two and a third of 7.  The model: a full round costs 1 (0.62 with 16-word rings, measured on
// dense streams), the last, partly filled one 0.55 + 0.45 x its fill (fewer waves per unit run faster).
static bool deep_ring_pays(uint32_t waves)
{
    static const uint32_t units = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return (uint32_t)n;
    }();
    if (waves > 4u * 8u * units) return true;  // many rounds: the tail is small change
    auto cost = [waves](uint32_t slots, double round) {
        const uint32_t rem = waves % slots;
        return ((double)(waves / slots) + (rem ? 0.55 + 0.45 * (double)rem / (double)slots : 0.0)) * round;
    };
    return cost(7u * units, 0.62) <= cost(8u * units, 1.0);
}

// Which kernels decode a batch follows from the batch alone.  Series with a capacity of at least long_points points
// ("long" blocks) may get a wave or a workgroup each instead of a lane: plan_decode_kernel counts them and measures
// their density on the device, and every kernel launched here asks block_owner() about each series -- the three
// kernels run over the batch and each takes its own blocks (the workgroup-per-block one on a stream of its own where
// the caller has one to give: see DecodeFork).  A uniform batch of short series (the
// headline workload) launches the lane-per-series kernel alone.
static hipError_t launch_decode_lanes(const DecodeArgs& a, int format, hipStream_t stream, bool ring_auto, uint32_t& did);

hipError_t launch_decode(const DecodeArgs& a0, int format, hipStream_t stream, const DecodeFork* fork)
{
    if (a0.nseries == 0) return hipSuccess;
    DecodeArgs a = a0;
    uint32_t none = 0;
    uint32_t& did = a.dispatch ? *a.dispatch : none;
    a.long_threshold = ~0ull; a.plan = nullptr; a.ring_by_plan = 0;
    const bool plain = a.dfcm_tables || a.tun.direct;
    const bool may_be_long = !plain && a.plan_buf && (a.pts.off ? true : a.pts.uniform >= a.tun.long_points);
    const bool ring_auto = !plain && a.plan_buf && a.tun.decode_ring == 0;
    if (may_be_long || ring_auto) {
        if (hipError_t e = launch_decode_plan(a, stream)) return e;
        a.plan = a.plan_buf;
    }
    if (may_be_long) {
        a.long_threshold = a.tun.long_points;
        // (a uniform batch: the host knows how many long blocks there are, and which kernels cannot get any)
        const uint64_t nlong = a.pts.off ? ~0ull : a.nseries;
        const bool wg_possible = a.pts.off || nlong <= a.tun.wg_decode_max || (a.pts.uniform >= a.tun.vlong_points && nlong <= a.tun.wg_decode_vlong_max);
        const uint64_t wave_most = a.tun.wave_decode_max > a.tun.wave_decode_max_dense ? a.tun.wave_decode_max : a.tun.wave_decode_max_dense;
        const bool wave_possible = a.pts.off || nlong <= wave_most;
        bool joined = true;
        if (wg_possible) {
            // (a uniform batch of long blocks goes to one kernel as a whole: nothing to run beside it; and the hand-over
            // between the streams costs a call of few blocks 30 us and buys it nothing)
            if (fork && a.pts.off && a.nseries >= 256) {
                if (hipError_t e = hipEventRecord(fork->planned, stream)) return e;
                if (hipError_t e = hipStreamWaitEvent(fork->side, fork->planned, 0)) return e;
                if (hipError_t e = launch_decode_blocks_wg(a, format, fork->side)) return e;
                if (hipError_t e = hipEventRecord(fork->joined, fork->side)) return e;
                (void)hipStreamQuery(fork->side);   // (hands the side stream's commands to the device now: nothing else will ask for them)
                joined = false;
            } else if (hipError_t e = launch_decode_blocks_wg(a, format, stream)) return e;
        }
        if (wave_possible) if (hipError_t e = launch_decode_blocks(a, format, stream)) return e;
        if (!a.pts.off && a.tun.wave_decode_max >= a.nseries && a.tun.wave_decode_max_dense >= a.nseries) {
            // a uniform batch of long blocks, few enough for a wave or a workgroup each whatever the plan says: nothing is
            // left for the lanes
            if (!joined) if (hipError_t e = hipStreamWaitEvent(stream, fork->joined, 0)) return e;
            return hipGetLastError();
        }
        if (!joined) {
            const hipError_t e = launch_decode_lanes(a, format, stream, ring_auto, did);
            if (hipError_t w = hipStreamWaitEvent(stream, fork->joined, 0)) return w;
            return e;
        }
    }
    return launch_decode_lanes(a, format, stream, ring_auto, did);
}

// the lane-per-series kernels of a batch (they leave alone what block_owner() gives to the other two)
static hipError_t launch_decode_lanes(const DecodeArgs& a, int format, hipStream_t stream, bool ring_auto, uint32_t& did)
{
    if (a.dfcm_tables) {
        if (format != FMT_V2) return hipErrorInvalidValue;
        const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
        did |= DISPATCH_LANE_DIRECT;
        hipLaunchKernelGGL((decode_direct_kernel<FMT_V2, PRED_DFCM>), dim3(blocks), dim3(256), 0, stream, a);
    } else if (a.tun.direct) {
        const uint32_t blocks = (uint32_t)((a.nseries + 255) / 256);
        did |= DISPATCH_LANE_DIRECT;
        if (format == FMT_V2) hipLaunchKernelGGL(decode_direct_kernel<FMT_V2>, dim3(blocks), dim3(256), 0, stream, a);
        else                  hipLaunchKernelGGL(decode_direct_kernel<FMT_V1>, dim3(blocks), dim3(256), 0, stream, a);
    } else {
        DecodeArgs g = a;
        g.group_shift = group_shift_of(a.pts);
        const uint32_t blocks = grouped_blocks(a.nseries, g.group_shift);
        g.run_skip = a.tun.run_skip;
        g.ring_by_plan = ring_auto ? 1u : 0u;
        // 16-word rings where asked for -- unless the batch is so few rounds of waves that 8 per unit beat 7 (see above)
        const bool deep = a.tun.decode_ring == 16 && deep_ring_pays(blocks);
        const bool both = ring_auto && deep_ring_pays(blocks);
        did |= DISPATCH_LANE_TILED | (both ? DISPATCH_LANE_BOTH_RINGS : deep ? DISPATCH_LANE_RING16 : 0u);
        if (!both) g.ring_by_plan = 0;
        if (format == FMT_V2) {
            if (deep || both) hipLaunchKernelGGL((decode_tiled_kernel<FMT_V2, 16>), dim3(blocks), dim3(64), 0, stream, g);
            if (!deep)        hipLaunchKernelGGL((decode_tiled_kernel<FMT_V2, 8>), dim3(blocks), dim3(64), 0, stream, g);
        } else {
            if (deep || both) hipLaunchKernelGGL((decode_tiled_kernel<FMT_V1, 16>), dim3(blocks), dim3(64), 0, stream, g);
            if (!deep)        hipLaunchKernelGGL((decode_tiled_kernel<FMT_V1, 8>), dim3(blocks), dim3(64), 0, stream, g);
        }
    }
    return hipGetLastError();
}

hipError_t launch_words_to_bytes(uint64_t nseries, const uint64_t* words, Layout wslots, const uint64_t* nbits,
                                 uint8_t* bytes, Layout bslots, uint64_t* out_nbytes, uint32_t* flags, hipStream_t stream)
{
    if (nseries == 0) return hipSuccess;
    uint64_t blocks = (nseries + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(words_to_bytes_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, nseries, words, wslots, nbits,
                       bytes, bslots, out_nbytes, flags);
    return hipGetLastError();
}

hipError_t launch_bytes_to_words(uint64_t nseries, const uint8_t* bytes, Layout bslots, const uint64_t* nbytes,
                                 uint64_t* words, Layout wslots, uint64_t* out_nwords, uint32_t* flags, hipStream_t stream)
{
    if (nseries == 0) return hipSuccess;
    uint64_t blocks = (nseries + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(bytes_to_words_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, nseries, bytes, bslots, nbytes,
                       words, wslots, out_nwords, flags);
    return hipGetLastError();
}

uint64_t scan_scratch_words(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

// Scratch for the compaction entry point that allocates nothing: the partial sums of its
// exclusive scan, one word per SCAN_TILE series.  Every GPU the library runs on holds its own
// copy of this array, at its own address, until the process ends.  In host code the array's
// name is only a handle for hipGetSymbolAddress; kernels get the address that call returns.
static __device__ __align__(256) uint64_t scratch[NOALLOC_SCRATCH_WORDS];

hipError_t noalloc_scratch_of_current_device(uint64_t** out)
{
    constexpr int MAX_DEVICES = 64;
    static uint64_t* ptr_of[MAX_DEVICES] = {};
    static std::mutex mu;
    *out = nullptr;
    int dev = -1;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr_of[dev]) {
        void* p = nullptr;
        if (hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch))) return e;
        ptr_of[dev] = static_cast<uint64_t*>(p);
    }
    *out = ptr_of[dev];
    return hipSuccess;
}

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
