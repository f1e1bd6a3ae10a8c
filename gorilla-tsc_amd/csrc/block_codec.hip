// Wave-per-block form of the Gorilla codec on gfx950: the 64 lanes of a wavefront work on 64
// consecutive points of ONE block.  It serves batches of few and/or long blocks, where the
// lane-per-series kernels (kernels.hip) leave most of the machine idle: a block there is a serial
// chain walked by a single lane.
//
// Compressor.  Everything about a point except the value window is a function of the point and
// its two predecessors (closed form of G/GorillaCompressor.java:49-143 and
// G/ValueCompressor.java:32-91, G/ = src/main/java/fi/iki/yak/ts/compression/gorilla/ in the
// reference tree), so the lanes compute delta-of-delta codes, XORs and leading/trailing zero
// counts for 64 points at once; neighbours' values come by data-parallel-primitive moves.  The
// window (storedLeadingZeros, storedTrailingZeros) is the one serial dependence
// (G/ValueCompressor.java:46-50,89-90): it is wave-uniform state, and one round of the loop in
// resolve_window() handles all points up to the next one that opens a new window.  An exclusive
// prefix sum of the code lengths gives every code its bit position; the lanes OR their codes
// into a staging area in LDS, and whole words leave for HBM as contiguous 8-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "codec_device.hpp"
#include "kernels.hpp"
#include "wave_util.hpp"

namespace gtsc {

namespace {

// Staging area of a wave: up to 63 pending bits + 64 codes of at most 114 bits = 115 words, plus the
// two words a code may spill into.
constexpr uint32_t STAGE_WORDS = 128;
constexpr uint32_t BLOCK_MAX_POINTS = 0x7FFFFF00u;  // points per block (the same limit as the lane-per-series kernels)

// What a lane contributes to the stream for its point: `head` (hlen <= 50 bits: timestamp code and
// value header, or the 27-bit first delta) followed by `pay` (plen <= 64 bits); both right-aligned
// with zeros above, both zero when their length is zero.
struct LaneCode {
    uint64_t head, pay;
    uint32_t hlen, plen;
};

// Bit writer of one block, wave-uniform state: words already in HBM, bits waiting in stage[0].
struct BlockWriter {
    uint64_t* stage;   // LDS, STAGE_WORDS words, zero except for the pending bits in word 0
    uint64_t* out;     // the block's slot
    uint64_t cap;      // words in the slot
    uint64_t wcount;   // whole words written (or dropped beyond cap) so far
    uint32_t base;     // pending bits (< 64)

    // Appends the lanes' codes in lane order and writes out the words they complete.
    __device__ __forceinline__ void append(const LaneCode& c, uint32_t lane)
    {
        const uint32_t len = c.hlen + c.plen;
        const uint32_t incl = wave_inclusive_sum(len);
        const uint32_t total = read_lane(incl, 63);
        const uint32_t pos = base + incl - len;       // bit position of this lane's code in the staging area
        const uint32_t w0 = pos >> 6, sh = pos & 63;
        // the code left-aligned in 128 bits (hi:lo)
        const uint64_t A = c.head << ((64u - c.hlen) & 63u);
        const uint64_t B = c.pay << ((64u - c.plen) & 63u);
        const uint64_t hi = A | (B >> c.hlen);
        const uint64_t lo = (B << 1) << (63u - c.hlen);
        // ... and shifted to its place: three words at most
        const uint64_t x0 = hi >> sh;
        const uint64_t x1 = ((hi << 1) << (63u - sh)) | (lo >> sh);
        const uint64_t x2 = (lo << 1) << (63u - sh);
        unsigned long long* const st = reinterpret_cast<unsigned long long*>(stage);
        if (x0) atomicOr(st + w0, (unsigned long long)x0);
        if (x1) atomicOr(st + w0 + 1, (unsigned long long)x1);
        if (x2) atomicOr(st + w0 + 2, (unsigned long long)x2);
        const uint32_t tot = base + total;
        const uint32_t nfull = tot >> 6;              // <= 115
        wave_sync();
        if (nfull) {
            const uint64_t partial = stage[nfull];    // every lane reads the same word
            const uint32_t k1 = lane + 64;
            uint64_t a0 = 0, a1 = 0;
            if (lane < nfull) a0 = stage[lane];
            if (k1 < nfull) a1 = stage[k1];
            wave_sync();
            if (lane < nfull) { stage[lane] = 0; if (wcount + lane < cap) out[wcount + lane] = a0; }
            if (k1 < nfull) { stage[k1] = 0; if (wcount + k1 < cap) out[wcount + k1] = a1; }
            if (lane == 0) stage[nfull] = 0;
            wave_sync();
            if (lane == 0) stage[0] = partial;
            wave_sync();
        }
        wcount += nfull;
        base = tot & 63;
    }
};

// The timestamp code of an ordinary point (2.x): int arithmetic on the low halves,
// G/GorillaCompressor.java:99-143.  Returns the code (<= 36 bits) and its length.
__device__ __forceinline__ void ts_code_v2(uint32_t t, uint32_t t1, uint32_t stored_delta, uint64_t& code, uint32_t& len)
{
    const uint32_t nd = t - t1;
    const int32_t dod = (int32_t)(nd - stored_delta);
    const uint32_t z = (((uint32_t)dod << 1) ^ (uint32_t)(dod >> 31)) - 1u;  // zigzag - 1
    uint32_t need = 32u - (uint32_t)__clz((int)z);  // bits required; 0 when z == 0 (dod == -1)
    need = need > 13u ? 13u : need;
    // two bits per entry: 0 -> 36-bit code (need 0 or > 12), 1 -> '10'+7, 2 -> '110'+9, 3 -> '1110'+12
    constexpr uint32_t BUCKET = (1u * 0x5554u) | (2u << 16) | (2u << 18) | (3u << 20) | (3u << 22) | (3u << 24);
    const uint32_t bk = (BUCKET >> (2u * need)) & 3u;
    constexpr uint32_t LEN = 36u | (9u << 8) | (12u << 16) | (16u << 24);
    constexpr uint64_t PRE = 0ull | (0x100ull << 16) | (0xC00ull << 32) | (0xE000ull << 48);
    const bool zero = dod == 0;
    len = zero ? 1u : (LEN >> (8u * bk)) & 0xFFu;
    const uint32_t pre = (uint32_t)(PRE >> (16u * bk)) & 0xFFFFu;
    const uint32_t lo = zero ? 0u : (z | pre);
    const uint32_t hi = (zero || bk != 0) ? 0u : 0xFu;
    code = ((uint64_t)hi << 32) | lo;
}

// The same for the 1.x format: long arithmetic, two's complement payloads, G/Compressor.java:94-118.
__device__ __forceinline__ void ts_code_v1(uint64_t t, uint64_t t1, uint64_t stored_delta, uint64_t& code, uint32_t& len)
{
    const uint64_t dod = (t - t1) - stored_delta;
    const bool zero = dod == 0;
    const bool b7 = (dod + 63u) <= 127u, b9 = (dod + 255u) <= 511u, b12 = (dod + 2047u) <= 4095u;
    len = zero ? 1u : b7 ? 9u : b9 ? 12u : b12 ? 16u : 36u;
    const uint32_t l = (uint32_t)dod;
    const uint32_t lo = zero ? 0u : b7 ? (0x100u | (l & 0x7Fu)) : b9 ? (0xC00u | (l & 0x1FFu)) : b12 ? (0xE000u | (l & 0xFFFu)) : l;
    const uint32_t hi = (zero || b7 || b9 || b12) ? 0u : 0xFu;
    code = ((uint64_t)hi << 32) | lo;
}

// The serial part of the value codec, for 64 points at once.  A point with a nonzero XOR reuses
// the stored window when its own leading and trailing zero counts are both at least the stored
// ones, else it opens a new window with its own counts (G/ValueCompressor.java:46-50; no cost
// heuristic).  (lead, trail) is the wave-uniform window on entry and on exit; every lane gets
// the window that holds at its point (wl, wt) and whether its point opens one.  One round of the
// loop settles all points up to the next window change.
__device__ __forceinline__ void resolve_window(bool nz, uint32_t lz, uint32_t tz, uint32_t lane, uint32_t& lead,
                                               uint32_t& trail, uint32_t& wl, uint32_t& wt, bool& fresh)
{
    uint64_t pend = lanes(nz);
    uint64_t fresh_m = 0;
    wl = lead; wt = trail;
    while (pend) {
        const uint64_t viol = lanes(nz && !(lz >= lead && tz >= trail)) & pend;
        if (!viol) break;
        const uint32_t j = (uint32_t)__builtin_ctzll(viol);
        lead = read_lane(lz, j);
        trail = read_lane(tz, j);
        fresh_m |= 1ull << j;
        pend &= (~1ull) << j;   // the lanes after j
        if (lane > j) { wl = lead; wt = trail; }
    }
    fresh = (fresh_m >> lane) & 1u;
}

// What the wave carries from one chunk of 64 points to the next.
struct ChunkCarry {
    uint32_t lead, trail;       // the value window (wave-uniform)
    uint64_t t1, t2, v1;        // the two timestamps and the value before the chunk's lane 0
};

// What a lane knows about its point before the value window is settled.
struct PointPrep {
    uint64_t t, v, x;           // timestamp, value, XOR with the previous value
    uint64_t tcode;             // timestamp code and its length
    uint32_t tlen;
    uint32_t lz, tz;            // leading / trailing zeros of x (leading ones capped for the 1.x format)
    bool active, first, nz;     // a point at all; written as a "first" point; changed value
};

// Lane l holds point i0 + l of the block (t, v), if that is a point at all; t1, t2, v1 are the
// two timestamps and the value before it.
template <int FMT>
__device__ __forceinline__ PointPrep prepare_point(uint64_t t, uint64_t v, uint64_t t1, uint64_t t2, uint64_t v1, int32_t i,
                                                   uint32_t np, int64_t bts)
{
    PointPrep p;
    p.t = t; p.v = v;
    p.active = (uint32_t)i < np;
    // "first" is whoever follows a stored timestamp of zero (G/GorillaCompressor.java:50,65)
    p.first = i == 0 || t1 == 0;
    const bool prev_first = i == 1 || t2 == 0;
    p.tcode = 0; p.tlen = 1;  // regular cadence: the single bit '0'
    if (FMT == FMT_V2) {
        const uint32_t sd = (uint32_t)t1 - (prev_first ? (uint32_t)bts : (uint32_t)t2);
        if (any_lane(p.active && !p.first && (uint32_t)t - (uint32_t)t1 != sd)) ts_code_v2((uint32_t)t, (uint32_t)t1, sd, p.tcode, p.tlen);
    } else {
        const uint64_t sd = t1 - (prev_first ? (uint64_t)bts : t2);
        if (any_lane(p.active && !p.first && t - t1 != sd)) ts_code_v1(t, t1, sd, p.tcode, p.tlen);
    }
    p.x = v ^ v1;
    p.nz = p.active && !p.first && p.x != 0;
    p.lz = (uint32_t)__clzll((long long)p.x);
    p.tz = (uint32_t)(__ffsll((long long)p.x) - 1);
    if (FMT == FMT_V1) p.lz = p.lz >= 32 ? 31u : p.lz;     // G/Compressor.java:132-134
    return p;
}

// The lane's code once the window that holds at its point (wl, wt) is known, and whether the point opens it.
template <int FMT>
__device__ __forceinline__ LaneCode point_code(const PointPrep& p, uint32_t wl, uint32_t wt, bool fresh, int64_t bts)
{
    const bool reuse = p.nz && !fresh;
    const uint32_t sig = 64u - p.lz - p.tz;
    const uint32_t vh_new = (FMT == FMT_V2) ? ((3u << 12) | (p.lz << 6) | (sig - 1u)) : ((3u << 11) | (p.lz << 6) | (sig & 63u));
    const uint32_t vh = p.nz ? (reuse ? 2u : vh_new) : 0u;
    const uint32_t vhlen = p.nz ? (reuse ? 2u : (FMT == FMT_V2 ? 14u : 13u)) : 1u;
    LaneCode code;
    code.head = p.active ? (p.tcode << vhlen) | vh : 0ull;
    code.hlen = p.active ? p.tlen + vhlen : 0u;
    code.pay = p.nz ? p.x >> ((reuse ? wt : p.tz) & 63u) : 0ull;
    code.plen = p.nz ? (reuse ? 64u - wl - wt : sig) : 0u;
    if (any_lane(p.active && p.first)) {
        // writeFirst: 27 low bits of the delta to the block timestamp, then the raw value
        if (p.active && p.first) {
            code.head = (p.t - (uint64_t)bts) & 0x7FFFFFFull; code.hlen = 27;
            code.pay = p.v; code.plen = 64;
        }
    }
    return code;
}

// Codes one chunk: lane l holds point i0 + l of the block (t, v), if that is a point at all.
template <int FMT>
__device__ __forceinline__ void encode_chunk(BlockWriter& w, ChunkCarry& k, uint64_t t, uint64_t v, int32_t i0, uint32_t np,
                                             int64_t bts, uint32_t lane)
{
    const uint64_t t1 = lane_shift_up(t, k.t1);
    const uint64_t t2 = lane_shift_up(t1, k.t2);
    const uint64_t v1 = lane_shift_up(v, k.v1);
    k.t1 = read_lane(t, 63); k.t2 = read_lane(t1, 63); k.v1 = read_lane(v, 63);
    const PointPrep p = prepare_point<FMT>(t, v, t1, t2, v1, i0 + (int32_t)lane, np, bts);
    uint32_t wl, wt; bool fresh;
    resolve_window(p.nz, p.lz, p.tz, lane, k.lead, k.trail, wl, wt, fresh);
    w.append(point_code<FMT>(p, wl, wt, fresh, bts), lane);
}

// Compresses block `s` with the whole wave.
template <int FMT>
__device__ __forceinline__ void encode_block(const EncodeArgs& a, uint64_t s, uint64_t* stage, uint32_t lane)
{
    const uint64_t p0 = a.pts.begin(s), np64 = a.pts.size(s);
    const uint32_t np = np64 > BLOCK_MAX_POINTS ? BLOCK_MAX_POINTS : (uint32_t)np64;
    const int64_t bts = a.block_ts[s];
    BlockWriter w{stage, a.out_words + a.slots.begin(s), a.slots.size(s), 1, 0};
    if (lane == 0 && w.cap > 0) w.out[0] = (uint64_t)bts;  // addHeader, G/GorillaCompressor.java:39-41

    // Chunks of 64 points on the 512-byte grid of the timestamp array: lane l of chunk c holds array
    // index g0 + 64 c + l, i.e. point 64 c + l - phase of the block.
    const uint32_t phase = (uint32_t)((p0 + ((uintptr_t)a.ts >> 3)) & 63u);
    const uint32_t nchunks = np ? (uint32_t)(((uint64_t)phase + np + 63) / 64) : 0u;
    const int64_t* const pt = a.ts + p0;
    const uint64_t* const pv = a.val + p0;

    // Three chunks are in flight in three sets of registers; a set is loaded again as soon as its
    // chunk is coded (a rotation through one set would copy registers at the loop's end, and the
    // copy would wait for the load just issued).  The loads are unconditional -- lanes without a
    // point fetch the block's nearest point instead -- so that their number between any two places
    // of the loop is fixed and a wait for one set does not have to wait for the sets issued after it.
    uint64_t tA = 0, vA = 0, tB = 0, vB = 0, tC = 0, vC = 0;
    const uint32_t last = np ? np - 1 : 0;
#define GTSC_BLOCK_LOAD(C, T_, V_)                                                             \
    {                                                                                          \
        const uint32_t u_ = (uint32_t)(C) * 64 + lane;                                         \
        const uint32_t d_ = u_ - phase;                                                        \
        const uint32_t i_ = u_ < phase ? 0u : (d_ < last ? d_ : last);                         \
        T_ = (uint64_t)__builtin_nontemporal_load(pt + i_);                                    \
        V_ = __builtin_nontemporal_load(pv + i_);                                              \
    }
    if (np) {
    GTSC_BLOCK_LOAD(0, tA, vA)
    GTSC_BLOCK_LOAD(1, tB, vB)
    GTSC_BLOCK_LOAD(2, tC, vC)
    }
    ChunkCarry k{NO_WINDOW, 0, 0, 0, 0};
    // (a block's last round may hold one or two chunks without points: they run empty rather than
    // branch around, again for the sake of a fixed number of loads between waits)
    for (uint32_t c = 0; c < nchunks; c += 3) {
        encode_chunk<FMT>(w, k, tA, vA, (int32_t)(c * 64) - (int32_t)phase, np, bts, lane);
        GTSC_BLOCK_LOAD(c + 3, tA, vA)
        encode_chunk<FMT>(w, k, tB, vB, (int32_t)(c * 64 + 64) - (int32_t)phase, np, bts, lane);
        GTSC_BLOCK_LOAD(c + 4, tB, vB)
        encode_chunk<FMT>(w, k, tC, vC, (int32_t)(c * 64 + 128) - (int32_t)phase, np, bts, lane);
        GTSC_BLOCK_LOAD(c + 5, tC, vC)
    }
#undef GTSC_BLOCK_LOAD
    uint64_t nbits = w.wcount * 64 + w.base;
    if (!a.no_close) {
        // close(): '1111' + 32 ones + '0' (G/GorillaCompressor.java:84-89)
        LaneCode m{0, 0, 0, 0};
        if (lane == 0) { m.head = 0xFFFFFFFFFull << 1; m.hlen = 37; }
        w.append(m, lane);
        nbits += 37;
    }
    // getLongArray(): the partial word (a zero word when the stream ended on a word boundary), and
    // after close() one more zero word (G/LongArrayOutput.java:136-145)
    const uint64_t nwords = w.wcount + (a.no_close ? 1 : 2);
    if (lane == 0) {
        if (w.wcount < w.cap) w.out[w.wcount] = stage[0];
        if (!a.no_close && w.wcount + 1 < w.cap) w.out[w.wcount + 1] = 0;
        stage[0] = 0;
        // a block beyond the point limit was coded up to the limit: reported, not silent
        const uint32_t st = (nwords > w.cap || np64 > BLOCK_MAX_POINTS) ? ST_OVERFLOW : 0u;
        a.out_nwords[s] = nwords;
        if (a.out_nbits) a.out_nbits[s] = nbits;
        if (a.status) a.status[s] = st;
        if (st) atomicOr(a.flags, st);
    }
    wave_sync();
}

// One wave per workgroup.  Wave w looks at the series w + k * gridDim.x, k < per_wave, and
// compresses those with at least long_threshold points (the others belong to the
// lane-per-series kernel).
template <int FMT>
__global__ __launch_bounds__(64) void encode_block_kernel(EncodeArgs a, uint32_t per_wave)
{
    __shared__ uint64_t stage[STAGE_WORDS];
    const uint32_t lane = threadIdx.x;
    stage[lane] = 0; stage[lane + 64] = 0;
    wave_sync();
    const uint64_t cand = blockIdx.x + (uint64_t)lane * gridDim.x;
    const bool mine = lane < per_wave && cand < a.nseries && a.pts.size(cand) >= a.long_threshold;
    uint64_t todo = lanes(mine);
    while (todo) {
        const uint32_t j = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        encode_block<FMT>(a, blockIdx.x + (uint64_t)j * gridDim.x, stage, lane);
    }
}


// ------------------------------------------------------------------------
// Cooperative form of the compressor for batches of so few blocks that a wave per block leaves the
// machine empty: a workgroup of COOP_WAVES waves per block, all on one compute unit (the waves talk
// through LDS).  Wave w of round r takes chunk r * COOP_WAVES + w (64 points).  Everything but the
// window and the bit position is worked out by the waves side by side.  The window: a chunk without
// a point that violates the window in force leaves it unchanged, so all chunks of a round test
// themselves against it at once, and only the chunks that do change it take turns (see the loop in
// GTSC_COOP_ROUND).  The bit position: every wave adds up the code lengths of the chunks before its
// own.  Barriers separate the steps.  The codes go into a staging ring shared by the workgroup at
// their absolute bit positions; after each round the words the round has completed leave for HBM.
// ------------------------------------------------------------------------
constexpr uint32_t COOP_WAVES = 16;
constexpr uint32_t COOP_STAGE_WORDS = COOP_WAVES * 128;   // a round's codes (64 x 114 bits = 114 words per chunk at most) and the word before them

struct CoopLds {
    uint64_t stage[COOP_STAGE_WORDS];   // image word i lives in stage[i % COOP_STAGE_WORDS] until it is written out
    uint32_t lead, trail;               // the value window after the chunks settled so far
    uint32_t flags[2];                  // bit w: wave w's chunk has a point that does not fit the window (two words taking turns)
    uint32_t totals[COOP_WAVES];        // bits of each wave's chunk in this round
};

// LDS words that the waves of a workgroup exchange: accesses the compiler may neither merge nor move
// (the value read is the same in every lane and comes back as a scalar)
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_poke(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// All waves of the workgroup have carried out their LDS accesses up to here.  (Not __syncthreads(): that
// would also wait for global memory, and the next rounds' loads are meant to stay in flight.)
__device__ __forceinline__ void coop_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int FMT>
__global__ __launch_bounds__(64 * COOP_WAVES) void encode_block_coop_kernel(EncodeArgs a)
{
    __shared__ CoopLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint64_t s = blockIdx.x;
    const uint64_t np64 = a.pts.size(s);
    if (np64 < a.long_threshold) return;   // the lane-per-series kernel's (the whole workgroup leaves together)
    for (uint32_t i = tid; i < COOP_STAGE_WORDS; i += 64 * COOP_WAVES) L.stage[i] = 0;
    if (tid == 0) { L.lead = NO_WINDOW; L.trail = 0; L.flags[0] = 0; L.flags[1] = 0; }
    __syncthreads();

    const uint64_t p0 = a.pts.begin(s);
    const uint32_t np = np64 > BLOCK_MAX_POINTS ? BLOCK_MAX_POINTS : (uint32_t)np64;
    const int64_t bts = a.block_ts[s];
    uint64_t* const out = a.out_words + a.slots.begin(s);
    const uint64_t cap = a.slots.size(s);
    if (tid == 0 && cap > 0) out[0] = (uint64_t)bts;  // addHeader, G/GorillaCompressor.java:39-41
    const uint32_t phase = (uint32_t)((p0 + ((uintptr_t)a.ts >> 3)) & 63u);
    const uint32_t nchunks = np ? (uint32_t)(((uint64_t)phase + np + 63) / 64) : 0u;
    const uint32_t nrounds = (nchunks + COOP_WAVES - 1) / COOP_WAVES;
    const int64_t* const pt = a.ts + p0;
    const uint64_t* const pv = a.val + p0;
    const uint32_t last = np ? np - 1 : 0;

    // A wave's chunks of three rounds are in flight in three sets of registers, loaded from clamped
    // addresses without conditions (as in encode_block); with each chunk come the two timestamps and
    // the value before its first point, which another wave's chunk holds.
    uint64_t tA = 0, vA = 0, tB = 0, vB = 0, tC = 0, vC = 0;
    uint64_t pA1 = 0, pA2 = 0, qA = 0, pB1 = 0, pB2 = 0, qB = 0, pC1 = 0, pC2 = 0, qC = 0;
#define GTSC_COOP_LOAD(R, T_, V_, P1_, P2_, Q_)                                                \
    {                                                                                          \
        const uint32_t u_ = ((uint32_t)(R) * COOP_WAVES + wave) * 64u;                         \
        const uint32_t ul_ = u_ + lane;                                                        \
        const uint32_t d_ = ul_ - phase;                                                       \
        const uint32_t i_ = ul_ < phase ? 0u : (d_ < last ? d_ : last);                        \
        T_ = (uint64_t)__builtin_nontemporal_load(pt + i_);                                    \
        V_ = __builtin_nontemporal_load(pv + i_);                                              \
        const uint32_t j1_ = u_ < phase + 1u ? 0u : (u_ - phase - 1u < last ? u_ - phase - 1u : last); \
        const uint32_t j2_ = u_ < phase + 2u ? 0u : (u_ - phase - 2u < last ? u_ - phase - 2u : last); \
        P1_ = (uint64_t)pt[j1_]; P2_ = (uint64_t)pt[j2_]; Q_ = pv[j1_];                        \
    }
    if (np) {
        GTSC_COOP_LOAD(0, tA, vA, pA1, pA2, qA)
        GTSC_COOP_LOAD(1, tB, vB, pB1, pB2, qB)
        GTSC_COOP_LOAD(2, tC, vC, pC1, pC2, qC)
    }
    uint64_t flushed = 1;   // image words written out so far (every thread keeps count)
    uint64_t base = 64;     // bit position where the next round's codes begin (likewise)
    uint32_t turn = 0;      // which of the two flag words is in use
    unsigned long long* const st = reinterpret_cast<unsigned long long*>(L.stage);

#define GTSC_COOP_ROUND(R, T_, V_, P1_, P2_, Q_)                                               \
    {                                                                                          \
        const uint32_t c_ = (uint32_t)(R) * COOP_WAVES + wave;                                 \
        const bool have_ = c_ < nchunks;                                                       \
        PointPrep p_{};                                                                        \
        if (have_) {                                                                           \
            const int32_t i0_ = (int32_t)(c_ * 64u) - (int32_t)phase;                          \
            const uint64_t t1_ = lane_shift_up(T_, P1_);                                       \
            const uint64_t t2_ = lane_shift_up(t1_, P2_);                                      \
            const uint64_t v1_ = lane_shift_up(V_, Q_);                                        \
            p_ = prepare_point<FMT>(T_, V_, t1_, t2_, v1_, i0_ + (int32_t)lane, np, bts);      \
        }                                                                                      \
        /* ---- windows.  A chunk none of whose points violates the window in force leaves it as it is, so   \
           all chunks test themselves against it side by side; the first one that has a violator settles      \
           its points serially (resolve_window) and publishes the window it ends with, the chunks after it    \
           test again -- one turn of this loop per chunk that changes the window. */            \
        uint32_t lead_ = lds_peek(&L.lead), trail_ = lds_peek(&L.trail);                       \
        uint32_t wl_ = lead_, wt_ = trail_; bool fresh_ = false;                               \
        if (have_ && any_lane(p_.nz && !(p_.lz >= lead_ && p_.tz >= trail_)) && lane == 0) atomicOr(&L.flags[turn & 1u], 1u << wave); \
        coop_barrier();                                                                        \
        for (;;) {                                                                             \
            const uint32_t f_ = lds_peek(&L.flags[turn & 1u]);                                 \
            if (f_ == 0) break;                                                                \
            const uint32_t first_ = (uint32_t)__builtin_ctz(f_);                               \
            if (wave == first_) {                                                              \
                resolve_window(p_.nz, p_.lz, p_.tz, lane, lead_, trail_, wl_, wt_, fresh_);    \
                if (lane == 0) { lds_poke(&L.lead, lead_); lds_poke(&L.trail, trail_); lds_poke(&L.flags[(turn + 1u) & 1u], 0u); } \
            }                                                                                  \
            coop_barrier();                                                                    \
            turn++;                                                                            \
            if (wave > first_ && have_) {                                                      \
                lead_ = lds_peek(&L.lead); trail_ = lds_peek(&L.trail);                        \
                wl_ = lead_; wt_ = trail_;                                                     \
                if (any_lane(p_.nz && !(p_.lz >= lead_ && p_.tz >= trail_)) && lane == 0) atomicOr(&L.flags[turn & 1u], 1u << wave); \
            }                                                                                  \
            coop_barrier();                                                                    \
        }                                                                                      \
        /* ---- codes and where they go: every wave adds up the lengths of the chunks before its own */ \
        const LaneCode code_ = point_code<FMT>(p_, wl_, wt_, fresh_, bts);                     \
        const uint32_t len_ = code_.hlen + code_.plen;                                         \
        const uint32_t incl_ = wave_inclusive_sum(len_);                                       \
        if (lane == 0) lds_poke(&L.totals[wave], have_ ? read_lane(incl_, 63) : 0u);           \
        coop_barrier();                                                                        \
        const uint32_t tot_ = lane < COOP_WAVES ? __hip_atomic_load(&L.totals[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u; \
        const uint32_t run_ = wave_inclusive_sum(tot_);                                        \
        const uint64_t pos_ = base + (wave ? read_lane(run_, wave - 1u) : 0u);                 \
        base += read_lane(run_, COOP_WAVES - 1u);                                              \
        if (have_) {                                                                           \
            /* the codes into the ring: 128 bits left-aligned, shifted to their place, three words at most */ \
            const uint64_t at_ = pos_ + incl_ - len_;                                          \
            const uint32_t w0_ = (uint32_t)(at_ >> 6), sh_ = (uint32_t)at_ & 63u;              \
            const uint64_t A_ = code_.head << ((64u - code_.hlen) & 63u);                      \
            const uint64_t B_ = code_.pay << ((64u - code_.plen) & 63u);                       \
            const uint64_t hi_ = A_ | (B_ >> code_.hlen);                                      \
            const uint64_t lo_ = (B_ << 1) << (63u - code_.hlen);                              \
            const uint64_t x0_ = hi_ >> sh_;                                                   \
            const uint64_t x1_ = ((hi_ << 1) << (63u - sh_)) | (lo_ >> sh_);                   \
            const uint64_t x2_ = (lo_ << 1) << (63u - sh_);                                    \
            if (x0_) atomicOr(st + (w0_ & (COOP_STAGE_WORDS - 1)), (unsigned long long)x0_);   \
            if (x1_) atomicOr(st + ((w0_ + 1) & (COOP_STAGE_WORDS - 1)), (unsigned long long)x1_); \
            if (x2_) atomicOr(st + ((w0_ + 2) & (COOP_STAGE_WORDS - 1)), (unsigned long long)x2_); \
        }                                                                                      \
        /* ---- the round is packed: its whole words leave */                                  \
        coop_barrier();                                                                        \
        {                                                                                      \
            const uint64_t e_ = base >> 6;                                                     \
            for (uint64_t w_ = flushed + tid; w_ < e_; w_ += 64 * COOP_WAVES) {                \
                const uint64_t x_ = L.stage[w_ & (COOP_STAGE_WORDS - 1)];                      \
                L.stage[w_ & (COOP_STAGE_WORDS - 1)] = 0;                                      \
                if (w_ < cap) out[w_] = x_;                                                    \
            }                                                                                  \
            flushed = e_;                                                                      \
        }                                                                                      \
        coop_barrier();                                                                        \
    }
    for (uint32_t r = 0; r < nrounds; r += 3) {
        GTSC_COOP_ROUND(r, tA, vA, pA1, pA2, qA)
        GTSC_COOP_LOAD(r + 3, tA, vA, pA1, pA2, qA)
        if (r + 1 < nrounds) GTSC_COOP_ROUND(r + 1, tB, vB, pB1, pB2, qB)
        GTSC_COOP_LOAD(r + 4, tB, vB, pB1, pB2, qB)
        if (r + 2 < nrounds) GTSC_COOP_ROUND(r + 2, tC, vC, pC1, pC2, qC)
        GTSC_COOP_LOAD(r + 5, tC, vC, pC1, pC2, qC)
    }
#undef GTSC_COOP_ROUND
#undef GTSC_COOP_LOAD
    if (wave != 0) return;
    // ---- wave 0: close marker, the partial word and the zero word(s) of getLongArray()
    uint64_t nbits = base;
    if (!a.no_close) {
        // close(): '1111' + 32 ones + '0' (G/GorillaCompressor.java:84-89)
        if (lane == 0) {
            const uint64_t m = 0xFFFFFFFFFull << 1;   // 37 bits
            const uint32_t w0 = (uint32_t)(nbits >> 6), sh = (uint32_t)nbits & 63u;
            const uint64_t hi = m << 27;              // left-aligned
            L.stage[w0 & (COOP_STAGE_WORDS - 1)] |= hi >> sh;
            L.stage[(w0 + 1) & (COOP_STAGE_WORDS - 1)] |= (hi << 1) << (63u - sh);
        }
        nbits += 37;
        wave_sync();
    }
    const uint64_t wcount = nbits >> 6;               // whole words of the stream
    const uint64_t nwords = wcount + (a.no_close ? 1 : 2);
    if (lane == 0) {
        // (at most one whole word is left beside the partial one: the marker is shorter than a word)
        for (uint64_t w = flushed; w <= wcount; w++)
            if (w < cap) out[w] = L.stage[w & (COOP_STAGE_WORDS - 1)];
        if (!a.no_close && wcount + 1 < cap) out[wcount + 1] = 0;
        const uint32_t stt = (nwords > cap || np64 > BLOCK_MAX_POINTS) ? ST_OVERFLOW : 0u;
        a.out_nwords[s] = nwords;
        if (a.out_nbits) a.out_nbits[s] = nbits;
        if (a.status) a.status[s] = stt;
        if (stt) atomicOr(a.flags, stt);
    }
}

// ------------------------------------------------------------------------
// Decompressor.  Where a point starts is only known by decoding what precedes it, so the chain
// of start positions is the serial part here.  It is found speculatively: the 64 lanes take the
// 64 bit positions that follow the current one as candidate starts and each works out how long
// a point starting there would be (timestamp bucket from the leading ones, value code from its
// two control bits and, for a reused window, the wave-uniform window size:
// G/GorillaDecompressor.java:74-112, G/ValueDecompressor.java:32-54).  A scalar loop then hops
// from real start to real start through those lengths (runs of unchanged points, '00' per point,
// in one hop).  Real starts are queued; every 64 queued points the lanes extract delta-of-deltas
// and XOR payloads in parallel, and prefix sums / a prefix XOR over the wave turn them into
// timestamps and values, stored as contiguous lines.
//
// Anything out of the ordinary -- end of the image inside a point, a reused window that was never
// defined, a decoded timestamp of zero (which makes the reference read the next point as a
// "first" one), more points than the caller's capacity -- hands the block to lane 0, which
// walks on from the last point stored with the serial decoder of the lane-per-series kernels, so
// that counts, status words and the order in which the reference's reads meet a problem are the same.
// ------------------------------------------------------------------------
constexpr uint32_t RING_WORDS_D = 512;   // image words staged in LDS (circular): up to 127 queued points behind, 64 candidates ahead
constexpr uint32_t QUEUE_SLOTS = 128;    // queued point starts

struct BlockDecodeLds {
    uint64_t ring[RING_WORDS_D];
    uint32_t qpos[QUEUE_SLOTS];   // start of a queued point, in bits from the block's committed position
    uint32_t qwin[QUEUE_SLOTS];   // the value window in force there: significant bits | (trailing zeros & 127) << 8
};

template <int FMT>
__device__ __forceinline__ void decode_block(const DecodeArgs& a, uint64_t s, BlockDecodeLds& lds, uint32_t lane)
{
    constexpr uint32_t HDR = (FMT == FMT_V2) ? 14u : 13u;
    const uint64_t ssz64 = a.slots.size(s);
    const uint64_t* const image = a.words + a.slots.begin(s);
    uint64_t nw64 = a.in_nwords ? a.in_nwords[s] : ssz64;
    if (nw64 > ssz64) nw64 = ssz64;                     // an image cannot be longer than its slot
    const uint32_t nw = nw64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nw64;
    const uint64_t p0 = a.pts.begin(s);
    const uint64_t cap64 = a.pts.size(s);
    const uint32_t cap = cap64 > 0x7FFFFF00ull ? 0x7FFFFF00u : (uint32_t)cap64;   // as in the lane-per-series kernels
    const uint64_t limit = (uint64_t)nw * 64;           // a field must end before this bit (G/LongArrayInput.java:48-52)

    // ---- state at the last point handed to the output ("committed"): where the serial decoder takes over
    uint64_t pos = 0;                 // bit position of the next unread bit
    uint32_t n = 0;                   // points decoded
    int64_t stored_delta = 0;         // (the block timestamp is word 0 of the image whenever it is needed again)
    uint64_t stored_ts = 0, val = 0;
    uint32_t nbits = 0;               // value window: significant bits (0: none yet) and trailing zeros
    int32_t trail = 0;
    bool serial = nw < 4;             // tiny images: the serial decoder from the start
    bool finished = false;

    // ---- the image through LDS: ring[w % RING_WORDS_D] holds word w for w in [filled - RING_WORDS_D, filled);
    // words beyond the image read as zero
    uint32_t filled = 0;
    uint64_t inflight = 0;            // the next 64 words, requested one refill ahead
#define GTSC_RING_REQUEST(W0) { const uint64_t w_ = (uint64_t)(W0) + lane; inflight = w_ < nw ? image[w_] : 0ull; }
#define GTSC_RING_LAND() { lds.ring[(filled + lane) & (RING_WORDS_D - 1)] = inflight; filled += 64; GTSC_RING_REQUEST(filled) }
    if (!serial) {
        GTSC_RING_REQUEST(0)
        GTSC_RING_LAND()
        GTSC_RING_LAND()
        wave_sync();
        // header and first point (G/GorillaDecompressor.java:27-35,62-72): nw >= 4, so neither runs off the image
        const uint64_t w0 = lds.ring[0], w1 = lds.ring[1], w2 = lds.ring[2];
        const uint32_t d = (uint32_t)(w1 >> 37);
        if (d == 0x7FFFFFFu) {
            finished = true;                             // empty block
        } else {
            val = (w1 << 27) | (w2 >> 37);
            stored_delta = (int64_t)d;
            stored_ts = w0 + d;
            pos = 64 + 91;
            n = 1;
            if (cap == 0 || stored_ts == 0) { serial = true; pos = 64; n = 0; stored_ts = 0; stored_delta = 0; val = 0; }
            else if (lane == 0) { a.ts[p0] = (int64_t)stored_ts; a.val[p0] = val; }
        }
    }

    uint32_t qn = 0;                  // queued points
    uint64_t cur = pos;               // walk position (pos is the committed one, where queue entry 0 starts)
    uint32_t wnb = nbits; int32_t wtr = trail;   // window at the walk position
    uint32_t last_len = 0;            // length of the last point the walk passed
    bool last_reuse = false;          // it reused the value window ('10')
    uint32_t last_found = 0;          // points the last step found, if it went by strides
    uint32_t hold = 0;                // steps for which sparse windows stay with bit positions
    bool stop = finished || serial;   // the walk has met the end of the stream or something for the serial decoder
    bool clean_end = finished;

    while (!(stop && qn == 0)) {
        if (!stop) {
            // ---- one step: 64 candidate starts.  Sparse streams: the 64 bit positions from cur on.
            // Streams of equal-sized points: the positions cur + j * stride -- while the points that
            // follow have the predicted form, all 64 candidates are real starts.
            // The stride is that of a point with an unchanged delta and a reused value window
            // ('0', '10', payload): what a dense stream mostly consists of.  A step that ends at a point
            // of another kind (an unchanged value, a new window, another timestamp bucket) takes that
            // point along and the next step predicts again -- as long as such steps find points at all.
            // Windows under 29 bits: 64 bit positions hold several points, so strides are only taken
            // while they find more than that (a counter's stream -- '10' and a payload of equal size at
            // every point -- goes 64 points a step; one with unchanged values mixed in does not).
            // After a point of the predicted kind the stride is that point's length (whatever its timestamp
            // bucket), after any other 3 + window bits.
            const uint32_t pstride = last_reuse ? last_len : 3u + wnb;   // (found * pstride > 64: more points than 64 bit positions hold)
            const bool strided = wnb == 0   ? false
                                 : wnb >= 29 ? (last_len >= 32 || last_found >= 2)
                                             : (hold == 0 && (last_len == pstride || last_found * pstride > 64u));
            const uint32_t stride = strided ? pstride : 1u;
            while ((uint64_t)filled * 64 < cur + 64 * (uint64_t)stride + 64 * 4) { GTSC_RING_LAND() }
            wave_sync();
            const uint64_t b = cur + (uint64_t)lane * stride;
            const uint32_t k = (uint32_t)(b >> 6), o = (uint32_t)b & 63u;
            const uint64_t r0 = lds.ring[k & (RING_WORDS_D - 1)], r1 = lds.ring[(k + 1) & (RING_WORDS_D - 1)];
            const uint64_t hi = WindowDecoder<FMT>::window(r0, r1, o);
            const uint32_t top = (uint32_t)(hi >> 32);
            uint32_t ones = (uint32_t)__clz((int)~top);
            ones = ones > 4 ? 4u : ones;
            constexpr uint32_t CLEN = 1u | (9u << 6) | (12u << 12) | (16u << 18) | (36u << 24);
            const uint32_t clen = (CLEN >> (ones * 6u)) & 63u;
            const bool eos = ones == 4 && (uint32_t)(hi >> 28) == 0xFFFFFFFFu;
            const uint32_t xh = (uint32_t)((hi << clen) >> 32);
            const uint32_t vtype = xh >= 0xC0000000u ? 2u : (xh >> 31);   // 0: '0', 1: '10', 2: '11'
            uint32_t own_sig; int32_t own_trail;
            if (FMT == FMT_V2) {
                const uint32_t l = (xh >> 24) & 63u;
                own_sig = ((xh >> 18) & 63u) + 1u;
                own_trail = 64 - (int32_t)own_sig - (int32_t)l;
            } else {
                const uint32_t l = (xh >> 25) & 31u;
                own_sig = (xh >> 19) & 63u;
                own_sig = own_sig ? own_sig : 64u;
                own_trail = 64 - (int32_t)own_sig - (int32_t)l;
            }
            // length of a point starting here, given the window size in force
            const uint32_t fixed = clen + (vtype == 0 ? 1u : vtype == 1 ? 2u : HDR + own_sig);
            uint32_t len = fixed + (vtype == 1 ? wnb : 0u);
            uint32_t lane_nb = wnb; int32_t lane_tr = wtr;   // the window in force at this candidate
            uint64_t real = 0;
            uint32_t found = 0;
            uint64_t adv = 0;                                // bits the walk moves on by
            if (strided) {
                // the chain holds while every candidate has the predicted length and opens no window
                const bool ok = !eos && vtype != 2 && !(vtype == 1 && wnb == 0) && len == stride && b + len < limit && n + qn + lane < cap;
                const uint64_t bad = lanes(!ok);
                const uint32_t js = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
                found = js;
                adv = (uint64_t)js * stride;
                if (js < 64) {
                    // the first candidate that differs starts where predicted, but is another kind of point
                    const uint64_t here = cur + adv;
                    const uint32_t l = read_lane(len, js);
                    if (read_lane((uint32_t)eos, js)) {
                        stop = true;
                        clean_end = here + 36 < limit;
                    } else if ((read_lane(vtype, js) == 1 && wnb == 0) || here + l >= limit || n + qn + js >= cap) {
                        stop = true;   // a window reused before any was defined, the end of the image, or of the capacity
                    } else {
                        found = js + 1;
                        adv += l;
                        last_len = l;
                        last_reuse = read_lane(vtype, js) == 1;
                        if (read_lane(vtype, js) == 2) {
                            wnb = read_lane(own_sig, js);
                            wtr = (int32_t)read_lane((uint32_t)own_trail, js);
                        }
                    }
                }
                real = found >= 64 ? ~0ull : (1ull << found) - 1ull;
                last_found = found;
                if (pstride < 32u && found * pstride <= 64u) hold = 4;   // not worth it here: bit positions for the next steps
            } else {
            last_found = 0;
            hold -= hold ? 1u : 0u;
            const uint64_t span = read_lane(hi, 0);          // the 64 bits at cur
            uint32_t p = 0;
            uint32_t packed = len | (vtype << 8);            // what a hop needs to know about a candidate
            // Far from the end of the image and of the capacity, with a window defined and no end marker
            // among the candidates, a hop has nothing to check.
            const bool safe = wnb != 0 && !any_lane(eos) && cur + 64 + 128 < limit && n + qn + 64 < cap;
            auto walk = [&](auto safe_tag) {
                constexpr bool SAFE = decltype(safe_tag)::value;
                while (p < 64) {
                    const uint64_t here = cur + p;
                    // runs of unchanged points: '00' each
                    const uint64_t rest = span << p;
                    uint32_t zpairs = (rest ? (uint32_t)__builtin_clzll(rest) : 64u - p) >> 1;
                    if (zpairs) {
                        bool cut = false;
                        if (!SAFE) {
                            const uint64_t room = limit > here ? (limit - here - 1) >> 1 : 0;   // points of 2 bits that end before the limit
                            const uint32_t left = cap - n - qn - found;                        // and fit the capacity
                            if (zpairs > room) { zpairs = (uint32_t)room; cut = true; }
                            if (zpairs > left) { zpairs = (uint32_t)left; cut = true; }
                        }
                        if (zpairs) {
                            const uint64_t m = (0x5555555555555555ull >> (64 - 2 * zpairs)) << p;   // bits p, p+2, ...
                            real |= m;
                            found += zpairs;
                            p += 2 * zpairs;
                            last_len = 2;
                            last_reuse = false;
                        }
                        if (cut) { stop = true; break; }
                        continue;
                    }
                    const uint32_t pk = read_lane(packed, p);
                    const uint32_t l = pk & 255u, ty = pk >> 8;
                    if (!SAFE) {
                        if (read_lane((uint32_t)eos, p)) {
                            stop = true;
                            clean_end = here + 36 < limit;   // the marker itself must not run off the image
                            break;
                        }
                        if ((ty == 1 && wnb == 0) || here + l >= limit || n + qn + found >= cap) { stop = true; break; }
                    }
                    real |= 1ull << p;
                    found++;
                    if (ty == 2) {
                        // a new window from here on: candidates that reuse the window change length
                        wnb = read_lane(own_sig, p);
                        wtr = (int32_t)read_lane((uint32_t)own_trail, p);
                        len = fixed + (vtype == 1 ? wnb : 0u);
                        packed = len | (vtype << 8);
                        if (lane > p) { lane_nb = wnb; lane_tr = wtr; }
                    }
                    p += l;
                    last_len = l;
                    last_reuse = ty == 1;
                }
            };
            if (safe) {
                // Every candidate works out its own hop beforehand -- a run of unchanged points ('00' each,
                // counted inside the 64 positions) or its length -- and the starts it stands for, so that a
                // hop of the scalar chain is three lane reads, an OR and an add.
                const uint32_t zbits = hi ? (uint32_t)__builtin_clzll(hi) : 64u;
                const uint32_t zp = (zbits < 64u - lane ? zbits : 64u - lane) >> 1;
                const uint64_t own = zp ? (0x5555555555555555ull >> (64u - 2u * zp)) << lane : 1ull << lane;
                uint32_t hop = zp ? (2u * zp) | 0x200u : len | (vtype == 2 ? 0x100u : vtype == 1 ? 0x400u : 0u);
                uint32_t k = 0;
                while (p < 64) {
                    k = read_lane(hop, p);
                    real |= read_lane(own, p);
                    if (k & 0x100u) {
                        // a new window from here on: candidates that reuse the window change length
                        wnb = read_lane(own_sig, p);
                        wtr = (int32_t)read_lane((uint32_t)own_trail, p);
                        len = fixed + (vtype == 1 ? wnb : 0u);
                        hop = zp ? hop : len | (vtype == 2 ? 0x100u : vtype == 1 ? 0x400u : 0u);
                        if (lane > p) { lane_nb = wnb; lane_tr = wtr; }
                    }
                    p += k & 255u;
                }
                found = (uint32_t)__builtin_popcountll(real);
                last_len = (k & 0x200u) ? 2u : (k & 255u);
                last_reuse = (k & 0x400u) != 0;
            } else {
                walk(std::false_type{});
            }
            adv = p;
            }
            // queue the real starts
            if ((real >> lane) & 1u) {
                const uint32_t at = qn + (uint32_t)__builtin_popcountll(real & ((1ull << lane) - 1ull));
                lds.qpos[at] = (uint32_t)(b - pos);
                lds.qwin[at] = lane_nb | (((uint32_t)lane_tr & 127u) << 8);
            }
            qn += found;
            cur += adv;
            wave_sync();
        }
        if (qn >= 64 || (stop && qn)) {
            // ---- emit up to 64 queued points
            const uint32_t m = qn < 64 ? qn : 64u;
            const bool have = lane < m;
            const uint32_t rel = have ? lds.qpos[lane] : 0u;
            const uint32_t win = have ? lds.qwin[lane] : 0u;
            const uint64_t b = pos + rel;
            const uint32_t k = (uint32_t)(b >> 6), o = (uint32_t)b & 63u;
            const uint64_t r0 = lds.ring[k & (RING_WORDS_D - 1)], r1 = lds.ring[(k + 1) & (RING_WORDS_D - 1)],
                           r2 = lds.ring[(k + 2) & (RING_WORDS_D - 1)];
            WindowDecoder<FMT> d;
            d.off = 0; d.widx = 0; d.nwords = 0xFFFFFFFFu;
            d.nbits = win & 255u;
            d.trail = (int32_t)((win >> 8) & 127u);
            d.trail = (d.trail ^ 64) - 64;   // sign-extend 7 bits
            const uint64_t hi = WindowDecoder<FMT>::window(r0, r1, o), lo = WindowDecoder<FMT>::window(r1, r2, o);
            const typename WindowDecoder<FMT>::Parsed q = d.template parse<false>(hi, [lo] { return lo; });
            const uint64_t x = have ? q.bits << ((uint32_t)q.trail & 63u) : 0ull;
            const uint64_t dd = have ? (uint64_t)q.dd : 0ull;
            // timestamps: deltas are the running sum of the delta-of-deltas, timestamps of the deltas
            uint64_t delta, ts;
            if (!any_lane(dd != 0)) {
                // regular cadence throughout: no sums needed
                delta = (uint64_t)stored_delta;
                ts = stored_ts + (uint64_t)(lane + 1) * delta;
            } else {
                delta = (uint64_t)stored_delta + wave_inclusive_scan64<false>(dd);
                ts = stored_ts + wave_inclusive_scan64<false>(have ? delta : 0ull);
            }
            const uint64_t v = val ^ wave_inclusive_scan64<true>(x);
            // a timestamp of exactly zero turns the next point into a "first" one: not this path's business
            if (any_lane(have && ts == 0)) {
                serial = true; stop = true; qn = 0;
                break;
            }
            if (have) {
                __builtin_nontemporal_store((int64_t)ts, a.ts + p0 + n + lane);
                __builtin_nontemporal_store(v, a.val + p0 + n + lane);
            }
            // commit: the state after point m - 1
            const uint32_t lastl = m - 1;
            stored_delta = (int64_t)read_lane(delta, lastl);
            stored_ts = read_lane(ts, lastl);
            val = read_lane(v, lastl);
            nbits = read_lane(q.nbits, lastl);
            trail = (int32_t)read_lane((uint32_t)q.trail, lastl);
            const uint32_t endrel = read_lane(rel + q.o3, lastl);
            n += m;
            wave_sync();
            // the rest of the queue moves down; positions are counted from the new committed position
            const uint32_t remain = qn - m;
            uint32_t mp = 0, mw = 0;
            if (lane < remain) { mp = lds.qpos[64 + lane] - endrel; mw = lds.qwin[64 + lane]; }
            wave_sync();
            if (lane < remain) { lds.qpos[lane] = mp; lds.qwin[lane] = mw; }
            wave_sync();
            qn = remain;
            pos += endrel;
        }
    }
#undef GTSC_RING_REQUEST
#undef GTSC_RING_LAND
    uint32_t st = 0;
    if (!clean_end || serial) {
        // ---- the serial decoder, lane 0, from the committed state (exact order of the reference's checks)
        if (lane == 0) {
            auto src = [&](uint32_t i) -> uint64_t { return i < nw ? image[i] : 0ull; };
            Decoder<FMT> d;
            if (nw < 4 || pos < 64) {
                d.begin(nw, src);
            } else {
                d.nwords = nw;
                d.widx = (uint32_t)(pos >> 6); d.off = (uint32_t)pos & 63u;
                d.r0 = src(d.widx); d.r1 = src(d.widx + 1); d.r2 = src(d.widx + 2); d.r3 = src(d.widx + 3);
                d.st = 0;
                d.block_ts = (int64_t)src(0); d.stored_delta = stored_delta; d.stored_ts = stored_ts; d.val = val; d.pred_last = 0;
                d.lead = nbits ? (uint32_t)(64 - (int32_t)nbits - trail) : NO_WINDOW;
                d.trail = trail;
            }
            int64_t t; uint64_t v;
            while (d.next(t, v, src)) {
                if (n < cap) { a.ts[p0 + n] = t; a.val[p0 + n] = v; } else st = ST_OVERFLOW;
                n++;
            }
            st |= d.st;
        }
    }
    if (lane == 0) {
        a.out_npoints[s] = n;
        if (a.status) a.status[s] = st;
        if (st) atomicOr(a.flags, st);
    }
    wave_sync();
}

template <int FMT>
__global__ __launch_bounds__(64) void decode_block_kernel(DecodeArgs a, uint32_t per_wave)
{
    __shared__ BlockDecodeLds lds;
    const uint32_t lane = threadIdx.x;
    const uint64_t cand = blockIdx.x + (uint64_t)lane * gridDim.x;
    const bool mine = lane < per_wave && cand < a.nseries && block_owner<FMT>(a, cand) == OWNER_WAVE;
    uint64_t todo = lanes(mine);
    while (todo) {
        const uint32_t j = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        decode_block<FMT>(a, blockIdx.x + (uint64_t)j * gridDim.x, lds, lane);
    }
}

// The plan of a batch of images (DecodePlan): how many series are long and very long, how dense the long ones'
// images are, how dense every 64th image of the batch is, and how much the very long sparse ones come to.  All
// decompressors of the batch read it.
__global__ __launch_bounds__(256) void plan_decode_kernel(DecodeArgs a, DecodePlan* plan)
{
    // Every thread sums over its series, a workgroup adds its threads up through LDS, and one thread adds the six
    // sums to the plan: few atomics on the same six words however many series there are.
    __shared__ unsigned long long part[4][8], part2[4];
    unsigned long long acc[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long longest = 0, widest = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.nseries; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t cap = a.pts.size(s);
        const bool is_long = cap >= a.tun.long_points, is_vlong = cap >= a.tun.vlong_points;
        const bool sampled = (s & 63u) == 0;
        const uint64_t nw = (is_long || sampled) ? (a.in_nwords ? a.in_nwords[s] : a.slots.size(s)) : 0;
        if (is_long) { acc[0]++; acc[2] += nw; acc[3] += cap; widest = nw > widest ? nw : widest; }
        if (is_vlong) {
            acc[1]++;
            if (nw * 64 < cap * a.tun.wg_dense_bits) { acc[6] += cap; longest = cap > longest ? cap : longest; }
        }
        if (sampled) { acc[4] += nw; acc[5] += cap; }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int i = 0; i < 7; i++) {
        const uint64_t t = wave_inclusive_scan64<false>(acc[i]);
        if (lane == 63) part[wave][i] = t;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(longest >> 32), d, 64) << 32) |
                                     (uint32_t)__shfl_xor((int)(uint32_t)longest, d, 64);
        longest = o > longest ? o : longest;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(widest >> 32), d, 64) << 32) |
                                     (uint32_t)__shfl_xor((int)(uint32_t)widest, d, 64);
        widest = o > widest ? o : widest;
    }
    if (lane == 0) { part[wave][7] = longest; part2[wave] = widest; }
    __syncthreads();
    if (threadIdx.x < 7) {
        const unsigned long long t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (t) atomicAdd(reinterpret_cast<unsigned long long*>(plan) + threadIdx.x, t);
    } else if (threadIdx.x == 7) {
        unsigned long long t = part[0][7];
        for (int w = 1; w < 4; w++) t = part[w][7] > t ? part[w][7] : t;
        if (t) atomicMax(reinterpret_cast<unsigned long long*>(plan) + 7, t);
    } else if (threadIdx.x == 8) {
        unsigned long long t = part2[0];
        for (int w = 1; w < 4; w++) t = part2[w] > t ? part2[w] : t;
        if (t) atomicMax(reinterpret_cast<unsigned long long*>(plan) + 8, t);
    }
}

}  // namespace

hipError_t launch_encode_blocks(const EncodeArgs& a, int format, hipStream_t stream)
{
    if (a.nseries == 0) return hipSuccess;
    // So few blocks that a wave each would leave most SIMDs empty: a workgroup of sixteen waves each.  One such
    // workgroup keeps a compute unit busy, so the form pays up to about one and a half blocks per unit (a lone
    // 100,000-point block: 0.25 ms against 0.95; 1,024 of them: 1.0 ms either way; 7,200-point blocks break even
    // at about 400).
    if (a.nseries <= a.tun.coop_encode_max) {
        if (a.dispatch) *a.dispatch |= DISPATCH_COOP;
        if (format == FMT_V2) hipLaunchKernelGGL(encode_block_coop_kernel<FMT_V2>, dim3((uint32_t)a.nseries), dim3(64 * COOP_WAVES), 0, stream, a);
        else                  hipLaunchKernelGGL(encode_block_coop_kernel<FMT_V1>, dim3((uint32_t)a.nseries), dim3(64 * COOP_WAVES), 0, stream, a);
        return hipGetLastError();
    }
    // enough waves to fill the machine several times over when there are that many series
    if (a.dispatch) *a.dispatch |= DISPATCH_WAVE;
    uint64_t per_wave = a.nseries / 8192;
    per_wave = per_wave < 1 ? 1 : per_wave > 64 ? 64 : per_wave;
    const uint32_t grid = (uint32_t)((a.nseries + per_wave - 1) / per_wave);
    if (format == FMT_V2) hipLaunchKernelGGL(encode_block_kernel<FMT_V2>, dim3(grid), dim3(64), 0, stream, a, (uint32_t)per_wave);
    else                  hipLaunchKernelGGL(encode_block_kernel<FMT_V1>, dim3(grid), dim3(64), 0, stream, a, (uint32_t)per_wave);
    return hipGetLastError();
}

}  // namespace gtsc

namespace gtsc {

hipError_t launch_decode_plan(const DecodeArgs& a, hipStream_t stream)
{
    if (hipError_t e = hipMemsetAsync(a.plan_buf, 0, DecodePlan::HEADER_BYTES, stream)) return e;
    if (a.nseries == 0) return hipSuccess;
    if (a.dispatch) *a.dispatch |= DISPATCH_PLAN;
    const uint64_t groups = (a.nseries + 255) / 256;
    hipLaunchKernelGGL(plan_decode_kernel, dim3((uint32_t)(groups < 256 ? groups : 256)), dim3(256), 0, stream, a, a.plan_buf);
    return hipGetLastError();
}

hipError_t launch_decode_blocks(const DecodeArgs& a, int format, hipStream_t stream)
{
    if (a.nseries == 0) return hipSuccess;
    if (a.dispatch) *a.dispatch |= DISPATCH_WAVE;
    uint64_t per_wave = a.nseries / 8192;
    per_wave = per_wave < 1 ? 1 : per_wave > 64 ? 64 : per_wave;
    const uint32_t grid = (uint32_t)((a.nseries + per_wave - 1) / per_wave);
    if (format == FMT_V2) hipLaunchKernelGGL(decode_block_kernel<FMT_V2>, dim3(grid), dim3(64), 0, stream, a, (uint32_t)per_wave);
    else                  hipLaunchKernelGGL(decode_block_kernel<FMT_V1>, dim3(grid), dim3(64), 0, stream, a, (uint32_t)per_wave);
    return hipGetLastError();
}

}  // namespace gtsc
