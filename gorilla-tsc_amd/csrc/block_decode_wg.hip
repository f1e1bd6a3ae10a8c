// Workgroup-per-block form of the Gorilla decompressor on gfx950, for batches of few long blocks.
//
// Where a point starts is only known from the points before it (G/GorillaDecompressor.java:74-112,
// G/ValueDecompressor.java:32-54; G/ = src/main/java/fi/iki/yak/ts/compression/gorilla/ in the reference
// tree), so the chain of start positions is the serial part of a block.  A lone wavefront issues an
// instruction every five cycles at best: doing everything else in the same wave (wave-per-block kernel,
// block_codec.hip) leaves a long block slower than one host thread.  Here a block has a workgroup, and
// its waves split the work by kind:
//
//   * classifier waves work out, for EVERY bit position of the next 4,096 bits of the image, what a point
//     starting there would look like -- timestamp bucket, kind of value code, length apart from the
//     window bits of a '10' code, the window a '11' code declares, the run of '00' points that begins
//     there -- and leave one word per position in LDS.  None of that depends on what came before;
//   * one walker wave follows the chain through those words: a hop is a lane read and an add (a run of
//     unchanged points is one hop); where the points have the same length it tests 64 starts a stride
//     apart with one gathered read.  It queues the starts with the value window in force at each;
//   * emitter waves (the classifiers, once they are done) parse 64 queued points each, and prefix
//     sums / a prefix XOR over the wave and then over the batches turn delta-of-deltas and XOR payloads
//     into timestamps and values, stored as 512-byte runs.
//
// The three run as a pipeline over regions of 4,096 bit positions, one step ("tick") per region, with
// workgroup barriers in between: in tick k the walker is in region k while the others classify region
// k + 1 and emit what the walker queued in region k - 1.
//
// Anything out of the ordinary -- end of the image inside a point, a reused window that was never
// defined, a decoded timestamp of zero (which makes the reference read the next point as a "first"
// one), more points than the caller's capacity -- hands the block to one lane, which walks on from the
// last point stored with the serial decoder of the lane-per-series kernels, so that counts, status
// words and the order in which the reference's reads meet a problem are the same.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#include "codec_device.hpp"
#include "diag.hpp"   // measurement hooks: empty in the shipped library
#include "kernels.hpp"
#include "wave_util.hpp"

namespace gtsc {

namespace {

constexpr uint32_t RPOS = 4096;            // bit positions per region
constexpr uint32_t RW = RPOS / 64;         // image words per region
constexpr uint32_t RINGW = 8 * RW;         // image words staged in LDS (circular)
constexpr uint32_t QCAP = RPOS / 2;        // a region holds at most that many points ('00' each)
constexpr uint32_t MAXB = QCAP / 64;       // batches of 64 queued points per region

// What a point starting at a bit position looks like ("hop word"):
//   bits 0..8    where the chain lands from here, counted from the start of the position's row of 64: the
//                position's place in the row plus the point's length without the payload of a reused window
//                ('10' codes: + window bits) -- or, where a run of '00' points (regular interval, unchanged
//                value) begins, plus the bits of the whole run as far as it stays inside the row
//   bit 9        such a run;  bit 10 / 11: the value code is '10' / '11';  bit 12: the position holds the end marker
//   bits 13..19  significant bits, bits 20..26 trailing zeros (7 bits, signed) of a '11' code's window
//   bit 31       a run, a '11' code or an end marker: not a plain hop
constexpr uint32_t HW_RUN = 0x200u, HW_T1 = 0x400u, HW_T2 = 0x800u, HW_EOS = 0x1000u, HW_SPECIAL = 0x80000000u;
constexpr uint32_t HW_NEXT = 0x1FFu;
__device__ __forceinline__ uint32_t hw_sig(uint32_t hw) { return (hw >> 13) & 127u; }
__device__ __forceinline__ uint32_t hw_tr7(uint32_t hw) { return (hw >> 20) & 127u; }
// the length of the point itself at place `at` of its row (a position inside a run holds a '00' point)
__device__ __forceinline__ uint32_t hw_fixed(uint32_t hw, uint32_t at) { return (hw & HW_RUN) ? 2u : (hw & HW_NEXT) - at; }
__device__ __forceinline__ uint32_t hw_zpc(uint32_t hw, uint32_t at) { return (hw & HW_RUN) ? ((hw & HW_NEXT) - at) >> 1 : 0u; }

// `at`: the position's place in its row of 64 (a run is cut where the row ends)
template <int FMT>
__device__ __forceinline__ uint32_t classify_position(uint64_t hi, uint32_t at)
{
    constexpr uint32_t HDR = (FMT == FMT_V2) ? 14u : 13u;
    const uint32_t top = (uint32_t)(hi >> 32);
    uint32_t ones = (uint32_t)__clz((int)~top);  // 32 when all ones
    ones = ones > 4 ? 4u : ones;
    constexpr uint32_t CLEN = 1u | (9u << 6) | (12u << 12) | (16u << 18) | (36u << 24);
    const uint32_t clen = (CLEN >> (ones * 6u)) & 63u;
    const bool eos = top == 0xFFFFFFFFu && (uint32_t)hi >= 0xF0000000u;  // '1111' and 32 ones
    const uint32_t xh = (uint32_t)((hi << clen) >> 32);
    const bool t2 = xh >= 0xC0000000u;
    const bool t1 = !t2 && (int32_t)xh < 0;
    uint32_t lead, sig;
    if (FMT == FMT_V2) {
        lead = (xh >> 24) & 63u;
        sig = ((xh >> 18) & 63u) + 1u;
    } else {
        lead = (xh >> 25) & 31u;
        sig = (xh >> 19) & 63u;
        sig = sig ? sig : 64u;
    }
    const uint32_t trail = (64u - sig - lead) & 127u;
    const uint32_t fixed = clen + (t2 ? HDR + sig : t1 ? 2u : 1u);
    const uint32_t zbits = hi ? (uint32_t)__builtin_clzll(hi) : 64u;
    const uint32_t room = 64u - at;
    const uint32_t zp = (zbits < room ? zbits : room) >> 1;
    return (at + (zp ? 2u * zp : fixed)) | (zp ? HW_RUN : 0u) | (t1 ? HW_T1 : 0u) | (t2 ? HW_T2 : 0u) | (eos ? HW_EOS : 0u) | (sig << 13) |
           (trail << 20) | ((zp || t2 || eos) ? HW_SPECIAL : 0u);
}

struct WgLds {
    uint64_t ring[RINGW];          // ring[w % RINGW] = image word w (zero beyond the image)
    uint32_t hopw[2 * RPOS];       // hop word of bit position p at hopw[p % (2 RPOS)]
    uint32_t queue[2][QCAP];       // queued starts of a region: position in the region | window << 16
    uint64_t aggD[MAXB], aggS[MAXB], aggX[MAXB];  // per batch: sum of delta-of-deltas, sum of their running sums, XOR of the payloads
    uint32_t qcount[2];
    uint32_t kb[2];                // first batch that decoded a timestamp of zero (or none)
    uint32_t anydd[2];             // some point of the region changed the interval
    uint32_t stop, clean;          // the walker met the end of the stream or something for the serial decoder
    uint32_t endrel[2], endwin[2]; // where the last queued point of a region ends, and the window after it (by the region's parity:
                                   // a wave may still be reading one region's while the next one's is written)
    uint64_t sbits[RW];            // the walker's own: starts found by strides in the region under way (bit l of word r: position 64 r + l)
    uint8_t land[2 * RPOS];        // landing tables: land[p % (2 RPOS)] = where the chain that enters p's row at p leaves the row
    uint32_t landwnb[2];           // the window size (significant bits) the tables of a region were made for
    uint32_t curwnb;               // the walker's window size when it last reported
    uint32_t want_tables;          // ... and whether it has use for tables (not while strides carry it)
    uint32_t pad_[14];             // (a row's table is read one row ahead, past the last row too)
#ifndef GTSC_WG_SHARE_UNIT
    // More than half a compute unit's 160 KB, so that a workgroup has its unit to itself: two on one unit run at half
    // speed each, and a block is a chain -- what a batch waits for is its slowest block, not the sum.
    uint32_t alone_[1800];
#endif
};
static_assert(sizeof(WgLds) <= 160 * 1024, "more LDS than a compute unit has");

// Where the chain leaves a row, for each of the 64 places it may enter it -- worked out by the classifier waves for
// the rows they classified, with one pointer-doubling pass per row: lane l starts with where position l lands and
// replaces it six times by where that place lands, until it is outside the row (a hop is two bits at least, a run
// of '00' points is one hop: 32 hops at most).  A '10' code's length needs the window size, which the walker only
// knows for certain when it gets there: the tables are made for the size it last reported (`wnb`; 0: no window yet)
// and the walker uses them while that is the size in force.  LAND_STOP: the path from that place meets a '11' code,
// an end marker or a '10' code without a window -- the walker goes through such a row hop by hop.
constexpr uint32_t LAND_STOP = 255;
// the tables of rows [row0, row0 + 4) of a region (four chains of dependent lane reads in flight together)
__device__ __forceinline__ void make_landing_tables(WgLds& L, uint32_t region, uint32_t row0, uint32_t wnb, uint32_t lane)
{
    const uint32_t at = (region & 1u) * RPOS + row0 * 64 + lane;
    uint32_t n[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t hw = L.hopw[at + 64 * i];
        n[i] = (hw & HW_NEXT) + ((hw >> 10) & 1u) * wnb;   // < 255
        if ((hw & (HW_T2 | HW_EOS)) || ((hw & HW_T1) && wnb == 0)) n[i] = LAND_STOP;
    }
    // (six doublings at most: 64 hops; rows of few points are through after two or three)
    for (int step = 0; step < 6 && any_lane((n[0] & n[1] & n[2] & n[3]) < 64 || n[0] < 64 || n[1] < 64 || n[2] < 64 || n[3] < 64); step++) {
        uint32_t m[4];
#pragma unroll
        for (int i = 0; i < 4; i++) m[i] = (uint32_t)__shfl((int)n[i], (int)(n[i] & 63u), 64);
#pragma unroll
        for (int i = 0; i < 4; i++) n[i] = n[i] < 64 ? m[i] : n[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) L.land[at + 64 * i] = (uint8_t)n[i];
}

// Follows the chain through rows by their landing tables: one lane read per row.  Lane r of `entry` receives the
// place where the chain entered row r.  lbase: LDS byte address of this lane's entry in the table of row 0.
// Returns 0 with r >= stoprow (q: the place in that row), 1 at a row whose table says LAND_STOP for the place the
// chain enters it (r, q: that row and place; nothing recorded for it).
__device__ __forceinline__ uint32_t chase_rows(uint32_t& r, uint32_t& q, uint32_t& entry, uint32_t lbase, uint32_t stoprow)
{
    uint32_t code, t, adv, ja, jb, va;
    asm volatile(
        "v_lshl_add_u32 %[va], %[r], 6, %[lbase]\n\t"
        "ds_read_u8 %[ja], %[va]\n\t"
        "ds_read_u8 %[jb], %[va] offset:64\n\t"
        "s_waitcnt lgkmcnt(1)\n"
        "1:\n\t"
        "v_readlane_b32 %[t], %[ja], %[q]\n\t"
        "s_cmp_eq_u32 %[t], 255\n\t"
        "s_cbranch_scc1 9f\n\t"
        "s_mov_b32 m0, %[r]\n\t"
        "s_lshr_b32 %[adv], %[t], 6\n\t"
        "v_writelane_b32 %[entry], %[q], m0\n\t"
        "s_add_u32 %[r], %[r], %[adv]\n\t"
        "s_and_b32 %[q], %[t], 63\n\t"
        "s_cmp_ge_u32 %[r], %[stoprow]\n\t"
        "s_cbranch_scc1 8f\n\t"
        "s_cmp_eq_u32 %[adv], 1\n\t"
        "s_cbranch_scc0 5f\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[ja], %[jb]\n\t"
        "v_add_u32 %[va], 64, %[va]\n\t"
        "ds_read_u8 %[jb], %[va] offset:64\n\t"
        "s_branch 1b\n"
        "5:\n\t"   // the chain hopped over a whole row
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshl_add_u32 %[va], %[r], 6, %[lbase]\n\t"
        "ds_read_u8 %[ja], %[va]\n\t"
        "ds_read_u8 %[jb], %[va] offset:64\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "s_branch 1b\n"
        "8:\n\t"
        "s_mov_b32 %[code], 0\n\t"
        "s_branch 10f\n"
        "9:\n\t"
        "s_mov_b32 %[code], 1\n"
        "10:\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        : [code] "=&s"(code), [t] "=&s"(t), [adv] "=&s"(adv), [ja] "=&v"(ja), [jb] "=&v"(jb), [va] "=&v"(va),
          [r] "+s"(r), [q] "+s"(q), [entry] "+v"(entry)
        : [lbase] "v"(lbase), [stoprow] "s"(stoprow)
        : "scc", "m0", "memory");
    return code;
}


// A word of LDS that another wave wrote before the last barrier, as a scalar.
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ uint64_t lds_peek64(const uint64_t* p)
{
    const uint64_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
// All waves of the workgroup have carried out their LDS accesses up to here.  (Not __syncthreads(): that
// would also wait for global memory, and the stores of decoded points are meant to stay in flight.)
__device__ __forceinline__ void wg_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


__device__ __forceinline__ int32_t sext7(uint32_t x) { return (int32_t)((x & 127u) ^ 64u) - 64; }

// The walker's state: wave-uniform.
struct Walk {
    uint64_t cur;            // bit position where the next point starts
    uint32_t wnb;            // value window in force: significant bits (0: none yet) ...
    uint32_t wtr7;           // ... and trailing zeros, 7 bits
    uint32_t last_len;       // length of the last point passed
    bool last_t1;            // it reused the value window ('10')
    uint32_t cool;           // rows to pass before strides are tried again
    uint32_t found_total;    // points found so far, the block's first one included
    bool stop, clean_end;
    uint32_t backoff;        // rows to the stride attempt after the next one that fails
    bool last_known;         // last_len / last_t1 say what the last point was
    bool want_tables;        // the last region was not one that strides carried
    uint32_t pause_after;    // regions in a row that strides must carry before the walker asks for a pause (0: never)
    uint32_t carried;        // regions in a row that strides have carried
    bool pause;              // the walker asks for the pipeline to stop here so that stride_phase() takes over
    GTSC_WG_WALK_DECL
};

// Follows the chain of point starts through region k (positions [k RPOS, (k+1) RPOS)) from w.cur on and
// queues them behind the qn points already there, looking at everything the reference's reads could meet:
// the form for the ends of a block (its last regions, the points before the first value window, the last
// points before the capacity is full).  Returns the number of points queued.
__device__ __forceinline__ uint32_t walk_region_careful(Walk& w, WgLds& L, uint32_t k, uint64_t limit, uint32_t cap, uint32_t lane, uint32_t qn)
{
    uint32_t* const q = L.queue[k & 1u];
    const uint64_t g0 = (uint64_t)k * RPOS, rend = g0 + RPOS;
    while (!w.stop && w.cur < rend) {
        // ---- points of equal length: 64 candidate starts a stride apart, tested with one gathered read.
        // The stride is the last point's length when that point reused the value window ('0'/'10'/'110'...,
        // '10', payload): what a dense stream mostly consists of.
        if (w.last_t1 && w.cool == 0 && w.wnb != 0) {
            const uint32_t s = w.last_len;
            const uint64_t cand = w.cur + (uint64_t)lane * s;
            const bool inreg = cand < rend;
            const uint32_t hw = inreg ? L.hopw[(uint32_t)cand & (2 * RPOS - 1)] : (HW_EOS | 0u);
            const uint32_t len = hw_fixed(hw, (uint32_t)cand & 63u) + ((hw >> 10) & 1u) * w.wnb;
            const bool ok = inreg && (hw & (HW_T2 | HW_EOS)) == 0 && len == s && cand + len < limit && w.found_total + lane < cap;
            const uint64_t bad = lanes(!ok);
            const uint32_t js = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
            if (js >= 4) {
                if (lane < js) q[qn + lane] = (uint32_t)(cand - g0) | ((w.wnb | (w.wtr7 << 7)) << 16);
                qn += js;
                w.found_total += js;
                w.cur += (uint64_t)js * s;
                w.last_t1 = (read_lane(hw, js - 1) & HW_T1) != 0;
                continue;
            }
            // the chain broke right away (not merely at the region's end): hops for a while
            if (read_lane((uint32_t)inreg, js)) w.cool = 8;
        }
        // ---- one row of 64 positions: lane l looks at position row + l
        const uint64_t row = w.cur & ~63ull;
        uint32_t p = (uint32_t)w.cur & 63u;
        const uint32_t hw = L.hopw[((uint32_t)row & (2 * RPOS - 1)) + lane];
        const uint32_t t1 = (hw >> 10) & 1u;
        uint32_t len = hw_fixed(hw, lane) + t1 * w.wnb;
        const uint32_t zpc = hw_zpc(hw, lane);            // '00' points from here on, inside the row
        const uint32_t sig = hw_sig(hw), tr7 = hw_tr7(hw);
        uint32_t lane_win = w.wnb | (w.wtr7 << 7);        // the window in force at this lane's position
        uint64_t real = 0;
        uint32_t found = 0;
        // Far from the end of the image and of the capacity, with a window defined and no end marker in the
        // row, a hop has nothing to check.
        const bool safe = w.wnb != 0 && !any_lane((hw & HW_EOS) != 0) && row + 64 + 128 < limit && w.found_total + 64 < cap;
        if (safe) {
            // every position knows its own hop -- a run of unchanged points or its length -- and the starts it
            // stands for, so that a hop of the scalar chain is three lane reads, an OR and an add
            const uint64_t own = zpc ? (0x5555555555555555ull >> (64u - 2u * zpc)) << lane : 1ull << lane;
            const uint32_t flags = hw & (HW_T1 | HW_T2);   // 0x400: '10', 0x800: '11'
            uint32_t hop = zpc ? (2u * zpc) | 0x200u : len | flags;
            uint32_t kk = 0;
            while (p < 64) {
                kk = read_lane(hop, p);
                real |= read_lane(own, p);
                if (kk & 0x800u) {
                    // a new window from here on: positions that reuse the window change length
                    w.wnb = read_lane(sig, p);
                    w.wtr7 = read_lane(tr7, p);
                    len = hw_fixed(hw, lane) + t1 * w.wnb;
                    hop = zpc ? hop : len | flags;
                    if (lane > p) lane_win = w.wnb | (w.wtr7 << 7);
                }
                p += kk & 255u;
            }
            found = (uint32_t)__builtin_popcountll(real);
            w.last_len = (kk & 0x200u) ? 2u : (kk & 255u);
            w.last_t1 = (kk & 0x400u) != 0;
        } else {
            // near an end, or before the first window: every hop looks at what the reference's reads would meet
            uint32_t packed = len | (hw & (HW_T1 | HW_T2 | HW_EOS)) >> 2;   // 0x100: '10', 0x200: '11', 0x400: end marker
            while (p < 64) {
                const uint64_t here = row + p;
                uint32_t zz = read_lane(zpc, p);
                if (zz) {
                    bool cut = false;
                    const uint64_t room = limit > here ? (limit - here - 1) >> 1 : 0;   // points of 2 bits that end before the limit
                    const uint32_t left = cap - w.found_total - found;                  // and fit the capacity
                    if (zz > room) { zz = (uint32_t)room; cut = true; }
                    if (zz > left) { zz = left; cut = true; }
                    if (zz) {
                        real |= (0x5555555555555555ull >> (64u - 2u * zz)) << p;
                        found += zz;
                        p += 2u * zz;
                        w.last_len = 2;
                        w.last_t1 = false;
                    }
                    if (cut) { w.stop = true; break; }
                    continue;
                }
                const uint32_t pk = read_lane(packed, p);
                const uint32_t l = pk & 255u;
                if (pk & 0x400u) {
                    w.stop = true;
                    w.clean_end = here + 36 < limit;   // the marker itself must not run off the image
                    break;
                }
                if (((pk & 0x100u) && w.wnb == 0) || here + l >= limit || w.found_total + found >= cap) { w.stop = true; break; }
                real |= 1ull << p;
                found++;
                if (pk & 0x200u) {
                    w.wnb = read_lane(sig, p);
                    w.wtr7 = read_lane(tr7, p);
                    len = hw_fixed(hw, lane) + t1 * w.wnb;
                    packed = len | (hw & (HW_T1 | HW_T2 | HW_EOS)) >> 2;
                    if (lane > p) lane_win = w.wnb | (w.wtr7 << 7);
                }
                p += l;
                w.last_len = l;
                w.last_t1 = (pk & 0x100u) != 0;
            }
        }
        // queue the real starts
        if ((real >> lane) & 1u) {
            const uint32_t at = qn + (uint32_t)__builtin_popcountll(real & ((1ull << lane) - 1ull));
            q[at] = ((uint32_t)(row - g0) + lane) | (lane_win << 16);
        }
        qn += found;
        w.found_total += found;
        w.cur = row + p;
        w.cool -= w.cool ? 1u : 0u;
    }
    return qn;
}

// The chain of hops through the rows of a region that lies well inside the image (no field of a point that starts
// in it can run off the image; the capacity is the caller's to check afterwards), as one piece of assembly: the compiler's own
// code for such a loop costs ten taken branches a row, and a lone wave pays a refill of its instruction buffer for
// each.  Here a plain hop is a lane read, a bit set and a compare, every second hop a taken branch, and a row's end
// or a '11' code one more each.
//
// State: row r (< 64) and position q in it; `real`: the starts found in the row so far; lane l of `nxt` holds the
// place in the row (64 and more: in the rows after it) where the chain lands from position l, or something above
// 511 where position l is not a plain hop: a run of '00' points, a '11' code, an end marker -- or a '10' code while
// there is no window for it to reuse.  Lane r of blo/bhi receives the finished mask of row r and lane r of rowwin
// the window in force where the row begins.  base: LDS byte address of this lane's hop word in row 0.
// win: the value window in force (significant bits | trailing zeros << 7; 0: none yet); a '11' code replaces it and
// with it the length of the positions that reuse the window.
// Starts at (r, q) with a fresh row.  Returns 0 with r >= stoprow (the region's end, or where the caller wants a
// word), the rows' masks stored and kk = the last hop (its length | HW_T1 or HW_RUN); 2 at an end marker or a '10'
// code without a window (r/q/real say where and what the row had before it): the careful walk's.
__device__ __forceinline__ uint32_t walk_rows(uint32_t& r, uint32_t& q, uint64_t& real, uint32_t& kk, uint32_t stoprow,
                                              uint32_t& win, uint32_t& blo, uint32_t& bhi, uint32_t& rowwin, uint32_t base)
{
    uint32_t code, t, q2, len, lastpos, wnbx, vt, hw, hwn, nxt, va;
    uint32_t rlo = (uint32_t)real, rhi = (uint32_t)(real >> 32);
    const uint64_t c5 = 0x5555555555555555ull;
    asm volatile(
        "s_mov_b32 s96, %[rlo]\n\t"
        "s_mov_b32 s97, %[rhi]\n\t"
        "s_and_b32 %[wnbx], %[win], 127\n\t"
        "s_cmp_eq_u32 %[wnbx], 0\n\t"
        "s_cselect_b32 %[wnbx], 0x800000, %[wnbx]\n\t"   // no window yet: a '10' code lands nowhere
        // ---- a fresh row: its hop words and those of the row after it
        "v_lshl_add_u32 %[va], %[r], 8, %[base]\n\t"
        "ds_read_b32 %[hw], %[va]\n\t"
        "v_add_u32 %[va], 0x100, %[va]\n\t"
        "ds_read_b32 %[hwn], %[va]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "s_branch 1f\n"
        // (The blocks are laid out so that the common ways round -- a row done, a '11' code -- pay one taken branch
        // each: a lone wave refills its instruction buffer for every one of them.)
        "5:\n\t"   // the chain hopped over a whole row
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshl_add_u32 %[va], %[r], 8, %[base]\n\t"
        "ds_read_b32 %[hw], %[va]\n\t"
        "v_add_u32 %[va], 0x100, %[va]\n\t"
        "ds_read_b32 %[hwn], %[va]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "s_branch 1f\n"
        // ---- the row is done: the chain went from q to q2 >= 64
        "30:\n\t"
        "s_sub_u32 %[len], %[q2], %[q]\n\t"
        "s_mov_b32 %[lastpos], %[q]\n\t"
        "v_writelane_b32 %[blo], s96, m0\n\t"
        "v_writelane_b32 %[bhi], s97, m0\n\t"
        "s_mov_b64 s[96:97], 0\n\t"
        "s_lshr_b32 %[t], %[q2], 6\n\t"
        "s_add_u32 %[r], %[r], %[t]\n\t"
        "s_and_b32 %[q], %[q2], 63\n\t"
        "s_cmp_ge_u32 %[r], %[stoprow]\n\t"
        "s_cbranch_scc1 8f\n\t"
        "s_cmp_eq_u32 %[t], 1\n\t"
        "s_cbranch_scc0 5b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[hw], %[hwn]\n\t"
        "v_add_u32 %[va], 0x100, %[va]\n\t"
        "ds_read_b32 %[hwn], %[va]\n"
        "1:\n\t"
        "s_mov_b32 m0, %[r]\n\t"
        "v_bfe_u32 %[vt], %[hw], 10, 1\n\t"
        "v_and_b32 %[nxt], 0x800001ff, %[hw]\n\t"
        "v_writelane_b32 %[rowwin], %[win], m0\n\t"
        "v_mad_u32_u24 %[nxt], %[vt], %[wnbx], %[nxt]\n\t"
        "s_nop 1\n"
        // ---- plain hops, two to a round (a lane read with a lane number that a lane read produced needs four
        // instructions in between)
        "2:\n\t"
        "v_readlane_b32 %[q2], %[nxt], %[q]\n\t"
        "s_bitset1_b64 s[96:97], %[q]\n\t"
        "s_cmp_lt_u32 %[q2], 64\n\t"
        "s_cbranch_scc0 20f\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[q], %[nxt], %[q2]\n\t"
        "s_bitset1_b64 s[96:97], %[q2]\n\t"
        "s_cmp_lt_u32 %[q], 64\n\t"
        "s_nop 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "s_mov_b32 %[t], %[q]\n\t"      // the same way out as the first half: from q to q2
        "s_mov_b32 %[q], %[q2]\n\t"
        "s_mov_b32 %[q2], %[t]\n"
        "20:\n\t"
        "s_cmp_lt_u32 %[q2], 512\n\t"
        "s_cbranch_scc1 30b\n\t"
        // ---- position q is not a plain hop
        "v_readlane_b32 %[kk], %[hw], %[q]\n\t"
        "s_bitcmp1_b32 %[kk], 9\n\t"
        "s_cbranch_scc1 23f\n\t"
        "s_and_b32 %[t], %[kk], 0x1400\n\t"      // an end marker, or a '10' code with no window to reuse: not this loop's
        "s_cbranch_scc1 29f\n\t"
        // a '11' code: the window it declares is in force from the next point on, and the positions of the row that
        // reuse the window change length
        "s_lshr_b32 %[t], %[kk], 13\n\t"
        "s_and_b32 %[win], %[t], 0x3fff\n\t"
        "s_and_b32 %[wnbx], %[win], 127\n\t"
        "v_and_b32 %[nxt], 0x800001ff, %[hw]\n\t"
        "s_and_b32 %[q2], %[kk], 0x1ff\n\t"
        "v_mad_u32_u24 %[nxt], %[vt], %[wnbx], %[nxt]\n\t"
        "s_cmp_lt_u32 %[q2], 64\n\t"
        "s_cbranch_scc0 30b\n\t"
        "s_mov_b32 %[q], %[q2]\n\t"
        "s_branch 2b\n"
        "23:\n\t"
        "s_and_b32 %[q2], %[kk], 0x1ff\n\t"      // a run of '00' points: bits q, q + 2, ... up to where it lands
        "s_sub_u32 %[t], %[q2], %[q]\n\t"
        "s_sub_u32 %[t], 64, %[t]\n\t"
        "s_lshr_b64 s[98:99], %[c5], %[t]\n\t"
        "s_lshl_b64 s[98:99], s[98:99], %[q]\n\t"
        "s_or_b64 s[96:97], s[96:97], s[98:99]\n\t"
        "s_sub_u32 %[q], %[q2], 2\n\t"           // (the run's last point)
        "s_cmp_lt_u32 %[q2], 64\n\t"
        "s_cbranch_scc0 30b\n\t"
        "s_mov_b32 %[q], %[q2]\n\t"
        "s_branch 2b\n"
        "29:\n\t"
        "s_bitset0_b64 s[96:97], %[q]\n\t"
        "s_mov_b32 %[kk], 2\n\t"
        "s_mov_b32 %[code], 2\n\t"
        "s_branch 10f\n"
        "8:\n\t"   // the last hop, for the caller: its length and whether it was a '10' code or part of a run
        "v_readlane_b32 %[t], %[hw], %[lastpos]\n\t"
        "s_and_b32 %[t], %[t], 0x600\n\t"
        "s_or_b32 %[kk], %[len], %[t]\n\t"
        "s_mov_b32 %[code], 0\n"
        "10:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %[rlo], s96\n\t"
        "s_mov_b32 %[rhi], s97\n"
        : [code] "=&s"(code), [t] "=&s"(t), [q2] "=&s"(q2), [len] "=&s"(len), [lastpos] "=&s"(lastpos), [wnbx] "=&s"(wnbx),
          [vt] "=&v"(vt), [hw] "=&v"(hw), [hwn] "=&v"(hwn), [nxt] "=&v"(nxt), [va] "=&v"(va),
          [kk] "+s"(kk), [r] "+s"(r), [q] "+s"(q), [win] "+s"(win), [rlo] "+s"(rlo), [rhi] "+s"(rhi),
          [blo] "+v"(blo), [bhi] "+v"(bhi), [rowwin] "+v"(rowwin)
        : [stoprow] "s"(stoprow), [c5] "s"(c5), [base] "v"(base)
        : "scc", "m0", "s96", "s97", "s98", "s99", "memory");
    real = ((uint64_t)rhi << 32) | rlo;
    return code;
}

// A region that lies well inside the image: nothing to check but the end marker and '10' codes before the first
// window.  (The capacity is the caller's to check afterwards: w.found_total.)  Three ways through it:
// chase_rows() while the landing tables fit the window in force, a lane read per row -- the starts inside those rows
// are found afterwards, lane r walking row r from where the chain entered it, all rows at once; walk_rows(), hop by
// hop, through rows whose table says stop and wherever the tables were made for another window size; and strides:
// where the last point reused the window, candidate starts that point's length apart are tested with gathered
// reads, up to 256 at a time -- a stream of long equal-sized points is taken a region at a time.  The queue is
// written once per region, lane r handing over the starts of row r in order with the window in force at each (a
// '11' point changes it for the points after it).
// Returns the number of points queued; `handover`: the walk stopped before the region's end (w.cur says where)
// and walk_region_careful() has the rest.
__device__ __forceinline__ uint32_t walk_region_fast(Walk& w, WgLds& L, uint32_t k, uint32_t lane, bool& handover)
{
    uint32_t* const qd = L.queue[k & 1u];
    const uint32_t* const hrow = L.hopw + (k & 1u) * RPOS + lane;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)hrow;
    uint32_t p = (uint32_t)w.cur & (RPOS - 1);     // position inside the region
    uint32_t win = w.wnb | (w.wtr7 << 7);          // the value window in force: significant bits | trailing zeros << 7
    uint32_t kk = w.last_t1 ? (w.last_len | HW_T1) : w.last_len;   // the last hop
    uint32_t blo = 0, bhi = 0, rowwin = 0;         // lane r: the starts of row r, the window in force where the row begins
    uint32_t next_try = (p >> 6) + w.cool;         // the row from which strides are tried again
    uint32_t backoff = w.backoff;                  // rows to the next attempt after one that failed (8 ... 1024)
    bool kk_known = w.last_known;                  // kk says what the last hop was (chase_rows() does not look)
    bool strided = false;
    uint32_t stride_points = 0;
    uint32_t lastgen = ~0u;                        // the last row the chain went through hop by hop
    handover = false;
    uint32_t r = p >> 6, q = p & 63u;
    uint64_t real = 0;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)(L.land + (k & 1u) * RPOS + lane);
    const uint32_t table_wnb = lds_peek(&L.landwnb[k & 1u]);
    uint32_t entry = ~0u;                          // lane r: where the chain entered row r by its landing table (none: ~0)
    GTSC_WG_WALK_BEGIN(w)
    // hops through rows [r, stop): false when the careful walk has to take over
    auto hop_rows = [&](uint32_t stop) -> bool {
        const uint32_t code = walk_rows(r, q, real, kk, stop, win, blo, bhi, rowwin, base);
        GTSC_WG_WALK_COUNT(w, tw_calls)
        GTSC_WG_WALK_TICK(w, tw_rows)
        kk_known = true;
        if (code == 2) {
            // what looks like the end marker, or a '10' code before any window (the starts of the row so far go along)
            blo = write_lane(blo, (uint32_t)real, r);
            bhi = write_lane(bhi, (uint32_t)(real >> 32), r);
            handover = true;
            return false;
        }
        lastgen = r - 1;
        return true;
    };
    while (r < RW) {
        if (r >= next_try) {
            // ---- time to see whether the points have become equal in length (once in a while: an attempt costs as
            // much as twenty rows by table)
            if (!kk_known) {
                if (!hop_rows(r + 1)) break;
                continue;
            }
            bool moved = false, broke_inside = true;
            if ((kk & HW_T1) && !(kk & HW_RUN)) {
                // candidate starts a stride apart, 64 per gathered read and as many reads as the rest of the region
                // has room for (all in flight together)
                const uint32_t s = kk & 255u, start = r * 64 + q;
                const uint32_t groups = (RPOS - start + 64 * s - 1) / (64 * s);   // >= 1
                const uint32_t G = groups < 4 ? groups : 4u;
                uint32_t hc[4];
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    const uint32_t cand = start + (lane + 64 * g) * s;
                    hc[g] = (g < G && cand < RPOS) ? hrow[(int32_t)cand - (int32_t)lane] : HW_EOS;
                }
                uint32_t js = 0, last_hc = 0;
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) {
                    if (g < G && js == 64 * g) {
                        const uint32_t cand = start + (lane + 64 * g) * s;
                        const uint32_t lc = hw_fixed(hc[g], cand & 63u) + ((hc[g] >> 10) & 1u) * (win & 127u);
                        const bool ok = (hc[g] & (HW_T2 | HW_EOS)) == 0 && lc == s;
                        const uint64_t bad = lanes(!ok);
                        const uint32_t jg = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
                        if (lane < jg && (g > 0 || jg >= 4)) atomicOr(reinterpret_cast<unsigned long long*>(&L.sbits[cand >> 6]), 1ull << (cand & 63u));
                        if (jg) last_hc = read_lane(hc[g], jg - 1);
                        if (jg < 64) broke_inside = read_lane((uint32_t)(cand < RPOS), jg) != 0;
                        js += jg;
                    }
                }
                GTSC_WG_WALK_TICK(w, tw_stride)
                if (js >= 4) {
                    // (rows the stride passes through get the window in force: it does not change on the way; row r
                    // has its own from the hops that came through it, if any did)
                    if (lane > r || (lane == r && lastgen != r)) rowwin = win;
                    strided = true;
                    stride_points += js;
                    const uint32_t pn = start + js * s;
                    r = pn >> 6; q = pn & 63u;
                    kk = (last_hc & HW_T1) ? (s | HW_T1) : s;
                    moved = true;
                    GTSC_WG_WALK_COUNT(w, tw_strides)
                    if (js >= 32 || !broke_inside) { backoff = 8; continue; }   // worth it (or cut short by the region's end): again right away
                }
            }
            // not worth it: rows by table or by hops for a while (longer every time)
            if (broke_inside) { next_try = r + backoff; backoff = backoff < 1024 ? backoff * 2 : 1024u; }
            if (moved) continue;
        }
        const uint32_t stop = next_try > r && next_try < RW ? next_try : RW;
        if ((win & 127u) == table_wnb) {
            // ---- by the landing tables, while they fit the window
            const uint32_t before = entry;
            const uint32_t code = chase_rows(r, q, entry, lbase, stop);
            if (entry != before) rowwin = win;     // (the rows just entered: the window does not change on the way)
            kk = 0; kk_known = false;
            GTSC_WG_WALK_TICK(w, tw_chase)
            // a row the table leaves to the hops: this one row
            if (code == 1 && !hop_rows(r + 1)) break;
            continue;
        }
        // ---- hops: to the region's end, or to the row where a stride may be tried again
        if (!hop_rows(stop)) break;
    }
    p = r * 64 + q;
    GTSC_WG_WALK_TICK(w, tw_stride)
    // ---- the starts inside the rows entered by table: lane r walks row r (no '11' code and no end marker on the way)
    uint64_t wb = ((uint64_t)bhi << 32) | blo;
    {
        uint32_t pos = entry < 64 ? entry : 64u;
        const uint32_t wn = rowwin & 127u;
        while (any_lane(pos < 64)) {
            if (pos < 64) {
                const uint32_t h = hrow[(int32_t)(lane * 64 + pos) - (int32_t)lane];
                const uint32_t nx = (h & HW_NEXT) + ((h >> 10) & 1u) * wn;
                wb |= (h & HW_RUN) ? (0x5555555555555555ull >> (64u - (nx - pos))) << pos : 1ull << pos;
                pos = nx;
            }
        }
    }
    GTSC_WG_WALK_TICK(w, tw_enum)
    // ---- the queue: lane r hands over the starts of row r
    if (strided) { wave_sync(); wb |= L.sbits[lane]; L.sbits[lane] = 0; }
    const uint32_t c = (uint32_t)__builtin_popcountll(wb);
    const uint32_t incl = wave_inclusive_sum(c);
    uint32_t at = incl - c;
    uint32_t cw = rowwin;
    while (any_lane(wb != 0)) {
        if (wb) {
            const uint32_t t = (uint32_t)__builtin_ctzll(wb);
            wb &= wb - 1;
            const uint32_t h = hrow[(int32_t)(lane * 64 + t) - (int32_t)lane];
            qd[at++] = (lane * 64 + t) | (cw << 16);
            if (h & HW_T2) cw = (h >> 13) & 0x3FFFu;
        }
    }
    w.cur = (w.cur & ~(uint64_t)(RPOS - 1)) + p;
    w.wnb = win & 127u; w.wtr7 = win >> 7;
    w.last_len = (kk & HW_RUN) ? 2u : (kk & 255u);
    w.last_t1 = (kk & HW_T1) != 0 && !(kk & HW_RUN);
    w.cool = next_try > r ? next_try - r : 0u;
    w.backoff = backoff;
    w.last_known = kk_known;
    const uint32_t found = read_lane(incl, 63);
    w.found_total += found;
    w.want_tables = 4 * stride_points < 3 * found || found == 0;
    w.carried = (!handover && found >= 32 && !w.want_tables) ? w.carried + 1 : 0u;
    w.pause = w.pause_after != 0 && w.carried >= w.pause_after;

    GTSC_WG_WALK_TICK(w, tw_queue)
    return found;
}

// The serial decoder, set to go on from a committed state (the point after the last one handed to the output).
template <int FMT, class Src>
__device__ __forceinline__ void decoder_from_state(Decoder<FMT>& d, uint32_t nw, Src& src, uint64_t pos, uint64_t stored_delta,
                                                   uint64_t stored_ts, uint64_t val, uint32_t win)
{
    const uint32_t nbits = win & 127u;
    const int32_t trail = sext7(win >> 7);
    d.nwords = nw;
    d.widx = (uint32_t)(pos >> 6); d.off = (uint32_t)pos & 63u;
    d.r0 = src(d.widx); d.r1 = src(d.widx + 1); d.r2 = src(d.widx + 2); d.r3 = src(d.widx + 3);
    d.st = 0;
    d.block_ts = (int64_t)src(0); d.stored_delta = (int64_t)stored_delta; d.stored_ts = stored_ts; d.val = val; d.pred_last = 0;
    d.lead = nbits ? (uint32_t)(64 - (int32_t)nbits - trail) : NO_WINDOW;
    d.trail = trail;
}

// Streams of plain points of equal length -- regular interval ('0'), changed value inside the window in force ('10' and
// the window's bits): `(double) i` of the reference's own benchmark (G/benchmark/EncodingBenchmark.java:60-69), counters
// that move at every sample.  Where such a run begins nothing is serial any more: point j starts j lengths on, every
// thread checks one candidate (the three control bits, the image's end, the capacity, a timestamp of zero), the first
// that fails ends the run, and the values before it are a prefix XOR of the payloads, the timestamps a multiple of the
// interval.  Before the pipeline of decode_block_wg() starts, the block is taken as far as that goes: runs of up to
// 1,024 points a round (the first round of a run by wave 0 alone, 64 candidates: most blocks are not of this kind and
// should not pay for more), and between runs the point that ended one is decoded by one lane with the serial decoder
// (a '11' code that declares the next window, as a rule).  The phase ends when the runs stay short, and the pipeline
// goes on from the state it leaves (or the serial decoder, after a decoded timestamp of zero).
// `proven`: the pipeline has seen the stream go by strides (it pauses to let this phase take over): short runs are
// put up with as long as the last few come to something.  Returns the points decoded by runs.
template <int FMT, uint32_t NW>
__device__ __forceinline__ uint32_t stride_phase(const DecodeArgs& a, const uint64_t* image, uint32_t nw, uint64_t p0, uint32_t cap, uint64_t limit,
                                                 WgLds& L, uint32_t tid, uint64_t& pos, uint32_t& n, uint64_t& stored_delta,
                                                 uint64_t& stored_ts, uint64_t& val, uint32_t& win, bool& serial, bool proven)
{
    constexpr uint32_t NT = NW * 64;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    auto src = [&](uint32_t i) -> uint64_t { return i < nw ? image[i] : 0ull; };
    uint32_t empty_runs = 0, accepted = 0, parity = 0;
    uint32_t recent = 1024;   // about eight times the length of the last few runs
    for (;;) {
        // ---- a run from `pos` on: unchanged points at a regular interval first ('00' each: as many as half the zero
        // bits ahead), every thread looking at one image word a round
        const uint32_t wnb = win & 127u;
        uint32_t run = 0;
        const uint64_t inside_end = limit > 195 ? limit - 195 : 0;   // a point must start before this (as `inside` below)
        if (pos < inside_end && n < cap) {
            const uint32_t kw = (uint32_t)(pos >> 6), o = (uint32_t)pos & 63u;
            const uint64_t first = image[kw] << o;
            uint64_t zeros = first ? clz64(first) : 64u - o;
            if (!first) {
                // the rest of the word is zero: the words after it, NT a round, up to the first that is not
                const uint32_t wend = (uint32_t)(inside_end >> 6) + 1;   // (< nw)
                for (uint32_t wbase = kw + 1; wbase < wend; wbase += NT, parity ^= 1u) {
                    const uint32_t w = wbase + tid;
                    const uint64_t x = w < wend ? image[w] : 1ull;
                    const uint64_t nz = lanes(x != 0);
                    const uint32_t fl = nz ? (uint32_t)__builtin_ctzll(nz) : 64u;
                    if (lane == 0) L.aggD[parity * 16 + wave] = fl < 64 ? (uint64_t)(wave * 64 + fl) * 64 + clz64(read_lane(x, fl)) : 64ull * NT;
                    wg_barrier();
                    const uint64_t mine = lane < NW ? L.aggD[parity * 16 + lane] : ~0ull;
                    // (the first wave that met a word other than zero says how many zero bits lie before its first one)
                    const uint64_t stopped = lanes(mine < 64ull * NT);
                    const uint64_t more = stopped ? read_lane(mine, (uint32_t)__builtin_ctzll(stopped)) : 64ull * NT;
                    zeros += more;
                    if (more < 64ull * NT) break;
                }
                parity ^= 1u;
            }
            uint64_t count = zeros >> 1;
            const uint64_t room = (inside_end - pos + 1) >> 1;
            count = count < room ? count : room;
            count = count < cap - n ? count : cap - n;
            // a timestamp of exactly zero ends the run before it (the point after it is read as a "first" one): the
            // smallest m > 0 with stored_ts + m * stored_delta == 0 (mod 2^64), if any
            if (stored_delta != 0 && count) {
                const uint32_t v = (uint32_t)__builtin_ctzll(stored_delta);
                const uint64_t odd = stored_delta >> v, target = 0ull - stored_ts;
                if ((target & ((1ull << v) - 1ull)) == 0) {
                    uint64_t inv = odd;   // inverse of an odd number mod 2^64: three bits right to begin with, doubled five times
                    for (int i = 0; i < 5; i++) inv *= 2ull - odd * inv;
                    uint64_t m = (target >> v) * inv;
                    if (v) m &= (1ull << (64u - v)) - 1ull;
                    if (m != 0 && m - 1 < count) count = m - 1;
                }
            }
            for (uint64_t j = tid; j < count; j += NT) {
                __builtin_nontemporal_store((int64_t)(stored_ts + (j + 1) * stored_delta), a.ts + p0 + n + j);
                __builtin_nontemporal_store(val, a.val + p0 + n + j);
            }
            run = (uint32_t)count;
            n += run;
            pos += 2ull * count;
            stored_ts += count * stored_delta;
        }
        // ---- then changed values inside the window in force ('0', '10', the window's bits)
        if (run == 0 && wnb != 0) {
            const uint32_t len = 3u + wnb, trail = (uint32_t)sext7(win >> 7) & 63u;
            for (uint32_t round = 0;; round++, parity ^= 1u) {
                const uint32_t width = round == 0 ? 64u : NT;
                const bool mine = round != 0 || wave == 0;
                const uint32_t j = round == 0 ? lane : tid;
                uint64_t x = 0, scan = 0;
                if (mine) {
                    const uint64_t p = pos + (uint64_t)j * len;
                    const bool inside = p + 67 + 128 < limit && n + j < cap;   // (no field of the point can run off the image)
                    const uint32_t kw = inside ? (uint32_t)(p >> 6) : 0u, o = (uint32_t)p & 63u;
                    const uint64_t w0 = image[kw], w1 = image[kw + 1], w2 = image[kw + 2];
                    const uint64_t hi = WindowDecoder<FMT>::window(w0, w1, o), lo = WindowDecoder<FMT>::window(w1, w2, o);
                    const uint64_t pay = ((hi << 3) | (lo >> 61)) >> (64u - wnb);
                    const uint64_t tsj = stored_ts + (uint64_t)(j + 1) * stored_delta;
                    const bool ok = inside && (hi >> 61) == 2u && tsj != 0;
                    const uint64_t bad = lanes(!ok);
                    const uint32_t fb = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
                    x = lane < fb ? pay << trail : 0ull;
                    scan = wave_inclusive_scan64<true>(x);
                    if (lane == 63) { L.aggD[parity * 16 + wave] = fb; L.aggX[parity * 16 + wave] = scan; }
                }
                wg_barrier();
                // the first wave with a candidate that failed, and what the waves before this one add up to
                const uint32_t waves_in = round == 0 ? 1u : NW;
                const uint64_t fbw = lane < waves_in ? L.aggD[parity * 16 + lane] : 64ull;
                const uint64_t tot = lane < waves_in ? L.aggX[parity * 16 + lane] : 0ull;
                const uint64_t short_m = lanes(fbw < 64);
                const uint32_t fw = short_m ? (uint32_t)__builtin_ctzll(short_m) : waves_in;   // (waves_in: none failed)
                const uint32_t got = fw < waves_in ? fw * 64 + (uint32_t)read_lane((uint32_t)fbw, fw) : width;
                const uint64_t before = read_lane(wave_inclusive_scan64<true>(lane < wave ? tot : 0ull), 63);
                const uint64_t all = read_lane(wave_inclusive_scan64<true>(lane <= fw ? tot : 0ull), 63);
                if (mine && j < got) {
                    __builtin_nontemporal_store((int64_t)(stored_ts + (uint64_t)(j + 1) * stored_delta), a.ts + p0 + n + j);
                    __builtin_nontemporal_store(val ^ before ^ scan, a.val + p0 + n + j);
                }
                n += got;
                pos += (uint64_t)got * len;
                stored_ts += (uint64_t)got * stored_delta;
                val ^= all;
                run += got;
                if (got < width) break;
            }
            parity ^= 1u;
        }
        accepted += run;
        empty_runs = run ? 0u : empty_runs + 1u;
        recent = recent - (recent >> 3) + (run < 1024 ? run : 1024u);
        // most blocks are not of this kind: three points in a row that start no run (eight, later on), a short run (at a block's start:
        // one look is all such a block should pay for) or runs that have become short (later on)
        if (empty_runs >= (proven ? 8u : 3u)) break;
        if (proven ? recent < 128 : (run != 0 && run < 128)) break;
        if (n >= cap || pos + 512 >= limit) break;
        // ---- the point that ended the run, by the serial decoder
        if (tid == 0) {
            Decoder<FMT> d;
            decoder_from_state(d, nw, src, pos, stored_delta, stored_ts, val, win);
            int64_t t; uint64_t v;
            uint64_t flag = 1;   // 1: no point (end of stream or something wrong: the pipeline's checked walk has it)
            if (d.next(t, v, src)) {
                a.ts[p0 + n] = t; a.val[p0 + n] = v;
                const uint32_t nb = d.lead == NO_WINDOW ? 0u : (uint32_t)(64 - (int32_t)d.lead - d.trail);
                L.aggS[0] = (uint64_t)d.widx * 64 + d.off;
                L.aggS[1] = (uint64_t)d.stored_delta;
                L.aggS[2] = d.stored_ts;
                L.aggS[3] = d.val;
                L.aggS[4] = nb | (((uint32_t)d.trail & 127u) << 7);
                flag = d.stored_ts == 0 ? 2 : 0;   // 2: a timestamp of zero: the serial decoder has the rest
            }
            L.aggS[5] = flag;
        }
        wg_barrier();
        const uint64_t flag = lds_peek64(&L.aggS[5]);
        if (flag != 1) {
            pos = lds_peek64(&L.aggS[0]); stored_delta = lds_peek64(&L.aggS[1]); stored_ts = lds_peek64(&L.aggS[2]);
            val = lds_peek64(&L.aggS[3]); win = (uint32_t)lds_peek64(&L.aggS[4]);
            n++;
        }
        wg_barrier();   // (the next serial point overwrites what was just read)
        if (flag == 2) serial = true;
        if (flag != 0) break;
    }
    // (every thread holds the same state; saying so keeps it in scalar registers from here on)
    auto uniform64 = [](uint64_t v) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    pos = uniform64(pos); stored_delta = uniform64(stored_delta); stored_ts = uniform64(stored_ts); val = uniform64(val);
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    win = (uint32_t)__builtin_amdgcn_readfirstlane((int)win);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)accepted);
}

// Decompresses block `s` with the whole workgroup of NW waves.
template <int FMT, uint32_t NW>
__device__ __forceinline__ void decode_block_wg(const DecodeArgs& a, uint64_t s, WgLds& L, uint32_t tid)
{
    constexpr uint32_t NT = NW * 64, NH = NW - 1;           // threads; helper waves
    constexpr uint32_t MB = (MAXB + NH - 1) / NH;           // batches a helper wave may have to emit per region
    const uint32_t lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint64_t ssz64 = a.slots.size(s);
    const uint64_t* const image = a.words + a.slots.begin(s);
    uint64_t nw64 = a.in_nwords ? a.in_nwords[s] : ssz64;
    if (nw64 > ssz64) nw64 = ssz64;                     // an image cannot be longer than its slot
    const uint32_t nw = nw64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nw64;
    const uint64_t p0 = a.pts.begin(s);
    const uint64_t cap64 = a.pts.size(s);
    const uint32_t cap = cap64 > 0x7FFFFF00ull ? 0x7FFFFF00u : (uint32_t)cap64;   // as in the lane-per-series kernels
    const uint64_t limit = (uint64_t)nw * 64;           // a field must end before this bit (G/LongArrayInput.java:48-52)

    // ---- state at the last point handed to the output ("committed"): where the serial decoder takes over.
    // Every wave keeps its own copy and moves it on in the same way.
    uint64_t pos = 0;                 // bit position of the next unread bit
    uint32_t n = 0;                   // points decoded
    uint64_t stored_delta = 0, stored_ts = 0, val = 0;
    uint32_t win = 0;                 // value window: significant bits (0: none yet) | trailing zeros (7 bits) << 7
    bool serial = nw < 4;             // tiny images: the serial decoder from the start
    bool finished = false;

    if (!serial) {
        // header and first point (G/GorillaDecompressor.java:27-35,62-72): nw >= 4, so neither runs off the image
        const uint64_t w0 = image[0], w1 = image[1], w2 = image[2];
        const uint32_t d = (uint32_t)(w1 >> 37);
        if (d == 0x7FFFFFFu) {
            finished = true;                             // empty block
        } else {
            val = (w1 << 27) | (w2 >> 37);
            stored_delta = d;
            stored_ts = w0 + d;
            pos = 64 + 91;
            n = 1;
            if (cap == 0 || stored_ts == 0) { serial = true; pos = 64; n = 0; stored_ts = 0; stored_delta = 0; val = 0; }
            else if (tid == 0) { a.ts[p0] = (int64_t)stored_ts; a.val[p0] = val; }
        }
    }
    bool clean_end = finished;

    // The block goes through two kinds of phases in turn: runs of plain points of equal length as far as they go
    // (stride_phase()), then the pipeline -- until the stream ends, or the walker sees it go by strides again and pauses.
    uint32_t pause_after = 1;   // doubles with every pause that bought little
    bool proven = false;
    for (;;) {
    if (!finished && !serial) {
        const uint32_t by_runs = stride_phase<FMT, NW>(a, image, nw, p0, cap, limit, L, tid, pos, n, stored_delta, stored_ts, val, win, serial, proven);
        if (proven) pause_after = by_runs >= 512 ? 1u : pause_after < 64 ? pause_after * 2 : 0u;
        GTSC_WG_RUN_PHASE_PRINT(by_runs, n, pos, win, serial)
    }
    bool paused = false;

    if (!finished && !serial) {
        auto word = [&](uint32_t w) -> uint64_t { return w < nw ? image[w] : 0ull; };
        // ---- prologue: the pipeline starts in the region that holds `pos`: its image words and those of the region
        // after it, its hop words and landing tables
        const uint32_t k0 = (uint32_t)(pos / RPOS);
        for (uint32_t w = tid; w < 2 * RW + 1; w += NT) L.ring[(k0 * RW + w) & (RINGW - 1)] = word(k0 * RW + w);
        if (tid == 0) { L.stop = 0; L.clean = 0; L.qcount[0] = 0; L.qcount[1] = 0; }
        if (tid < RW) L.sbits[tid] = 0;
        wg_barrier();
        auto classify_region = [&](uint32_t r, uint32_t first_task, uint32_t tasks_step) {
            // four consecutive positions per task: they share the two image words they look at
            const uint64_t g = (uint64_t)r * RPOS;
            for (uint32_t t = first_task; t < RPOS / 4; t += tasks_step) {
                const uint64_t P = g + 4u * t;
                const uint32_t kw = (uint32_t)(P >> 6), o = (uint32_t)P & 63u;
                const uint64_t r0 = L.ring[kw & (RINGW - 1)], r1 = L.ring[(kw + 1) & (RINGW - 1)];
                uint4 h;
                h.x = classify_position<FMT>(WindowDecoder<FMT>::window(r0, r1, o), o);
                h.y = classify_position<FMT>(WindowDecoder<FMT>::window(r0, r1, o + 1), o + 1);
                h.z = classify_position<FMT>(WindowDecoder<FMT>::window(r0, r1, o + 2), o + 2);
                h.w = classify_position<FMT>(WindowDecoder<FMT>::window(r0, r1, o + 3), o + 3);
                *reinterpret_cast<uint4*>(&L.hopw[((uint32_t)P & (2 * RPOS - 1))]) = h;
            }
        };
        classify_region(k0, tid, NT);
        // (a wave classifies whole rows -- four positions per lane, 256 per pass: the tables of its rows are its own work)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        make_landing_tables(L, k0, 4 * wave, win & 127u, lane);
        if (tid == 0) { L.landwnb[k0 & 1u] = win & 127u; L.landwnb[(k0 & 1u) ^ 1u] = 0xFFFFFFFFu; L.curwnb = win & 127u; L.want_tables = 1; }
        wg_barrier();

        Walk wk{pos, win & 127u, win >> 7, 0u, false, 0u, n, false, false, 8u, true, true, pause_after, 0u, false};
        GTSC_WG_STAMP_DECL
        bool stop_prev = false;       // the walker stopped in the tick before: this one only emits
        uint32_t qprev = 0;           // points the walker queued in the tick before
        // what the walker reported at the end of the tick before (read by every wave between that tick's barriers, so
        // that all of them make this tick's tables for the same window size)
        uint32_t wnb_seen = win & 127u;
        bool want_seen = true;
        for (uint32_t k = k0;; k++) {
            const uint32_t slot = k & 1u, pslot = slot ^ 1u;
            uint64_t s_loc[MB], x_loc[MB];
            // ---- the walker in region k; the others classify region k + 1 and parse what was queued in region k - 1
            GTSC_WG_TICK(tk_other)
            if (wave == 0) {
                uint32_t qn = 0;
                if (!stop_prev) {
                    bool rest = true;
                    if ((uint64_t)(k + 1) * RPOS + 256 < limit && wk.found_total < cap) {
                        // (the fast walk does not count: if the region turns out to fill the capacity, it is walked
                        // again by the form that looks at every point)
                        const Walk before = wk;
                        qn = walk_region_fast(wk, L, k, lane, rest);
                        if (wk.found_total > cap) { wk = before; qn = 0; rest = true; }
                    }
                    if (rest) {
                        GTSC_WG_WALK_COUNT(wk, tw_handover)
                        GTSC_WG_WALK_BEGIN(wk)
                        qn = walk_region_careful(wk, L, k, limit, cap, lane, qn);
                        GTSC_WG_WALK_TICK(wk, tw_careful)
                    }
                }
                if (lane == 0) {
                    L.qcount[slot] = qn;
                    L.stop = wk.stop ? 1u : wk.pause ? 2u : 0u;
                    L.clean = wk.clean_end ? 1u : 0u;
                    L.curwnb = wk.wnb;
                    L.want_tables = wk.want_tables ? 1u : 0u;
                    L.kb[slot] = 0xFFFFFFFFu;   // (for the tick after this one)
                    L.anydd[slot] = 0;
                }
            } else {
                uint64_t landing = 0;
                const uint32_t lw = (k + 2) * RW + 1 + lane;   // words [(k+2) RW + 1, (k+3) RW]: what region k + 2's positions look at
                if (!stop_prev) {
                    if (wave == 1) landing = word(lw);
                    // four rows per helper wave -- and the four that leaves over to the last wave, which is the last to
                    // get points to parse
                    classify_region(k + 1, (wave - 1) * 64 + lane, RPOS);
                    if (wave == NH) classify_region(k + 1, NH * 64 + lane, RPOS);
                    // ... and the landing tables of the same rows, for the window size the walker had when the tick
                    // began -- unless the stream is one the strides take, which do not use them
                    if (want_seen) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        make_landing_tables(L, k + 1, 4 * (wave - 1), wnb_seen, lane);
                        if (wave == NH) make_landing_tables(L, k + 1, 4 * NH, wnb_seen, lane);
                    }
                }
                GTSC_WG_TICK(tk_x1)
#pragma unroll
                for (uint32_t j = 0; j < MB; j++) {
                    const uint32_t b = (wave - 1) + j * NH;
                    s_loc[j] = 0; x_loc[j] = 0;
                    if (b * 64 < qprev) {
                        const uint32_t m = qprev - b * 64 < 64 ? qprev - b * 64 : 64u;
                        const bool have = lane < m;
                        const uint32_t e = have ? L.queue[pslot][b * 64 + lane] : 0u;
                        const uint32_t rel = e & 0xFFFFu;
                        const uint64_t bp = (uint64_t)(k - 1) * RPOS + rel;
                        const uint32_t kw = (uint32_t)(bp >> 6), o = (uint32_t)bp & 63u;
                        const uint64_t r0 = L.ring[kw & (RINGW - 1)], r1 = L.ring[(kw + 1) & (RINGW - 1)],
                                       r2 = L.ring[(kw + 2) & (RINGW - 1)];
                        WindowDecoder<FMT> d;
                        d.off = 0; d.widx = 0; d.nwords = 0xFFFFFFFFu;
                        d.nbits = (e >> 16) & 127u;
                        d.trail = sext7(e >> 23);
                        const uint64_t hi = WindowDecoder<FMT>::window(r0, r1, o), lo = WindowDecoder<FMT>::window(r1, r2, o);
                        const typename WindowDecoder<FMT>::Parsed pq = d.template parse<false>(hi, [lo] { return lo; });
                        const uint64_t x = have ? pq.bits << ((uint32_t)pq.trail & 63u) : 0ull;
                        const uint64_t dd = have ? (uint64_t)pq.dd : 0ull;
                        uint64_t dsum = 0, ssum = 0;
                        if (any_lane(dd != 0)) {
                            dsum = wave_inclusive_scan64<false>(dd);
                            ssum = wave_inclusive_scan64<false>(dsum);
                            if (lane == 0) L.anydd[pslot] = 1u;
                        }
                        const uint64_t xsum = wave_inclusive_scan64<true>(x);
                        s_loc[j] = ssum; x_loc[j] = xsum;
                        if (lane == m - 1) {
                            L.aggD[b] = dsum; L.aggS[b] = ssum; L.aggX[b] = xsum;
                            if (b * 64 + m == qprev) {   // the region's last point: where it ends, and the window after it
                                L.endrel[pslot] = rel + pq.o3;
                                L.endwin[pslot] = pq.nbits | (((uint32_t)pq.trail & 127u) << 7);
                            }
                        }
                    }
                }
                if (!stop_prev && wave == 1) L.ring[lw & (RINGW - 1)] = landing;
            }
            GTSC_WG_TICK(tk_x2)
            wg_barrier();
            GTSC_WG_TICK(tk_b1)
            const uint32_t qnow = lds_peek(&L.qcount[slot]);
            const uint32_t stop_code = lds_peek(&L.stop);
            const bool stop_now = stop_code != 0;
            if (stop_code == 1) clean_end = lds_peek(&L.clean) != 0;
            if (stop_code == 2) paused = true;
            // ---- the walker has changed its window size under way: the landing tables of region k + 1 once more, four
            // rows per wave
            {
                const uint32_t wnb_now = lds_peek(&L.curwnb);
                const bool want_now = lds_peek(&L.want_tables) != 0;
                if (!stop_prev && !stop_now) {
                    if (want_seen && wnb_now != wnb_seen) make_landing_tables(L, k + 1, 4 * wave, wnb_now, lane);
                    if (tid == 0) L.landwnb[pslot] = want_seen ? wnb_now : 0xFFFFFFFFu;
                }
                wnb_seen = wnb_now;
                want_seen = want_now;
                // Without points to emit there is no second barrier below: one here, so that no wave is still reading what
                // the walker reported (or making tables) when the walker goes on into the next region.
                if (!qprev) wg_barrier();
            }
            uint32_t kbv = 0xFFFFFFFFu;
            const uint32_t nb = (qprev + 63) / 64;
            if (qprev) {
                // ---- from the batches' sums to the state before each batch (every wave works this out for itself):
                // deltas are the running sum of the delta-of-deltas, timestamps of the deltas
                const bool mineb = lane < nb;
                const uint64_t cnt = mineb ? (qprev - lane * 64 < 64 ? qprev - lane * 64 : 64u) : 0u;
                uint64_t delta_in = stored_delta, ts_in, dinc = 0, tinc;
                if (lds_peek(&L.anydd[pslot])) {
                    const uint64_t D = mineb ? L.aggD[lane] : 0ull, S = mineb ? L.aggS[lane] : 0ull;
                    dinc = wave_inclusive_scan64<false>(D);
                    delta_in = stored_delta + dinc - D;
                    const uint64_t T = cnt * delta_in + S;
                    tinc = wave_inclusive_scan64<false>(T);
                    ts_in = stored_ts + tinc - T;
                } else {
                    // regular cadence throughout: no sums needed
                    tinc = (uint64_t)(lane * 64u + (uint32_t)cnt) * stored_delta;
                    ts_in = stored_ts + (uint64_t)(lane * 64u) * stored_delta;
                }
                const uint64_t X = mineb ? L.aggX[lane] : 0ull;
                const uint64_t xinc = wave_inclusive_scan64<true>(X);
                const uint64_t val_in = val ^ xinc ^ X;
                // ---- the helpers' points
                uint64_t ts_out[MB], v_out[MB];
                if (wave != 0) {
#pragma unroll
                    for (uint32_t j = 0; j < MB; j++) {
                        const uint32_t b = (wave - 1) + j * NH;
                        ts_out[j] = 1; v_out[j] = 0;
                        if (b * 64 < qprev) {
                            const uint32_t m = qprev - b * 64 < 64 ? qprev - b * 64 : 64u;
                            const uint64_t db = read_lane(delta_in, b), tb = read_lane(ts_in, b), vb = read_lane(val_in, b);
                            ts_out[j] = tb + (uint64_t)(lane + 1) * db + s_loc[j];
                            v_out[j] = vb ^ x_loc[j];
                            // a timestamp of exactly zero turns the next point into a "first" one: not this path's business
                            if (any_lane(lane < m && ts_out[j] == 0) && lane == 0) atomicMin(&L.kb[pslot], b);
                        }
                    }
                }
                GTSC_WG_TICK(tk_y)
                wg_barrier();
                GTSC_WG_TICK(tk_b2)
                kbv = lds_peek(&L.kb[pslot]);
                if (wave != 0) {
#pragma unroll
                    for (uint32_t j = 0; j < MB; j++) {
                        const uint32_t b = (wave - 1) + j * NH;
                        if (b * 64 < qprev && b < kbv) {
                            const uint32_t m = qprev - b * 64 < 64 ? qprev - b * 64 : 64u;
                            if (lane < m) {
                                __builtin_nontemporal_store((int64_t)ts_out[j], a.ts + p0 + n + b * 64 + lane);
                                __builtin_nontemporal_store(v_out[j], a.val + p0 + n + b * 64 + lane);
                            }
                        }
                    }
                }
                // ---- commit
                const uint64_t g = (uint64_t)(k - 1) * RPOS;
                if (kbv >= nb) {
                    stored_delta = stored_delta + read_lane(dinc, nb - 1);
                    stored_ts = stored_ts + read_lane(tinc, nb - 1);
                    val = val ^ read_lane(xinc, nb - 1);
                    n += qprev;
                    pos = g + lds_peek(&L.endrel[pslot]);
                    win = lds_peek(&L.endwin[pslot]);
                } else {
                    // up to the batch that met a zero timestamp; the serial decoder has the rest
                    stored_delta = read_lane(delta_in, kbv);
                    stored_ts = read_lane(ts_in, kbv);
                    val = read_lane(val_in, kbv);
                    n += kbv * 64;
                    const uint32_t e = lds_peek(&L.queue[pslot][kbv * 64]);
                    pos = g + (e & 0xFFFFu);
                    win = e >> 16;
                    serial = true;
                }
            }
            GTSC_WG_TICK(tk_z)
            if (serial || stop_prev) break;
            stop_prev = stop_now;
            qprev = qnow;
        }
        GTSC_WG_STAMP_PRINT
        GTSC_WG_WALK_PRINT(wk)
    }
    // (paused: everything the walker queued has been handed to the output, the committed state is where the runs go on)
    if (!paused || serial || finished) break;
    proven = true;
    wg_barrier();   // (the pipeline's last reads of LDS before the run phase reuses it)
    }
    uint32_t st = 0;
    if (!clean_end || serial) {
        // ---- the serial decoder, one lane, from the committed state (exact order of the reference's checks)
        if (tid == 0) {
            auto src = [&](uint32_t i) -> uint64_t { return i < nw ? image[i] : 0ull; };
            Decoder<FMT> d;
            if (nw < 4 || pos < 64) d.begin(nw, src);
            else decoder_from_state(d, nw, src, pos, stored_delta, stored_ts, val, win);
            int64_t t; uint64_t v;
            while (d.next(t, v, src)) {
                if (n < cap) { a.ts[p0 + n] = t; a.val[p0 + n] = v; } else st = ST_OVERFLOW;
                n++;
            }
            st |= d.st;
        }
    }
    if (tid == 0) {
        a.out_npoints[s] = n;
        if (a.status) a.status[s] = st;
        if (st) atomicOr(a.flags, st);
    }
}

template <int FMT, uint32_t NW>
__global__ __launch_bounds__(64 * NW) void decode_block_wg_kernel(DecodeArgs a, uint32_t per_group)
{
    __shared__ WgLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t cand = blockIdx.x + (uint64_t)lane * gridDim.x;
    const bool mine = lane < per_group && cand < a.nseries && block_owner<FMT>(a, cand) == OWNER_WG;
    uint64_t todo = lanes(mine);   // (the same in every wave)
    while (todo) {
        const uint32_t j = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        decode_block_wg<FMT, NW>(a, blockIdx.x + (uint64_t)j * gridDim.x, L, tid);
        __syncthreads();   // the next block starts over in LDS
    }
}

// The blocks of the workgroup-per-block decompressor, listed by size class (DecodePlan::queue_*): class 0 holds the
// largest long image of the batch, class c those of at least (15 - c) fifteenths of its words.  (The order inside
// a class is whatever the atomics make it.)
template <int FMT>
__global__ __launch_bounds__(256) void wg_queue_kernel(DecodeArgs a, DecodePlan* plan)
{
    const uint64_t widest = plan->words_longest;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.nseries; s += (uint64_t)gridDim.x * blockDim.x) {
        if (block_owner<FMT>(a, s) != OWNER_WG) continue;
        uint64_t nw = a.in_nwords ? a.in_nwords[s] : a.slots.size(s);
        nw = nw < widest ? nw : widest;
        const uint32_t cls = DecodePlan::QUEUE_CLASSES - 1 - (uint32_t)(nw * (DecodePlan::QUEUE_CLASSES - 1) / (widest ? widest : 1));
        const uint32_t at = atomicAdd(&plan->queue_count[cls], 1u);
        if (at < DecodePlan::QUEUE_CAP) plan->queue_list[cls][at] = (uint32_t)s;
        else atomicOr(&plan->queue_overflow, 1u);
    }
}

// As many workgroups as the device has compute units, each drawing its next block from the plan's lists, largest
// images first.  Where a list overflowed the workgroups go through the series by number instead.
template <int FMT, uint32_t NW>
__global__ __launch_bounds__(64 * NW) void decode_block_wg_queue_kernel(DecodeArgs a, DecodePlan* plan)
{
    __shared__ WgLds L;
    __shared__ uint32_t drawn;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (plan->queue_overflow) {
        // (workgroup g: series g, g + G, g + 2 G, ... -- neighbours in a batch tend to be alike, and this way they go
        // to different workgroups)
        for (uint64_t base = blockIdx.x; base < a.nseries; base += (uint64_t)gridDim.x * 64) {
            const uint64_t cand = base + (uint64_t)lane * gridDim.x;
            const bool mine = cand < a.nseries && block_owner<FMT>(a, cand) == OWNER_WG;
            uint64_t todo = lanes(mine);   // (the same in every wave)
            while (todo) {
                const uint32_t j = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                decode_block_wg<FMT, NW>(a, base + (uint64_t)j * gridDim.x, L, tid);
                __syncthreads();   // the next block starts over in LDS
            }
        }
        return;
    }
    // lane c: the tickets of the classes before class c
    const uint32_t mine = lane < DecodePlan::QUEUE_CLASSES ? plan->queue_count[lane] : 0u;
    const uint32_t upto = wave_inclusive_sum(mine);
    const uint32_t total = read_lane(upto, 63);
    for (;;) {
        if (tid == 0) drawn = atomicAdd(&plan->queue_ticket, 1u);
        __syncthreads();
        const uint32_t t = drawn;
        __syncthreads();
        if (t >= total) break;
        const uint32_t cls = (uint32_t)__builtin_popcountll(lanes(lane < DecodePlan::QUEUE_CLASSES && upto <= t));
        const uint32_t first = cls ? read_lane(upto, cls - 1) : 0u;
        decode_block_wg<FMT, NW>(a, plan->queue_list[cls][t - first], L, tid);
        __syncthreads();
    }
}

// compute units of the current device (remembered per device)
uint32_t compute_units()
{
    static std::atomic<uint32_t> known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    uint32_t n = known[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int v = 0;
        n = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? (uint32_t)v : 256u;
        known[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

}  // namespace

hipError_t launch_decode_blocks_wg(const DecodeArgs& a, int format, hipStream_t stream)
{
    if (a.nseries == 0) return hipSuccess;
    if (a.dispatch) *a.dispatch |= DISPATCH_WG;
    const uint32_t units = compute_units();
    if (a.nseries <= units || a.nseries > 0xFFFFFFFFull) {
        // a workgroup per series (they all start at once), or sixty-four series each where there are billions
        const uint64_t per_group = a.nseries <= units ? 1 : 64;
        const uint32_t grid = (uint32_t)((a.nseries + per_group - 1) / per_group);
        if (format == FMT_V2) hipLaunchKernelGGL((decode_block_wg_kernel<FMT_V2, 16>), dim3(grid), dim3(64 * 16), 0, stream, a, (uint32_t)per_group);
        else                  hipLaunchKernelGGL((decode_block_wg_kernel<FMT_V1, 16>), dim3(grid), dim3(64 * 16), 0, stream, a, (uint32_t)per_group);
        return hipGetLastError();
    }
    const uint64_t groups = (a.nseries + 255) / 256;
    const uint32_t lgrid = (uint32_t)(groups < 1024 ? groups : 1024);
    if (format == FMT_V2) {
        hipLaunchKernelGGL(wg_queue_kernel<FMT_V2>, dim3(lgrid), dim3(256), 0, stream, a, a.plan_buf);
        hipLaunchKernelGGL((decode_block_wg_queue_kernel<FMT_V2, 16>), dim3(units), dim3(64 * 16), 0, stream, a, a.plan_buf);
    } else {
        hipLaunchKernelGGL(wg_queue_kernel<FMT_V1>, dim3(lgrid), dim3(256), 0, stream, a, a.plan_buf);
        hipLaunchKernelGGL((decode_block_wg_queue_kernel<FMT_V1, 16>), dim3(units), dim3(64 * 16), 0, stream, a, a.plan_buf);
    }
    return hipGetLastError();
}

}  // namespace gtsc
