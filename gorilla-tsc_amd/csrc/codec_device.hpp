// Per-lane Gorilla codec state machines (device code, gfx950).
//
// One lane owns one series block and walks it point by point; 64 series advance
// in lock step per wavefront.  The state machines are independent of how points
// and words reach the lane (direct global access or LDS-staged tiles): callers
// supply a word sink / word source.
//
// Bit-exactness targets (reference tree, G/ = src/main/java/fi/iki/yak/ts/compression/gorilla/):
//   v2 encode: G/GorillaCompressor.java:49-143, G/ValueCompressor.java:27-91
//   v2 decode: G/GorillaDecompressor.java:51-127, G/ValueDecompressor.java:26-54
//   v1 encode: G/Compressor.java:42-181      v1 decode: G/Decompressor.java:44-130
//   word image: G/LongArrayOutput.java:96-145, reader G/LongArrayInput.java:35-91
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gtsc {

constexpr int FMT_V2 = 0;
constexpr int FMT_V1 = 1;

constexpr uint32_t ST_OVERFLOW = 1u;
constexpr uint32_t ST_TRUNCATED = 2u;
constexpr uint32_t ST_MALFORMED = 4u;

constexpr uint32_t NO_WINDOW = 255u;  // stands for Integer.MAX_VALUE (G/ValueCompressor.java:12)

// CSR offsets, or a uniform stride when off == nullptr.
struct Layout {
    const uint64_t* off;
    uint64_t uniform;
    __device__ __forceinline__ uint64_t begin(uint64_t s) const { return off ? off[s] : s * uniform; }
    __device__ __forceinline__ uint64_t size(uint64_t s) const { return off ? off[s + 1] - off[s] : uniform; }
};

// Wave votes straight on the condition's lane mask (the library forms go through an integer
// per lane: two more instructions each, in loops where a dozen of them run per point).
// lanes(p): the mask of lanes where p holds.  Given a comparison it is that comparison's own
// result; given a stored or combined flag the compiler rebuilds it through a register, so the
// hot loops keep flags as integers, vote on comparisons and combine the masks.
__device__ __forceinline__ uint64_t lanes(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
__device__ __forceinline__ bool all_lanes(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }

__device__ __forceinline__ uint32_t clz64(uint64_t x) { return (uint32_t)__builtin_clzll(x); }  // x != 0
__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }  // x != 0

// ---------------------------------------------------------------- predictors
// PRED_LAST: the prediction is the previous value (G/predictors/LastValuePredictor.java:10-22).
// PRED_DFCM: differential finite context method (G/predictors/DifferentialFCM.java:23-46): a table
// of value differences indexed by a hash of recent differences; the table of a series lives in
// global memory (2^k entries, zero-filled before the block starts).
constexpr int PRED_LAST = 0, PRED_DFCM = 1;
struct Dfcm {
    int64_t* table;
    uint32_t hash, mask;
    __device__ __forceinline__ uint64_t predict(uint64_t last) const { return (uint64_t)table[hash] + last; }
    __device__ __forceinline__ void update(uint64_t last, uint64_t v)
    {
        const int64_t d = (int64_t)(v - last);
        table[hash] = d;
        // (int)(((lastHash << 5) ^ ((value - lastValue) >> 50)) & mask): the mask keeps low bits only
        hash = ((hash << 5) ^ (uint32_t)(d >> 50)) & mask;
    }
};

// ---------------------------------------------------------------- encoder
// Pending bits sit left-aligned in `acc` (its top `nacc` bits, zeros below).
// A completed 64-bit word is handed to the sink: sink(word_index, word).
// The per-point path is written with selects instead of branches: 64 series
// advance in lock step and take different code paths at every point, so a
// branchy form pays for every path anyway plus the exec-mask bookkeeping.
// put() keeps the dependency chain through `acc` to an OR and a select: the
// shifts depend only on the bit count, which runs ahead as a chain of adds.
template <int FMT, int PRED = PRED_LAST>
struct Encoder {
    Dfcm dfcm;  // PRED_DFCM only
    uint64_t acc;
    uint32_t nacc;
    uint32_t widx;
    // previous timestamp: v2 does int arithmetic, so only its low half is kept; "started"
    // stands for the reference's storedTimestamp != 0 test (G/GorillaCompressor.java:50,65)
    uint32_t pts_lo, pts_hi;
    bool started;
    int64_t stored_delta;  // v2 keeps an int (sign-extended here), v1 a long
    uint64_t prev;
    uint32_t lead, trail;

    // n in [0,64]; bits of v above n must be zero.
    template <class Sink>
    __device__ __forceinline__ void put(uint64_t v, uint32_t n, Sink& sink)
    {
        const uint64_t va = v << ((64 - n) & 63);          // left-aligned code (n == 0: v == 0)
        const uint64_t word = acc | (va >> nacc);
        const uint64_t rest = (va << 1) << (63 - nacc);    // what did not fit (nacc == 0: nothing)
        const uint32_t total = nacc + n;                   // <= 127
        const uint32_t full = total >> 6;                  // 1: the word is complete
        if (Sink::ALWAYS || full) sink(widx, word);  // an LDS sink takes every candidate: cheaper than a branch
        widx += full;
        // rest is zero unless bits did not fit, so the choice between rest and word is a mask, not a
        // select (a 64-bit select is two v_cndmask in a row, which this part executes 15 cycles slower)
        const uint32_t keep = full - 1u;                   // all ones while the word is still open
        acc = rest | (word & (((uint64_t)keep << 32) | keep));
        nacc = total & 63;
    }

    template <class Sink>
    __device__ __forceinline__ void begin(int64_t block_ts, Sink& sink)
    {
        acc = 0; nacc = 0; widx = 0;
        pts_lo = 0; pts_hi = 0; started = false; stored_delta = 0; prev = 0;
        lead = NO_WINDOW; trail = 0;
        put((uint64_t)block_ts, 64, sink);  // addHeader, G/GorillaCompressor.java:39-41
    }

    // writeFirst: 27 low bits of the delta, then the raw value
    template <class Sink>
    __device__ __forceinline__ void first(uint64_t d, uint32_t ts_lo, uint32_t ts_hi, bool ts_zero, uint64_t v, Sink& sink)
    {
        stored_delta = (FMT == FMT_V2) ? (int64_t)(int32_t)(uint32_t)d : (int64_t)d;
        pts_lo = ts_lo; pts_hi = ts_hi;
        started = !ts_zero;
        if (PRED == PRED_DFCM) dfcm.update(prev, v);  // writeFirst: predictor.update(value), G/ValueCompressor.java:27-30
        prev = v;
        put(d & 0x7FFFFFFull, 27, sink);
        put(v, 64, sink);
    }

    template <class Sink>
    __device__ __forceinline__ void add(int64_t block_ts, int64_t ts, uint64_t v, Sink& sink)
    {
        if (__builtin_expect(!started, 0)) {
            first((uint64_t)ts - (uint64_t)block_ts, (uint32_t)ts, (uint32_t)((uint64_t)ts >> 32), ts == 0, v, sink);
            return;
        }
        started = ts != 0;
        add_next((uint32_t)ts, (uint32_t)((uint64_t)ts >> 32), v, sink);
    }

    // v2 only needs the low half of a timestamp plus whether the whole of it is zero
    template <class Sink>
    __device__ __forceinline__ void add_lo(uint32_t block_ts_lo, uint32_t ts_lo, bool ts_zero, uint64_t v, Sink& sink)
    {
        static_assert(FMT == FMT_V2, "v1 does long arithmetic on timestamps");
        if (__builtin_expect(!started, 0)) {
            first((uint64_t)(ts_lo - block_ts_lo), ts_lo, 0u, ts_zero, v, sink);
            return;
        }
        started = !ts_zero;
        add_next(ts_lo, 0u, v, sink);
    }

    // The codes of one point that is known not to be a "first" point: timestamp code and value
    // header together (head, hlen <= 36 + 14 bits), then the value's payload (pay, plen).
    struct Code {
        uint64_t head, pay;
        uint32_t hlen, plen;
    };

    // Works out a point's codes and moves the compressor's state past it; nothing is written.
    __device__ __forceinline__ Code code_next(uint32_t ts_lo, uint32_t ts_hi, uint64_t v)
    {
        // ---- timestamp: code (thi:tlo) of tlen bits
        uint32_t tlo, thi, tlen;
        if (FMT == FMT_V2) {
            // int arithmetic: only the low 32 bits of the timestamps take part
            const uint32_t nd = ts_lo - pts_lo;
            const int32_t dod = (int32_t)(nd - (uint32_t)(int32_t)stored_delta);
            if (all_lanes(dod == 0)) {
                // regular cadence in every lane (the common case by far): the code is the single bit '0'
                tlen = 1u; tlo = 0u; thi = 0u;
            } else {
                const uint32_t z = (((uint32_t)dod << 1) ^ (uint32_t)(dod >> 31)) - 1u;  // zigzag - 1
                uint32_t need = 32u - (uint32_t)__clz((int)z);  // bits required; 0 when z == 0 (dod == -1)
                need = need > 13u ? 13u : need;
                // bucket by bits required, two bits per entry: 0 -> 36-bit code (need 0 or > 12),
                // 1 -> '10'+7, 2 -> '110'+9, 3 -> '1110'+12   (G/GorillaCompressor.java:110-137)
                constexpr uint32_t BUCKET = (1u * 0x5554u) | (2u << 16) | (2u << 18) | (3u << 20) | (3u << 22) | (3u << 24);
                const uint32_t bk = (BUCKET >> (2u * need)) & 3u;
                constexpr uint32_t LEN = 36u | (9u << 8) | (12u << 16) | (16u << 24);
                constexpr uint64_t PRE = 0ull | (0x100ull << 16) | (0xC00ull << 32) | (0xE000ull << 48);
                const bool zero = dod == 0;
                tlen = zero ? 1u : (LEN >> (8u * bk)) & 0xFFu;
                const uint32_t pre = (uint32_t)(PRE >> (16u * bk)) & 0xFFFFu;
                tlo = zero ? 0u : (z | pre);
                thi = (zero || bk != 0) ? 0u : 0xFu;
            }
            stored_delta = (int64_t)(int32_t)nd;
        } else {
            const uint64_t ts = ((uint64_t)ts_hi << 32) | ts_lo, pts = ((uint64_t)pts_hi << 32) | pts_lo;
            const uint64_t nd = ts - pts;
            const uint64_t dod = nd - (uint64_t)stored_delta;
            if (all_lanes(dod == 0)) {
                tlen = 1u; tlo = 0u; thi = 0u;  // regular cadence in every lane
            } else {
                const bool zero = dod == 0;
                const bool b7 = (dod + 63u) <= 127u, b9 = (dod + 255u) <= 511u, b12 = (dod + 2047u) <= 4095u;
                tlen = zero ? 1u : b7 ? 9u : b9 ? 12u : b12 ? 16u : 36u;
                const uint32_t lo = (uint32_t)dod;
                tlo = zero ? 0u : b7 ? (0x100u | (lo & 0x7Fu)) : b9 ? (0xC00u | (lo & 0x1FFu)) : b12 ? (0xE000u | (lo & 0xFFFu)) : lo;
                thi = (zero || b7 || b9 || b12) ? 0u : 0xFu;
            }
            stored_delta = (int64_t)nd;
        }
        pts_lo = ts_lo; pts_hi = ts_hi;
        // ---- value: header vh of vhlen bits, payload pay of plen bits
        const uint64_t x = (PRED == PRED_DFCM ? dfcm.predict(prev) : prev) ^ v;  // G/ValueCompressor.java:32-34
        if (PRED == PRED_DFCM) dfcm.update(prev, v);
        prev = v;
        const bool nz = x != 0;
        uint32_t lz = (uint32_t)__clzll((long long)x);         // 64 when x == 0 (unused then)
        const uint32_t tz = (uint32_t)(__ffsll((long long)x) - 1);
        if (FMT == FMT_V1) lz = lz >= 32 ? 31u : lz;           // G/Compressor.java:132-134
        const bool reuse = nz && lz >= lead && tz >= trail;
        const bool fresh = nz && !reuse;
        const uint32_t sig = 64u - lz - tz;
        const uint32_t sh = reuse ? trail : tz;
        const uint64_t pay = x >> (sh & 63);                    // x == 0 gives 0 whatever the count
        const uint32_t plen = nz ? (reuse ? 64u - lead - trail : sig) : 0u;
        const uint32_t vh_new = (FMT == FMT_V2) ? ((3u << 12) | (lz << 6) | (sig - 1u)) : ((3u << 11) | (lz << 6) | (sig & 63u));
        const uint32_t vh = nz ? (reuse ? 2u : vh_new) : 0u;
        const uint32_t vhlen = nz ? (reuse ? 2u : (FMT == FMT_V2 ? 14u : 13u)) : 1u;
        lead = fresh ? lz : lead;
        trail = fresh ? tz : trail;
        const uint64_t tcode = ((uint64_t)thi << 32) | tlo;
        return Code{(tcode << vhlen) | vh, pay, tlen + vhlen, plen};
    }

    template <class Sink>
    __device__ __forceinline__ void emit(const Code& c, Sink& sink)
    {
        if (all_lanes(c.hlen + c.plen <= 64u)) {
            put((c.head << c.plen) | c.pay, c.hlen + c.plen, sink);  // one step when every lane's point fits a word (plen <= 62 then)
        } else {
            put(c.head, c.hlen, sink);
            put(c.pay, c.plen, sink);
        }
    }

    // A point that is known not to be a "first" point: straight-line code.
    template <class Sink>
    __device__ __forceinline__ void add_next(uint32_t ts_lo, uint32_t ts_hi, uint64_t v, Sink& sink)
    {
        emit(code_next(ts_lo, ts_hi, v), sink);
    }

    // Two such points.  Slowly varying series take a few bits per point: when both points of
    // every lane fit one word together they go out in one step.
    template <class Sink>
    __device__ __forceinline__ void add_next2(uint32_t ts0_lo, uint32_t ts0_hi, uint64_t v0,
                                              uint32_t ts1_lo, uint32_t ts1_hi, uint64_t v1, Sink& sink)
    {
        const Code a = code_next(ts0_lo, ts0_hi, v0);
        const Code b = code_next(ts1_lo, ts1_hi, v1);
        const uint32_t la = a.hlen + a.plen, lb = b.hlen + b.plen;
        if (all_lanes(la + lb <= 64u)) {
            const uint64_t ca = (a.head << a.plen) | a.pay, cb = (b.head << b.plen) | b.pay;  // plen <= 60 in both
            put((ca << lb) | cb, la + lb, sink);
        } else {
            emit(a, sink);
            emit(b, sink);
        }
    }

    // close() + getLongArray(): marker, then the partial word and the trailing
    // zero word(s); returns the image length floor(N/64)+2.
    template <class Sink>
    __device__ __forceinline__ uint32_t finish(Sink& sink)
    {
        put((0xFFFFFFFFFull << 1), 37, sink);  // '1111' + 32 ones + '0'
        sink(widx, acc); widx++;  // the partial word, or a zero word when the stream ended on a boundary
        sink(widx, 0ull); widx++;
        return widx;
    }

    // Bits written so far (N of the stream).
    __device__ __forceinline__ uint64_t bits() const { return (uint64_t)widx * 64u + nacc; }

    // getLongArray() without close(): the completed words plus the current partial word
    // (G/LongArrayOutput.java:141-145; T/EncodeGorillaTest.java:483-500).
    template <class Sink>
    __device__ __forceinline__ uint32_t snapshot(Sink& sink)
    {
        sink(widx, acc); widx++;
        return widx;
    }
};

// ---------------------------------------------------------------- decoder
// Source: src(word_index) -> word, returning 0 beyond the image.
// r0..r3 hold words widx..widx+3 of the image, `off` is the bit offset of the
// next unread bit inside r0.  One point needs at most 63 + 114 bits, i.e. r0..r2;
// r3 is look-ahead so that the load issued when the window slides is not needed
// before the following point.  All fields of a point are extracted speculatively
// with selects; the rare outcomes (end of stream, truncation, malformed) share
// one branch.
template <int FMT, int PRED = PRED_LAST>
struct Decoder {
    Dfcm dfcm;  // PRED_DFCM only
    uint64_t r0, r1, r2, r3;
    uint32_t widx, off, nwords;
    uint32_t st;
    int64_t block_ts, stored_delta;
    uint64_t stored_ts, val;
    // PRED_DFCM: the predictor's own last value.  The reference's decoder does not update the
    // predictor when a value code is the single bit '0' (G/ValueDecompressor.java:53 returns
    // predictor.predict() as it is), so after such a point the predictor lags behind the
    // values; that is reproduced, not repaired.
    uint64_t pred_last;
    uint32_t lead;
    int32_t trail;

    // 64 bits starting o bits into the window, o in [0,127]
    __device__ __forceinline__ uint64_t peek_at(uint32_t o) const
    {
        const bool up = o >= 64;
        const uint64_t a = up ? r1 : r0, b = up ? r2 : r1;
        const uint32_t s = o & 63;
        return (a << s) | ((b >> 1) >> (63 - s));
    }
    // slide the window by o bits, o in [0,191]
    template <class Src>
    __device__ __forceinline__ void advance(uint32_t o, Src& src)
    {
        const uint32_t k = o >> 6;  // 0..2 words
        off = o & 63;
        if (k) {
            const uint64_t n0 = src(widx + 4), n1 = (k == 2) ? src(widx + 5) : 0ull;
            const bool two = k == 2;
            r0 = two ? r2 : r1;
            r1 = two ? r3 : r2;
            r2 = two ? n0 : r3;
            r3 = two ? n1 : n0;
            widx += k;
        }
    }
    // A read that ends at or beyond the last bit of the image is what makes
    // LongArrayInput throw (eager load of the next word, G/LongArrayInput.java:48-52):
    // true when a field ending o bits into the window runs off the image.
    __device__ __forceinline__ bool runs_off(uint32_t o) const { return widx + (o >> 6) >= nwords; }

    template <class Src>
    __device__ __forceinline__ void begin(uint32_t nwords_, Src& src)
    {
        nwords = nwords_;
        widx = 0; off = 0;
        r0 = src(0); r1 = src(1); r2 = src(2); r3 = src(3);
        stored_ts = 0; stored_delta = 0; val = 0; pred_last = 0;
        lead = NO_WINDOW; trail = 0;
        block_ts = (int64_t)r0;
        st = runs_off(64) ? ST_TRUNCATED : 0u;  // also covers nwords == 0
        advance(64, src);
    }

    // Decodes the next pair into (ts_out, val_out).  Returns true when a pair
    // was produced; false at end of stream or on error (st != 0).
    template <class Src>
    __device__ __forceinline__ bool next(int64_t& ts_out, uint64_t& val_out, Src& src)
    {
        if (st) return false;
        if (__builtin_expect(stored_ts == 0, 0)) {
            const uint32_t d = (uint32_t)(peek_at(off) >> 37);
            if (runs_off(off + 27)) { st |= ST_TRUNCATED; return false; }
            if (d == 0x7FFFFFFu) return false;  // empty block
            const uint64_t v = peek_at(off + 27);
            if (runs_off(off + 91)) { st |= ST_TRUNCATED; return false; }
            if (PRED == PRED_DFCM) { dfcm.update(pred_last, v); pred_last = v; }  // readFirst: predictor.update(value), G/ValueDecompressor.java:26-30
            val = v;
            advance(off + 91, src);
            stored_delta = (int64_t)d;
            stored_ts = (uint64_t)block_ts + (uint64_t)d;
            ts_out = (int64_t)stored_ts; val_out = val;
            return true;
        }
        // ---- timestamp: leading ones of the next four bits select the bucket
        const uint64_t top = peek_at(off);
        uint32_t ones = (uint32_t)__clz((int)~(uint32_t)(top >> 32));  // 32 when all ones
        ones = ones > 4 ? 4u : ones;
        const uint32_t plen = ones == 0 ? 0u : ones == 1 ? 7u : ones == 2 ? 9u : ones == 3 ? 12u : 32u;
        const uint32_t clen = (ones < 4 ? ones + 1u : 4u) + plen;  // 1, 9, 12, 16, 36
        const uint32_t raw = (uint32_t)(top >> (64 - clen)) & (uint32_t)~(~0ull << plen);
        const uint32_t o1 = off + clen;
        const bool eos = ones == 4 && raw == 0xFFFFFFFFu;
        int64_t dd;
        if (FMT == FMT_V2) {
            const uint32_t z = raw + 1u;
            dd = (int64_t)(int32_t)((z >> 1) ^ (0u - (z & 1u)));  // decodeZigZag32
        } else {
            int64_t xr = (int64_t)raw;
            if (plen != 32 && xr > ((int64_t)1 << ((plen - 1) & 31))) xr -= ((int64_t)1 << plen);
            dd = (int64_t)(int32_t)(uint32_t)xr;
        }
        dd = ones ? dd : 0;
        // ---- value header: next 14 bits
        const uint32_t h = (uint32_t)(peek_at(o1) >> 50);
        const bool c0 = (h >> 13) & 1u, c1 = (h >> 12) & 1u;
        const bool fresh = c0 && c1;
        uint32_t nlead;
        int32_t ntrail;
        uint32_t hlen;
        if (FMT == FMT_V2) {
            nlead = (h >> 6) & 63u;
            ntrail = 64 - (int32_t)((h & 63u) + 1u) - (int32_t)nlead;
            hlen = 14;
        } else {
            nlead = (h >> 7) & 31u;
            uint32_t sig = (h >> 1) & 63u;
            sig = sig ? sig : 64u;
            ntrail = 64 - (int32_t)sig - (int32_t)nlead;
            hlen = 13;
        }
        nlead = fresh ? nlead : lead;
        ntrail = fresh ? ntrail : trail;
        hlen = c0 ? (c1 ? hlen : 2u) : 1u;
        const uint32_t o2 = o1 + hlen;
        const bool malformed = c0 && nlead == NO_WINDOW;
        const uint32_t n = c0 ? (uint32_t)(64 - (int32_t)nlead - ntrail) & 127u : 0u;  // 1..64 once a window exists
        const uint64_t bits = n ? (peek_at(o2) >> ((64u - n) & 63u)) : 0ull;
        const uint32_t o3 = o2 + (malformed ? 0u : n);
        if (__builtin_expect(eos || malformed || runs_off(o3), 0)) {
            // rare outcomes, in the order the reference's reads meet them
            if (runs_off(o1)) st |= ST_TRUNCATED;
            else if (eos) { /* end of stream */ }
            else if (runs_off(o2)) st |= ST_TRUNCATED;
            else if (malformed) st |= ST_MALFORMED;
            else st |= ST_TRUNCATED;
            return false;
        }
        lead = nlead;
        trail = ntrail;
        if (PRED == PRED_DFCM) {
            // G/ValueDecompressor.java:32-54: codes '10' and '11' xor the prediction and update the
            // predictor; code '0' returns the prediction and leaves the predictor as it was
            if (c0) {
                const uint64_t v = dfcm.predict(pred_last) ^ (bits << ((uint32_t)ntrail & 63u));
                dfcm.update(pred_last, v);
                pred_last = v;
                val = v;
            } else {
                val = dfcm.predict(pred_last);
            }
        } else {
            val ^= bits << ((uint32_t)ntrail & 63u);  // Java shifts by the count mod 64
        }
        stored_delta = (int64_t)((uint64_t)stored_delta + (uint64_t)dd);
        stored_ts += (uint64_t)stored_delta;
        advance(o3, src);
        ts_out = (int64_t)stored_ts; val_out = val;
        return true;
    }
};

// ---------------------------------------------------------------- window decoder
// The same parser over an explicit three-word window: the caller supplies the image
// words widx..widx+2 (from an LDS ring) before every point and takes back the new
// position.  A point spans at most 63 + 114 bits, so three words always suffice.
template <int FMT>
struct WindowDecoder {
    uint32_t widx, off, nwords;  // word index of w0, bit offset inside it, image length
    uint32_t st;
    int64_t block_ts, stored_delta;
    uint64_t stored_ts, val;
    // The value window as (significant bits, trailing zeros): the reference keeps (leading,
    // trailing) and reads 64 - leading - trailing bits (G/ValueDecompressor.java:44-52), which is
    // the significant-bit count of the header that set the window.  0 = no window yet.
    uint32_t nbits;
    int32_t trail;

    // true when a field ending o bits into the window runs off the image (the reader's
    // eager load of the following word, G/LongArrayInput.java:48-52)
    __device__ __forceinline__ bool runs_off(uint32_t o) const { return widx + (o >> 6) >= nwords; }
    __device__ __forceinline__ void advance(uint32_t o) { widx += o >> 6; off = o & 63; }

    // header word: w0 is word 0 of the image
    __device__ __forceinline__ void begin(uint32_t nwords_, uint64_t w0)
    {
        nwords = nwords_;
        widx = 0; off = 0;
        stored_ts = 0; stored_delta = 0; val = 0;
        nbits = 0; trail = 0;
        block_ts = (int64_t)w0;
        st = runs_off(64) ? ST_TRUNCATED : 0u;  // also covers nwords == 0
        advance(64);
    }

    // What one ordinary (not first) point parses to, before anything is committed.
    struct Parsed {
        uint32_t o1, o2, o3;  // window offsets after the timestamp code, the value header, the payload
        uint32_t nbits;
        int32_t trail;
        uint64_t bits;        // payload, right-aligned
        int64_t dd;           // delta of delta
        bool eos, malformed;
        bool changed;         // the value code starts with '1' (malformed = changed and no window yet)
    };

    // Branch-free parse of an ordinary point from 128 bits of stream (hi,lo) at the current position.
    // TS_ZERO: the caller has seen that the timestamp code is the single bit '0' (regular
    // cadence, the common case by far), so that part folds away.
    // `lo` is a callable giving the second 64 bits of the window: it is only called when some
    // lane's payload reaches past the first 64 (never, for well compressible data).
    template <bool TS_ZERO = false, class Lo>
    __device__ __forceinline__ Parsed parse(uint64_t hi, Lo lo) const
    {
        Parsed p;
        if (TS_ZERO) {
            p.eos = false;
            p.dd = 0;
            parse_value(p, hi, lo, 1u);
            return p;
        }
        // ---- timestamp: leading ones of the next four bits select the bucket; payload width
        // 0, 7, 9, 12, 32 and code length 1, 9, 12, 16, 36 sit in 6-bit fields of two constants
        const uint32_t top = (uint32_t)(hi >> 32);
        uint32_t ones = (uint32_t)__clz((int)~top);  // 32 when all ones
        ones = ones > 4 ? 4u : ones;
        constexpr uint32_t PLEN = 0u | (7u << 6) | (9u << 12) | (12u << 18) | (32u << 24);
        constexpr uint32_t CLEN = 1u | (9u << 6) | (12u << 12) | (16u << 18) | (36u << 24);
        const uint32_t f = ones * 6u;
        const uint32_t plen = (PLEN >> f) & 63u;
        const uint32_t clen = (CLEN >> f) & 63u;
        const uint32_t raw = (uint32_t)(hi >> (64 - clen)) & (uint32_t)~(~0ull << plen);
        p.eos = raw == 0xFFFFFFFFu;  // only the 32-bit payload can hold it
        int64_t dd;
        if (FMT == FMT_V2) {
            const uint32_t z = raw + 1u;
            dd = (int64_t)(int32_t)((z >> 1) ^ (0u - (z & 1u)));  // decodeZigZag32
        } else {
            // two's complement payloads; exactly 2^(n-1) stays positive (G/Decompressor.java:98-100)
            const uint32_t half = 1u << ((plen - 1) & 31);
            const uint32_t ext = (plen != 32 && raw > half) ? raw - (half << 1) : raw;
            dd = (int64_t)(int32_t)ext;
        }
        p.dd = ones ? dd : 0;
        parse_value(p, hi, lo, clen);
        return p;
    }

    // The value part of a point whose timestamp code took clen bits.
    template <class Lo>
    __device__ __forceinline__ void parse_value(Parsed& p, uint64_t hi, Lo lo, uint32_t clen) const
    {
        // ---- value header: the bits after the timestamp code, left-aligned in xh (clen + 14 <= 50)
        const uint32_t xh = (uint32_t)((hi << clen) >> 32);
        const bool c0 = (int32_t)xh < 0;          // '1': the value changed
        const bool fresh = xh >= 0xC0000000u;     // '11': a new window follows
        uint32_t nlead, sig, hlen;
        if (FMT == FMT_V2) {
            nlead = (xh >> 24) & 63u;             // 6 bits of leading zeros, 6 bits of (significant - 1)
            sig = ((xh >> 18) & 63u) + 1u;
            hlen = 14;
        } else {
            nlead = (xh >> 25) & 31u;             // 5 bits of leading zeros, 6 bits of significant, 0 = 64
            sig = (xh >> 19) & 63u;
            sig = sig ? sig : 64u;
            hlen = 13;
        }
        p.nbits = fresh ? sig : nbits;
        p.trail = fresh ? 64 - (int32_t)sig - (int32_t)nlead : trail;
        hlen = c0 ? (fresh ? hlen : 2u) : 1u;
        const uint32_t p2 = clen + hlen;  // <= 50: start of the payload inside (hi,lo)
        p.malformed = c0 & (p.nbits == 0);
        p.changed = c0;
        const uint32_t n = c0 ? p.nbits : 0u;  // 1..64 once a window exists
        uint64_t pay = hi << p2;
        if (any_lane(p2 + n > 64)) pay |= (lo() >> 1) >> (63 - p2);  // wave-uniform: the second half of the window is needed
        // n == 0 gives 0 through a mask, not a select: two selects in a row on the condition register cost 15 cycles extra
        const uint32_t some = (uint32_t)((int32_t)(0u - n) >> 31);  // all ones when n > 0
        const uint64_t sh = pay >> ((64u - n) & 63u);
        p.bits = ((uint64_t)((uint32_t)(sh >> 32) & some) << 32) | ((uint32_t)sh & some);
        p.o1 = off + clen;
        p.o2 = off + p2;
        p.o3 = p.o2 + n;
    }
    __device__ __forceinline__ bool rare(const Parsed& p) const { return p.eos || p.malformed || runs_off(p.o3); }
    __device__ __forceinline__ void commit(const Parsed& p)
    {
        nbits = p.nbits;
        trail = p.trail;
        val ^= p.bits << ((uint32_t)p.trail & 63u);  // Java shifts by the count mod 64
        stored_delta = (int64_t)((uint64_t)stored_delta + (uint64_t)p.dd);
        stored_ts += (uint64_t)stored_delta;
        advance(p.o3);
    }
    // 64 bits of stream starting `off` bits into word a (b is the word after it)
    __device__ __forceinline__ static uint64_t window(uint64_t a, uint64_t b, uint32_t off)
    {
        return (a << off) | ((b >> 1) >> (63 - off));
    }

    // Decodes the next pair from the window (w0,w1,w2 = words widx..widx+2).
    // Returns true when a pair was produced; false at end of stream or on error (st != 0).
    __device__ __forceinline__ bool next(uint64_t w0, uint64_t w1, uint64_t w2, int64_t& ts_out, uint64_t& val_out)
    {
        if (st) return false;
        const uint64_t hi = window(w0, w1, off), lo = window(w1, w2, off);
        if (__builtin_expect(stored_ts == 0, 0)) {
            // first(): 27-bit delta, 64-bit value
            const uint32_t d = (uint32_t)(hi >> 37);
            if (runs_off(off + 27)) { st |= ST_TRUNCATED; return false; }
            if (d == 0x7FFFFFFu) return false;  // empty block
            if (runs_off(off + 91)) { st |= ST_TRUNCATED; return false; }
            val = (hi << 27) | (lo >> 37);
            advance(off + 91);
            stored_delta = (int64_t)d;
            stored_ts = (uint64_t)block_ts + (uint64_t)d;
            ts_out = (int64_t)stored_ts; val_out = val;
            return true;
        }
        const Parsed p = parse(hi, [lo] { return lo; });
        if (__builtin_expect(rare(p), 0)) {
            // rare outcomes, in the order the reference's reads meet them
            if (runs_off(p.o1)) st |= ST_TRUNCATED;
            else if (p.eos) { /* end of stream */ }
            else if (runs_off(p.o2)) st |= ST_TRUNCATED;
            else if (p.malformed) st |= ST_MALFORMED;
            else st |= ST_TRUNCATED;
            return false;
        }
        commit(p);
        ts_out = (int64_t)stored_ts; val_out = val;
        return true;
    }
};

}  // namespace gtsc
