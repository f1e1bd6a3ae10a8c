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

__device__ __forceinline__ uint32_t clz64(uint64_t x) { return (uint32_t)__builtin_clzll(x); }  // x != 0
__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }  // x != 0

// ---------------------------------------------------------------- encoder
// Pending bits sit right-aligned in `acc` (exactly `nacc` low bits, zeros above).
// A completed 64-bit word is handed to the sink: sink(word_index, word).
template <int FMT>
struct Encoder {
    uint64_t acc;
    uint32_t nacc;
    uint32_t widx;
    uint64_t stored_ts;   // 0 = no point yet (G/GorillaCompressor.java:50,65)
    int64_t stored_delta; // v2 keeps an int (sign-extended here), v1 a long
    uint64_t prev;
    uint32_t lead, trail;

    // n in [0,64]; bits of v above n must be zero.
    template <class Sink>
    __device__ __forceinline__ void put(uint64_t v, uint32_t n, Sink& sink)
    {
        uint32_t total = nacc + n;
        if (total < 64) {
            acc = (acc << (n & 63)) | v;
            nacc = total;
        } else {
            uint32_t spill = total - 64;            // 0..63
            uint64_t word = (acc << ((64 - nacc) & 63)) | (v >> spill);
            sink(widx, word);
            widx++;
            acc = v & ((1ull << spill) - 1ull);
            nacc = spill;
        }
    }

    template <class Sink>
    __device__ __forceinline__ void begin(int64_t block_ts, Sink& sink)
    {
        acc = 0; nacc = 0; widx = 0;
        stored_ts = 0; stored_delta = 0; prev = 0;
        lead = NO_WINDOW; trail = 0;
        put((uint64_t)block_ts, 64, sink);  // addHeader, G/GorillaCompressor.java:39-41
    }

    template <class Sink>
    __device__ __forceinline__ void add(int64_t block_ts, int64_t ts, uint64_t v, Sink& sink)
    {
        if (stored_ts == 0) {
            // writeFirst: 27 low bits of the delta, then the raw value
            uint64_t d = (uint64_t)ts - (uint64_t)block_ts;
            stored_delta = (FMT == FMT_V2) ? (int64_t)(int32_t)(uint32_t)d : (int64_t)d;
            stored_ts = (uint64_t)ts;
            prev = v;
            put(d & 0x7FFFFFFull, 27, sink);
            put(v, 64, sink);
            return;
        }
        // ---- timestamp: code `tcode` of `tlen` bits
        uint64_t tcode;
        uint32_t tlen;
        if (FMT == FMT_V2) {
            // int arithmetic: only the low 32 bits of the timestamps take part
            int32_t nd = (int32_t)((uint32_t)ts - (uint32_t)stored_ts);
            int32_t dod = (int32_t)((uint32_t)nd - (uint32_t)(int32_t)stored_delta);
            uint32_t z = (((uint32_t)dod << 1) ^ (uint32_t)(dod >> 31)) - 1u;  // zigzag - 1
            uint32_t need = 32u - (uint32_t)__clz((int)z);  // bits required; 0 when z == 0
            if (dod == 0)                     { tcode = 0; tlen = 1; }
            else if (need >= 1 && need <= 7)  { tcode = 0x100u | z; tlen = 9; }
            else if (need >= 8 && need <= 9)  { tcode = 0xC00u | z; tlen = 12; }
            else if (need >= 10 && need <= 12){ tcode = 0xE000u | z; tlen = 16; }
            else                              { tcode = (0xFull << 32) | z; tlen = 36; }  // need 0 (dod -1) or > 12
            stored_delta = nd;
        } else {
            int64_t nd = (int64_t)((uint64_t)ts - stored_ts);
            int64_t dod = (int64_t)((uint64_t)nd - (uint64_t)stored_delta);
            if (dod == 0)                         { tcode = 0; tlen = 1; }
            else if (dod >= -63 && dod <= 64)     { tcode = 0x100u | ((uint64_t)dod & 0x7F); tlen = 9; }
            else if (dod >= -255 && dod <= 256)   { tcode = 0xC00u | ((uint64_t)dod & 0x1FF); tlen = 12; }
            else if (dod >= -2047 && dod <= 2048) { tcode = 0xE000u | ((uint64_t)dod & 0xFFF); tlen = 16; }
            else                                  { tcode = (0xFull << 32) | ((uint64_t)dod & 0xFFFFFFFFull); tlen = 36; }
            stored_delta = nd;
        }
        stored_ts = (uint64_t)ts;
        // ---- value: header `vh` of `vhlen` bits, payload `pay` of `plen` bits
        uint64_t x = prev ^ v;
        prev = v;
        uint32_t vh, vhlen, plen;
        uint64_t pay;
        if (x == 0) {
            vh = 0; vhlen = 1; pay = 0; plen = 0;
        } else {
            uint32_t lz = clz64(x), tz = ctz64(x);
            if (FMT == FMT_V1 && lz >= 32) lz = 31;  // G/Compressor.java:132-134
            if (lz >= lead && tz >= trail) {
                vh = 2; vhlen = 2;
                pay = x >> trail; plen = 64 - lead - trail;
            } else {
                uint32_t sig = 64 - lz - tz;
                if (FMT == FMT_V2) { vh = (3u << 12) | (lz << 6) | (sig - 1); vhlen = 14; }
                else               { vh = (3u << 11) | (lz << 6) | (sig & 63); vhlen = 13; }
                pay = x >> tz; plen = sig;
                lead = lz; trail = tz;
            }
        }
        put((tcode << vhlen) | vh, tlen + vhlen, sink);  // <= 36 + 14 bits
        put(pay, plen, sink);
    }

    // close() + getLongArray(): marker, then the partial word and the trailing
    // zero word(s); returns the image length floor(N/64)+2.
    template <class Sink>
    __device__ __forceinline__ uint32_t finish(Sink& sink)
    {
        put((0xFFFFFFFFFull << 1), 37, sink);  // '1111' + 32 ones + '0'
        if (nacc) { sink(widx, acc << (64 - nacc)); widx++; }
        else      { sink(widx, 0ull); widx++; }
        sink(widx, 0ull); widx++;
        return widx;
    }
};

// ---------------------------------------------------------------- decoder
// Source: src(word_index) -> word, returning 0 beyond the image.
// `w0` holds word widx, `w1` word widx+1, `off` the bit offset inside w0.
template <int FMT>
struct Decoder {
    uint64_t w0, w1;
    uint32_t widx, off, nwords;
    uint32_t st;
    int64_t block_ts, stored_delta;
    uint64_t stored_ts, val;
    uint32_t lead;
    int32_t trail;

    __device__ __forceinline__ uint64_t peek() const
    {
        return off ? ((w0 << off) | (w1 >> (64 - off))) : w0;
    }
    template <class Src>
    __device__ __forceinline__ void skip(uint32_t n, Src& src)  // n in [0,64]
    {
        off += n;
        if (off >= 64) {
            off -= 64;
            widx++;
            w0 = w1;
            w1 = src(widx + 1);
        }
    }
    // a read that ends at or beyond the last bit of the image is what makes
    // LongArrayInput throw (eager load of the next word, G/LongArrayInput.java:48-52)
    __device__ __forceinline__ bool ran_off() const { return widx >= nwords; }

    template <class Src>
    __device__ __forceinline__ void begin(uint32_t nwords_, Src& src)
    {
        nwords = nwords_;
        st = nwords ? 0u : ST_TRUNCATED;
        widx = 0; off = 0;
        w0 = src(0); w1 = src(1);
        stored_ts = 0; stored_delta = 0; val = 0;
        lead = NO_WINDOW; trail = 0;
        block_ts = (int64_t)peek();
        skip(64, src);
        if (ran_off()) st |= ST_TRUNCATED;
    }

    // Decodes the next pair into (ts_out, val_out).  Returns true when a pair
    // was produced; false at end of stream or on error (st != 0).
    template <class Src>
    __device__ __forceinline__ bool next(int64_t& ts_out, uint64_t& val_out, Src& src)
    {
        if (st) return false;
        uint64_t top = peek();
        if (stored_ts == 0) {
            uint32_t d = (uint32_t)(top >> 37);
            skip(27, src);
            if (ran_off()) { st |= ST_TRUNCATED; return false; }
            if (d == 0x7FFFFFFu) return false;  // empty block
            val = peek();
            skip(64, src);
            if (ran_off()) { st |= ST_TRUNCATED; return false; }
            stored_delta = (int64_t)d;
            stored_ts = (uint64_t)block_ts + (uint64_t)d;
            ts_out = (int64_t)stored_ts; val_out = val;
            return true;
        }
        // ---- timestamp: leading ones of the next four bits select the bucket
        uint32_t hi = (uint32_t)(top >> 32);
        uint32_t ones = (uint32_t)__clz((int)~hi);  // 32 when hi is all ones
        ones = ones > 4 ? 4 : ones;
        int64_t dd = 0;
        if (ones == 0) {
            skip(1, src);
        } else {
            uint32_t plen = ones == 1 ? 7u : ones == 2 ? 9u : ones == 3 ? 12u : 32u;
            uint32_t clen = (ones < 4 ? ones + 1 : 4u) + plen;  // 9, 12, 16, 36
            uint32_t raw = (uint32_t)(top >> (64 - clen)) & (plen == 32 ? 0xFFFFFFFFu : ((1u << plen) - 1u));
            skip(clen, src);
            if (ran_off()) { st |= ST_TRUNCATED; return false; }
            if (plen == 32 && raw == 0xFFFFFFFFu) return false;  // end of stream
            if (FMT == FMT_V2) {
                uint32_t z = raw + 1u;
                dd = (int64_t)(int32_t)((z >> 1) ^ (0u - (z & 1u)));  // decodeZigZag32
            } else {
                int64_t xr = (int64_t)raw;
                if (plen != 32 && xr > ((int64_t)1 << (plen - 1))) xr -= ((int64_t)1 << plen);
                dd = (int64_t)(int32_t)(uint32_t)xr;
            }
        }
        if (ran_off()) { st |= ST_TRUNCATED; return false; }
        stored_delta = (int64_t)((uint64_t)stored_delta + (uint64_t)dd);
        stored_ts += (uint64_t)stored_delta;
        // ---- value
        top = peek();
        uint32_t h = (uint32_t)(top >> 50);  // next 14 bits
        if ((h >> 13) == 0) {
            skip(1, src);
        } else {
            uint32_t hlen = 2;
            if ((h >> 12) & 1u) {
                if (FMT == FMT_V2) {
                    lead = (h >> 6) & 63u;
                    uint32_t sig = (h & 63u) + 1u;
                    trail = 64 - (int32_t)sig - (int32_t)lead;
                    hlen = 14;
                } else {
                    lead = (h >> 7) & 31u;
                    uint32_t sig = (h >> 1) & 63u;
                    if (sig == 0) sig = 64;
                    trail = 64 - (int32_t)sig - (int32_t)lead;
                    hlen = 13;
                }
            }
            skip(hlen, src);
            if (ran_off()) { st |= ST_TRUNCATED; return false; }
            if (lead == NO_WINDOW) { st |= ST_MALFORMED; return false; }
            uint32_t n = (uint32_t)(64 - (int32_t)lead - trail);  // 1..64 once a window exists
            uint64_t bits = peek() >> (64 - n);
            skip(n, src);
            val ^= bits << ((uint32_t)trail & 63u);  // Java shifts by the count mod 64
        }
        if (ran_off()) { st |= ST_TRUNCATED; return false; }
        ts_out = (int64_t)stored_ts; val_out = val;
        return true;
    }
};

}  // namespace gtsc
