// Host-side mirror of the reference's public classes over the gtsc C ABI.
//
// Same class and method names, argument meaning and end/error behaviour as the Java library
// ("G/" = src/main/java/fi/iki/yak/ts/compression/gorilla/ of the reference):
//
//   LongArrayOutput        G/LongArrayOutput.java:44-56,141-145     getLongArray()
//   LongArrayInput         G/LongArrayInput.java:30                 wraps a long[]
//   GorillaCompressor      G/GorillaCompressor.java:28-37,49-71,84  addValue(long,double|long), close()
//   GorillaDecompressor    G/GorillaDecompressor.java:23-35,42-49   readPair() -> null at the end
//   Compressor/Decompressor (1.x format)  G/Compressor.java:24-84, G/Decompressor.java:22-42
//   Pair                   G/Pair.java:8-28
//
// The codec itself runs on the GPU.  addValue() only appends to host arrays; the block is coded
// when close() is called, and several compressors attached to one Batch are coded together in
// a single gtsc_encode_host call by Batch::flush() (one foreign call per point would cost more
// than the codec).  A decompressor decodes its whole block on the first readPair().
//
// Errors: what the Java classes throw unchecked (ArrayIndexOutOfBoundsException on a truncated
// array, G/LongArrayInput.java:44) is std::out_of_range here; failures of the library or the GPU
// are std::runtime_error.  There is no CPU fallback.
#ifndef GORILLA_HPP
#define GORILLA_HPP

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "gtsc.h"

namespace gorilla {

class Context {
public:
    explicit Context(int device = -1)
    {
        if (gtsc_create(device, &ctx_) != GTSC_OK) throw std::runtime_error(gtsc_last_error());
    }
    ~Context() { if (ctx_) gtsc_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    gtsc_ctx* get() const { return ctx_; }
    // one lazily created context per host thread, for objects constructed without one
    static Context& thread_default()
    {
        thread_local Context c;
        return c;
    }

private:
    gtsc_ctx* ctx_ = nullptr;
};

struct Pair {
    Pair(int64_t timestamp, int64_t value) : timestamp_(timestamp), value_(value) {}
    int64_t getTimestamp() const { return timestamp_; }
    int64_t getLongValue() const { return value_; }
    double getDoubleValue() const  // Double.longBitsToDouble
    {
        double d;
        std::memcpy(&d, &value_, sizeof d);
        return d;
    }

private:
    int64_t timestamp_, value_;
};

// Receives a block's word image.
class LongArrayOutput {
public:
    LongArrayOutput() = default;
    explicit LongArrayOutput(int /*initialSize*/) {}  // capacity hint of the Java class; not needed here
    // The image as the Java method returns it; empty until the owning compressor was closed
    // (and, in a Batch, flushed).
    const std::vector<int64_t>& getLongArray() const { return words_; }

private:
    friend class BlockEncoder;
    friend class Batch;
    std::vector<int64_t> words_;
};

class LongArrayInput {
public:
    explicit LongArrayInput(std::vector<int64_t> array) : words_(std::move(array)) {}
    LongArrayInput(const int64_t* array, size_t n) : words_(array, array + n) {}
    const std::vector<int64_t>& array() const { return words_; }

private:
    std::vector<int64_t> words_;
};

class BlockEncoder;

// Several blocks coded by one library call.
class Batch {
public:
    explicit Batch(Context& ctx = Context::thread_default()) : ctx_(ctx) {}
    // Codes every closed, not yet coded block attached to this batch; their outputs are filled.
    inline void flush();
    Context& context() const { return ctx_; }

private:
    friend class BlockEncoder;
    Context& ctx_;
    std::vector<BlockEncoder*> closed_;
};

// Common part of GorillaCompressor (2.x) and Compressor (1.x).
class BlockEncoder {
public:
    void addValue(int64_t timestamp, int64_t value)
    {
        if (closed_) throw std::logic_error("addValue after close");
        ts_.push_back(timestamp);
        val_.push_back((uint64_t)value);
    }
    void addValue(int64_t timestamp, double value)  // Double.doubleToRawLongBits
    {
        int64_t bits;
        std::memcpy(&bits, &value, sizeof bits);
        addValue(timestamp, bits);
    }
    // Writes the end marker; without a Batch the block is coded right here.
    void close()
    {
        if (closed_) return;
        closed_ = true;
        if (batch_) {
            batch_->closed_.push_back(this);
        } else {
            Batch one(ctx_);
            one.closed_.push_back(this);
            one.flush();
        }
    }

protected:
    BlockEncoder(int format, int64_t blockTimestamp, LongArrayOutput& out, Context& ctx, Batch* batch)
        : format_(format), block_ts_(blockTimestamp), out_(out), ctx_(ctx), batch_(batch) {}

private:
    friend class Batch;
    int format_;
    int64_t block_ts_;
    LongArrayOutput& out_;
    Context& ctx_;
    Batch* batch_;
    bool closed_ = false;
    std::vector<int64_t> ts_;
    std::vector<uint64_t> val_;
};

class GorillaCompressor : public BlockEncoder {
public:
    static constexpr int FIRST_DELTA_BITS = 27;
    GorillaCompressor(int64_t timestamp, LongArrayOutput& output, Context& ctx = Context::thread_default())
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, ctx, nullptr) {}
    GorillaCompressor(int64_t timestamp, LongArrayOutput& output, Batch& batch)
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, batch.context(), &batch) {}
    static int32_t encodeZigZag32(int32_t n) { return (int32_t)(((uint32_t)n << 1) ^ (uint32_t)(n >> 31)); }
};

class Compressor : public BlockEncoder {
public:
    Compressor(int64_t timestamp, LongArrayOutput& output, Context& ctx = Context::thread_default())
        : BlockEncoder(GTSC_FMT_V1, timestamp, output, ctx, nullptr) {}
    Compressor(int64_t timestamp, LongArrayOutput& output, Batch& batch)
        : BlockEncoder(GTSC_FMT_V1, timestamp, output, batch.context(), &batch) {}
};

inline void Batch::flush()
{
    if (closed_.empty()) return;
    // one call per stream format present
    for (int format : {GTSC_FMT_V2, GTSC_FMT_V1}) {
        std::vector<BlockEncoder*> blocks;
        for (BlockEncoder* b : closed_) if (b->format_ == format) blocks.push_back(b);
        if (blocks.empty()) continue;
        const uint64_t S = blocks.size();
        std::vector<int64_t> block_ts(S);
        std::vector<uint64_t> pt_off(S + 1, 0), out_off(S + 1, 0);
        uint64_t cap = 0;
        for (uint64_t s = 0; s < S; s++) {
            block_ts[s] = blocks[s]->block_ts_;
            pt_off[s + 1] = pt_off[s] + blocks[s]->ts_.size();
            cap += gtsc_max_words(blocks[s]->ts_.size());
        }
        std::vector<int64_t> ts(pt_off[S] ? pt_off[S] : 1);
        std::vector<uint64_t> val(ts.size());
        for (uint64_t s = 0; s < S; s++) {
            std::copy(blocks[s]->ts_.begin(), blocks[s]->ts_.end(), ts.begin() + pt_off[s]);
            std::copy(blocks[s]->val_.begin(), blocks[s]->val_.end(), val.begin() + pt_off[s]);
        }
        std::vector<uint64_t> words(cap);
        const int rc = gtsc_encode_host(ctx_.get(), format, S, block_ts.data(), ts.data(), val.data(), pt_off.data(), 0,
                                        words.data(), cap, out_off.data());
        if (rc != GTSC_OK) throw std::runtime_error(gtsc_last_error());
        for (uint64_t s = 0; s < S; s++) {
            const uint64_t* w = words.data() + out_off[s];
            blocks[s]->out_.words_.assign(reinterpret_cast<const int64_t*>(w),
                                          reinterpret_cast<const int64_t*>(w + (out_off[s + 1] - out_off[s])));
            blocks[s]->ts_.clear();
            blocks[s]->val_.clear();
        }
    }
    closed_.clear();
}

// Common part of GorillaDecompressor (2.x) and Decompressor (1.x).
class BlockDecoder {
public:
    // The next pair, or nullptr at the end of the block (Java: null).  The pointer stays valid
    // until the next call.
    const Pair* readPair()
    {
        if (!decoded_) decode();
        if (next_ >= n_) {
            if (truncated_) throw std::out_of_range("compressed array ended inside a value");
            if (malformed_) throw std::out_of_range("value window used before it was defined");
            return nullptr;
        }
        current_ = Pair(ts_[next_], (int64_t)val_[next_]);
        next_++;
        return &current_;
    }

protected:
    BlockDecoder(int format, const LongArrayInput& in, Context& ctx) : format_(format), in_(in), ctx_(ctx) {}

private:
    void decode()
    {
        decoded_ = true;
        const std::vector<int64_t>& w = in_.array();
        // a point takes at least 2 bits, so 32 per word bounds the count
        const uint64_t cap = (uint64_t)w.size() * 32 + 1;
        ts_.resize(cap);
        val_.resize(cap);
        const uint64_t in_off[2] = {0, (uint64_t)w.size()};
        uint64_t npoints = 0;
        uint32_t status = 0;
        static const int64_t none = 0;
        const uint64_t* words = reinterpret_cast<const uint64_t*>(w.empty() ? &none : w.data());
        const int rc = gtsc_decode_host(ctx_.get(), format_, 1, words, in_off, ts_.data(), val_.data(), nullptr, cap,
                                        &npoints, &status);
        if (rc != GTSC_OK && rc != GTSC_ERR_STREAM) throw std::runtime_error(gtsc_last_error());
        n_ = npoints;
        truncated_ = status & GTSC_ST_TRUNCATED;
        malformed_ = status & GTSC_ST_MALFORMED;
    }

    int format_;
    const LongArrayInput& in_;
    Context& ctx_;
    bool decoded_ = false, truncated_ = false, malformed_ = false;
    uint64_t n_ = 0, next_ = 0;
    std::vector<int64_t> ts_;
    std::vector<uint64_t> val_;
    Pair current_{0, 0};
};

class GorillaDecompressor : public BlockDecoder {
public:
    explicit GorillaDecompressor(const LongArrayInput& input, Context& ctx = Context::thread_default())
        : BlockDecoder(GTSC_FMT_V2, input, ctx) {}
    static int32_t decodeZigZag32(int32_t n) { return (int32_t)(((uint32_t)n >> 1) ^ (uint32_t)(-(n & 1))); }
};

class Decompressor : public BlockDecoder {
public:
    explicit Decompressor(const LongArrayInput& input, Context& ctx = Context::thread_default())
        : BlockDecoder(GTSC_FMT_V1, input, ctx) {}
};

}  // namespace gorilla
#endif  // GORILLA_HPP
