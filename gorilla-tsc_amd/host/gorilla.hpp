// Host-side mirror of the reference's public classes over the gtsc C ABI.
//
// Same class and method names, argument meaning and end/error behaviour as the Java library
// ("G/" = src/main/java/fi/iki/yak/ts/compression/gorilla/ of the reference):
//
//   BitOutput, BitInput      G/BitOutput.java:8-32, G/BitInput.java:8-32   the container plug-in points
//   LongArrayOutput          G/LongArrayOutput.java:44-56,141-145          getLongArray(), also before close()
//   LongArrayInput           G/LongArrayInput.java:30                      wraps a long[]
//   ByteBufferBitOutput      G/ByteBufferBitOutput.java:20-32,103          getByteBuffer()
//   ByteBufferBitInput       G/ByteBufferBitInput.java:20-27               wraps bytes
//   Predictor, LastValuePredictor, DifferentialFCM
//                            G/Predictor.java:22-37, G/predictors/*.java
//   GorillaCompressor        G/GorillaCompressor.java:28-37,49-71,84       (ts, out[, predictor]); addValue; close
//   GorillaDecompressor      G/GorillaDecompressor.java:23-35,42-49        (in[, predictor]); readPair() -> null at the end
//   Compressor/Decompressor  G/Compressor.java:24-84, G/Decompressor.java:22-42   the 1.x format
//   Pair                     G/Pair.java:8-28
//
// The codec itself runs on the GPU.  addValue() only appends to host arrays; the block is coded
// when close() is called, and several compressors attached to one Batch are coded together in
// a single library call by Batch::flush() (one foreign call per point would cost more than the
// codec).  An output asked for its contents while its compressor is still open returns the
// snapshot the reference returns there (the words written so far plus the partial word).  A
// decompressor decodes its whole block on the first readPair().
//
// Errors: what the Java classes throw unchecked (ArrayIndexOutOfBoundsException on a truncated
// array, G/LongArrayInput.java:44; BufferUnderflowException, G/ByteBufferBitInput.java:87) is
// std::out_of_range here; IllegalArgumentException is std::invalid_argument; failures of the
// library or the GPU are std::runtime_error.  There is no CPU fallback.
#ifndef GORILLA_HPP
#define GORILLA_HPP

#include <algorithm>
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

// ---------------------------------------------------------------- predictors
// The value predictor of a 2.x compressor / decompressor.  The prediction itself happens on the
// GPU; these objects only say which one (gtsc_set_predictor).
class Predictor {
public:
    virtual ~Predictor() = default;
    virtual int kind() const = 0;       // GTSC_PRED_*
    virtual int tableSize() const { return 0; }
};

class LastValuePredictor : public Predictor {  // G/predictors/LastValuePredictor.java:10-22
public:
    int kind() const override { return GTSC_PRED_LAST; }
};

class DifferentialFCM : public Predictor {  // G/predictors/DifferentialFCM.java:23-46
public:
    explicit DifferentialFCM(int size) : size_(size)
    {
        if (size <= 0) throw std::invalid_argument("Size cannot be negative and should be power of 2");
    }
    int kind() const override { return GTSC_PRED_DFCM; }
    int tableSize() const override { return size_; }

private:
    int size_;
};

class BlockEncoder;

// ---------------------------------------------------------------- outputs
// Where a compressor's bit stream goes.  The stream arrives as a word image plus its length in
// bits when the block is coded; the concrete classes present it as the reference's containers do.
class BitOutput {
public:
    virtual ~BitOutput() = default;

protected:
    // words of the stream, its length in bits, and whether the block was closed
    const std::vector<uint64_t>& stream() const { bring_up_to_date(); return words_; }
    uint64_t bits() const { bring_up_to_date(); return nbits_; }
    bool closed() const { return final_; }

private:
    friend class BlockEncoder;
    friend class Batch;
    inline void bring_up_to_date() const;
    mutable std::vector<uint64_t> words_;
    mutable uint64_t nbits_ = 0;
    mutable bool final_ = false;
    BlockEncoder* writer_ = nullptr;   // the compressor writing into this output
};

class LongArrayOutput : public BitOutput {
public:
    LongArrayOutput() = default;
    explicit LongArrayOutput(int /*initialSize*/) {}  // capacity hint of the Java class; not needed here
    // A copy of the image, as the Java method returns it: after close() the whole block, before it
    // the words completed so far plus the current partial word.
    std::vector<int64_t> getLongArray() const
    {
        const std::vector<uint64_t>& w = stream();
        return std::vector<int64_t>(reinterpret_cast<const int64_t*>(w.data()), reinterpret_cast<const int64_t*>(w.data()) + w.size());
    }
};

class ByteBufferBitOutput : public BitOutput {
public:
    ByteBufferBitOutput() = default;
    explicit ByteBufferBitOutput(int /*initialSize*/) {}
    // The bytes between 0 and the buffer's position: floor(N/8)+1 once the block is closed (flush()
    // stores the partial byte), floor(N/8) while it is open.  Byte i holds stream bits 8i..8i+7.
    std::vector<uint8_t> getByteBuffer() const
    {
        const std::vector<uint64_t>& w = stream();
        const uint64_t n = bits() / 8 + (closed() ? 1 : 0);
        std::vector<uint8_t> b(n);
        for (uint64_t i = 0; i < n; i++) b[i] = (i >> 3) < w.size() ? (uint8_t)(w[i >> 3] >> (56 - 8 * (i & 7))) : 0;
        return b;
    }
};

// ---------------------------------------------------------------- inputs
class BitInput {
public:
    virtual ~BitInput() = default;
    const std::vector<uint64_t>& words() const { return words_; }

protected:
    std::vector<uint64_t> words_;
};

class LongArrayInput : public BitInput {
public:
    explicit LongArrayInput(const std::vector<int64_t>& array) { words_.assign(array.begin(), array.end()); }
    LongArrayInput(const int64_t* array, size_t n) { words_.assign(array, array + n); }
};

class ByteBufferBitInput : public BitInput {
public:
    // The words a LongArrayInput would hold for these bytes: the bytes packed big-endian, zero-padded,
    // plus the one word the word reader loads ahead (G/LongArrayInput.java:48-52).
    explicit ByteBufferBitInput(const std::vector<uint8_t>& bytes) : ByteBufferBitInput(bytes.data(), bytes.size()) {}
    ByteBufferBitInput(const uint8_t* bytes, size_t n)
    {
        words_.assign((n + 7) / 8 + 1, 0);
        for (size_t i = 0; i < n; i++) words_[i >> 3] |= (uint64_t)bytes[i] << (56 - 8 * (i & 7));
    }
};

// ---------------------------------------------------------------- compressors
// Several blocks coded by one library call.
class Batch {
public:
    explicit Batch(Context& ctx = Context::thread_default()) : ctx_(ctx) {}
    // Codes every closed, not yet coded block attached to this batch; their outputs are filled.
    inline void flush();
    Context& context() const { return ctx_; }

private:
    friend class BlockEncoder;
    friend class BitOutput;
    inline void code(const std::vector<BlockEncoder*>& blocks, bool close);
    Context& ctx_;
    std::vector<BlockEncoder*> closed_;
};

// Common part of GorillaCompressor (2.x) and Compressor (1.x).
class BlockEncoder {
public:
    BlockEncoder(const BlockEncoder&) = delete;
    BlockEncoder& operator=(const BlockEncoder&) = delete;
    virtual ~BlockEncoder() { if (out_.writer_ == this) out_.writer_ = nullptr; }

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
    BlockEncoder(int format, int64_t blockTimestamp, BitOutput& out, int predictor, int table_size, Context& ctx, Batch* batch)
        : format_(format), predictor_(predictor), table_size_(table_size), block_ts_(blockTimestamp), out_(out), ctx_(ctx), batch_(batch)
    {
        out_.writer_ = this;
        out_.final_ = false;
    }

private:
    friend class Batch;
    friend class BitOutput;
    int format_, predictor_, table_size_;
    int64_t block_ts_;
    BitOutput& out_;
    Context& ctx_;
    Batch* batch_;
    bool closed_ = false, coded_ = false;
    std::vector<int64_t> ts_;
    std::vector<uint64_t> val_;
};

class GorillaCompressor : public BlockEncoder {
public:
    static constexpr int FIRST_DELTA_BITS = 27;
    GorillaCompressor(int64_t timestamp, BitOutput& output, Context& ctx = Context::thread_default())
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, GTSC_PRED_LAST, 0, ctx, nullptr) {}
    GorillaCompressor(int64_t timestamp, BitOutput& output, const Predictor& predictor, Context& ctx = Context::thread_default())
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, predictor.kind(), predictor.tableSize(), ctx, nullptr) {}
    GorillaCompressor(int64_t timestamp, BitOutput& output, Batch& batch)
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, GTSC_PRED_LAST, 0, batch.context(), &batch) {}
    GorillaCompressor(int64_t timestamp, BitOutput& output, const Predictor& predictor, Batch& batch)
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, predictor.kind(), predictor.tableSize(), batch.context(), &batch) {}
    static int32_t encodeZigZag32(int32_t n) { return (int32_t)(((uint32_t)n << 1) ^ (uint32_t)(n >> 31)); }
};

class Compressor : public BlockEncoder {
public:
    Compressor(int64_t timestamp, BitOutput& output, Context& ctx = Context::thread_default())
        : BlockEncoder(GTSC_FMT_V1, timestamp, output, GTSC_PRED_LAST, 0, ctx, nullptr) {}
    Compressor(int64_t timestamp, BitOutput& output, Batch& batch)
        : BlockEncoder(GTSC_FMT_V1, timestamp, output, GTSC_PRED_LAST, 0, batch.context(), &batch) {}
};

// One library call for blocks of one format and predictor: closed blocks, or snapshots of open ones.
inline void Batch::code(const std::vector<BlockEncoder*>& blocks, bool close)
{
    const uint64_t S = blocks.size();
    std::vector<int64_t> block_ts(S);
    std::vector<uint64_t> pt_off(S + 1, 0), out_off(S + 1, 0), nbits(S, 0);
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
    const BlockEncoder& first = *blocks[0];
    if (gtsc_set_predictor(ctx_.get(), first.predictor_, first.table_size_) != GTSC_OK) throw std::runtime_error(gtsc_last_error());
    const int rc = gtsc_encode_host_ex(ctx_.get(), first.format_, close ? 0u : GTSC_ENC_NO_CLOSE, S, block_ts.data(), ts.data(),
                                       val.data(), pt_off.data(), 0, words.data(), cap, out_off.data(), nbits.data());
    (void)gtsc_set_predictor(ctx_.get(), GTSC_PRED_LAST, 0);
    if (rc != GTSC_OK) throw std::runtime_error(gtsc_last_error());
    for (uint64_t s = 0; s < S; s++) {
        BitOutput& o = blocks[s]->out_;
        o.words_.assign(words.begin() + out_off[s], words.begin() + out_off[s + 1]);
        o.nbits_ = nbits[s];
        o.final_ = close;
        if (close) {
            blocks[s]->coded_ = true;
            blocks[s]->ts_.clear();
            blocks[s]->val_.clear();
        }
    }
}

inline void Batch::flush()
{
    // one call per stream format and predictor present
    while (!closed_.empty()) {
        const BlockEncoder* k = closed_.front();
        std::vector<BlockEncoder*> same, rest;
        for (BlockEncoder* b : closed_)
            (b->format_ == k->format_ && b->predictor_ == k->predictor_ && b->table_size_ == k->table_size_ ? same : rest).push_back(b);
        closed_.swap(rest);
        code(same, true);
    }
}

// An output read while its compressor has not been coded yet: an open block gives its snapshot
// (T/EncodeGorillaTest.java:483-500), a closed block waiting in a Batch makes the batch flush.
inline void BitOutput::bring_up_to_date() const
{
    BlockEncoder* w = writer_;
    if (!w || w->coded_) return;
    if (w->closed_) {
        if (w->batch_) w->batch_->flush();
        return;
    }
    Batch one(w->ctx_);
    one.code({w}, false);
}

// ---------------------------------------------------------------- decompressors
// Common part of GorillaDecompressor (2.x) and Decompressor (1.x).
class BlockDecoder {
public:
    virtual ~BlockDecoder() = default;
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
    BlockDecoder(int format, const BitInput& in, int predictor, int table_size, Context& ctx)
        : format_(format), predictor_(predictor), table_size_(table_size), in_(in), ctx_(ctx) {}

private:
    void decode()
    {
        decoded_ = true;
        const std::vector<uint64_t>& w = in_.words();
        // a point takes at least 2 bits, so 32 per word bounds the count
        const uint64_t cap = (uint64_t)w.size() * 32 + 1;
        ts_.resize(cap);
        val_.resize(cap);
        const uint64_t in_off[2] = {0, (uint64_t)w.size()};
        uint64_t npoints = 0;
        uint32_t status = 0;
        static const uint64_t none = 0;
        if (gtsc_set_predictor(ctx_.get(), predictor_, table_size_) != GTSC_OK) throw std::runtime_error(gtsc_last_error());
        const int rc = gtsc_decode_host(ctx_.get(), format_, 1, w.empty() ? &none : w.data(), in_off, ts_.data(), val_.data(), nullptr,
                                        cap, &npoints, &status);
        (void)gtsc_set_predictor(ctx_.get(), GTSC_PRED_LAST, 0);
        if (rc != GTSC_OK && rc != GTSC_ERR_STREAM) throw std::runtime_error(gtsc_last_error());
        n_ = npoints;
        truncated_ = status & GTSC_ST_TRUNCATED;
        malformed_ = status & GTSC_ST_MALFORMED;
    }

    int format_, predictor_, table_size_;
    const BitInput& in_;
    Context& ctx_;
    bool decoded_ = false, truncated_ = false, malformed_ = false;
    uint64_t n_ = 0, next_ = 0;
    std::vector<int64_t> ts_;
    std::vector<uint64_t> val_;
    Pair current_{0, 0};
};

class GorillaDecompressor : public BlockDecoder {
public:
    explicit GorillaDecompressor(const BitInput& input, Context& ctx = Context::thread_default())
        : BlockDecoder(GTSC_FMT_V2, input, GTSC_PRED_LAST, 0, ctx) {}
    GorillaDecompressor(const BitInput& input, const Predictor& predictor, Context& ctx = Context::thread_default())
        : BlockDecoder(GTSC_FMT_V2, input, predictor.kind(), predictor.tableSize(), ctx) {}
    static int32_t decodeZigZag32(int32_t n) { return (int32_t)(((uint32_t)n >> 1) ^ (uint32_t)(-(n & 1))); }
};

class Decompressor : public BlockDecoder {
public:
    explicit Decompressor(const BitInput& input, Context& ctx = Context::thread_default())
        : BlockDecoder(GTSC_FMT_V1, input, GTSC_PRED_LAST, 0, ctx) {}
};

}  // namespace gorilla
#endif  // GORILLA_HPP
