// Host-side mirror of the reference's public classes over the gtsc C ABI.
//
// Same class and method names, argument meaning and end/error behaviour as the Java library
// ("G/" = src/main/java/fi/iki/yak/ts/compression/gorilla/ of the reference):
//
//   BitOutput                G/BitOutput.java:8-32                         the reference's interface itself: writeBit, skipBit,
//                                                                          writeBits, flush -- any implementation receives the stream
//   BitInput                 G/BitInput.java:8-32                          the library's two containers (a decompressor has to hand
//                                                                          the whole image to the GPU, so it cannot pull bits one by one)
//   LongArrayOutput          G/LongArrayOutput.java:44-56,141-145          getLongArray(), also before close()
//   LongArrayInput           G/LongArrayInput.java:30                      wraps a long[]
//   ByteBufferBitOutput      G/ByteBufferBitOutput.java:20-32,103          getByteBuffer()
//   ByteBufferBitInput       G/ByteBufferBitInput.java:20-27               wraps bytes
//   Predictor, LastValuePredictor, DifferentialFCM
//                            G/Predictor.java:22-37, G/predictors/*.java   update/predict as in the reference; the GPU runs these two
//   GorillaCompressor        G/GorillaCompressor.java:28-37,49-71,84       (ts, out[, predictor]); addValue; close
//   GorillaDecompressor      G/GorillaDecompressor.java:23-35,42-49        (in[, predictor]); readPair() -> null at the end
//   Compressor/Decompressor  G/Compressor.java:24-84, G/Decompressor.java:22-42   the 1.x format
//   Pair                     G/Pair.java:8-28
//
// The codec itself runs on the GPU.  addValue() only appends to host arrays; the block is coded
// when close() is called, and several compressors attached to one Batch are coded together in
// a single library call by Batch::flush() (one foreign call per point would cost more than the
// codec).  The coded stream is then written into the compressor's BitOutput through the interface's own
// methods -- writeBits(word, 64) per whole word, the rest of the bits, and flush() as the reference's close()
// does -- so a caller's own container works as it does with the Java library.  One of the library's outputs
// asked for its contents while its compressor is still open returns the snapshot the reference returns
// there (the words written so far plus the partial word): the bits coded since the last time are written
// into it first.  A decompressor decodes its whole block on the first readPair().
// A batch of blocks spread over several GPUs: MultiBatch.
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
// The value predictor of a 2.x compressor / decompressor: the reference's interface (G/Predictor.java:22-37).
// The prediction itself happens on the GPU, which runs the reference's two predictors; another implementation
// cannot be sent there, and a compressor given one throws std::invalid_argument.  update/predict of the two
// work as in the reference for callers that use them directly.
class Predictor {
public:
    virtual ~Predictor() = default;
    virtual void update(int64_t value) = 0;
    virtual int64_t predict() = 0;
};

class LastValuePredictor : public Predictor {  // G/predictors/LastValuePredictor.java:10-22
public:
    void update(int64_t value) override { stored_ = value; }
    int64_t predict() override { return stored_; }

private:
    int64_t stored_ = 0;
};

class DifferentialFCM : public Predictor {  // G/predictors/DifferentialFCM.java:23-46
public:
    explicit DifferentialFCM(int size) : size_(size)
    {
        if (size <= 0) throw std::invalid_argument("Size cannot be negative and should be power of 2");
        uint64_t n = 1;   // rounded up to a power of two, as the Java constructor does
        while (n < (uint64_t)size) n <<= 1;
        table_.assign(n, 0);
        mask_ = (uint32_t)(n - 1);
    }
    void update(int64_t value) override
    {
        const int64_t d = (int64_t)((uint64_t)value - (uint64_t)last_value_);
        table_[last_hash_] = d;
        last_hash_ = ((last_hash_ << 5) ^ (uint32_t)(d >> 50)) & mask_;
        last_value_ = value;
    }
    int64_t predict() override { return (int64_t)((uint64_t)table_[last_hash_] + (uint64_t)last_value_); }
    int tableSize() const { return size_; }

private:
    int size_;
    std::vector<int64_t> table_;
    uint32_t mask_ = 0, last_hash_ = 0;
    int64_t last_value_ = 0;
};

// which of the GPU's predictors a Predictor object stands for
inline void gpu_predictor_of(const Predictor& p, int& kind, int& table_size)
{
    if (dynamic_cast<const LastValuePredictor*>(&p)) { kind = GTSC_PRED_LAST; table_size = 0; return; }
    if (const DifferentialFCM* d = dynamic_cast<const DifferentialFCM*>(&p)) { kind = GTSC_PRED_DFCM; table_size = d->tableSize(); return; }
    throw std::invalid_argument("only LastValuePredictor and DifferentialFCM run on the GPU");
}

class BlockEncoder;

// ---------------------------------------------------------------- outputs
// Where a compressor's bit stream goes: the reference's interface, method for method (G/BitOutput.java:8-32).
// Any implementation may be handed to a compressor; it receives the stream's bits in order (in pieces of up
// to 64 bits) and a flush() when the block is closed.
class BitOutput {
public:
    virtual ~BitOutput() = default;
    virtual void writeBit() = 0;                            // a '1' bit
    virtual void skipBit() = 0;                             // a '0' bit
    virtual void writeBits(int64_t value, int bits) = 0;    // the low `bits` bits of value, most significant first
    virtual void flush() = 0;
};

// The library's own outputs also know the compressor that writes into them, so that their contents can be
// asked for at any time.
class TrackedOutput : public BitOutput {
protected:
    inline void bring_up_to_date() const;   // has the bits coded so far written into this output

private:
    friend class BlockEncoder;
    BlockEncoder* writer_ = nullptr;
};

class LongArrayOutput : public TrackedOutput {  // G/LongArrayOutput.java:44-145
public:
    LongArrayOutput() = default;
    explicit LongArrayOutput(int /*initialSize*/) {}  // capacity hint of the Java class; not needed here
    void writeBit() override { writeBits(1, 1); }
    void skipBit() override { writeBits(0, 1); }
    void writeBits(int64_t value, int bits) override
    {
        uint64_t v = bits >= 64 ? (uint64_t)value : (uint64_t)value & ((1ull << bits) - 1);
        while (bits > 0) {
            const int take = bits < left_ ? bits : left_;
            const uint64_t piece = take == 64 ? v : (v >> (bits - take)) & ((1ull << take) - 1);
            cur_ |= take == 64 ? piece : piece << (left_ - take);
            left_ -= take;
            bits -= take;
            if (left_ == 0) flip();
        }
    }
    void flush() override { flip(); }   // stores the current word whatever it holds (G/LongArrayOutput.java:136-139)
    // A copy of the image, as the Java method returns it (G/LongArrayOutput.java:141-145): the words completed so
    // far plus the current partial word -- after close() that is the whole block and a zero word.
    std::vector<int64_t> getLongArray() const
    {
        bring_up_to_date();
        std::vector<int64_t> a(done_.begin(), done_.end());
        a.push_back((int64_t)cur_);
        return a;
    }

private:
    void flip() { done_.push_back((int64_t)cur_); cur_ = 0; left_ = 64; }
    std::vector<int64_t> done_;
    uint64_t cur_ = 0;
    int left_ = 64;
};

class ByteBufferBitOutput : public TrackedOutput {  // G/ByteBufferBitOutput.java:20-106
public:
    ByteBufferBitOutput() = default;
    explicit ByteBufferBitOutput(int /*initialSize*/) {}
    void writeBit() override { writeBits(1, 1); }
    void skipBit() override { writeBits(0, 1); }
    // (every write is masked to its width: the Java class's unmasked short write, G/ByteBufferBitOutput.java:79-83,
    // which damages its own 1.x streams, is not reproduced)
    void writeBits(int64_t value, int bits) override
    {
        for (int i = bits - 1; i >= 0; i--) {
            cur_ = (uint8_t)(cur_ | ((((uint64_t)value >> i) & 1u) << (left_ - 1)));
            if (--left_ == 0) flip();
        }
    }
    void flush() override { flip(); }   // puts the current byte whatever it holds (G/ByteBufferBitOutput.java:93-96)
    // The bytes between 0 and the buffer's position: floor(N/8)+1 once the block is closed, floor(N/8) while
    // it is open (the partial byte is not in the buffer yet).  Byte i holds stream bits 8i..8i+7.
    std::vector<uint8_t> getByteBuffer() const
    {
        bring_up_to_date();
        return done_;
    }

private:
    void flip() { done_.push_back(cur_); cur_ = 0; left_ = 8; }
    std::vector<uint8_t> done_;
    uint8_t cur_ = 0;
    int left_ = 8;
};

// ---------------------------------------------------------------- inputs
// What a decompressor reads.  The reference's BitInput is pulled bit by bit (G/BitInput.java:8-32); the GPU needs
// the whole image at once, so the inputs here are the library's two containers, which hold it.
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
    virtual ~Batch() = default;
    // Codes every closed, not yet coded block attached to this batch; their outputs receive their streams.
    inline void flush();
    Context& context() const { return ctx_; }

protected:
    // One library call: images back to back in `words` (out_off[S+1]) and each stream's length in bits.
    virtual void encode(int format, int predictor, int table_size, bool close, uint64_t S, const int64_t* block_ts, const int64_t* ts,
                        const uint64_t* val, const uint64_t* pt_off, std::vector<uint64_t>& words, uint64_t* out_off, uint64_t* nbits)
    {
        if (gtsc_set_predictor(ctx_.get(), predictor, table_size) != GTSC_OK) throw std::runtime_error(gtsc_last_error());
        const int rc = gtsc_encode_host_ex(ctx_.get(), format, close ? 0u : GTSC_ENC_NO_CLOSE, S, block_ts, ts, val, pt_off, 0, words.data(),
                                           words.size(), out_off, nbits);
        (void)gtsc_set_predictor(ctx_.get(), GTSC_PRED_LAST, 0);
        if (rc != GTSC_OK) throw std::runtime_error(gtsc_last_error());
    }

private:
    friend class BlockEncoder;
    friend class TrackedOutput;
    inline void code(const std::vector<BlockEncoder*>& blocks, bool close);
    Context& ctx_;
    std::vector<BlockEncoder*> closed_;
};

// Common part of GorillaCompressor (2.x) and Compressor (1.x).
class BlockEncoder {
public:
    BlockEncoder(const BlockEncoder&) = delete;
    BlockEncoder& operator=(const BlockEncoder&) = delete;
    virtual ~BlockEncoder() { if (tracked_ && tracked_->writer_ == this) tracked_->writer_ = nullptr; }

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
        : format_(format), predictor_(predictor), table_size_(table_size), block_ts_(blockTimestamp), out_(out),
          tracked_(dynamic_cast<TrackedOutput*>(&out)), ctx_(ctx), batch_(batch)
    {
        if (tracked_) tracked_->writer_ = this;
    }

private:
    friend class Batch;
    friend class TrackedOutput;
    // Writes stream bits [written_, nbits) of the image into the output through the reference's interface, whole
    // words where it can; `final`: the block is closed -- flush() follows, as in the reference's close()
    // (G/GorillaCompressor.java:84-89).
    void deliver(const uint64_t* words, uint64_t nbits, bool final)
    {
        while (written_ < nbits) {
            const uint64_t w = words[written_ >> 6];
            const int at = (int)(written_ & 63), room = 64 - at;
            const int n = nbits - written_ < (uint64_t)room ? (int)(nbits - written_) : room;   // bits at..at+n-1 of this word
            out_.writeBits((int64_t)(n == 64 ? w : (w >> (room - n)) & ((1ull << n) - 1)), n);
            written_ += (uint64_t)n;
        }
        if (final) out_.flush();
    }
    int format_, predictor_, table_size_;
    int64_t block_ts_;
    BitOutput& out_;
    TrackedOutput* tracked_;   // the output again, when it is one of the library's
    Context& ctx_;
    Batch* batch_;
    bool closed_ = false, coded_ = false;
    uint64_t written_ = 0;     // bits of the stream the output has received
    std::vector<int64_t> ts_;
    std::vector<uint64_t> val_;
};

class GorillaCompressor : public BlockEncoder {
public:
    static constexpr int FIRST_DELTA_BITS = 27;
    GorillaCompressor(int64_t timestamp, BitOutput& output, Context& ctx = Context::thread_default())
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, GTSC_PRED_LAST, 0, ctx, nullptr) {}
    GorillaCompressor(int64_t timestamp, BitOutput& output, const Predictor& predictor, Context& ctx = Context::thread_default())
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, kind_of(predictor), size_of(predictor), ctx, nullptr) {}
    GorillaCompressor(int64_t timestamp, BitOutput& output, Batch& batch)
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, GTSC_PRED_LAST, 0, batch.context(), &batch) {}
    GorillaCompressor(int64_t timestamp, BitOutput& output, const Predictor& predictor, Batch& batch)
        : BlockEncoder(GTSC_FMT_V2, timestamp, output, kind_of(predictor), size_of(predictor), batch.context(), &batch) {}
    static int32_t encodeZigZag32(int32_t n) { return (int32_t)(((uint32_t)n << 1) ^ (uint32_t)(n >> 31)); }

private:
    static int kind_of(const Predictor& p) { int k, n; gpu_predictor_of(p, k, n); return k; }
    static int size_of(const Predictor& p) { int k, n; gpu_predictor_of(p, k, n); return n; }
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
    encode(first.format_, first.predictor_, first.table_size_, close, S, block_ts.data(), ts.data(), val.data(), pt_off.data(), words,
           out_off.data(), nbits.data());
    for (uint64_t s = 0; s < S; s++) {
        blocks[s]->deliver(words.data() + out_off[s], nbits[s], close);
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

// One of the library's outputs read while its compressor has not been coded yet: an open block is coded as far
// as it goes and the bits not yet delivered are written (the snapshot of T/EncodeGorillaTest.java:483-500); a
// closed block waiting in a Batch makes the batch flush.
inline void TrackedOutput::bring_up_to_date() const
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

// A batch whose blocks are spread over several GPUs.  Blocks are independent (one compressor instance per block,
// no shared state: G/GorillaCompressor.java:13-26), so the batch is cut into contiguous runs of blocks with about
// equal numbers of points, one run per context, each coded by a host thread of its own through its context's
// pipeline; the images come back in the order of the blocks.  Contexts may sit on different devices or share one.
class MultiBatch : public Batch {
public:
    explicit MultiBatch(const std::vector<Context*>& contexts) : Batch(*contexts.at(0))
    {
        for (Context* c : contexts) shards_.push_back(c->get());
    }

protected:
    void encode(int format, int predictor, int table_size, bool close, uint64_t S, const int64_t* block_ts, const int64_t* ts,
                const uint64_t* val, const uint64_t* pt_off, std::vector<uint64_t>& words, uint64_t* out_off, uint64_t* nbits) override
    {
        for (gtsc_ctx* c : shards_)
            if (gtsc_set_predictor(c, predictor, table_size) != GTSC_OK) throw std::runtime_error(gtsc_last_error());
        const int rc = gtsc_encode_host_multi((int)shards_.size(), shards_.data(), format, close ? 0u : GTSC_ENC_NO_CLOSE, S, block_ts, ts, val,
                                              pt_off, 0, words.data(), words.size(), out_off, nbits);
        for (gtsc_ctx* c : shards_) (void)gtsc_set_predictor(c, GTSC_PRED_LAST, 0);
        if (rc != GTSC_OK) throw std::runtime_error(gtsc_last_error());
    }

private:
    std::vector<gtsc_ctx*> shards_;
};

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
        : BlockDecoder(GTSC_FMT_V2, input, kind_of(predictor), size_of(predictor), ctx) {}
    static int32_t decodeZigZag32(int32_t n) { return (int32_t)(((uint32_t)n >> 1) ^ (uint32_t)(-(n & 1))); }

private:
    static int kind_of(const Predictor& p) { int k, n; gpu_predictor_of(p, k, n); return k; }
    static int size_of(const Predictor& p) { int k, n; gpu_predictor_of(p, k, n); return n; }
};

class Decompressor : public BlockDecoder {
public:
    explicit Decompressor(const BitInput& input, Context& ctx = Context::thread_default())
        : BlockDecoder(GTSC_FMT_V1, input, GTSC_PRED_LAST, 0, ctx) {}
};

}  // namespace gorilla
#endif  // GORILLA_HPP
