// Drives the C++ mirror of the reference classes the way the reference's own tests drive the
// Java classes (construct, addValue..., close, getLongArray; then readPair until null), on
// series read from a text file, and writes what it got for the caller to compare.
//
//   mirror_test <in.txt> <out.txt>
//   mirror_test shapes <outdir>     the shapes of three of the reference's tests (see shapes())
//   mirror_test plug <in.txt> <out.txt>   the same input through a caller's own BitOutput and over two contexts (see plug())
// in.txt : "<nseries>" then per series "<format> <block_ts> <npoints>" and npoints lines
//          "<ts> <value bits as hex>"
// out.txt: per series "<nwords>" + words (hex), then "<npairs>" + pairs "<ts> <hex>"
#include <cinttypes>
#include <cstdio>
#include <memory>
#include <string>

#include "gorilla.hpp"

// The value of point i in the large shapes: the reference's tests use i * Math.random(); here a
// multiplicative hash stands in for the random number, so that the checker can make the same series.
static double shape_value(uint32_t i) { return (double)i * ((double)(uint32_t)(i * 2654435761u) / 4294967296.0); }

static void write_file(const std::string& path, const void* p, size_t bytes)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, 1, bytes, f) != bytes) throw std::runtime_error("cannot write " + path);
    std::fclose(f);
}

// Three shapes of the reference's test suite (T/ = src/test/java/fi/iki/yak/ts/compression/gorilla/):
//   testCopyFlush                          T/EncodeGorillaTest.java:483-500  getLongArray() before close() is decodable
//   testEncodeLargeAmountOfDataOldBuffer   T/EncodeGorillaTest.java:420-462  100,000 points through ByteBufferBitOutput/Input
//   testDifferentialFCM                    T/EncodeGorillaTest.java:550-589  100,000 points with DifferentialFCM(1024) on both sides
// Each checks here what the reference's test asserts, and leaves images and decoded values in
// <outdir> for the caller to compare with the oracle.
static int shapes(const std::string& dir)
{
    const int64_t B = 1500400800000LL;
    {   // ---- testCopyFlush
        gorilla::LongArrayOutput out;
        gorilla::GorillaCompressor c(B, out);
        c.addValue(B + 1, 1.0);
        c.addValue(B + 2, 1.0);
        const std::vector<int64_t> snap = out.getLongArray();   // no close()
        gorilla::LongArrayInput in(snap);
        gorilla::GorillaDecompressor d(in);
        const gorilla::Pair* p = d.readPair();
        if (!p || p->getTimestamp() != B + 1 || p->getDoubleValue() != 1.0) throw std::runtime_error("copyflush: first pair");
        p = d.readPair();
        if (!p || p->getTimestamp() != B + 2 || p->getDoubleValue() != 1.0) throw std::runtime_error("copyflush: second pair");
        write_file(dir + "/copyflush_words.bin", snap.data(), snap.size() * 8);
        // the block can still be added to and closed afterwards
        c.addValue(B + 3, 2.0);
        c.close();
        const std::vector<int64_t> whole = out.getLongArray();
        write_file(dir + "/copyflush_closed_words.bin", whole.data(), whole.size() * 8);
    }
    const uint32_t N = 100000;
    {   // ---- testEncodeLargeAmountOfDataOldBuffer
        gorilla::ByteBufferBitOutput out;
        gorilla::GorillaCompressor c(B, out);
        for (uint32_t i = 0; i < N; i++) c.addValue(B + 60 * (int64_t)(i + 1), shape_value(i));
        c.close();
        const std::vector<uint8_t> bytes = out.getByteBuffer();
        write_file(dir + "/oldbuffer_bytes.bin", bytes.data(), bytes.size());
        gorilla::ByteBufferBitInput in(bytes);
        gorilla::GorillaDecompressor d(in);
        for (uint32_t i = 0; i < N; i++) {
            const gorilla::Pair* p = d.readPair();
            if (!p || p->getTimestamp() != B + 60 * (int64_t)(i + 1) || p->getDoubleValue() != shape_value(i))
                throw std::runtime_error("oldbuffer: pair " + std::to_string(i));
        }
        if (d.readPair()) throw std::runtime_error("oldbuffer: no end of stream");
    }
    {   // ---- testDifferentialFCM
        gorilla::LongArrayOutput out;
        gorilla::DifferentialFCM enc_pred(1024), dec_pred(1024);
        gorilla::GorillaCompressor c(B, out, enc_pred);
        for (uint32_t i = 0; i < N; i++) c.addValue(B + 60 * (int64_t)(i + 1), shape_value(i));
        c.close();
        const std::vector<int64_t> words = out.getLongArray();
        write_file(dir + "/dfcm_words.bin", words.data(), words.size() * 8);
        gorilla::LongArrayInput in(words);
        gorilla::GorillaDecompressor d(in, dec_pred);
        std::vector<int64_t> vals;
        while (const gorilla::Pair* p = d.readPair()) {
            if (p->getTimestamp() != B + 60 * (int64_t)(vals.size() + 1)) throw std::runtime_error("dfcm: timestamp");
            vals.push_back(p->getLongValue());
        }
        if (vals.size() != N) throw std::runtime_error("dfcm: pair count");
        write_file(dir + "/dfcm_values.bin", vals.data(), vals.size() * 8);
        bool rejected = false;
        try { gorilla::DifferentialFCM bad(0); } catch (const std::invalid_argument&) { rejected = true; }
        if (!rejected) throw std::runtime_error("dfcm: size 0 accepted");
    }
    std::printf("shapes ok\n");
    return 0;
}

// A container of the caller's own: it implements the reference's BitOutput interface (G/BitOutput.java:8-32) and
// simply records what it is told, one character per bit, 'F' per flush() -- what a user's container (or the
// reference jar's own LongArrayOutput) would see.
class RecordingBitOutput : public gorilla::BitOutput {
public:
    void writeBit() override { log += '1'; calls++; }
    void skipBit() override { log += '0'; calls++; }
    void writeBits(int64_t value, int bits) override
    {
        if (bits < 1 || bits > 64) throw std::logic_error("writeBits: bit count out of range");
        for (int i = bits - 1; i >= 0; i--) log += (char)('0' + (((uint64_t)value >> i) & 1u));
        calls++;
    }
    void flush() override { log += 'F'; calls++; }
    std::string log;
    size_t calls = 0;
};

// A predictor of the caller's own: the GPU cannot run it, and a compressor says so.
class MeanPredictor : public gorilla::Predictor {
public:
    void update(int64_t value) override { sum += value; n++; }
    int64_t predict() override { return n ? sum / n : 0; }
    int64_t sum = 0, n = 0;
};

// The series of in.txt (a) each through a RecordingBitOutput, including a look at an open block's stream half way
// (the recorder then holds a prefix of the final stream), and (b) all together through a MultiBatch over two contexts
// on the current device, standing in for two GPUs.  out.txt: per series the recorded characters, then per series the
// word image from the MultiBatch.
static int plug(const char* in_path, const char* out_path)
{
    FILE* in = std::fopen(in_path, "r");
    FILE* out = std::fopen(out_path, "w");
    if (!in || !out) { std::perror("open"); return 2; }
    size_t S;
    if (std::fscanf(in, "%zu", &S) != 1) return 2;
    gorilla::Context c0, c1;
    gorilla::MultiBatch multi({&c0, &c1});
    std::vector<std::unique_ptr<gorilla::LongArrayOutput>> outs;
    std::vector<std::unique_ptr<gorilla::BlockEncoder>> encs;
    for (size_t s = 0; s < S; s++) {
        int fmt; int64_t block; size_t P;
        if (std::fscanf(in, "%d %" SCNd64 " %zu", &fmt, &block, &P) != 3) return 2;
        RecordingBitOutput rec;
        std::unique_ptr<gorilla::BlockEncoder> enc;
        if (fmt == GTSC_FMT_V2) enc.reset(new gorilla::GorillaCompressor(block, rec));
        else                    enc.reset(new gorilla::Compressor(block, rec));
        outs.emplace_back(new gorilla::LongArrayOutput());
        if (fmt == GTSC_FMT_V2) encs.emplace_back(new gorilla::GorillaCompressor(block, *outs.back(), multi));
        else                    encs.emplace_back(new gorilla::Compressor(block, *outs.back(), multi));
        for (size_t i = 0; i < P; i++) {
            int64_t t; uint64_t v;
            if (std::fscanf(in, "%" SCNd64 " %" SCNx64, &t, &v) != 2) return 2;
            enc->addValue(t, (int64_t)v);
            encs.back()->addValue(t, (int64_t)v);
        }
        enc->close();
        encs.back()->close();
        if (rec.log.empty() || rec.log.back() != 'F') throw std::runtime_error("no flush() at close()");
        std::fprintf(out, "%s\n", rec.log.c_str());
    }
    multi.flush();
    for (size_t s = 0; s < S; s++) {
        const std::vector<int64_t> w = outs[s]->getLongArray();
        std::fprintf(out, "%zu", w.size());
        for (int64_t x : w) std::fprintf(out, " %016" PRIx64, (uint64_t)x);
        std::fprintf(out, "\n");
    }
    bool rejected = false;
    try {
        MeanPredictor mp;
        gorilla::LongArrayOutput o;
        gorilla::GorillaCompressor c(1500400800000LL, o, mp);
    } catch (const std::invalid_argument&) { rejected = true; }
    // the library's two predictors answer update/predict as the reference's do
    gorilla::DifferentialFCM d(8);
    for (int64_t v : {0, 10, 20, 30}) d.update(v);
    gorilla::LastValuePredictor l;
    l.update(7);
    std::fprintf(out, "predictors %d %" PRId64 " %" PRId64 "\n", (int)rejected, d.predict(), l.predict());
    std::fclose(out);
    std::fclose(in);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 4 && std::string(argv[1]) == "plug") {
        try { return plug(argv[2], argv[3]); } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc == 3 && std::string(argv[1]) == "shapes") {
        try { return shapes(argv[2]); } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc != 3) { std::fprintf(stderr, "usage: %s in out\n", argv[0]); return 2; }
    FILE* in = std::fopen(argv[1], "r");
    FILE* out = std::fopen(argv[2], "w");
    if (!in || !out) { std::perror("open"); return 2; }
    try {
        size_t S;
        if (std::fscanf(in, "%zu", &S) != 1) return 2;
        gorilla::Batch batch;
        std::vector<std::unique_ptr<gorilla::LongArrayOutput>> outs;
        std::vector<std::unique_ptr<gorilla::BlockEncoder>> encs;
        std::vector<int> fmts;
        for (size_t s = 0; s < S; s++) {
            int fmt; int64_t block; size_t P;
            if (std::fscanf(in, "%d %" SCNd64 " %zu", &fmt, &block, &P) != 3) return 2;
            outs.emplace_back(new gorilla::LongArrayOutput());
            if (fmt == GTSC_FMT_V2) encs.emplace_back(new gorilla::GorillaCompressor(block, *outs.back(), batch));
            else                    encs.emplace_back(new gorilla::Compressor(block, *outs.back(), batch));
            fmts.push_back(fmt);
            for (size_t i = 0; i < P; i++) {
                int64_t t; uint64_t v;
                if (std::fscanf(in, "%" SCNd64 " %" SCNx64, &t, &v) != 2) return 2;
                if (i & 1) {  // exercise both overloads
                    double d; std::memcpy(&d, &v, 8);
                    encs.back()->addValue(t, d);
                } else {
                    encs.back()->addValue(t, (int64_t)v);
                }
            }
            encs.back()->close();
        }
        batch.flush();
        for (size_t s = 0; s < S; s++) {
            const std::vector<int64_t> w = outs[s]->getLongArray();
            std::fprintf(out, "%zu\n", w.size());
            for (int64_t x : w) std::fprintf(out, "%016" PRIx64 "\n", (uint64_t)x);
            gorilla::LongArrayInput input(w);
            std::unique_ptr<gorilla::BlockDecoder> dec;
            if (fmts[s] == GTSC_FMT_V2) dec.reset(new gorilla::GorillaDecompressor(input));
            else                        dec.reset(new gorilla::Decompressor(input));
            std::vector<gorilla::Pair> pairs;
            while (const gorilla::Pair* p = dec->readPair()) pairs.push_back(*p);
            std::fprintf(out, "%zu\n", pairs.size());
            for (const gorilla::Pair& p : pairs)
                std::fprintf(out, "%" PRId64 " %016" PRIx64 "\n", p.getTimestamp(), (uint64_t)p.getLongValue());
        }
        // a standalone compressor (no batch) and a truncated image
        gorilla::LongArrayOutput o1;
        gorilla::GorillaCompressor c1(1500400800000LL, o1);
        c1.addValue(1500400801000LL, 1.5);
        c1.addValue(1500400802000LL, 2.5);
        c1.close();
        const std::vector<int64_t> whole = o1.getLongArray();
        std::vector<int64_t> cut(whole.begin(), whole.begin() + 2);
        gorilla::LongArrayInput in_cut(cut);
        gorilla::GorillaDecompressor dcut(in_cut);
        int got = 0; bool threw = false;
        try { while (dcut.readPair()) got++; } catch (const std::out_of_range&) { threw = true; }
        std::fprintf(out, "standalone %zu truncated_pairs %d threw %d zigzag %d %d\n", o1.getLongArray().size(), got, (int)threw,
                     gorilla::GorillaCompressor::encodeZigZag32(-3), gorilla::GorillaDecompressor::decodeZigZag32(5));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    std::fclose(out);
    return 0;
}
