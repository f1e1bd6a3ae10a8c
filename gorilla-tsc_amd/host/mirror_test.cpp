// Drives the C++ mirror of the reference classes the way the reference's own tests drive the
// Java classes (construct, addValue..., close, getLongArray; then readPair until null), on
// series read from a text file, and writes what it got for the caller to compare.
//
//   mirror_test <in.txt> <out.txt>
// in.txt : "<nseries>" then per series "<format> <block_ts> <npoints>" and npoints lines
//          "<ts> <value bits as hex>"
// out.txt: per series "<nwords>" + words (hex), then "<npairs>" + pairs "<ts> <hex>"
#include <cinttypes>
#include <cstdio>
#include <memory>

#include "gorilla.hpp"

int main(int argc, char** argv)
{
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
            const std::vector<int64_t>& w = outs[s]->getLongArray();
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
        std::vector<int64_t> cut(o1.getLongArray().begin(), o1.getLongArray().begin() + 2);
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
