/*
 * CPU oracle for the Gorilla time-series codec: TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the algorithm implemented by burmanm/gorilla-tsc
 * (reference tree: src/main/java/fi/iki/yak/ts/compression/gorilla/, cited
 * below as G/).  It exists to check the HIP product path and to serve as the
 * timed CPU baseline.  Nothing in the shipped library links, loads or calls
 * it; only tests/, the smoke check and the benchmark's cpu_baseline leg do.
 *
 * PARITY STATUS: the reference is Java and no JVM exists in the build or GPU
 * images, and the reference's own tests (src/test/java/.../EncodeGorillaTest.java,
 * EncodeTest.java) assert only decode(encode(x)) == x -- they hold no golden
 * compressed bytes.  Hence:
 *   - round-trip identity on the reference's test inputs: PINNED (tests/golden);
 *   - compressed-byte image: PARITY UNPINNED against a running reference; it is
 *     cross-checked against an independently written bit-string restatement
 *     (oracle/spec_codec.py) and hand-derived vectors.
 *
 * Formats
 *   GOR_FMT_V2: GorillaCompressor/ValueCompressor + GorillaDecompressor/ValueDecompressor
 *   GOR_FMT_V1: Compressor + Decompressor (the 1.x stream)
 * Containers
 *   word image : LongArrayOutput.getLongArray()  (G/LongArrayOutput.java:141-145)
 *   byte image : ByteBufferBitOutput after flush + caller flip(), without the
 *                unmasked-short-write defect of G/ByteBufferBitOutput.java:79-83
 *                (i.e. the big-endian serialisation of the clean bit stream).
 */
#ifndef GORILLA_ORACLE_H
#define GORILLA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GOR_FMT_V2 = 0, GOR_FMT_V1 = 1 };
enum { GOR_PRED_LAST = 0, GOR_PRED_DFCM = 1 };

/* decoder return codes */
enum { GOR_OK = 0, GOR_END = 1, GOR_ERR_TRUNCATED = -1, GOR_ERR_MALFORMED = -2 };

/* ---------------------------------------------------------------- encoder */
typedef struct gor_encoder gor_encoder;

/* G/GorillaCompressor.java:32-41 (v2), G/Compressor.java:24-34 (v1): the
 * constructor emits the 64-bit block timestamp immediately.
 * dfcm_size > 0 selects DifferentialFCM(dfcm_size) (v2 only), else last-value. */
gor_encoder *gor_enc_new(int format, int64_t block_ts, int dfcm_size);
void gor_enc_free(gor_encoder *e);
/* addValue(long ts, long rawValueBits)  G/GorillaCompressor.java:49-71 */
void gor_enc_add(gor_encoder *e, int64_t ts, uint64_t value_bits);
/* close()  G/GorillaCompressor.java:84-89 */
void gor_enc_close(gor_encoder *e);
/* number of stream bits written so far */
uint64_t gor_enc_bits(const gor_encoder *e);
/* getLongArray(): returns the number of words of the image and copies at most
 * cap of them to dst (dst may be NULL to query).  Valid before or after close. */
size_t gor_enc_words(const gor_encoder *e, uint64_t *dst, size_t cap);
/* byte image (ByteBuffer after flush()+flip()): floor(N/8)+1 bytes once closed. */
size_t gor_enc_bytes(const gor_encoder *e, uint8_t *dst, size_t cap);

/* ---------------------------------------------------------------- decoder */
typedef struct gor_decoder gor_decoder;

/* LongArrayInput(long[]) + GorillaDecompressor(BitInput)
 * G/LongArrayInput.java:30-33, G/GorillaDecompressor.java:27-35 */
gor_decoder *gor_dec_new(int format, const uint64_t *words, size_t nwords, int dfcm_size);
void gor_dec_free(gor_decoder *d);
/* readPair(): GOR_OK with (*ts,*val) filled, GOR_END at end of stream
 * (Java returns null), negative on a stream Java would throw on. */
int gor_dec_next(gor_decoder *d, int64_t *ts, uint64_t *value_bits);

/* ------------------------------------------------------------- batch forms
 * One codec instance per series, series split over nthreads POSIX threads in
 * contiguous ranges (the reference leaves this to its caller).  Layouts match
 * the device C-ABI: CSR point offsets, caller-provided output slots. */

/* Q10 bound: floor((64 + (P ? 91 + 114*(P-1) : 0) + 37)/64) + 2 */
uint64_t gor_max_words(uint64_t npoints);

/* returns 0, or -1 if some series overflowed its slot (status[s] != 0 there) */
int gor_encode_batch(int format, uint64_t nseries,
                     const int64_t *block_ts,   /* [S]   */
                     const int64_t *ts,         /* [sum P] */
                     const uint64_t *val,       /* [sum P] */
                     const uint64_t *pt_off,    /* [S+1] */
                     uint64_t *out_words,
                     const uint64_t *out_off,   /* [S+1] slot of series s = [out_off[s], out_off[s+1]) */
                     uint64_t *out_nwords,      /* [S]   */
                     uint32_t *status,          /* [S] or NULL */
                     int nthreads);

int gor_decode_batch(int format, uint64_t nseries,
                     const uint64_t *words,
                     const uint64_t *in_off,    /* [S+1]: image of s starts at in_off[s] */
                     const uint64_t *in_nwords, /* [S] or NULL (then in_off[s+1]-in_off[s]) */
                     int64_t *ts, uint64_t *val,
                     const uint64_t *pt_off,    /* [S+1] capacity slots */
                     uint64_t *out_npoints,     /* [S] */
                     uint32_t *status,          /* [S] or NULL */
                     int nthreads);

#ifdef __cplusplus
}
#endif
#endif
