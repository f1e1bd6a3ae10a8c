/*
 * gtsc -- Gorilla time-series codec for AMD Instinct MI355X (gfx950), C ABI.
 *
 * Batch-shaped drop-in for the compress/decompress path of burmanm/gorilla-tsc.
 * The reference has no FFI; its public Java classes are the boundary, and each
 * entry point below names the reference interface it stands in for ("G/" =
 * src/main/java/fi/iki/yak/ts/compression/gorilla/ in the reference tree).
 * One reference compressor instance handles one series block; one call here
 * handles many independent blocks ("series") at once.
 *
 * Data layout (all little-endian host/device memory, plain arrays):
 *   block_ts[S]      int64   block start timestamp per series (constructor arg of
 *                            GorillaCompressor, G/GorillaCompressor.java:28-37)
 *   ts[], val[]      int64 / uint64, points of all series back to back; val holds
 *                            Double.doubleToRawLongBits(v) or the long itself
 *                            (addValue, G/GorillaCompressor.java:49-71)
 *   pt_off[S+1]      uint64  series s owns points [pt_off[s], pt_off[s+1])
 *   words[]          uint64  compressed images; word i of an image equals element
 *                            i of LongArrayOutput.getLongArray()
 *                            (G/LongArrayOutput.java:141-145), i.e. stream bit k
 *                            is bit 63-(k%64) of word k/64
 *   *_off[S+1]       uint64  image of series s lives in slot
 *                            [off[s], off[s+1]) of words[]; nwords[s] says how
 *                            many words of the slot are the image
 * Passing NULL for an offset array selects the uniform layout given by the
 * accompanying uniform_* argument (every series has that many points / words).
 *
 * All "device" entry points take device pointers, enqueue on the context's
 * stream and return without waiting; per-series problems are reported in
 * status[] (optional) and summarised by gtsc_sync().
 */
#ifndef GTSC_H
#define GTSC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes */
#define GTSC_OK            0
#define GTSC_ERR_ARG      (-1)  /* NULL / inconsistent argument */
#define GTSC_ERR_CAPACITY (-2)  /* some series did not fit its output slot */
#define GTSC_ERR_HIP      (-3)  /* HIP runtime error, see gtsc_last_error() */
#define GTSC_ERR_STREAM   (-4)  /* some compressed image was truncated or malformed */

/* stream formats */
#define GTSC_FMT_V2 0  /* GorillaCompressor / GorillaDecompressor (2.x) */
#define GTSC_FMT_V1 1  /* Compressor / Decompressor (1.x) */

/* per-series status bits */
#define GTSC_ST_OVERFLOW  1u  /* output slot too small; output for that series is incomplete */
#define GTSC_ST_TRUNCATED 2u  /* decoder ran off the image: Java would throw
                                 ArrayIndexOutOfBoundsException (G/LongArrayInput.java:44) */
#define GTSC_ST_MALFORMED 4u  /* window reused before any was defined: Java would throw on the
                                 mask-table index (G/LongArrayInput.java:60) */

typedef struct gtsc_ctx gtsc_ctx;

/* Library/ABI version (major*1000 + minor). */
int gtsc_version(void);

/* Thread-local description of the last error returned on this thread. */
const char *gtsc_last_error(void);

/* Worst-case image size in words for a block of npoints points: 36+78 = 114
 * bits per point after the first, plus header, first point, close marker and
 * the trailing word(s) of getLongArray().  No reference counterpart
 * (LongArrayOutput grows by doubling, G/LongArrayOutput.java:58-62). */
uint64_t gtsc_max_words(uint64_t npoints);

/* A context owns one HIP stream on one device and a little scratch.  One per
 * host thread and device, like the reference's per-instance, non-thread-safe
 * objects.  device_id < 0 selects the current device. */
int gtsc_create(int device_id, gtsc_ctx **ctx);
int gtsc_destroy(gtsc_ctx *ctx);

/*
 * Predictor of the value codec for the calls that follow on this context (2.x format
 * only): the third constructor argument of GorillaCompressor / second of
 * GorillaDecompressor (G/GorillaCompressor.java:34-37, G/GorillaDecompressor.java:27-31).
 * GTSC_PRED_LAST is LastValuePredictor (the default; table_size ignored).
 * GTSC_PRED_DFCM is DifferentialFCM(table_size) (G/predictors/DifferentialFCM.java:23-46):
 * table_size > 0 is rounded up to a power of two as the Java constructor does; the library
 * keeps one table per block in device memory (nseries * size * 8 bytes per call).  Blocks
 * coded with a predictor must be decoded with the same one.  DFCM runs on the plain
 * lane-per-series kernels (its table look-ups are per-lane memory accesses).
 */
#define GTSC_PRED_LAST 0
#define GTSC_PRED_DFCM 1
int gtsc_set_predictor(gtsc_ctx *ctx, int predictor, int table_size);

/*
 * Tunables.  Which kernel codes a batch is decided from the batch itself -- its layout, the number and length of
 * its long blocks and the density of their images, the last two worked out on the device before the kernels start --
 * never from what the context coded before: a fresh context and one with history choose alike.  The thresholds of
 * those decisions are per-context tunables; results never depend on them, only speed.  Each has a default, which an
 * environment variable (read when the context is created) or gtsc_set_tunable() replaces.  No reference counterpart.
 *
 *   GTSC_TUNE_LONG_POINTS (GTSC_LONG_POINTS, 1024): blocks of at least this many points ("long" blocks) may get a
 *     wavefront or a workgroup each instead of one lane of a wavefront that walks 64 blocks in lock step.
 *   GTSC_TUNE_VLONG_POINTS (GTSC_VLONG_POINTS, 16384): from this many points on a block is "very long".
 *   GTSC_TUNE_WG_DECODE_MAX (GTSC_WG_DECODE_MAX, 512) / GTSC_TUNE_WG_DECODE_VLONG_MAX (GTSC_WG_DECODE_VLONG_MAX, 640):
 *     in batches of up to that many long blocks, the long blocks whose images are sparse are decompressed a
 *     workgroup per block -- and in any batch the very long ones with sparse images, while their capacities together
 *     come to no more than that many times the largest of them (640 blocks of equal length; more of mixed lengths);
 *     GTSC_TUNE_WG_DENSE_BITS (GTSC_WG_DENSE_BITS, 40): sparse = fewer compressed bits per point of capacity than this.
 *   GTSC_TUNE_WAVE_DECODE_MAX (GTSC_LONG_DECODE_MAX, 12288): in batches of up to that many long blocks the other
 *     long blocks are decompressed a wavefront per block; GTSC_TUNE_WAVE_DECODE_MAX_DENSE (GTSC_LONG_DECODE_MAX_DENSE,
 *     2048): the limit when the long blocks' images come to more than GTSC_TUNE_DENSE_BITS (GTSC_DENSE_BITS, 64)
 *     bits per point.  Beyond the limit every block has a lane.
 *   GTSC_TUNE_COOP_ENCODE_MAX (GTSC_COOP_MAX, 384): batches of up to that many long blocks are compressed a workgroup
 *     per block, larger ones a wavefront per block.
 *   GTSC_TUNE_SPARSE_HINT (GTSC_SPARSE_HINT, 0): 1 tells the compressor that the streams are sparse (a few bits per
 *     point: mostly unchanged values at a regular interval); batches of at least GTSC_TUNE_SPARSE_ENCODE_MIN
 *     (GTSC_SPARSE_TILED_MIN, 57344) long blocks are then compressed a lane per block, which skips runs of such points.
 *   GTSC_TUNE_DECODE_RING (GTSC_RING, 0): the lane-per-series decompressor exists with input rings of 8 words per
 *     series (fastest on streams of a few bits per point) and of 16 words (15-30 % faster on streams of tens of bits
 *     per point); 0 = the batch's density decides, on the device: both forms are launched and one of them returns
 *     at once (about 20 us per call: 3-5 % of a million short series; a caller who knows its streams pins 8 or 16).
 *     gtsc_decode_host knows each chunk's density beforehand and treats 0 as "by density" at no cost.
 *   GTSC_TUNE_RUN_SKIP (GTSC_RUN_SKIP, 1): 0 keeps the lane-per-series kernels on their lock-step loops.
 *   GTSC_TUNE_DIRECT_KERNELS (GTSC_DIRECT, 0): 1 selects the plain lane-per-series kernels (no LDS staging) for everything.
 *
 * gtsc_set_decode_ring(ctx, n) is gtsc_set_tunable(ctx, GTSC_TUNE_DECODE_RING, n).
 */
#define GTSC_TUNE_LONG_POINTS            1
#define GTSC_TUNE_VLONG_POINTS           2
#define GTSC_TUNE_WAVE_DECODE_MAX        3
#define GTSC_TUNE_WAVE_DECODE_MAX_DENSE  4
#define GTSC_TUNE_WG_DECODE_MAX          5
#define GTSC_TUNE_WG_DECODE_VLONG_MAX    6
#define GTSC_TUNE_WG_DENSE_BITS          7
#define GTSC_TUNE_DENSE_BITS             8
#define GTSC_TUNE_SPARSE_ENCODE_MIN      9
#define GTSC_TUNE_COOP_ENCODE_MAX       10
#define GTSC_TUNE_SPARSE_HINT           11
#define GTSC_TUNE_DECODE_RING           12
#define GTSC_TUNE_RUN_SKIP              13
#define GTSC_TUNE_DIRECT_KERNELS        14
int gtsc_set_tunable(gtsc_ctx *ctx, int which, uint64_t value);
int gtsc_get_tunable(gtsc_ctx *ctx, int which, uint64_t *value);
int gtsc_set_decode_ring(gtsc_ctx *ctx, int ring_words);

/*
 * Diagnostic: which kernels the context's last gtsc_encode_device(_ex) / gtsc_decode_device call enqueued (bits
 * below; the kernels that find no block of theirs in the batch return at once), and -- plan != NULL, decode only,
 * waits for the context's stream -- what the device made of that batch: plan[0] long blocks, [1] very long ones,
 * [2] image words and [3] capacity of the long ones, [4] image words and [5] capacity of every 64th series.
 * No reference counterpart.
 */
#define GTSC_DISPATCH_LANE_DIRECT      1u  /* plain lane-per-series kernel */
#define GTSC_DISPATCH_LANE_TILED       2u  /* lane-per-series kernel with LDS staging */
#define GTSC_DISPATCH_LANE_RING16      4u  /* ... its decompressor with 16-word input rings */
#define GTSC_DISPATCH_LANE_BOTH_RINGS  8u  /* ... in both forms, the device choosing */
#define GTSC_DISPATCH_WAVE            16u  /* wavefront-per-block kernel */
#define GTSC_DISPATCH_WG              32u  /* workgroup-per-block decompressor */
#define GTSC_DISPATCH_PLAN            64u  /* the kernel that works out the plan */
#define GTSC_DISPATCH_COOP           128u  /* workgroup-per-block compressor */
#define GTSC_DISPATCH_SPLIT          256u  /* the compressor's form for arrays on different phases of a 128-byte line */
int gtsc_last_dispatch(gtsc_ctx *ctx, uint32_t *kernels, uint64_t plan[6]);

/* Enqueue on a caller-owned hipStream_t instead of the context's own
 * (NULL restores the context's stream). */
int gtsc_set_stream(gtsc_ctx *ctx, void *hip_stream);

/* Wait for everything enqueued through ctx; returns GTSC_ERR_CAPACITY or
 * GTSC_ERR_STREAM if any series of any call since the last sync raised a
 * status bit, else GTSC_OK. */
int gtsc_sync(gtsc_ctx *ctx);

/*
 * Compress nseries blocks.  Stands in for, per series s:
 *   new GorillaCompressor(block_ts[s], new LongArrayOutput())   G/GorillaCompressor.java:28-37
 *   addValue(ts[i], val[i]) for i in the series                 G/GorillaCompressor.java:49-71
 *   close(); getLongArray()                                     G/GorillaCompressor.java:84-89,
 *                                                               G/LongArrayOutput.java:141-145
 * (GTSC_FMT_V1: Compressor, G/Compressor.java:24-84.)
 * out_nwords[s] receives the image length; the image is written to
 * out_words + out_off[s] (or + s*uniform_slot_words).  A series whose image does
 * not fit its slot gets GTSC_ST_OVERFLOW; out_nwords[s] is then the length the image
 * would have had, and only the slot's words are written.
 * Limits and access pattern: a block holds at most 2,147,483,392 points (longer ones are
 * coded up to that point and flagged GTSC_ST_OVERFLOW).  The kernels fetch ts[] and val[] in
 * aligned 16-byte pieces, so up to 8 bytes before the first and after the last point of the
 * arrays are read (never used, never written): harmless for any allocation, since a
 * 16-byte aligned piece cannot cross a page, but visible to a memory checker.
 */
int gtsc_encode_device(gtsc_ctx *ctx, int format, uint64_t nseries,
                       const int64_t *block_ts,
                       const int64_t *ts, const uint64_t *val,
                       const uint64_t *pt_off, uint64_t uniform_npoints,
                       uint64_t *out_words,
                       const uint64_t *out_off, uint64_t uniform_slot_words,
                       uint64_t *out_nwords, uint32_t *status);

/*
 * gtsc_encode_device with two extras:
 *  - options & GTSC_ENC_NO_CLOSE: the blocks are left open; each image is what
 *    getLongArray() returns before close(), the completed words plus the current partial
 *    word (G/LongArrayOutput.java:141-145; the reference's testCopyFlush,
 *    src/test/java/.../EncodeGorillaTest.java:483-500).  Such an image is not decodable
 *    (it has no end marker), it is a snapshot of a block still being filled.
 *  - out_nbits (may be NULL): the length N of each stream in bits, which
 *    gtsc_words_to_bytes_device needs.
 */
#define GTSC_ENC_NO_CLOSE 1u
int gtsc_encode_device_ex(gtsc_ctx *ctx, int format, uint32_t options, uint64_t nseries,
                          const int64_t *block_ts,
                          const int64_t *ts, const uint64_t *val,
                          const uint64_t *pt_off, uint64_t uniform_npoints,
                          uint64_t *out_words,
                          const uint64_t *out_off, uint64_t uniform_slot_words,
                          uint64_t *out_nwords, uint64_t *out_nbits, uint32_t *status);

/*
 * Byte images.  ByteBufferBitOutput (G/ByteBufferBitOutput.java:42-103) holds the
 * same bit stream as LongArrayOutput byte by byte: byte i is stream bits 8i..8i+7, and
 * after close() the buffer holds floor(N/8)+1 bytes.  gtsc_words_to_bytes_device
 * writes that image for every series (out_nbytes[s] bytes at out_bytes +
 * byte_off[s] or + s*uniform_slot_bytes; offsets here are in BYTES) from the word
 * images and their bit lengths.  gtsc_bytes_to_words_device goes the other way
 * for input that was stored through a ByteBuffer (ByteBufferBitInput,
 * G/ByteBufferBitInput.java:20-27): it rebuilds word images, ceil(nbytes/8)+1
 * words each, that gtsc_decode_device reads.  nbytes may be NULL (whole slot).
 * The 1.x ByteBufferBitOutput defect that corrupts a bit when a 64-bit value follows a
 * 5-bit field (G/ByteBufferBitOutput.java:79-83) is not reproduced: bytes are those
 * of the clean stream.
 */
int gtsc_words_to_bytes_device(gtsc_ctx *ctx, uint64_t nseries,
                               const uint64_t *words,
                               const uint64_t *word_off, uint64_t uniform_slot_words,
                               const uint64_t *nbits,
                               uint8_t *out_bytes,
                               const uint64_t *byte_off, uint64_t uniform_slot_bytes,
                               uint64_t *out_nbytes);
int gtsc_bytes_to_words_device(gtsc_ctx *ctx, uint64_t nseries,
                               const uint8_t *bytes,
                               const uint64_t *byte_off, uint64_t uniform_slot_bytes,
                               const uint64_t *nbytes,
                               uint64_t *out_words,
                               const uint64_t *word_off, uint64_t uniform_slot_words,
                               uint64_t *out_nwords);

/*
 * Decompress nseries images.  Stands in for, per series s:
 *   new GorillaDecompressor(new LongArrayInput(image))          G/GorillaDecompressor.java:23-35
 *   readPair() until null                                       G/GorillaDecompressor.java:42-49
 * (GTSC_FMT_V1: Decompressor, G/Decompressor.java:22-42.)
 * Pairs come back as structure-of-arrays (Pair.getTimestamp / getLongValue,
 * G/Pair.java:8-28) at ts/val + pt_off[s] (or + s*uniform_capacity);
 * out_npoints[s] receives the number of pairs before end of stream.
 * in_nwords may be NULL: the whole slot is then taken as the image.
 */
int gtsc_decode_device(gtsc_ctx *ctx, int format, uint64_t nseries,
                       const uint64_t *words,
                       const uint64_t *in_off, uint64_t uniform_slot_words,
                       const uint64_t *in_nwords,
                       int64_t *ts, uint64_t *val,
                       const uint64_t *pt_off, uint64_t uniform_capacity,
                       uint64_t *out_npoints, uint32_t *status);

/*
 * Pack slotted images back to back: compact_off[0] = 0,
 * compact_off[s+1] = compact_off[s] + nwords[s], and the image of s copied to
 * compact_words + compact_off[s].  No reference counterpart (each Java block
 * owns its own long[]); this is what a caller uses to persist a batch.
 * compact_words must hold sum(nwords) words.  For a series that overflowed its slot
 * (nwords[s] larger than the slot) only the slot's words are copied; the offsets still
 * advance by nwords[s].
 */
int gtsc_compact_device(gtsc_ctx *ctx, uint64_t nseries,
                        const uint64_t *slot_words,
                        const uint64_t *slot_off, uint64_t uniform_slot_words,
                        const uint64_t *nwords,
                        uint64_t *compact_words, uint64_t *compact_off);

/*
 * gtsc_compact_device for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in
 * the reference.  Same arguments and results as gtsc_compact_device; the partial sums of its
 * scan live in a static device array of the library (64 KiB, held on every GPU the library
 * runs on until the process ends) instead of a buffer the context grows, and the call makes
 * no hipMalloc, hipFree or other allocating call.  At most GTSC_NOALLOC_MAX_SERIES series per
 * call (GTSC_ERR_ARG beyond).  All contexts of a GPU share the one array: calls of this entry
 * point that overlap in time on one GPU (different streams or contexts) race; calls on one
 * stream, or ordered by events, do not.
 */
#define GTSC_NOALLOC_MAX_SERIES 8000000ull
int gtsc_compact_device_noalloc(gtsc_ctx *ctx, uint64_t nseries,
                                const uint64_t *slot_words,
                                const uint64_t *slot_off, uint64_t uniform_slot_words,
                                const uint64_t *nwords,
                                uint64_t *compact_words, uint64_t *compact_off);

/*
 * Host-memory forms: every pointer is host memory and the call returns when
 * the results are in the caller's arrays.  The batch is cut into chunks of
 * whole series that pass through four streams -- one for all copies in (two chunks ahead of the kernels), one for
 * the kernels (and the compressor's copy out, which the host can only enqueue once it knows the chunk's size), one
 * beside it for the workgroup-per-block decompressor of a ragged chunk, one for the decompressor's copies out (the
 * two copy streams are created with the highest stream priority, which gives them hardware queues of their own) --
 * and four sets of device buffers that the chunks take in turn, tied together by events, so that copies overlap
 * kernels and each other.  Whatever the return code, nothing of the call is in
 * flight any more when it returns: the caller may free or reuse its arrays.  Host arrays obtained
 * from gtsc_host_alloc() are page-locked and move by DMA at PCIe rate; other
 * memory works but is staged by the HIP runtime.
 *
 * Encode produces the images back to back: out_off[0] = 0, out_off[s+1] =
 * out_off[s] + (image length of s), image s at out_words + out_off[s].
 * out_capacity_words is the room in out_words; if the batch needs more the
 * call returns GTSC_ERR_CAPACITY (out_off is still complete, so out_off[S]
 * tells the room required).
 * Decode reads such back-to-back images (in_off[S+1]) and fills ts/val slots
 * exactly as gtsc_decode_device does.
 */
int gtsc_encode_host(gtsc_ctx *ctx, int format, uint64_t nseries,
                     const int64_t *block_ts,
                     const int64_t *ts, const uint64_t *val,
                     const uint64_t *pt_off, uint64_t uniform_npoints,
                     uint64_t *out_words, uint64_t out_capacity_words,
                     uint64_t *out_off);

/*
 * gtsc_encode_host with the extras of gtsc_encode_device_ex: options & GTSC_ENC_NO_CLOSE
 * gives the snapshot getLongArray() returns before close() (G/LongArrayOutput.java:141-145,
 * testCopyFlush); out_nbits (host array of nseries entries, may be NULL) receives each
 * stream's length in bits, from which the ByteBufferBitOutput image is the first
 * floor(nbits/8)+1 bytes of the words written big-endian (G/ByteBufferBitOutput.java:42-103).
 */
int gtsc_encode_host_ex(gtsc_ctx *ctx, int format, uint32_t options, uint64_t nseries,
                        const int64_t *block_ts,
                        const int64_t *ts, const uint64_t *val,
                        const uint64_t *pt_off, uint64_t uniform_npoints,
                        uint64_t *out_words, uint64_t out_capacity_words,
                        uint64_t *out_off, uint64_t *out_nbits);

int gtsc_decode_host(gtsc_ctx *ctx, int format, uint64_t nseries,
                     const uint64_t *words, const uint64_t *in_off,
                     int64_t *ts, uint64_t *val,
                     const uint64_t *pt_off, uint64_t uniform_capacity,
                     uint64_t *out_npoints, uint32_t *status);

/*
 * The host-memory forms over several contexts -- normally one per GPU of the node.  Blocks are independent (one
 * compressor instance per block, no shared state: G/GorillaCompressor.java:13-26), so the batch is cut into nctx
 * contiguous runs of series with about equal numbers of points; run k goes through ctxs[k]'s pipeline on a host
 * thread of its own, and the results are exactly those of the single-context call: images back to back in series
 * order, out_off[S+1], decoded points in their slots.  No data moves between GPUs.  The contexts may sit on
 * different devices or share one; each must be used by this call only while it runs.  Predictors and tunables are
 * those of each context.  nctx == 1 is the single-context call.
 */
int gtsc_encode_host_multi(int nctx, gtsc_ctx *const *ctxs, int format, uint32_t options, uint64_t nseries,
                           const int64_t *block_ts,
                           const int64_t *ts, const uint64_t *val,
                           const uint64_t *pt_off, uint64_t uniform_npoints,
                           uint64_t *out_words, uint64_t out_capacity_words,
                           uint64_t *out_off, uint64_t *out_nbits);
int gtsc_decode_host_multi(int nctx, gtsc_ctx *const *ctxs, int format, uint64_t nseries,
                           const uint64_t *words, const uint64_t *in_off,
                           int64_t *ts, uint64_t *val,
                           const uint64_t *pt_off, uint64_t uniform_capacity,
                           uint64_t *out_npoints, uint32_t *status);

/* Diagnostic: 1 when none of the streams of the host entry points has work in flight, 0 when one has,
 * negative on error.  (After any host entry point has returned it is 1.)  No reference counterpart. */
int gtsc_host_pipeline_idle(gtsc_ctx *ctx);

/* Page-locked host memory for the host entry points (hipHostMalloc/hipHostFree). */
void *gtsc_host_alloc(size_t bytes);
void gtsc_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* GTSC_H */
