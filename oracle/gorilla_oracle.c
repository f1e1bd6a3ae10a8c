/*
 * CPU oracle for the Gorilla codec: TEST INFRASTRUCTURE ONLY (see header).
 * Streaming shape (one call per point through a bit writer), like the
 * reference's addValue -> BitOutput.writeBits call structure, so that it can
 * also be timed as the CPU baseline.  G/ = reference
 * src/main/java/fi/iki/yak/ts/compression/gorilla/.
 *
 * PARITY: round-trip pinned by the reference's test inputs; compressed bytes
 * unpinned against a running reference (no JVM available) -- see header.
 */
#include "gorilla_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ bit writer
 * Bit k of the stream lives in word k/64 at bit 63-(k%64)
 * (G/LongArrayOutput.java:96-131).  Only the low n bits of a value are kept,
 * as both branches of writeBits do (:118 mask after shift, :123 explicit). */
typedef struct {
    uint64_t *w;
    size_t cap;      /* words allocated (all zero beyond the stream) */
    uint64_t nbits;  /* stream length */
    uint64_t flips;  /* extra word flips caused by flush() (G/LongArrayOutput.java:136-139) */
} bitw;

static void bw_reserve(bitw *b, size_t words)
{
    if (words <= b->cap) return;
    size_t nc = b->cap ? b->cap : 256;            /* DEFAULT_ALLOCATION, :11 */
    while (nc < words) nc *= 2;                   /* doubling growth, :58-62 */
    b->w = (uint64_t *)realloc(b->w, nc * sizeof(uint64_t));
    memset(b->w + b->cap, 0, (nc - b->cap) * sizeof(uint64_t));
    b->cap = nc;
}

static inline void bw_put(bitw *b, uint64_t v, int n) /* 1 <= n <= 64 */
{
    size_t idx = (size_t)(b->nbits >> 6);
    int room = 64 - (int)(b->nbits & 63);
    if (idx + 3 > b->cap) bw_reserve(b, idx + 3);
    if (n < 64) v &= (1ULL << n) - 1;
    if (n <= room) {
        b->w[idx] |= v << (room - n);
    } else {
        int spill = n - room;
        b->w[idx] |= v >> spill;
        b->w[idx + 1] |= v << (64 - spill);
    }
    b->nbits += (uint64_t)n;
}

/* ------------------------------------------------------------ predictors
 * LastValuePredictor: G/predictors/LastValuePredictor.java:10-22
 * DifferentialFCM   : G/predictors/DifferentialFCM.java:23-46 */
typedef struct {
    uint64_t last;
    int64_t *table; /* NULL => last-value predictor */
    int32_t hash, mask;
} predictor;

static void pred_init(predictor *p, int dfcm_size)
{
    memset(p, 0, sizeof *p);
    if (dfcm_size > 0) {
        /* size rounded up to a power of two: 1 << (64 - clz64(size-1)) */
        uint64_t s = (uint64_t)(dfcm_size - 1);
        int lz = s ? __builtin_clzll(s) : 64;
        int32_t n = (int32_t)(1u << (64 - lz));
        p->table = (int64_t *)calloc((size_t)n, sizeof(int64_t));
        p->mask = n - 1;
    }
}
static inline uint64_t pred_predict(const predictor *p)
{
    return p->table ? (uint64_t)p->table[p->hash] + p->last : p->last;
}
static inline void pred_update(predictor *p, uint64_t v)
{
    if (p->table) {
        int64_t d = (int64_t)(v - p->last);
        p->table[p->hash] = d;
        /* (int)(((hash << 5) ^ (d >> 50)) & mask): int shift, long arithmetic shift */
        int64_t h = (int64_t)(int32_t)((uint32_t)p->hash << 5) ^ (d >> 50);
        p->hash = (int32_t)(h & (int64_t)p->mask);
    }
    p->last = v;
}

/* ---------------------------------------------------------------- encoder */
struct gor_encoder {
    int format;
    bitw out;
    int64_t block_ts;
    int64_t stored_ts;     /* 0 means "no point yet" (G/GorillaCompressor.java:50,65) */
    int64_t stored_delta;  /* int32 range in v2 (:14), long in v1 (G/Compressor.java:15) */
    int lead, trail;       /* window; lead starts at Integer.MAX_VALUE (G/ValueCompressor.java:12-13) */
    predictor pred;
};

gor_encoder *gor_enc_new(int format, int64_t block_ts, int dfcm_size)
{
    gor_encoder *e = (gor_encoder *)calloc(1, sizeof *e);
    e->format = format;
    e->block_ts = block_ts;
    e->lead = 0x7fffffff;
    e->trail = 0;
    pred_init(&e->pred, format == GOR_FMT_V2 ? dfcm_size : 0);
    bw_reserve(&e->out, 256);
    bw_put(&e->out, (uint64_t)block_ts, 64); /* addHeader */
    return e;
}

void gor_enc_free(gor_encoder *e)
{
    if (!e) return;
    free(e->out.w);
    free(e->pred.table);
    free(e);
}

static inline uint32_t zigzag32(int32_t n) /* G/GorillaCompressor.java:157-160 */
{
    return ((uint32_t)n << 1) ^ (uint32_t)(n >> 31);
}

/* v2 timestamp code, G/GorillaCompressor.java:99-143 */
static inline void v2_timestamp(gor_encoder *e, int64_t ts)
{
    int32_t new_delta = (int32_t)((uint64_t)ts - (uint64_t)e->stored_ts);
    int32_t dod = (int32_t)((uint32_t)new_delta - (uint32_t)(int32_t)e->stored_delta);
    if (dod == 0) {
        bw_put(&e->out, 0, 1);
    } else {
        uint32_t z = zigzag32(dod) - 1u;
        int need = z ? 32 - __builtin_clz(z) : 0;
        if (need >= 1 && need <= 7)        bw_put(&e->out, 0x100u | z, 9);   /* '10'   + 7  */
        else if (need == 8 || need == 9)   bw_put(&e->out, 0xC00u | z, 12);  /* '110'  + 9  */
        else if (need >= 10 && need <= 12) bw_put(&e->out, 0xE000u | z, 16); /* '1110' + 12 */
        else { bw_put(&e->out, 0xF, 4); bw_put(&e->out, z, 32); }            /* need==0 lands here too */
        e->stored_delta = new_delta;
    }
    e->stored_ts = ts;
}

/* v2 value code, G/ValueCompressor.java:32-91 */
static inline void v2_value(gor_encoder *e, uint64_t v)
{
    uint64_t x = pred_predict(&e->pred) ^ v;
    pred_update(&e->pred, v);
    if (x == 0) { bw_put(&e->out, 0, 1); return; }
    int lz = __builtin_clzll(x), tz = __builtin_ctzll(x);
    if (lz >= e->lead && tz >= e->trail) {
        bw_put(&e->out, 2, 2); /* '1','0' */
        bw_put(&e->out, x >> e->trail, 64 - e->lead - e->trail);
    } else {
        int sig = 64 - lz - tz;
        bw_put(&e->out, 3, 2); /* '1','1' */
        bw_put(&e->out, (uint64_t)lz, 6);
        bw_put(&e->out, (uint64_t)(sig - 1), 6);
        bw_put(&e->out, x >> tz, sig);
        e->lead = lz;
        e->trail = tz;
    }
}

/* v1 timestamp code, G/Compressor.java:94-118: long arithmetic, two's-complement payloads */
static inline void v1_timestamp(gor_encoder *e, int64_t ts)
{
    int64_t new_delta = (int64_t)((uint64_t)ts - (uint64_t)e->stored_ts);
    int64_t dod = (int64_t)((uint64_t)new_delta - (uint64_t)e->stored_delta);
    if (dod == 0)                          bw_put(&e->out, 0, 1);
    else if (dod >= -63 && dod <= 64)      { bw_put(&e->out, 2, 2);   bw_put(&e->out, (uint64_t)dod, 7); }
    else if (dod >= -255 && dod <= 256)    { bw_put(&e->out, 6, 3);   bw_put(&e->out, (uint64_t)dod, 9); }
    else if (dod >= -2047 && dod <= 2048)  { bw_put(&e->out, 0xE, 4); bw_put(&e->out, (uint64_t)dod, 12); }
    else                                   { bw_put(&e->out, 0xF, 4); bw_put(&e->out, (uint64_t)dod, 32); }
    e->stored_delta = new_delta;
    e->stored_ts = ts;
}

/* v1 value code, G/Compressor.java:120-181: lz capped at 31, 5-bit lz, 6-bit sig (64 -> 0) */
static inline void v1_value(gor_encoder *e, uint64_t v)
{
    uint64_t x = e->pred.last ^ v;
    e->pred.last = v;
    if (x == 0) { bw_put(&e->out, 0, 1); return; }
    int lz = __builtin_clzll(x), tz = __builtin_ctzll(x);
    if (lz >= 32) lz = 31;
    bw_put(&e->out, 1, 1);
    if (lz >= e->lead && tz >= e->trail) {
        bw_put(&e->out, 0, 1);
        bw_put(&e->out, x >> e->trail, 64 - e->lead - e->trail);
    } else {
        int sig = 64 - lz - tz;
        bw_put(&e->out, 1, 1);
        bw_put(&e->out, (uint64_t)lz, 5);
        bw_put(&e->out, (uint64_t)sig, 6);
        bw_put(&e->out, x >> tz, sig);
        e->lead = lz;
        e->trail = tz;
    }
}

void gor_enc_add(gor_encoder *e, int64_t ts, uint64_t v)
{
    if (e->stored_ts == 0) {
        /* writeFirst: G/GorillaCompressor.java:73-79 (delta truncated to int),
         * G/Compressor.java:66-73 (long); 27 low bits written either way */
        int64_t d = (int64_t)((uint64_t)ts - (uint64_t)e->block_ts);
        e->stored_delta = (e->format == GOR_FMT_V2) ? (int64_t)(int32_t)(uint64_t)d : d;
        e->stored_ts = ts;
        bw_put(&e->out, (uint64_t)e->stored_delta, 27);
        if (e->format == GOR_FMT_V2) pred_update(&e->pred, v); else e->pred.last = v;
        bw_put(&e->out, v, 64);
        return;
    }
    if (e->format == GOR_FMT_V2) { v2_timestamp(e, ts); v2_value(e, v); }
    else                         { v1_timestamp(e, ts); v1_value(e, v); }
}

void gor_enc_close(gor_encoder *e)
{
    bw_put(&e->out, 0xF, 4);
    bw_put(&e->out, 0xFFFFFFFFu, 32);
    bw_put(&e->out, 0, 1);
    e->out.flips += 1; /* flush() always stores the current word and starts a new one */
}

uint64_t gor_enc_bits(const gor_encoder *e) { return e->out.nbits; }

size_t gor_enc_words(const gor_encoder *e, uint64_t *dst, size_t cap)
{
    /* position = completed words; every flush adds one; image = position + 1.
     * A flush on a non-empty partial word completes it; on an empty word it
     * stores a zero word.  Either way: floor(N/64) + flips + 1. */
    size_t n = (size_t)(e->out.nbits >> 6) + (size_t)e->out.flips + 1;
    if (dst) {
        size_t m = n < cap ? n : cap;
        for (size_t i = 0; i < m; i++) dst[i] = i < e->out.cap ? e->out.w[i] : 0;
    }
    return n;
}

size_t gor_enc_bytes(const gor_encoder *e, uint8_t *dst, size_t cap)
{
    /* ByteBufferBitOutput: bytes completed = floor(N/8); flush() forces one
     * more put even when the current byte is empty (G/ByteBufferBitOutput.java:93-96). */
    size_t n = (size_t)(e->out.nbits >> 3) + (size_t)e->out.flips;
    if (dst) {
        size_t m = n < cap ? n : cap;
        for (size_t i = 0; i < m; i++) {
            size_t wi = i >> 3;
            uint64_t w = wi < e->out.cap ? e->out.w[wi] : 0;
            dst[i] = (uint8_t)(w >> (56 - 8 * (i & 7)));
        }
    }
    return n;
}

/* ---------------------------------------------------------------- decoder
 * Reader semantics of G/LongArrayInput.java:35-91: after every read the next
 * word is loaded eagerly when the current one is exhausted, so a read of n
 * bits at position p needs p + n < 64 * nwords (strictly). */
struct gor_decoder {
    int format;
    const uint64_t *w;
    uint64_t limit_bits; /* 64 * nwords */
    uint64_t pos;
    int err;
    int64_t block_ts, stored_ts, stored_delta;
    uint64_t stored_val;
    int lead, trail;
    int eos;
    predictor pred;
};

static inline uint64_t br_get(gor_decoder *d, int n) /* 1 <= n <= 64 */
{
    if (d->err) return 0;
    if (d->pos + (uint64_t)n >= d->limit_bits) { d->err = GOR_ERR_TRUNCATED; return 0; }
    size_t idx = (size_t)(d->pos >> 6);
    int off = (int)(d->pos & 63);
    uint64_t v = d->w[idx] << off;
    if (off + n > 64) v |= d->w[idx + 1] >> (64 - off);
    d->pos += (uint64_t)n;
    return n == 64 ? v : v >> (64 - n);
}

/* nextClearBit(maxBits): G/LongArrayInput.java:76-91 */
static inline int br_prefix(gor_decoder *d, int max_bits)
{
    int val = 0;
    for (int i = 0; i < max_bits; i++) {
        val <<= 1;
        if (br_get(d, 1)) val |= 1; else break;
        if (d->err) break;
    }
    return val;
}

gor_decoder *gor_dec_new(int format, const uint64_t *words, size_t nwords, int dfcm_size)
{
    gor_decoder *d = (gor_decoder *)calloc(1, sizeof *d);
    d->format = format;
    d->w = words;
    d->limit_bits = (uint64_t)nwords * 64;
    d->lead = 0x7fffffff;
    pred_init(&d->pred, format == GOR_FMT_V2 ? dfcm_size : 0);
    if (nwords == 0) d->err = GOR_ERR_TRUNCATED; /* constructor loads word 0 */
    d->block_ts = (int64_t)br_get(d, 64);        /* readHeader */
    return d;
}

void gor_dec_free(gor_decoder *d)
{
    if (!d) return;
    free(d->pred.table);
    free(d);
}

/* v2 value: G/ValueDecompressor.java:32-54 */
static uint64_t v2_next_value(gor_decoder *d)
{
    int c = br_prefix(d, 2);
    if (c == 0 || d->err) return pred_predict(&d->pred);
    if (c == 3) {
        d->lead = (int)br_get(d, 6);
        int sig = (int)(int8_t)br_get(d, 6) + 1;
        d->trail = 64 - sig - d->lead;
    }
    int n = 64 - d->lead - d->trail;
    if (n < 1 || n > 64) { if (!d->err) d->err = GOR_ERR_MALFORMED; return 0; } /* Java: mask-table index out of range */
    uint64_t v = br_get(d, n);
    v <<= (d->trail & 63); /* Java long shift uses the count mod 64 */
    v ^= pred_predict(&d->pred);
    pred_update(&d->pred, v);
    return v;
}

/* v1 value: G/Decompressor.java:111-130 */
static void v1_next_value(gor_decoder *d)
{
    if (!br_get(d, 1) || d->err) return;
    if (br_get(d, 1)) {
        d->lead = (int)br_get(d, 5);
        int sig = (int)br_get(d, 6);
        if (sig == 0) sig = 64;
        d->trail = 64 - sig - d->lead;
    }
    if (d->err) return;
    int n = 64 - d->lead - d->trail;
    if (n < 1 || n > 64) { d->err = GOR_ERR_MALFORMED; return; }
    uint64_t v = br_get(d, n);
    v <<= (d->trail & 63);
    d->stored_val ^= v;
}

int gor_dec_next(gor_decoder *d, int64_t *ts, uint64_t *val)
{
    if (d->err) return d->err;
    if (d->eos) return GOR_END;
    if (d->stored_ts == 0) {
        /* first(): G/GorillaDecompressor.java:62-72, G/Decompressor.java:45-53 */
        d->stored_delta = (int64_t)br_get(d, 27);
        if (d->err) return d->err;
        if (d->stored_delta == (1 << 27) - 1) { d->eos = 1; return GOR_END; }
        uint64_t v = br_get(d, 64);
        if (d->format == GOR_FMT_V2) pred_update(&d->pred, v);
        d->stored_val = v;
        d->stored_ts = (int64_t)((uint64_t)d->block_ts + (uint64_t)d->stored_delta);
    } else {
        /* nextTimestamp(): G/GorillaDecompressor.java:74-112, G/Decompressor.java:59-109 */
        int c = br_prefix(d, 4);
        int64_t dd = 0;
        int nb = c == 0x2 ? 7 : c == 0x6 ? 9 : c == 0xE ? 12 : c == 0xF ? 32 : 0;
        if (nb) {
            uint64_t raw = br_get(d, nb);
            if (d->err) return d->err;
            if (nb == 32 && (uint32_t)raw == 0xFFFFFFFFu) { d->eos = 1; return GOR_END; }
            if (d->format == GOR_FMT_V2) {
                uint32_t z = (uint32_t)(raw + 1);
                dd = (int64_t)(int32_t)((z >> 1) ^ (uint32_t)-(int32_t)(z & 1)); /* decodeZigZag32, :125-127 */
            } else {
                int64_t x = (int64_t)raw;
                if (nb != 32 && x > ((int64_t)1 << (nb - 1))) x -= ((int64_t)1 << nb); /* :98-100 */
                dd = (int64_t)(int32_t)(uint64_t)x;                                     /* :103 */
            }
        }
        d->stored_delta = (int64_t)((uint64_t)d->stored_delta + (uint64_t)dd);
        d->stored_ts = (int64_t)((uint64_t)d->stored_ts + (uint64_t)d->stored_delta);
        if (d->format == GOR_FMT_V2) d->stored_val = v2_next_value(d);
        else v1_next_value(d);
    }
    if (d->err) return d->err;
    *ts = d->stored_ts;
    *val = d->stored_val;
    return GOR_OK;
}

/* ------------------------------------------------------------ batch forms */
uint64_t gor_max_words(uint64_t p)
{
    uint64_t bits = 64 + (p ? 91 + 114 * (p - 1) : 0) + 37;
    return bits / 64 + 2;
}

typedef struct {
    int format, encode;
    uint64_t s0, s1;
    const int64_t *block_ts;
    const int64_t *ts_in; const uint64_t *val_in;
    int64_t *ts_out; uint64_t *val_out;
    const uint64_t *pt_off;
    uint64_t *words; const uint64_t *w_off; const uint64_t *w_n_in;
    uint64_t *out_count; uint32_t *status;
    int bad;
} job;

static void *job_run(void *arg)
{
    job *j = (job *)arg;
    for (uint64_t s = j->s0; s < j->s1; s++) {
        uint32_t st = 0;
        if (j->encode) {
            gor_encoder *e = gor_enc_new(j->format, j->block_ts[s], 0);
            for (uint64_t i = j->pt_off[s]; i < j->pt_off[s + 1]; i++) gor_enc_add(e, j->ts_in[i], j->val_in[i]);
            gor_enc_close(e);
            size_t slot = (size_t)(j->w_off[s + 1] - j->w_off[s]);
            size_t n = gor_enc_words(e, NULL, 0);
            if (n > slot) st |= 1u;
            gor_enc_words(e, j->words + j->w_off[s], slot);
            j->out_count[s] = n;
            gor_enc_free(e);
        } else {
            size_t nw = j->w_n_in ? (size_t)j->w_n_in[s] : (size_t)(j->w_off[s + 1] - j->w_off[s]);
            gor_decoder *d = gor_dec_new(j->format, j->words + j->w_off[s], nw, 0);
            uint64_t cap = j->pt_off[s + 1] - j->pt_off[s], n = 0;
            int64_t t; uint64_t v; int rc;
            while ((rc = gor_dec_next(d, &t, &v)) == GOR_OK) {
                if (n < cap) { j->ts_out[j->pt_off[s] + n] = t; j->val_out[j->pt_off[s] + n] = v; }
                else st |= 1u;
                n++;
            }
            if (rc == GOR_ERR_TRUNCATED) st |= 2u;
            if (rc == GOR_ERR_MALFORMED) st |= 4u;
            j->out_count[s] = n;
            gor_dec_free(d);
        }
        if (j->status) j->status[s] = st;
        if (st) j->bad = 1;
    }
    return NULL;
}

static int run_jobs(job *proto, uint64_t nseries, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > nseries) nthreads = nseries ? (int)nseries : 1;
    job *jobs = (job *)calloc((size_t)nthreads, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    /* contiguous ranges balanced by point count (by series count when there are no points) */
    uint64_t total = proto->pt_off[nseries] - proto->pt_off[0], s = 0;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = *proto;
        jobs[t].s0 = s;
        if (t == nthreads - 1) {
            s = nseries;
        } else if (total == 0) {
            s = nseries * (uint64_t)(t + 1) / (uint64_t)nthreads;
        } else {
            uint64_t target = proto->pt_off[0] + total / (uint64_t)nthreads * (uint64_t)(t + 1);
            while (s < nseries && proto->pt_off[s + 1] <= target) s++;
        }
        jobs[t].s1 = s;
    }
    for (int t = 1; t < nthreads; t++) pthread_create(&th[t], NULL, job_run, &jobs[t]);
    job_run(&jobs[0]);
    int bad = jobs[0].bad;
    for (int t = 1; t < nthreads; t++) { pthread_join(th[t], NULL); bad |= jobs[t].bad; }
    free(jobs); free(th);
    return bad ? -1 : 0;
}

int gor_encode_batch(int format, uint64_t nseries, const int64_t *block_ts, const int64_t *ts,
                     const uint64_t *val, const uint64_t *pt_off, uint64_t *out_words,
                     const uint64_t *out_off, uint64_t *out_nwords, uint32_t *status, int nthreads)
{
    job p; memset(&p, 0, sizeof p);
    p.format = format; p.encode = 1; p.block_ts = block_ts; p.ts_in = ts; p.val_in = val;
    p.pt_off = pt_off; p.words = out_words; p.w_off = out_off; p.out_count = out_nwords; p.status = status;
    return run_jobs(&p, nseries, nthreads);
}

int gor_decode_batch(int format, uint64_t nseries, const uint64_t *words, const uint64_t *in_off,
                     const uint64_t *in_nwords, int64_t *ts, uint64_t *val, const uint64_t *pt_off,
                     uint64_t *out_npoints, uint32_t *status, int nthreads)
{
    job p; memset(&p, 0, sizeof p);
    p.format = format; p.encode = 0; p.words = (uint64_t *)words; p.w_off = in_off; p.w_n_in = in_nwords;
    p.ts_out = ts; p.val_out = val; p.pt_off = pt_off; p.out_count = out_npoints; p.status = status;
    return run_jobs(&p, nseries, nthreads);
}
