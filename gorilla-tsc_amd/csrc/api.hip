// C ABI of the gtsc library (see include/gtsc.h).  Host code only: argument
// checks, stream/scratch ownership, chunked host<->device staging.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gtsc.h"
#include "kernels.hpp"

using namespace gtsc;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(GTSC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; bytes = 0; }
        size_t want = need + need / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // the failed attempt must not show up as a later launch's error
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// page-locked host memory that grows (the per-chunk offset arrays of the host pipeline: a copy from
// pageable memory would stall the stream it is enqueued on)
struct HostBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipHostFree(p); if (e != hipSuccess) return e; p = nullptr; bytes = 0; }
        const size_t want = need + need / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

constexpr size_t FLAG_BLOCK_BYTES = 64;
// Above this many compressed bits per point the decoder's 16-word input rings pay (measured cross-over:
// gauges at 5 bits lose 8 %, drifting gauges at 47 bits gain 3 %, random walks and random doubles 10 %).
constexpr uint64_t DENSE_BITS_PER_POINT = 40;

// The host pipeline has four streams -- all copy-ins, all kernels but one, the workgroup-per-block decompressor
// beside them, all copy-outs, each back to back -- and NSTAGE sets of device buffers that chunks take in turn, tied
// together by events.  The HIP runtime maps the streams of one priority onto four hardware queues by default
// (GPU_MAX_HW_QUEUES), and streams that share a queue run in order (a copy-out behind another chunk's kernels): see
// ensure_stage_streams() for how the copy streams are kept apart from the others.
constexpr int NSTAGE = 4;

struct gtsc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;   // the one launches go to
    // device block of FLAG_BLOCK_BYTES: word 0 = OR of status bits raised since the last sync
    uint32_t* d_flags = nullptr;
    uint32_t* h_flags = nullptr;    // pinned mirror
    // The host entry points raise their status bits in a block of their own (the second
    // FLAG_BLOCK_BYTES of the same allocations): they run on other streams than the device entry
    // points, whose flags and counts wait for gtsc_sync() untouched.
    uint32_t* d_flags_host = nullptr;
    uint32_t* h_flags_host = nullptr;
    // then the decode plans (what a batch of images looks like, worked out on the device before its kernels
    // start): one for the device entry point, one per stage of the host pipeline
    DecodePlan* d_plans = nullptr;
    uint64_t* noalloc_scratch = nullptr;  // this device's copy of the library's static scratch array
    Tunables tun;                   // gtsc_set_tunable; nothing the context has coded before enters a choice of kernel
    uint32_t last_dispatch = 0;     // gtsc_last_dispatch: the kernels the last device call enqueued
    DeviceBuf scan_scratch;
    // predictor of the value codec: 0 = last value, else DFCM with tables of dfcm_mask + 1 entries
    int predictor = GTSC_PRED_LAST;
    uint32_t dfcm_mask = 0;
    DeviceBuf dfcm_tables[1 + NSTAGE];  // device entry points, then one per stage of the host pipeline
    // staging for the host entry points
    hipStream_t copy_in = nullptr, work = nullptr, copy_out = nullptr, work2 = nullptr;
    bool holds_priority_pair = false;   // copy_in and copy_out were made with the highest stream priority
    struct Stage {
        // the chunk's input has arrived / its kernels are done / its results have left the stage's buffers
        hipEvent_t copied_in = nullptr, coded = nullptr, drained = nullptr, planned = nullptr, joined = nullptr;
        HostBuf h_off_a, h_off_b;   // this chunk's offset arrays, rebased to the chunk
        DeviceBuf block_ts, ts, val, pt_off, slots, slot_off, nwords, nbits, coff, compact, status, npoints;
    } stage[NSTAGE];
};

namespace {

void release_priority_pair(int device);

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int check_ctx(gtsc_ctx* ctx)
{
    if (!ctx) return fail(GTSC_ERR_ARG, "ctx is NULL");
    return GTSC_OK;
}

}  // namespace

namespace {

// the tunables by number (GTSC_TUNE_* in gtsc.h), with the environment variable that presets each
struct TunableSlot { int id; const char* env; uint64_t Tunables::* u64; uint32_t Tunables::* u32; };
const TunableSlot TUNABLES[] = {
    {GTSC_TUNE_LONG_POINTS, "GTSC_LONG_POINTS", &Tunables::long_points, nullptr},
    {GTSC_TUNE_VLONG_POINTS, "GTSC_VLONG_POINTS", &Tunables::vlong_points, nullptr},
    {GTSC_TUNE_WAVE_DECODE_MAX, "GTSC_LONG_DECODE_MAX", &Tunables::wave_decode_max, nullptr},
    {GTSC_TUNE_WAVE_DECODE_MAX_DENSE, "GTSC_LONG_DECODE_MAX_DENSE", &Tunables::wave_decode_max_dense, nullptr},
    {GTSC_TUNE_WG_DECODE_MAX, "GTSC_WG_DECODE_MAX", &Tunables::wg_decode_max, nullptr},
    {GTSC_TUNE_WG_DECODE_VLONG_MAX, "GTSC_WG_DECODE_VLONG_MAX", &Tunables::wg_decode_vlong_max, nullptr},
    {GTSC_TUNE_WG_DENSE_BITS, "GTSC_WG_DENSE_BITS", nullptr, &Tunables::wg_dense_bits},
    {GTSC_TUNE_DENSE_BITS, "GTSC_DENSE_BITS", nullptr, &Tunables::dense_bits},
    {GTSC_TUNE_SPARSE_ENCODE_MIN, "GTSC_SPARSE_TILED_MIN", &Tunables::sparse_encode_min, nullptr},
    {GTSC_TUNE_COOP_ENCODE_MAX, "GTSC_COOP_MAX", &Tunables::coop_encode_max, nullptr},
    {GTSC_TUNE_SPARSE_HINT, "GTSC_SPARSE_HINT", nullptr, &Tunables::sparse_hint},
    {GTSC_TUNE_DECODE_RING, "GTSC_RING", nullptr, &Tunables::decode_ring},
    {GTSC_TUNE_RUN_SKIP, "GTSC_RUN_SKIP", nullptr, &Tunables::run_skip},
    {GTSC_TUNE_DIRECT_KERNELS, "GTSC_DIRECT", nullptr, &Tunables::direct},
};

void set_tunable(Tunables& t, const TunableSlot& slot, uint64_t v)
{
    if (slot.u64) t.*slot.u64 = v;
    else t.*slot.u32 = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
}

bool tunable_value_ok(int id, uint64_t v)
{
    if (id == GTSC_TUNE_DECODE_RING) return v == 0 || v == 8 || v == 16;
    if (id == GTSC_TUNE_SPARSE_HINT || id == GTSC_TUNE_RUN_SKIP || id == GTSC_TUNE_DIRECT_KERNELS) return v <= 1;
    return true;
}

void tunables_from_environment(Tunables& t)
{
    for (const TunableSlot& slot : TUNABLES) {
        const char* e = getenv(slot.env);
        if (!e || !*e) continue;
        const uint64_t v = strtoull(e, nullptr, 10);
        if (tunable_value_ok(slot.id, v)) set_tunable(t, slot, v);
    }
    const char* k = getenv("GTSC_KERNEL");   // (the older spelling of GTSC_DIRECT=1)
    if (k && !strcmp(k, "direct")) t.direct = 1;
}

}  // namespace

// Zero-filled DFCM tables for nseries blocks in buffer `which`, on `stream`; null when the context
// uses the last-value predictor.  (G/predictors/DifferentialFCM.java:23-33: new long[size], hash 0.)
static hipError_t dfcm_prepare(gtsc_ctx* c, int which, uint64_t nseries, hipStream_t stream, int64_t** tables)
{
    *tables = nullptr;
    if (c->predictor != GTSC_PRED_DFCM || nseries == 0) return hipSuccess;
    const size_t bytes = (size_t)nseries * ((size_t)c->dfcm_mask + 1) * sizeof(int64_t);
    DeviceBuf& b = c->dfcm_tables[which];
    if (hipError_t e = b.reserve(bytes)) return e;
    if (hipError_t e = hipMemsetAsync(b.p, 0, bytes, stream)) return e;
    *tables = b.as<int64_t>();
    return hipSuccess;
}

// true when any of the given (possibly NULL) pointers is not 8-byte aligned
template <typename... P>
static bool misaligned8(P... p)
{
    return (((uintptr_t)0 | ... | (uintptr_t)p) & 7u) != 0;
}

extern "C" {

int gtsc_version(void) { return 1000; }

const char* gtsc_last_error(void) { return g_err; }

uint64_t gtsc_max_words(uint64_t npoints)
{
    const uint64_t bits = 64 + (npoints ? 91 + 114 * (npoints - 1) : 0) + 37;
    return bits / 64 + 2;
}

int gtsc_create(int device_id, gtsc_ctx** out)
{
    if (!out) return fail(GTSC_ERR_ARG, "ctx out-pointer is NULL");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(GTSC_ERR_HIP, "no HIP device available");
    if (device_id < 0) HIP_TRY(hipGetDevice(&device_id));
    if (device_id >= ndev) return fail(GTSC_ERR_ARG, "device %d out of range (%d devices)", device_id, ndev);
    DeviceGuard g(device_id);
    if (!g.ok) return fail(GTSC_ERR_HIP, "cannot select device %d", device_id);
    gtsc_ctx* c = new (std::nothrow) gtsc_ctx();
    if (!c) return fail(GTSC_ERR_ARG, "out of host memory");
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    constexpr size_t DEV_BYTES = 2 * FLAG_BLOCK_BYTES + (1 + NSTAGE) * sizeof(DecodePlan);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_flags, DEV_BYTES);
    if (e == hipSuccess) e = hipMemset(c->d_flags, 0, DEV_BYTES);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_flags, 2 * FLAG_BLOCK_BYTES, hipHostMallocDefault);
    if (e == hipSuccess) {
        c->d_flags_host = c->d_flags + FLAG_BLOCK_BYTES / sizeof(uint32_t);
        c->h_flags_host = c->h_flags + FLAG_BLOCK_BYTES / sizeof(uint32_t);
        c->d_plans = reinterpret_cast<DecodePlan*>(c->d_flags + 2 * FLAG_BLOCK_BYTES / sizeof(uint32_t));
        tunables_from_environment(c->tun);
    }
    if (e == hipSuccess) e = noalloc_scratch_of_current_device(&c->noalloc_scratch);
    if (e != hipSuccess) {
        int rc = fail(GTSC_ERR_HIP, "context setup: %s", hipGetErrorString(e));
        gtsc_destroy(c);
        return rc;
    }
    c->stream = c->own_stream;
    *out = c;
    return GTSC_OK;
}

int gtsc_destroy(gtsc_ctx* c)
{
    if (!c) return GTSC_OK;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipStream_t q : {c->copy_in, c->work, c->copy_out, c->work2})
        if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    if (c->holds_priority_pair) release_priority_pair(c->device);
    for (auto& s : c->stage) {
        for (hipEvent_t e : {s.copied_in, s.coded, s.drained, s.planned, s.joined}) if (e) (void)hipEventDestroy(e);
        s.h_off_a.release(); s.h_off_b.release();
        for (DeviceBuf* b : {&s.block_ts, &s.ts, &s.val, &s.pt_off, &s.slots, &s.slot_off, &s.nwords, &s.nbits, &s.coff,
                             &s.compact, &s.status, &s.npoints})
            b->release();
    }
    c->scan_scratch.release();
    for (DeviceBuf& b : c->dfcm_tables) b.release();
    if (c->d_flags) (void)hipFree(c->d_flags);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return GTSC_OK;
}

int gtsc_set_predictor(gtsc_ctx* c, int predictor, int table_size)
{
    if (int rc = check_ctx(c)) return rc;
    if (predictor == GTSC_PRED_LAST) {
        c->predictor = GTSC_PRED_LAST;
        c->dfcm_mask = 0;
        return GTSC_OK;
    }
    if (predictor != GTSC_PRED_DFCM) return fail(GTSC_ERR_ARG, "unknown predictor %d", predictor);
    if (table_size <= 0) return fail(GTSC_ERR_ARG, "table size must be positive");  // IllegalArgumentException in Java
    // rounded up to a power of two: 1 << (64 - numberOfLeadingZeros(size - 1))
    const uint64_t s1 = (uint64_t)table_size - 1;
    const int lz = s1 ? __builtin_clzll(s1) : 64;
    c->predictor = GTSC_PRED_DFCM;
    c->dfcm_mask = (uint32_t)((1ull << (64 - lz)) - 1);
    return GTSC_OK;
}

int gtsc_set_decode_ring(gtsc_ctx* c, int ring_words)
{
    return gtsc_set_tunable(c, GTSC_TUNE_DECODE_RING, ring_words < 0 ? ~0ull : (uint64_t)ring_words);
}

int gtsc_set_tunable(gtsc_ctx* c, int which, uint64_t value)
{
    if (int rc = check_ctx(c)) return rc;
    for (const TunableSlot& slot : TUNABLES) {
        if (slot.id != which) continue;
        if (!tunable_value_ok(which, value)) return fail(GTSC_ERR_ARG, "tunable %d does not take the value %llu", which, (unsigned long long)value);
        set_tunable(c->tun, slot, value);
        return GTSC_OK;
    }
    return fail(GTSC_ERR_ARG, "unknown tunable %d", which);
}

int gtsc_get_tunable(gtsc_ctx* c, int which, uint64_t* value)
{
    if (int rc = check_ctx(c)) return rc;
    if (!value) return fail(GTSC_ERR_ARG, "value out-pointer is NULL");
    for (const TunableSlot& slot : TUNABLES) {
        if (slot.id != which) continue;
        *value = slot.u64 ? c->tun.*slot.u64 : (uint64_t)(c->tun.*slot.u32);
        return GTSC_OK;
    }
    return fail(GTSC_ERR_ARG, "unknown tunable %d", which);
}

int gtsc_last_dispatch(gtsc_ctx* c, uint32_t* kernels, uint64_t plan[6])
{
    if (int rc = check_ctx(c)) return rc;
    if (kernels) *kernels = c->last_dispatch;
    if (plan && !(c->last_dispatch & DISPATCH_PLAN)) {
        for (int i = 0; i < 6; i++) plan[i] = 0;   // the last call needed no plan
    } else if (plan) {
        // what the device made of the last gtsc_decode_device batch (waits for the context's stream)
        DeviceGuard g(c->device);
        HIP_TRY(hipMemcpyAsync(plan, c->d_plans, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return GTSC_OK;
}

int gtsc_set_stream(gtsc_ctx* c, void* hip_stream)
{
    if (int rc = check_ctx(c)) return rc;
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return GTSC_OK;
}

int gtsc_sync(gtsc_ctx* c)
{
    if (int rc = check_ctx(c)) return rc;
    DeviceGuard g(c->device);
    HIP_TRY(hipMemcpyAsync(c->h_flags, c->d_flags, FLAG_BLOCK_BYTES, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_flags, 0, FLAG_BLOCK_BYTES, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t f = *c->h_flags;
    if (f & (ST_TRUNCATED | ST_MALFORMED)) return fail(GTSC_ERR_STREAM, "some images were truncated or malformed (status bits 0x%x)", f);
    if (f & ST_OVERFLOW) return fail(GTSC_ERR_CAPACITY, "some series overflowed their output slot");
    return GTSC_OK;
}

int gtsc_encode_device_ex(gtsc_ctx* c, int format, uint32_t options, uint64_t nseries, const int64_t* block_ts,
                          const int64_t* ts, const uint64_t* val, const uint64_t* pt_off, uint64_t uniform_npoints,
                          uint64_t* out_words, const uint64_t* out_off, uint64_t uniform_slot_words, uint64_t* out_nwords,
                          uint64_t* out_nbits, uint32_t* status)
{
    if (int rc = check_ctx(c)) return rc;
    if (format != GTSC_FMT_V2 && format != GTSC_FMT_V1) return fail(GTSC_ERR_ARG, "unknown format %d", format);
    if (options & ~(uint32_t)GTSC_ENC_NO_CLOSE) return fail(GTSC_ERR_ARG, "unknown encode options 0x%x", options);
    if (nseries == 0) return GTSC_OK;
    if (!block_ts || !out_words || !out_nwords) return fail(GTSC_ERR_ARG, "block_ts/out_words/out_nwords must not be NULL");
    if ((!ts || !val) && (pt_off || uniform_npoints)) return fail(GTSC_ERR_ARG, "ts/val must not be NULL");
    if (misaligned8(block_ts, ts, val, pt_off, out_words, out_off, out_nwords, out_nbits))
        return fail(GTSC_ERR_ARG, "64-bit arrays must be 8-byte aligned");
    if (!out_off && uniform_slot_words < 3) return fail(GTSC_ERR_ARG, "uniform_slot_words must be at least 3 (an empty block)");
    if (c->predictor != GTSC_PRED_LAST && format != GTSC_FMT_V2)
        return fail(GTSC_ERR_ARG, "only the 2.x format takes a predictor");
    DeviceGuard g(c->device);
    int64_t* tables = nullptr;
    HIP_TRY(dfcm_prepare(c, 0, nseries, c->stream, &tables));
    c->last_dispatch = 0;
    EncodeArgs a{nseries, block_ts, ts, val, Layout{pt_off, uniform_npoints}, out_words, Layout{out_off, uniform_slot_words},
                 out_nwords, status, c->d_flags, out_nbits, (options & GTSC_ENC_NO_CLOSE) ? 1u : 0u, tables, c->dfcm_mask, 0u, ~0ull, 0u,
                 c->tun, &c->last_dispatch};
    HIP_TRY(launch_encode(a, format, c->stream));
    return GTSC_OK;
}

int gtsc_encode_device(gtsc_ctx* c, int format, uint64_t nseries, const int64_t* block_ts, const int64_t* ts,
                       const uint64_t* val, const uint64_t* pt_off, uint64_t uniform_npoints, uint64_t* out_words,
                       const uint64_t* out_off, uint64_t uniform_slot_words, uint64_t* out_nwords, uint32_t* status)
{
    return gtsc_encode_device_ex(c, format, 0, nseries, block_ts, ts, val, pt_off, uniform_npoints, out_words, out_off,
                                 uniform_slot_words, out_nwords, nullptr, status);
}

int gtsc_words_to_bytes_device(gtsc_ctx* c, uint64_t nseries, const uint64_t* words, const uint64_t* word_off,
                               uint64_t uniform_slot_words, const uint64_t* nbits, uint8_t* out_bytes,
                               const uint64_t* byte_off, uint64_t uniform_slot_bytes, uint64_t* out_nbytes)
{
    if (int rc = check_ctx(c)) return rc;
    if (nseries == 0) return GTSC_OK;
    if (!words || !nbits || !out_bytes || !out_nbytes) return fail(GTSC_ERR_ARG, "words/nbits/out_bytes/out_nbytes must not be NULL");
    if (misaligned8(words, word_off, nbits, byte_off, out_nbytes)) return fail(GTSC_ERR_ARG, "64-bit arrays must be 8-byte aligned");
    DeviceGuard g(c->device);
    HIP_TRY(launch_words_to_bytes(nseries, words, Layout{word_off, uniform_slot_words}, nbits, out_bytes,
                                  Layout{byte_off, uniform_slot_bytes}, out_nbytes, c->d_flags, c->stream));
    return GTSC_OK;
}

int gtsc_bytes_to_words_device(gtsc_ctx* c, uint64_t nseries, const uint8_t* bytes, const uint64_t* byte_off,
                               uint64_t uniform_slot_bytes, const uint64_t* nbytes, uint64_t* out_words,
                               const uint64_t* word_off, uint64_t uniform_slot_words, uint64_t* out_nwords)
{
    if (int rc = check_ctx(c)) return rc;
    if (nseries == 0) return GTSC_OK;
    if (!bytes || !out_words || !out_nwords) return fail(GTSC_ERR_ARG, "bytes/out_words/out_nwords must not be NULL");
    if (misaligned8(byte_off, nbytes, out_words, word_off, out_nwords)) return fail(GTSC_ERR_ARG, "64-bit arrays must be 8-byte aligned");
    DeviceGuard g(c->device);
    HIP_TRY(launch_bytes_to_words(nseries, bytes, Layout{byte_off, uniform_slot_bytes}, nbytes, out_words,
                                  Layout{word_off, uniform_slot_words}, out_nwords, c->d_flags, c->stream));
    return GTSC_OK;
}

int gtsc_decode_device(gtsc_ctx* c, int format, uint64_t nseries, const uint64_t* words, const uint64_t* in_off,
                       uint64_t uniform_slot_words, const uint64_t* in_nwords, int64_t* ts, uint64_t* val,
                       const uint64_t* pt_off, uint64_t uniform_capacity, uint64_t* out_npoints, uint32_t* status)
{
    if (int rc = check_ctx(c)) return rc;
    if (format != GTSC_FMT_V2 && format != GTSC_FMT_V1) return fail(GTSC_ERR_ARG, "unknown format %d", format);
    if (nseries == 0) return GTSC_OK;
    if (!words || !out_npoints) return fail(GTSC_ERR_ARG, "words/out_npoints must not be NULL");
    if ((!ts || !val) && (pt_off || uniform_capacity)) return fail(GTSC_ERR_ARG, "ts/val must not be NULL");
    if (misaligned8(words, in_off, in_nwords, ts, val, pt_off, out_npoints))
        return fail(GTSC_ERR_ARG, "64-bit arrays must be 8-byte aligned");
    if (c->predictor != GTSC_PRED_LAST && format != GTSC_FMT_V2)
        return fail(GTSC_ERR_ARG, "only the 2.x format takes a predictor");
    DeviceGuard g(c->device);
    int64_t* tables = nullptr;
    HIP_TRY(dfcm_prepare(c, 0, nseries, c->stream, &tables));
    c->last_dispatch = 0;
    DecodeArgs a{nseries, words, Layout{in_off, uniform_slot_words}, in_nwords, ts, val, Layout{pt_off, uniform_capacity},
                 out_npoints, status, c->d_flags, tables, c->dfcm_mask, 0u, ~0ull, nullptr, c->d_plans, 0u, 0u, c->tun, &c->last_dispatch};
    // (No second stream here, unlike the host pipeline: a caller's blocking copy on this stream right after the call
    // -- hipMemcpy into pageable memory -- would wait inside the runtime for kernels on two streams, and that wait has
    // been seen not to return.)
    HIP_TRY(launch_decode(a, format, c->stream));
    return GTSC_OK;
}

int gtsc_compact_device(gtsc_ctx* c, uint64_t nseries, const uint64_t* slot_words, const uint64_t* slot_off,
                        uint64_t uniform_slot_words, const uint64_t* nwords, uint64_t* compact_words, uint64_t* compact_off)
{
    if (int rc = check_ctx(c)) return rc;
    if (!compact_off) return fail(GTSC_ERR_ARG, "compact_off must not be NULL");
    if (nseries && (!slot_words || !nwords || !compact_words)) return fail(GTSC_ERR_ARG, "slot_words/nwords/compact_words must not be NULL");
    DeviceGuard g(c->device);
    HIP_TRY(c->scan_scratch.reserve(scan_scratch_words(nseries) * sizeof(uint64_t)));
    HIP_TRY(launch_exclusive_scan(nwords, nseries, compact_off, c->scan_scratch.as<uint64_t>(), c->stream));
    HIP_TRY(launch_compact_copy(nseries, slot_words, Layout{slot_off, uniform_slot_words}, nwords, compact_words,
                                compact_off, c->stream));
    return GTSC_OK;
}

// The same compaction for callers that cannot allocate: the scan's partial sums go to the
// library's static device array instead of a buffer of the context.  Nothing in there has to
// start at zero (every partial sum is written before it is read), and no hipMalloc, hipFree or
// allocating routine is called on this path.
int gtsc_compact_device_noalloc(gtsc_ctx* c, uint64_t nseries, const uint64_t* slot_words, const uint64_t* slot_off,
                                uint64_t uniform_slot_words, const uint64_t* nwords, uint64_t* compact_words,
                                uint64_t* compact_off)
{
    if (int rc = check_ctx(c)) return rc;
    if (!compact_off) return fail(GTSC_ERR_ARG, "compact_off must not be NULL");
    if (nseries && (!slot_words || !nwords || !compact_words)) return fail(GTSC_ERR_ARG, "slot_words/nwords/compact_words must not be NULL");
    if (nseries > GTSC_NOALLOC_MAX_SERIES)
        return fail(GTSC_ERR_ARG, "gtsc_compact_device_noalloc takes at most %llu series per call (got %llu)",
                    (unsigned long long)GTSC_NOALLOC_MAX_SERIES, (unsigned long long)nseries);
    static_assert((GTSC_NOALLOC_MAX_SERIES + 1023) / 1024 + 1 <= NOALLOC_SCRATCH_WORDS, "scratch too small for the limit");
    if (!c->noalloc_scratch) return fail(GTSC_ERR_HIP, "the static scratch array was not found on device %d", c->device);
    DeviceGuard g(c->device);
    HIP_TRY(launch_exclusive_scan(nwords, nseries, compact_off, c->noalloc_scratch, c->stream));
    HIP_TRY(launch_compact_copy(nseries, slot_words, Layout{slot_off, uniform_slot_words}, nwords, compact_words,
                                compact_off, c->stream));
    return GTSC_OK;
}

// ------------------------------------------------------------------------
// Host-memory entry points: chunked, NSTAGE streams taking turns.
// ------------------------------------------------------------------------
int gtsc_host_pipeline_idle(gtsc_ctx* c)
{
    if (int rc = check_ctx(c)) return rc;
    DeviceGuard g(c->device);
    for (hipStream_t q : {c->copy_in, c->work, c->copy_out, c->work2}) {
        if (!q) continue;
        const hipError_t e = hipStreamQuery(q);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) return fail(GTSC_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
    }
    return 1;
}

void* gtsc_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(GTSC_ERR_HIP, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void gtsc_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

}  // extern "C"

namespace {

// Points per chunk of the host pipeline.  Compression has no chain along a block (a long block is spread over the
// lanes of a wave), and chunks of 256 MiB let the copy of one hide the work on the one before.  Decompression of a
// block is a chain (where a point starts is only known from the points before it), so a chunk's kernels take as long
// as its longest block whatever its size -- about 4 ms with 100,000-point blocks, the time it takes to copy out 14
// million points: chunks of 16 Mi points keep the copies out back to back.  But the runtime's copies out slow down
// once a call has made more than two dozen of them (every sixteenth, from about the twenty-eighth on, takes four and
// a half times as long, whatever the chunk size: 268 million points in 16 Mi chunks decode at 33 GB/s, in 64 Mi
// chunks at 54): a call is cut into six chunks or fewer where chunks of 64 Mi points allow that.
constexpr uint64_t CHUNK_POINTS = 64ull << 20;         // 1 GiB of raw points: the largest chunk
constexpr uint64_t ENCODE_CHUNK_POINTS = 16ull << 20;  // 256 MiB
constexpr uint64_t DECODE_CHUNK_POINTS = 16ull << 20;  // the smallest chunk of a large batch
constexpr uint64_t DECODE_CHUNKS = 6;
constexpr uint64_t CHUNK_SERIES = 4ull << 20;

// Chunk size in points; GTSC_CHUNK_POINTS overrides it (tests use it to get many chunks from a small batch).
uint64_t chunk_points(bool encode, uint64_t total_points)
{
    static const uint64_t forced = [] {
        const char* e = getenv("GTSC_CHUNK_POINTS");
        const uint64_t n = e ? strtoull(e, nullptr, 10) : 0;
        return n < CHUNK_POINTS ? n : CHUNK_POINTS;
    }();
    if (forced) return forced;
    if (encode) return ENCODE_CHUNK_POINTS;
    const uint64_t even = (total_points + DECODE_CHUNKS - 1) / DECODE_CHUNKS;
    return even < DECODE_CHUNK_POINTS ? DECODE_CHUNK_POINTS : even > CHUNK_POINTS ? CHUNK_POINTS : even;
}

struct Chunk { uint64_t s0, s1; };

// cut [0, nseries) into runs of whole series of at most CHUNK_POINTS points
std::vector<Chunk> plan_chunks(uint64_t nseries, const uint64_t* off, uint64_t uniform, bool encode)
{
    std::vector<Chunk> out;
    const uint64_t most = chunk_points(encode, off ? off[nseries] - off[0] : nseries * uniform);
    uint64_t s = 0;
    while (s < nseries) {
        uint64_t e = s, pts = 0;
        while (e < nseries && e - s < CHUNK_SERIES) {
            const uint64_t n = off ? off[e + 1] - off[e] : uniform;
            if (e > s && pts + n > most) break;
            pts += n;
            e++;
        }
        out.push_back({s, e});
        s = e;
    }
    return out;
}

constexpr int MAX_DEVICES = 64;
std::atomic<int> priority_pairs[MAX_DEVICES];   // contexts per device whose copy streams have the highest priority

void release_priority_pair(int device)
{
    if (device >= 0 && device < MAX_DEVICES) priority_pairs[device].fetch_sub(1);
}

int ensure_stage_streams(gtsc_ctx* c)
{
    // The runtime hands streams of one priority a small pool of hardware queues (four by default), and streams that
    // share a queue run in order -- a copy-out behind the next chunk's kernels.  The two copy streams are made with
    // the highest priority: that is a pool of their own, and what they move is what everything else waits for.
    // (Two and no more: with a third stream of that priority in the process, calls of many chunks stalled for
    // tens of seconds on the box this was measured on.)
    // One context per device holds such a pair; the copy streams of further contexts on the same device are ordinary
    // streams (they may share queues with others and run slower, but they cannot stall).
    if (!c->copy_in && !c->copy_out) {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int dev = c->device >= 0 && c->device < MAX_DEVICES ? c->device : -1;
        const bool first = dev >= 0 && priority_pairs[dev].fetch_add(1) == 0;
        if (dev >= 0 && !first) priority_pairs[dev].fetch_sub(1);
        c->holds_priority_pair = first;
        for (hipStream_t* q : {&c->copy_in, &c->copy_out})
            HIP_TRY(hipStreamCreateWithPriority(q, hipStreamNonBlocking, first ? greatest : 0));
    }
    for (hipStream_t* q : {&c->work, &c->work2})
        if (!*q) HIP_TRY(hipStreamCreateWithFlags(q, hipStreamNonBlocking));
    for (auto& st : c->stage)
        for (hipEvent_t* e : {&st.copied_in, &st.coded, &st.drained, &st.planned, &st.joined})
            if (!*e) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return GTSC_OK;
}

// Every way out of a host entry point goes through this: whatever the pipeline still has in flight -- copies out of
// the caller's input arrays, copies into its output arrays -- is waited for before the call returns, also when it
// returns an error half way (a buffer that could not grow, a failed launch), so that the caller may free or reuse
// its arrays whatever the return code.
struct PipelineDrain {
    gtsc_ctx* c;
    ~PipelineDrain()
    {
        for (hipStream_t q : {c->copy_in, c->work, c->copy_out, c->work2})
            if (q) (void)hipStreamSynchronize(q);
    }
};

// GTSC_TEST_FAIL_CHUNK=k makes the host pipeline fail (as if a staging buffer could not grow) when it stages its
// k-th chunk: for the test of the error path.
bool injected_failure(size_t chunk)
{
    static const long k = [] { const char* e = getenv("GTSC_TEST_FAIL_CHUNK"); return e && *e ? strtol(e, nullptr, 10) : -1L; }();
    return k >= 0 && (size_t)k == chunk;
}

// Whether p is page-locked (or registered) host memory: copies to and from such memory are asynchronous; with
// pageable memory hipMemcpyAsync blocks the calling thread until the copy is done.
bool page_locked(const void* p)
{
    if (!p) return true;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();   // (not an error of the call's: the pointer is simply not the runtime's)
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// status bits raised by the host pipeline (its streams are idle when this is called)
int read_flags(gtsc_ctx* c, hipStream_t stream, uint32_t* out)
{
    HIP_TRY(hipMemcpyAsync(c->h_flags_host, c->d_flags_host, FLAG_BLOCK_BYTES, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemsetAsync(c->d_flags_host, 0, FLAG_BLOCK_BYTES, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *out = *c->h_flags_host;
    return GTSC_OK;
}

}  // namespace

extern "C" {

int gtsc_encode_host_ex(gtsc_ctx* c, int format, uint32_t options, uint64_t nseries, const int64_t* block_ts,
                        const int64_t* ts, const uint64_t* val, const uint64_t* pt_off, uint64_t uniform_npoints,
                        uint64_t* out_words, uint64_t out_capacity_words, uint64_t* out_off, uint64_t* out_nbits)
{
    if (int rc = check_ctx(c)) return rc;
    if (format != GTSC_FMT_V2 && format != GTSC_FMT_V1) return fail(GTSC_ERR_ARG, "unknown format %d", format);
    if (options & ~(uint32_t)GTSC_ENC_NO_CLOSE) return fail(GTSC_ERR_ARG, "unknown encode options 0x%x", options);
    if (c->predictor != GTSC_PRED_LAST && format != GTSC_FMT_V2)
        return fail(GTSC_ERR_ARG, "only the 2.x format takes a predictor");
    if (!out_off) return fail(GTSC_ERR_ARG, "out_off must not be NULL");
    out_off[0] = 0;
    if (nseries == 0) return GTSC_OK;
    if (!block_ts || !out_words) return fail(GTSC_ERR_ARG, "block_ts/out_words must not be NULL");
    if ((!ts || !val) && (pt_off || uniform_npoints)) return fail(GTSC_ERR_ARG, "ts/val must not be NULL");
    DeviceGuard g(c->device);
    if (int rc = ensure_stage_streams(c)) return rc;
    PipelineDrain drain{c};

    const std::vector<Chunk> chunks = plan_chunks(nseries, pt_off, uniform_npoints, true);
    // Copy-in runs two chunks ahead of the kernels.  On the work stream a chunk's kernels are followed by
    // its copy-out, which the host can only enqueue once it knows the chunk's size -- hence one host wait
    // per chunk, during which the copy-in stream has the next two chunks to move.
    struct Geometry { uint64_t s0, s1, ns, p0, np, slot_total, uniform_slot; const uint64_t* d_pt_off; const uint64_t* d_slot_off; } geo[NSTAGE];
    bool used[NSTAGE] = {};
    auto copy_in_chunk = [&](size_t k) -> int {
        const int b = (int)(k % NSTAGE);
        auto& st = c->stage[b];
        if (used[b]) HIP_TRY(hipEventSynchronize(st.drained));  // the stage's previous chunk has left its buffers
        used[b] = true;
        if (injected_failure(k)) return fail(GTSC_ERR_HIP, "staging chunk %zu: injected failure", k);
        Geometry& g = geo[b];
        g.s0 = chunks[k].s0; g.s1 = chunks[k].s1; g.ns = g.s1 - g.s0;
        g.p0 = pt_off ? pt_off[g.s0] : g.s0 * uniform_npoints;
        g.np = (pt_off ? pt_off[g.s1] : g.s1 * uniform_npoints) - g.p0;
        g.uniform_slot = 0; g.d_pt_off = nullptr; g.d_slot_off = nullptr;
        const uint64_t ns = g.ns;
        if (pt_off) {
            HIP_TRY(st.h_off_a.reserve((ns + 1) * 8));
            HIP_TRY(st.h_off_b.reserve((ns + 1) * 8));
            uint64_t* const rebased = st.h_off_a.as<uint64_t>();
            uint64_t* const slot_off = st.h_off_b.as<uint64_t>();
            slot_off[0] = 0;
            for (uint64_t i = 0; i <= ns; i++) rebased[i] = pt_off[g.s0 + i] - g.p0;
            for (uint64_t i = 0; i < ns; i++) slot_off[i + 1] = slot_off[i] + gtsc_max_words(rebased[i + 1] - rebased[i]);
            g.slot_total = slot_off[ns];
            HIP_TRY(st.pt_off.reserve((ns + 1) * 8));
            HIP_TRY(st.slot_off.reserve((ns + 1) * 8));
            HIP_TRY(hipMemcpyAsync(st.pt_off.p, rebased, (ns + 1) * 8, hipMemcpyHostToDevice, c->copy_in));
            HIP_TRY(hipMemcpyAsync(st.slot_off.p, slot_off, (ns + 1) * 8, hipMemcpyHostToDevice, c->copy_in));
            g.d_pt_off = st.pt_off.as<uint64_t>();
            g.d_slot_off = st.slot_off.as<uint64_t>();
        } else {
            g.uniform_slot = gtsc_max_words(uniform_npoints);
            g.slot_total = g.uniform_slot * ns;
        }
        HIP_TRY(st.block_ts.reserve(ns * 8));
        HIP_TRY(st.ts.reserve(g.np * 8 + 8));
        HIP_TRY(st.val.reserve(g.np * 8 + 8));
        HIP_TRY(st.slots.reserve(g.slot_total * 8));
        HIP_TRY(st.compact.reserve(g.slot_total * 8));
        HIP_TRY(st.nwords.reserve(ns * 8));
        if (out_nbits) HIP_TRY(st.nbits.reserve(ns * 8));
        HIP_TRY(st.coff.reserve((ns + 1) * 8));
        HIP_TRY(hipMemcpyAsync(st.block_ts.p, block_ts + g.s0, ns * 8, hipMemcpyHostToDevice, c->copy_in));
        if (g.np) {
            HIP_TRY(hipMemcpyAsync(st.ts.p, ts + g.p0, g.np * 8, hipMemcpyHostToDevice, c->copy_in));
            HIP_TRY(hipMemcpyAsync(st.val.p, val + g.p0, g.np * 8, hipMemcpyHostToDevice, c->copy_in));
        }
        HIP_TRY(hipEventRecord(st.copied_in, c->copy_in));
        return GTSC_OK;
    };

    HIP_TRY(c->scan_scratch.reserve(scan_scratch_words(CHUNK_SERIES) * sizeof(uint64_t)));
    uint64_t base = 0;      // words produced by the chunks completed so far
    bool overflow = false;
    for (size_t k = 0; k < chunks.size() && k < 2; k++)
        if (int rc = copy_in_chunk(k)) return rc;
    for (size_t k = 0; k < chunks.size(); k++) {
        if (k + 2 < chunks.size())
            if (int rc = copy_in_chunk(k + 2)) return rc;
        const int b = (int)(k % NSTAGE);
        auto& st = c->stage[b];
        const Geometry& g = geo[b];
        HIP_TRY(hipStreamWaitEvent(c->work, st.copied_in, 0));
        EncodeArgs a{g.ns, st.block_ts.as<int64_t>(), st.ts.as<int64_t>(), st.val.as<uint64_t>(),
                     Layout{g.d_pt_off, uniform_npoints}, st.slots.as<uint64_t>(), Layout{g.d_slot_off, g.uniform_slot},
                     st.nwords.as<uint64_t>(), nullptr, c->d_flags_host, out_nbits ? st.nbits.as<uint64_t>() : nullptr,
                     (options & GTSC_ENC_NO_CLOSE) ? 1u : 0u, nullptr, c->dfcm_mask, 0u, ~0ull, 0u, c->tun, nullptr};
        HIP_TRY(dfcm_prepare(c, 1 + b, g.ns, c->work, &a.dfcm_tables));
        HIP_TRY(launch_encode(a, format, c->work));
        HIP_TRY(launch_exclusive_scan(st.nwords.as<uint64_t>(), g.ns, st.coff.as<uint64_t>(), c->scan_scratch.as<uint64_t>(), c->work));
        HIP_TRY(launch_compact_copy(g.ns, st.slots.as<uint64_t>(), Layout{g.d_slot_off, g.uniform_slot}, st.nwords.as<uint64_t>(),
                                    st.compact.as<uint64_t>(), st.coff.as<uint64_t>(), c->work));
        if (out_nbits) HIP_TRY(hipMemcpyAsync(out_nbits + g.s0, st.nbits.p, g.ns * 8, hipMemcpyDeviceToHost, c->work));
        HIP_TRY(hipMemcpyAsync(out_off + g.s0 + 1, st.coff.as<uint64_t>() + 1, g.ns * 8, hipMemcpyDeviceToHost, c->work));
        HIP_TRY(hipEventRecord(st.coded, c->work));
        HIP_TRY(hipEventSynchronize(st.coded));
        // out_off[s0+1 .. s1] hold chunk-local prefix sums; make them global
        for (uint64_t i = 1; i <= g.ns; i++) out_off[g.s0 + i] += base;
        const uint64_t total = out_off[g.s1] - base;
        if (base + total > out_capacity_words) overflow = true;
        else if (total)
            HIP_TRY(hipMemcpyAsync(out_words + base, st.compact.p, total * sizeof(uint64_t), hipMemcpyDeviceToHost, c->work));
        HIP_TRY(hipEventRecord(st.drained, c->work));
        base += total;
    }
    HIP_TRY(hipStreamSynchronize(c->work));
    uint32_t flags = 0;
    if (int rc = read_flags(c, c->work, &flags)) return rc;
    if (overflow) return fail(GTSC_ERR_CAPACITY, "out_words holds %llu words, the batch needs %llu",
                              (unsigned long long)out_capacity_words, (unsigned long long)out_off[nseries]);
    if (flags) return fail(GTSC_ERR_HIP, "internal: worst-case slot overflowed (flags 0x%x)", flags);
    return GTSC_OK;
}

int gtsc_encode_host(gtsc_ctx* c, int format, uint64_t nseries, const int64_t* block_ts, const int64_t* ts,
                     const uint64_t* val, const uint64_t* pt_off, uint64_t uniform_npoints, uint64_t* out_words,
                     uint64_t out_capacity_words, uint64_t* out_off)
{
    return gtsc_encode_host_ex(c, format, 0, nseries, block_ts, ts, val, pt_off, uniform_npoints, out_words,
                               out_capacity_words, out_off, nullptr);
}

int gtsc_decode_host(gtsc_ctx* c, int format, uint64_t nseries, const uint64_t* words, const uint64_t* in_off,
                     int64_t* ts, uint64_t* val, const uint64_t* pt_off, uint64_t uniform_capacity,
                     uint64_t* out_npoints, uint32_t* status)
{
    if (int rc = check_ctx(c)) return rc;
    if (format != GTSC_FMT_V2 && format != GTSC_FMT_V1) return fail(GTSC_ERR_ARG, "unknown format %d", format);
    if (nseries == 0) return GTSC_OK;
    if (!words || !in_off || !out_npoints) return fail(GTSC_ERR_ARG, "words/in_off/out_npoints must not be NULL");
    if ((!ts || !val) && (pt_off || uniform_capacity)) return fail(GTSC_ERR_ARG, "ts/val must not be NULL");
    if (c->predictor != GTSC_PRED_LAST && format != GTSC_FMT_V2)
        return fail(GTSC_ERR_ARG, "only the 2.x format takes a predictor");
    DeviceGuard g(c->device);
    if (int rc = ensure_stage_streams(c)) return rc;
    PipelineDrain drain{c};

    const std::vector<Chunk> chunks = plan_chunks(nseries, pt_off, uniform_capacity, false);
    // Copies into pageable memory block this thread inside hipMemcpyAsync, behind everything their stream waits
    // for -- here the chunk's kernels on two other streams.  That wait has been seen never to return (about one
    // run of the test suite in three, always at the first copy out of a chunk whose workgroup-per-block kernel ran
    // on the second work stream), while a hipStreamSynchronize of the work streams in the same place always does:
    // with pageable output the work streams are synchronised before the copies are made.
    const bool blocking_out = !(page_locked(ts) && page_locked(val) && page_locked(out_npoints) && page_locked(status));
    bool used[NSTAGE] = {};
    for (size_t k = 0; k < chunks.size(); k++) {
        const int b = (int)(k % NSTAGE);
        auto& st = c->stage[b];
        if (used[b]) HIP_TRY(hipEventSynchronize(st.drained));  // the stage's previous chunk has left its buffers
        used[b] = true;
        if (injected_failure(k)) return fail(GTSC_ERR_HIP, "staging chunk %zu: injected failure", k);
        const uint64_t s0 = chunks[k].s0, s1 = chunks[k].s1, ns = s1 - s0;
        const uint64_t w0 = in_off[s0], nw = in_off[s1] - w0;
        const uint64_t p0 = pt_off ? pt_off[s0] : s0 * uniform_capacity;
        const uint64_t np = (pt_off ? pt_off[s1] : s1 * uniform_capacity) - p0;
        HIP_TRY(st.h_off_a.reserve((ns + 1) * 8));
        uint64_t* const reb_in = st.h_off_a.as<uint64_t>();
        for (uint64_t i = 0; i <= ns; i++) reb_in[i] = in_off[s0 + i] - w0;
        HIP_TRY(st.slot_off.reserve((ns + 1) * 8));
        HIP_TRY(hipMemcpyAsync(st.slot_off.p, reb_in, (ns + 1) * 8, hipMemcpyHostToDevice, c->copy_in));
        const uint64_t* d_pt_off = nullptr;
        if (pt_off) {
            HIP_TRY(st.h_off_b.reserve((ns + 1) * 8));
            uint64_t* const reb_pt = st.h_off_b.as<uint64_t>();
            for (uint64_t i = 0; i <= ns; i++) reb_pt[i] = pt_off[s0 + i] - p0;
            HIP_TRY(st.pt_off.reserve((ns + 1) * 8));
            HIP_TRY(hipMemcpyAsync(st.pt_off.p, reb_pt, (ns + 1) * 8, hipMemcpyHostToDevice, c->copy_in));
            d_pt_off = st.pt_off.as<uint64_t>();
        }
        HIP_TRY(st.compact.reserve(nw * 8 + 8));
        HIP_TRY(st.ts.reserve(np * 8 + 8));
        HIP_TRY(st.val.reserve(np * 8 + 8));
        HIP_TRY(st.npoints.reserve(ns * 8));
        HIP_TRY(st.status.reserve(ns * 4));
        if (nw) HIP_TRY(hipMemcpyAsync(st.compact.p, words + w0, nw * 8, hipMemcpyHostToDevice, c->copy_in));
        HIP_TRY(hipEventRecord(st.copied_in, c->copy_in));
        const hipStream_t work = c->work;
        HIP_TRY(hipStreamWaitEvent(work, st.copied_in, 0));
        DecodeArgs a{ns, st.compact.as<uint64_t>(), Layout{st.slot_off.as<uint64_t>(), 0}, nullptr, st.ts.as<int64_t>(),
                     st.val.as<uint64_t>(), Layout{d_pt_off, uniform_capacity}, st.npoints.as<uint64_t>(),
                     st.status.as<uint32_t>(), c->d_flags_host, nullptr, c->dfcm_mask, 0u, ~0ull, nullptr, c->d_plans + 1 + b, 0u, 0u,
                     c->tun, nullptr};
        // the host form knows the chunk's density: no need to let the device choose the ring
        if (a.tun.decode_ring == 0) a.tun.decode_ring = nw * 64 > np * DENSE_BITS_PER_POINT ? 16u : 8u;
        HIP_TRY(dfcm_prepare(c, 1 + b, ns, work, &a.dfcm_tables));
        const DecodeFork fork{c->work2, st.planned, st.joined};
        HIP_TRY(launch_decode(a, format, work, &fork));
        HIP_TRY(hipEventRecord(st.coded, work));
        if (blocking_out) {
            HIP_TRY(hipStreamSynchronize(c->work2));
            HIP_TRY(hipStreamSynchronize(work));
        }
        HIP_TRY(hipStreamWaitEvent(c->copy_out, st.coded, 0));
        if (np) {
            HIP_TRY(hipMemcpyAsync(ts + p0, st.ts.p, np * 8, hipMemcpyDeviceToHost, c->copy_out));
            HIP_TRY(hipMemcpyAsync(val + p0, st.val.p, np * 8, hipMemcpyDeviceToHost, c->copy_out));
        }
        HIP_TRY(hipMemcpyAsync(out_npoints + s0, st.npoints.p, ns * 8, hipMemcpyDeviceToHost, c->copy_out));
        if (status) HIP_TRY(hipMemcpyAsync(status + s0, st.status.p, ns * 4, hipMemcpyDeviceToHost, c->copy_out));
        HIP_TRY(hipEventRecord(st.drained, c->copy_out));
    }
    HIP_TRY(hipStreamSynchronize(c->work));
    HIP_TRY(hipStreamSynchronize(c->work2));
    HIP_TRY(hipStreamSynchronize(c->copy_out));
    uint32_t flags = 0;
    if (int rc = read_flags(c, c->work, &flags)) return rc;
    if (flags & (ST_TRUNCATED | ST_MALFORMED)) return fail(GTSC_ERR_STREAM, "some images were truncated or malformed (status bits 0x%x)", flags);
    if (flags & ST_OVERFLOW) return fail(GTSC_ERR_CAPACITY, "some series hold more points than their ts/val slot");
    return GTSC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------
// One batch over several contexts (normally one per GPU of the node).  Series are independent, so the batch is cut
// into contiguous runs of series with about equal numbers of points, one per context; each run goes through its
// context's host pipeline on a host thread of its own, and the results land where a single call would have put them.
// ------------------------------------------------------------------------
namespace {

// boundaries of n contiguous shards of [0, nseries) with about equal weight (points, or capacity)
std::vector<uint64_t> shard_bounds(int n, uint64_t nseries, const uint64_t* off, uint64_t uniform)
{
    std::vector<uint64_t> b(n + 1, nseries);
    b[0] = 0;
    const uint64_t total = off ? off[nseries] - off[0] : nseries * uniform;
    uint64_t s = 0;
    for (int k = 1; k < n; k++) {
        const uint64_t want = off ? off[0] + (total / (uint64_t)n) * (uint64_t)k : 0;
        if (off) { while (s < nseries && off[s] < want) s++; }
        else s = nseries * (uint64_t)k / (uint64_t)n;
        b[k] = s;
    }
    return b;
}

struct ShardResult { int rc = GTSC_OK; std::string err; };

// the worst return code of the shards becomes the call's, with its message
int merge_shard_results(const std::vector<ShardResult>& r)
{
    int rc = GTSC_OK;
    const ShardResult* worst = nullptr;
    for (const ShardResult& x : r)
        if (x.rc != GTSC_OK && (rc == GTSC_OK || x.rc == GTSC_ERR_HIP || x.rc == GTSC_ERR_ARG)) { rc = x.rc; worst = &x; }
    return worst ? fail(rc, "%s", worst->err.c_str()) : GTSC_OK;
}

}  // namespace

extern "C" {

int gtsc_encode_host_multi(int nctx, gtsc_ctx* const* ctxs, int format, uint32_t options, uint64_t nseries, const int64_t* block_ts,
                           const int64_t* ts, const uint64_t* val, const uint64_t* pt_off, uint64_t uniform_npoints,
                           uint64_t* out_words, uint64_t out_capacity_words, uint64_t* out_off, uint64_t* out_nbits)
{
    if (nctx <= 0 || !ctxs) return fail(GTSC_ERR_ARG, "no contexts");
    for (int k = 0; k < nctx; k++) if (!ctxs[k]) return fail(GTSC_ERR_ARG, "context %d is NULL", k);
    if (!out_off) return fail(GTSC_ERR_ARG, "out_off must not be NULL");
    if (nctx == 1 || nseries < (uint64_t)nctx)
        return gtsc_encode_host_ex(ctxs[0], format, options, nseries, block_ts, ts, val, pt_off, uniform_npoints, out_words,
                                   out_capacity_words, out_off, out_nbits);
    if (!block_ts || !out_words) return fail(GTSC_ERR_ARG, "block_ts/out_words must not be NULL");
    const std::vector<uint64_t> b = shard_bounds(nctx, nseries, pt_off, uniform_npoints);
    // Where each shard's images go: shard k writes back to back from the place where it would start if every
    // earlier image had its worst-case size; afterwards the shards are moved together.  (A caller whose buffer is
    // smaller than the worst case of the whole batch gets private buffers per shard instead.)
    std::vector<uint64_t> worst(nctx + 1, 0);
    for (int k = 0; k < nctx; k++) {
        uint64_t w = 0;
        for (uint64_t s = b[k]; s < b[k + 1]; s++) w += gtsc_max_words(pt_off ? pt_off[s + 1] - pt_off[s] : uniform_npoints);
        worst[k + 1] = worst[k] + w;
    }
    const bool in_place = worst[nctx] <= out_capacity_words;
    std::vector<std::vector<uint64_t>> priv(in_place ? 0 : nctx);
    std::vector<std::vector<uint64_t>> offs(nctx);
    std::vector<ShardResult> res(nctx);
    std::vector<std::thread> threads;
    for (int k = 0; k < nctx; k++) {
        const uint64_t s0 = b[k], ns = b[k + 1] - b[k];
        offs[k].assign(ns + 1, 0);
        uint64_t* dst = out_words + worst[k];
        uint64_t cap = worst[k + 1] - worst[k];
        if (!in_place) { priv[k].resize(cap ? cap : 1); dst = priv[k].data(); }
        threads.emplace_back([=, &offs, &res] {
            // (offset arrays are absolute: the shard's slice of pt_off still indexes the whole ts/val arrays)
            const int rc = gtsc_encode_host_ex(ctxs[k], format, options, ns, block_ts + s0, pt_off ? ts : ts + s0 * uniform_npoints,
                                               pt_off ? val : val + s0 * uniform_npoints, pt_off ? pt_off + s0 : nullptr, uniform_npoints,
                                               dst, cap, offs[k].data(), out_nbits ? out_nbits + s0 : nullptr);
            res[k].rc = rc;
            if (rc != GTSC_OK) res[k].err = gtsc_last_error();
        });
    }
    for (std::thread& t : threads) t.join();
    if (int rc = merge_shard_results(res)) return rc;
    uint64_t base = 0;
    out_off[0] = 0;
    bool overflow = false;
    for (int k = 0; k < nctx; k++) {
        const uint64_t s0 = b[k], ns = b[k + 1] - b[k], total = offs[k][ns];
        if (base + total > out_capacity_words) overflow = true;
        else if (total) std::memmove(out_words + base, in_place ? out_words + worst[k] : priv[k].data(), total * sizeof(uint64_t));
        for (uint64_t i = 1; i <= ns; i++) out_off[s0 + i] = base + offs[k][i];
        base += total;
    }
    if (overflow) return fail(GTSC_ERR_CAPACITY, "out_words holds %llu words, the batch needs %llu",
                              (unsigned long long)out_capacity_words, (unsigned long long)base);
    return GTSC_OK;
}

int gtsc_decode_host_multi(int nctx, gtsc_ctx* const* ctxs, int format, uint64_t nseries, const uint64_t* words, const uint64_t* in_off,
                           int64_t* ts, uint64_t* val, const uint64_t* pt_off, uint64_t uniform_capacity, uint64_t* out_npoints,
                           uint32_t* status)
{
    if (nctx <= 0 || !ctxs) return fail(GTSC_ERR_ARG, "no contexts");
    for (int k = 0; k < nctx; k++) if (!ctxs[k]) return fail(GTSC_ERR_ARG, "context %d is NULL", k);
    if (nctx == 1 || nseries < (uint64_t)nctx)
        return gtsc_decode_host(ctxs[0], format, nseries, words, in_off, ts, val, pt_off, uniform_capacity, out_npoints, status);
    if (!words || !in_off || !out_npoints) return fail(GTSC_ERR_ARG, "words/in_off/out_npoints must not be NULL");
    const std::vector<uint64_t> b = shard_bounds(nctx, nseries, pt_off, uniform_capacity);
    std::vector<ShardResult> res(nctx);
    std::vector<std::thread> threads;
    for (int k = 0; k < nctx; k++) {
        const uint64_t s0 = b[k], ns = b[k + 1] - b[k];
        threads.emplace_back([=, &res] {
            const int rc = gtsc_decode_host(ctxs[k], format, ns, words, in_off + s0, pt_off ? ts : ts + s0 * uniform_capacity,
                                            pt_off ? val : val + s0 * uniform_capacity, pt_off ? pt_off + s0 : nullptr, uniform_capacity,
                                            out_npoints + s0, status ? status + s0 : nullptr);
            res[k].rc = rc;
            if (rc != GTSC_OK) res[k].err = gtsc_last_error();
        });
    }
    for (std::thread& t : threads) t.join();
    return merge_shard_results(res);
}

}  // extern "C"
