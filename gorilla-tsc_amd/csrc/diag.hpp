// Measurement hooks of the kernels.  In the shipped library every macro here is empty or the plain operation and
// every DIAG_* constant is false; a diagnostic build (never shipped: some of them give wrong results on purpose)
// defines one of
//   -DGTSC_STAMP        per-wave cycle stamps (s_memtime) left in the outputs, read by tools/stamps.py
//   -DGTSC_ABLATE=n     leaves one cost out to see what it is worth (1/3: encoder arithmetic, 4: decoder
//                       stores, 5: decoder tile writes, 6: decoder ring reads, 7: both, 8: the encoder's stores
//                       of compressed words before the last copy); results are wrong
//   -DGTSC_LDS_PAD=n    pads the encoder's LDS to lower its occupancy (-DGTSC_DEC_LDS_PAD=n: the decoder's)
//   -DGTSC_POISON=x     the decoder's input rings start out filled with x (nothing may depend on slots never written)
//   -DGTSC_RUNSTAT      one wave prints how many chunks it decoded by run skipping, in how many steps, and why not
//   -DGTSC_PLAIN_STORES the decoder's full-chunk stores as plain instead of streaming stores
//   -DGTSC_WG_STAMP     the workgroup-per-block decompressor prints where the cycles of waves 0 and 1 of a block went
//   -DGTSC_LAND_LOW=n   unread ring words at which the 16-word form lands a top-up between its regular checks
#pragma once

#ifndef GTSC_ABLATE
#define GTSC_ABLATE 0
#endif
#ifndef GTSC_LAND_LOW
#define GTSC_LAND_LOW 5u
#endif
#ifndef GTSC_DEC_LDS_PAD
#define GTSC_DEC_LDS_PAD 0
#endif
#ifndef GTSC_LDS_PAD
#define GTSC_LDS_PAD 0
#endif

namespace gtsc {
constexpr bool DIAG_NO_ENC_ARITH = GTSC_ABLATE == 1 || GTSC_ABLATE == 3;
constexpr bool DIAG_NO_ENC_STORES = GTSC_ABLATE == 8;
constexpr bool DIAG_NO_TILE_WRITES = GTSC_ABLATE == 5 || GTSC_ABLATE == 7;
constexpr bool DIAG_NO_RING_READS = GTSC_ABLATE == 6 || GTSC_ABLATE == 7;
#ifdef GTSC_STAMP
constexpr bool DIAG_STAMPED = true;
#else
constexpr bool DIAG_STAMPED = false;
#endif
}  // namespace gtsc

// ---- the decoder's stores of whole chunks: streaming stores (whole aligned lines that nothing reads again soon)
#if GTSC_ABLATE == 4
#define GTSC_CHUNK_STORE(T_, P_, X_) asm volatile("" ::"v"(X_), "v"(P_));
#elif defined(GTSC_PLAIN_STORES)
#define GTSC_CHUNK_STORE(T_, P_, X_) *reinterpret_cast<T_*>(P_) = (X_);
#else
#define GTSC_CHUNK_STORE(T_, P_, X_) __builtin_nontemporal_store((X_), reinterpret_cast<T_*>(P_));
#endif

// ---- the decoder's writes of decoded points to its LDS tiles
#define GTSC_TILE_WRITE(T_, P_, X_)                                  \
    {                                                                \
        if (DIAG_NO_TILE_WRITES) { asm volatile("" ::"v"(X_)); }     \
        else { *reinterpret_cast<T_*>(P_) = (T_)(X_); }              \
    }

#ifdef GTSC_POISON
#define GTSC_DIAG_POISON_RING(RING_, WORDS_) for (uint32_t i_ = 0; i_ < (WORDS_); i_++) (RING_)[i_] = GTSC_POISON;
#else
#define GTSC_DIAG_POISON_RING(RING_, WORDS_)
#endif

// ---- cycle stamps: where a wave's cycles go (results land in the tail of the wave's first slot / in its counts)
#ifdef GTSC_STAMP
#define GTSC_STAMP_NOW(VAR_) { __builtin_amdgcn_sched_barrier(0); VAR_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define GTSC_TICK(acc_) { __builtin_amdgcn_sched_barrier(0); const uint64_t now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - tk_prev; tk_prev = now_; __builtin_amdgcn_sched_barrier(0); }
#define GTSC_ENC_STAMPS_BEGIN uint64_t tk_stage = 0, tk_fetch = 0, tk_comp = 0, tk_prev = __builtin_amdgcn_s_memtime();
#define GTSC_ENC_STAMPS_END(A_, S_, LANE_)                                                   \
    {                                                                                        \
        uint64_t tk_flush = 0;                                                               \
        GTSC_TICK(tk_flush)                                                                  \
        if ((LANE_) == 0) {                                                                  \
            uint64_t* dbg = (A_).out_words + (A_).slots.begin(S_) + (A_).slots.size(S_) - 4; \
            dbg[0] = tk_stage; dbg[1] = tk_fetch; dbg[2] = tk_comp; dbg[3] = tk_flush;       \
        }                                                                                    \
    }
#define GTSC_DEC_STAMPS_ENTRY const uint64_t tk_entry = __builtin_amdgcn_s_memtime(); uint64_t tk_p1 = 0, tk_p2 = 0, tk_pk = 0;
#define GTSC_DEC_STAMP_PROBE(VALID_) { uint32_t probe_ = (VALID_) ? 1u : 0u; asm volatile("" : "+v"(probe_)); GTSC_STAMP_NOW(tk_pk) }
#define GTSC_DEC_STAMP_P1 GTSC_STAMP_NOW(tk_p1)
#define GTSC_DEC_STAMP_P2 GTSC_STAMP_NOW(tk_p2)
#define GTSC_DEC_STAMPS_BEGIN uint64_t tk_comp = 0, tk_flush = 0, tk_land = 0, tk_prev = __builtin_amdgcn_s_memtime(); const uint64_t tk_pro = tk_prev - tk_entry;
#define GTSC_LAND_STAMP() { __builtin_amdgcn_sched_barrier(0); const uint64_t t0_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tk_land += __builtin_amdgcn_s_memtime() - t0_; __builtin_amdgcn_sched_barrier(0); }
#define GTSC_DEC_STAMPS_END(A_, S_, LANE_)                                                   \
    {                                                                                        \
        wave_sync(); /* the stamps replace the counts of the series after the wave's first */ \
        if ((LANE_) == 0) {                                                                  \
            (A_).out_npoints[(S_) + 1] = tk_comp; (A_).out_npoints[(S_) + 2] = tk_flush; (A_).out_npoints[(S_) + 3] = tk_pro; \
            (A_).out_npoints[(S_) + 4] = tk_p1 - tk_entry; (A_).out_npoints[(S_) + 5] = tk_p2 - tk_p1;                       \
            (A_).out_npoints[(S_) + 6] = tk_pk - tk_entry; (A_).out_npoints[(S_) + 7] = tk_land;                             \
        }                                                                                    \
    }
#else
#define GTSC_TICK(acc_)
#define GTSC_ENC_STAMPS_BEGIN
#define GTSC_ENC_STAMPS_END(A_, S_, LANE_)
#define GTSC_DEC_STAMPS_ENTRY
#define GTSC_DEC_STAMP_PROBE(VALID_)
#define GTSC_DEC_STAMP_P1
#define GTSC_DEC_STAMP_P2
#define GTSC_DEC_STAMPS_BEGIN
#define GTSC_LAND_STAMP()
#define GTSC_DEC_STAMPS_END(A_, S_, LANE_)
#endif

// ---- run-skipping statistics of one wave of the decoder
#ifdef GTSC_RUNSTAT
#define GTSC_RUNSTAT_DECL uint32_t rs_try = 0, rs_ok = 0, rs_steps = 0, rs_chunks = 0; uint64_t rs_why = 0;
#define GTSC_RUNSTAT_CHUNK(WHY_) { rs_chunks++; rs_why |= (WHY_); }
#define GTSC_RUNSTAT_ATTEMPT(STEPS_, OK_) { rs_try++; rs_steps += (STEPS_); if (OK_) rs_ok++; }
#define GTSC_RUNSTAT_PRINT(LANE_) if (blockIdx.x == 1000 && (LANE_) == 0) printf("runstat: chunks %u tried %u ok %u steps %u why %llx\n", rs_chunks, rs_try, rs_ok, rs_steps, (unsigned long long)rs_why);
#else
#define GTSC_RUNSTAT_DECL
#define GTSC_RUNSTAT_CHUNK(WHY_)
#define GTSC_RUNSTAT_ATTEMPT(STEPS_, OK_)
#define GTSC_RUNSTAT_PRINT(LANE_)
#endif

// ---- cycle stamps of the workgroup-per-block decompressor (phases x1 x2 | barrier | y | barrier | z of a tick)
#ifdef GTSC_WG_STAMP
#define GTSC_WG_STAMP_DECL uint64_t tk_other = 0, tk_x1 = 0, tk_x2 = 0, tk_b1 = 0, tk_y = 0, tk_b2 = 0, tk_z = 0, tk_prev = __builtin_amdgcn_s_memtime(); uint32_t tk_ticks = 0;
#define GTSC_WG_TICK(acc_) { __builtin_amdgcn_sched_barrier(0); const uint64_t now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - tk_prev; tk_prev = now_; __builtin_amdgcn_sched_barrier(0); }
#define GTSC_WG_WALK_DECL uint64_t tw_rows = 0, tw_stride = 0, tw_queue = 0, tw_careful = 0, tw_chase = 0, tw_enum = 0, tw_prev = 0; uint32_t tw_calls = 0, tw_strides = 0, tw_handover = 0;
#define GTSC_WG_WALK_BEGIN(W_) { __builtin_amdgcn_sched_barrier(0); (W_).tw_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define GTSC_WG_WALK_TICK(W_, acc_) { __builtin_amdgcn_sched_barrier(0); const uint64_t now_ = __builtin_amdgcn_s_memtime(); (W_).acc_ += now_ - (W_).tw_prev; (W_).tw_prev = now_; __builtin_amdgcn_sched_barrier(0); }
#define GTSC_WG_WALK_COUNT(W_, cnt_) (W_).cnt_++;
#define GTSC_WG_WALK_PRINT(W_) if (lane == 0 && wave == 0) printf("walker: chase %llu enum %llu rows %llu (%u calls) strides %llu (%u) queue %llu careful %llu (%u handovers)\n", (unsigned long long)(W_).tw_chase, (unsigned long long)(W_).tw_enum, (unsigned long long)(W_).tw_rows, (W_).tw_calls, (unsigned long long)(W_).tw_stride, (W_).tw_strides, (unsigned long long)(W_).tw_queue, (unsigned long long)(W_).tw_careful, (W_).tw_handover);
#define GTSC_WG_RUN_PHASE_PRINT(by_runs_, n_, pos_, win_, serial_) if (tid == 0) printf("run phase: %u points by runs, now at point %u, bit %llu, window %u, serial %d\n", (by_runs_), (n_), (unsigned long long)(pos_), (win_) & 127u, (int)(serial_));
#define GTSC_WG_STAMP_PRINT if (lane == 0 && wave < 2) printf("wave %u: x1 %llu x2 %llu b1 %llu y %llu b2 %llu z %llu other %llu\n", wave, (unsigned long long)tk_x1, (unsigned long long)tk_x2, (unsigned long long)tk_b1, (unsigned long long)tk_y, (unsigned long long)tk_b2, (unsigned long long)tk_z, (unsigned long long)tk_other);
#else
#define GTSC_WG_STAMP_DECL
#define GTSC_WG_TICK(acc_)
#define GTSC_WG_WALK_DECL
#define GTSC_WG_WALK_BEGIN(W_)
#define GTSC_WG_WALK_TICK(W_, acc_)
#define GTSC_WG_WALK_COUNT(W_, cnt_)
#define GTSC_WG_WALK_PRINT(W_)
#define GTSC_WG_RUN_PHASE_PRINT(by_runs_, n_, pos_, win_, serial_)
#define GTSC_WG_STAMP_PRINT
#endif
