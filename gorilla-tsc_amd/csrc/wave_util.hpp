// Wave-level helpers shared by the kernels (device code, gfx950, 64-lane wavefronts).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gtsc {

// The tiled kernels run one wavefront per workgroup.  The LDS executes a wavefront's
// instructions in order, so lanes exchanging data through LDS need no hardware wait, only
// that the compiler keeps the accesses in program order.  (__syncthreads() would also wait
// for every outstanding global store, which costs a memory round trip per tile flush.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Data-parallel-primitive moves: lane l reads a register of another lane without going through
// LDS.  `old` is what a lane keeps when it has no source (row or wave edge, masked-out row).
// All 64 lanes must be active where these are used (a disabled source lane delivers nothing).
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_move(uint32_t old, uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, ROW_MASK, BANK_MASK, false);
}
// lane l gets x of lane l - 1; lane 0 gets `carry` (wave_shr:1)
__device__ __forceinline__ uint32_t lane_shift_up(uint32_t x, uint32_t carry) { return dpp_move<0x138>(carry, x); }
__device__ __forceinline__ uint64_t lane_shift_up(uint64_t x, uint64_t carry)
{
    const uint32_t lo = lane_shift_up((uint32_t)x, (uint32_t)carry);
    const uint32_t hi = lane_shift_up((uint32_t)(x >> 32), (uint32_t)(carry >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// inclusive prefix sum over the 64 lanes: four steps inside each row of 16 (row_shr:1,2,4,8), then
// lane 15 of rows 0 and 2 added to rows 1 and 3 (row_bcast:15), then lane 31 to rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v += dpp_move<0x111>(0u, v);
    v += dpp_move<0x112>(0u, v);
    v += dpp_move<0x114>(0u, v);
    v += dpp_move<0x118>(0u, v);
    v += dpp_move<0x142, 0xA>(0u, v);
    v += dpp_move<0x143, 0xC>(0u, v);
    return v;
}
// x with lane `lane` (wave-uniform) replaced by the wave-uniform value v: v_writelane_b32 (this compiler has no
// builtin of that name; the declaration binds the intrinsic itself)
extern "C" __device__ int gtsc_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane");
__device__ __forceinline__ uint32_t write_lane(uint32_t x, uint32_t v, uint32_t lane) { return (uint32_t)gtsc_llvm_writelane((int)v, (int)lane, (int)x); }
__device__ __forceinline__ uint32_t read_lane(uint32_t x, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)lane); }
__device__ __forceinline__ uint64_t read_lane(uint64_t x, uint32_t lane)
{
    return ((uint64_t)read_lane((uint32_t)(x >> 32), lane) << 32) | read_lane((uint32_t)x, lane);
}

// inclusive prefix sum / xor of 64-bit values over the wave (all 64 lanes active; lanes without a
// value contribute zero)
template <bool XOR>
__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v)
{
#define GTSC_SCAN_STEP(CTRL, RM)                                                               \
    {                                                                                          \
        const uint32_t lo_ = dpp_move<CTRL, RM>(0u, (uint32_t)v);                              \
        const uint32_t hi_ = dpp_move<CTRL, RM>(0u, (uint32_t)(v >> 32));                      \
        const uint64_t o_ = ((uint64_t)hi_ << 32) | lo_;                                       \
        v = XOR ? (v ^ o_) : (v + o_);                                                         \
    }
    GTSC_SCAN_STEP(0x111, 0xF)
    GTSC_SCAN_STEP(0x112, 0xF)
    GTSC_SCAN_STEP(0x114, 0xF)
    GTSC_SCAN_STEP(0x118, 0xF)
    GTSC_SCAN_STEP(0x142, 0xA)
    GTSC_SCAN_STEP(0x143, 0xC)
#undef GTSC_SCAN_STEP
    return v;
}

}  // namespace gtsc
