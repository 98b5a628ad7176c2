// rqp_lanes.h -- wave-wide reductions in registers for the convergence checks (no LDS, no ds_bpermute): DPP inside a row of
// 16 lanes, v_permlane16_swap / v_permlane32_swap across the 4 rows.  The swaps are inline asm with the 2 wait states a
// VALU-written operand needs (hipcc 7.2 also miscompiles the builtin's two-result form, see rqp_resident2.hip).
// NaN handling of torch.max / norm(inf) -- NaN propagates -- is kept out of the value path: callers reduce a bit mask of
// NaN flags with lanes_or() and the values with v_max (which skips NaN), then put NaN back where the mask says so.
#pragma once
#include <hip/hip_runtime.h>

namespace {

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_max64(double v) {                  // max(v, v of the DPP partner); NaN-free inputs
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)dpp_i<CTRL>((int)(unsigned)u), hi = (unsigned)dpp_i<CTRL>((int)(unsigned)(u >> 32));
    return fmax(v, __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo));
}
template <int CTRL>
__device__ __forceinline__ float dpp_max32(float v) {
    return fmaxf(v, __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))));
}
__device__ __forceinline__ void swap16(int& a, int& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(int& a, int& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ double rows_max64(double v) {                 // max over the 4 rows of 16 lanes, same lane of each row
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    int alo = (int)(unsigned)u, blo = alo, ahi = (int)(unsigned)(u >> 32), bhi = ahi;
    swap16(alo, blo);
    swap16(ahi, bhi);
    double x = fmax(__builtin_bit_cast(double, ((unsigned long long)(unsigned)ahi << 32) | (unsigned)alo),
                    __builtin_bit_cast(double, ((unsigned long long)(unsigned)bhi << 32) | (unsigned)blo));
    const unsigned long long w = __builtin_bit_cast(unsigned long long, x);
    alo = (int)(unsigned)w; blo = alo; ahi = (int)(unsigned)(w >> 32); bhi = ahi;
    swap32(alo, blo);
    swap32(ahi, bhi);
    return fmax(__builtin_bit_cast(double, ((unsigned long long)(unsigned)ahi << 32) | (unsigned)alo),
                __builtin_bit_cast(double, ((unsigned long long)(unsigned)bhi << 32) | (unsigned)blo));
}
__device__ __forceinline__ float rows_max32(float v) {
    int a = __builtin_bit_cast(int, v), b = a;
    swap16(a, b);
    v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    a = __builtin_bit_cast(int, v); b = a;
    swap32(a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ int rows_or(int v) {
    int a = v, b = v;
    swap16(a, b);
    v = a | b;
    a = v; b = v;
    swap32(a, b);
    return a | b;
}
__device__ __forceinline__ double wave_max64(double v) {                 // every lane gets the max over the 64 lanes
    v = dpp_max64<0xB1>(v);                                            // quad_perm [1,0,3,2]
    v = dpp_max64<0x4E>(v);                                            // quad_perm [2,3,0,1]
    v = dpp_max64<0x141>(v);                                           // row_half_mirror
    v = dpp_max64<0x140>(v);                                           // row_mirror
    return rows_max64(v);
}
__device__ __forceinline__ float wave_max32(float v) {
    v = dpp_max32<0xB1>(v);
    v = dpp_max32<0x4E>(v);
    v = dpp_max32<0x141>(v);
    v = dpp_max32<0x140>(v);
    return rows_max32(v);
}
__device__ __forceinline__ double lanes_max(double v) { return wave_max64(v); }
__device__ __forceinline__ float lanes_max(float v) { return wave_max32(v); }
// v + (v of lane ^ 32), every lane: one v_permlane32_swap instead of a ds_bpermute round trip
__device__ __forceinline__ float xor32_sum(float v) {
    int a = __builtin_bit_cast(int, v), b = a;
    swap32(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ double xor32_sum(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    int alo = (int)(unsigned)u, blo = alo, ahi = (int)(unsigned)(u >> 32), bhi = ahi;
    swap32(alo, blo);
    swap32(ahi, bhi);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)ahi << 32) | (unsigned)alo) +
           __builtin_bit_cast(double, ((unsigned long long)(unsigned)bhi << 32) | (unsigned)blo);
}
__device__ __forceinline__ int wave_or_i(int v) {
    v |= dpp_i<0xB1>(v);
    v |= dpp_i<0x4E>(v);
    v |= dpp_i<0x141>(v);
    v |= dpp_i<0x140>(v);
    return rows_or(v);
}

}   // namespace
