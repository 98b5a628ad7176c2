// Microbenchmark (diagnostic, not product): issue cost of v_fma_f32 / v_pk_fma_f32 / DP fma / DPP add on gfx950
// with 1, 2, 4 waves per SIMD.  Prints cycles per wave-instruction measured with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
    float a[16]; f2 p[16]; double d[8];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = (f2){seed + i, seed - i}; }
    for (int i = 0; i < 8; ++i) d[i] = seed + i;
    const float m = seed * 0.5f; const f2 pm = {m, m + 1.f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < REP; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], m, 1.0f);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = __builtin_elementwise_fma(p[i], pm, pm);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = fma(d[i], 0.5, 1.0);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[(i + 8) & 15]), 0xB1, 0xF, 0xF, true));
        } else if (MODE == 4) {                 // v_permlane32_swap with its two wait states (8 independent pairs)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 8]));
        } else if (MODE == 5) {                 // the swap alone (operands not written by the previous instruction)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 8]));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 8]));
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("s_nop 1");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y; for (int i = 0; i < 8; ++i) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
void run(const char* name, int ninst, int threads) {
    float* out; unsigned long long* cyc;
    int blocks = 256;   // one block per CU (roughly); threads/256 = waves per SIMD
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
    k<MODE><<<blocks, threads>>>(out, cyc, 1.0f); hipDeviceSynchronize();
    k<MODE><<<blocks, threads>>>(out, cyc, 1.0f); hipDeviceSynchronize();
    unsigned long long h[16 * 256]; hipMemcpy(h, cyc, (size_t)blocks * (threads / 64) * 8, hipMemcpyDeviceToHost);
    double avg = 0; int nw = blocks * (threads / 64); for (int i = 0; i < nw; ++i) avg += (double)h[i]; avg /= nw;
    printf("%-14s waves/SIMD=%d  cycles per wave-instruction = %.2f  (per SIMD: %.2f cycles/instr)\n", name, threads / 256,
           avg / (REP * ninst), avg / (REP * ninst) / (threads / 256));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int t : {256, 512, 1024}) {
        run<0>("v_fma_f32", 16, t); run<1>("v_pk_fma_f32", 16, t); run<2>("v_fma_f64", 8, t); run<3>("v_add_f32_dpp", 16, t);
        run<4>("nop1+swap32", 8, t); run<5>("permlane32_swap", 8, t); run<6>("permlane16_swap", 8, t); run<7>("s_nop 1", 16, t);
    }
    return 0;
}
