// Microbenchmark (diagnostic, not product): v_mfma_f32_16x16x4_f32 issue/latency on gfx950 with one wave per SIMD --
// dependent chain, 5 independent chains, and a dependent chain with NV independent VALU ops between MFMAs.
// Prints s_memtime ticks per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP 32
template <int MODE, int NV>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, float seed) {
    f32x4 acc[5];
    float a[8], v[16];
    for (int i = 0; i < 5; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 16; ++i) v[i] = seed * i + threadIdx.x;
    const float b = seed * 0.25f, m = seed * 0.5f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            if (MODE == 0) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b, acc[0], 0, 0, 0);
            if (MODE == 1) acc[i % 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b, acc[i % 5], 0, 0, 0);
            if (MODE == 2) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b, acc[0], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = fmaf(v[j], m, 1.0f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
    for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
template <int MODE, int NV>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    const int blocks = 256, threads = 256;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    k<MODE, NV><<<blocks, threads>>>(out, cyc, 1.0f); hipDeviceSynchronize();
    k<MODE, NV><<<blocks, threads>>>(out, cyc, 1.0f); hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 1024; ++i) avg += (double)h[i]; avg /= 1024;
    printf("%-40s ticks per MFMA = %.2f\n", name, avg / (REP * 20));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0>("dependent chain");
    run<1, 0>("5 independent chains");
    run<2, 2>("dependent chain + 2 VALU per MFMA");
    run<2, 4>("dependent chain + 4 VALU per MFMA");
    run<2, 6>("dependent chain + 6 VALU per MFMA");
    run<2, 8>("dependent chain + 8 VALU per MFMA");
    run<2, 12>("dependent chain + 12 VALU per MFMA");
    run<2, 16>("dependent chain + 16 VALU per MFMA");
    return 0;
}
