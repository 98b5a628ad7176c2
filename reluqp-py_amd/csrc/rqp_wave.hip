// rqp_wave.hip -- ADMM hot loop for SMALL per-instance problems (n <= 32, m <= 64; BASELINE config 4 shape):
// ONE WAVEFRONT = ONE QP.  A 256-thread workgroup per instance is latency- and barrier-bound at this size (3072 MACs
// per iteration); here the whole solve of an instance runs inside one wave with every matrix in its registers, no
// barrier at all (LDS operations of a wave execute in order), and the CU keeps many instances in flight.
//
// Lane l plays two roles:
//   row role     r = l            (constraint row):  Ar[c] = A[r][c], the float64 row state z, lam, A x, and l, u, rho
//   column role  c = l & 31, h = l >> 5:  half h of column c:  Atc[j] = A[32 h + j][c],  Kc[j] = K[c][16 h + j],
//                                         Hc[j] = H[c][16 h + j];  x[c] (float64) and g[c] are kept by both halves
// so A is held twice (row-major for A dx, column-major for A' nu): 32 + 32 + 16 + 16 = 96 VGPRs of matrices.
// Vectors cross between the roles through 640 B of LDS read with wave-uniform-per-half addresses (broadcast reads,
// ds_read_b128); the two halves of a column meet with one cross-half shuffle.
// Same recurrence (oracle/reluqp_oracle.py: forward_refine), check logic and quirk dispositions as k_admm_generic
// (rqp_admm.hip; reference line citations there); float32 products, float64 state.
#include "rqp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int WNC = 32, WMC = 64;      // caps: n <= 32, m <= 64

__device__ __forceinline__ float wtmax(float a, float b) {            // torch.max / norm(inf): NaN propagates
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
__device__ __forceinline__ float wave_tmax(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = wtmax(v, __shfl_xor(v, off, 64));
    return v;
}

}   // namespace

__global__ void __launch_bounds__(64, 3) k_admm_wave(SolveArgs a) {      // 3 waves per SIMD: <= 168 VGPRs
    __shared__ __attribute__((aligned(16))) float nuL[WMC];            // nu (lam at a check) by row
    __shared__ __attribute__((aligned(16))) float xL[WNC];             // x by column
    __shared__ __attribute__((aligned(16))) float dL[WNC];             // d by column
    __shared__ __attribute__((aligned(16))) float dxL[WNC];            // dx by column
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = a.n, m = a.m, ldn = a.ldn, ldm = a.ldm;
    const int r = lane, c = lane & 31, h = lane >> 5;
    const bool rok = r < m, cok = c < n;
    const float* A = (const float*)a.A + (size_t)b * a.sA;
    const float* At = (const float*)a.At + (size_t)b * a.sAt;
    const float* Ht = (const float*)a.Ht + (size_t)b * a.sH;
    const float* Kb = (const float*)a.K + (size_t)b * a.sK;

    // ---- matrices into registers (leading dimensions are multiples of 4 floats, padding is zero)
    // (kept as float pairs: every product runs on v_pk_fma_f32 -- two accumulators, even / odd elements, added at the end)
    f32x2 Ar[WNC / 2], Atc[16], Hc[8], Kc[8];
#pragma unroll
    for (int q = 0; q < WNC / 4; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rok && 4 * q < ldn) v = *(const f32x4*)(A + (size_t)r * ldn + 4 * q);
        Ar[2 * q] = (f32x2){v[0], v[1]};
        Ar[2 * q + 1] = (f32x2){v[2], v[3]};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (cok && 32 * h + 4 * q < ldm) v = *(const f32x4*)(At + (size_t)c * ldm + 32 * h + 4 * q);
        Atc[2 * q] = (f32x2){v[0], v[1]};
        Atc[2 * q + 1] = (f32x2){v[2], v[3]};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (cok && 16 * h + 4 * q < ldn) v = *(const f32x4*)(Ht + (size_t)c * ldn + 16 * h + 4 * q);   // sym(H): row c = column c
        Hc[2 * q] = (f32x2){v[0], v[1]};
        Hc[2 * q + 1] = (f32x2){v[2], v[3]};
    }
    int ri = a.rho_ind[b];
    auto load_K = [&]() {
        const float* Kj = Kb + (size_t)ri * n * ldn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && 16 * h + 4 * q < ldn) v = *(const f32x4*)(Kj + (size_t)c * ldn + 16 * h + 4 * q);
            Kc[2 * q] = (f32x2){v[0], v[1]};
            Kc[2 * q + 1] = (f32x2){v[2], v[3]};
        }
    };
    load_K();

    // ---- vectors and state
    const float gc = cok ? ((const float*)a.g)[(size_t)b * n + c] : 0.f;
    const float lr = rok ? ((const float*)a.l)[(size_t)b * m + r] : 0.f;
    const float ur = rok ? ((const float*)a.u)[(size_t)b * m + r] : 0.f;
    const float cr = rok ? ((const float*)a.c)[(size_t)b * m + r] : 1.f;
    double x = cok ? a.x[(size_t)b * n + c] : 0.0;
    double z = rok ? a.z[(size_t)b * m + r] : 0.0;
    double lam = rok ? a.lam[(size_t)b * m + r] : 0.0;
    float rv = (float)a.rhos[ri] * cr;
    double inv = 1.0 / (double)rv;

    // products: column role reads by-row vectors (nuL) / by-column vectors (xL, dL); row role reads dxL
    auto at_times = [&](f32x2 acc) __attribute__((always_inline)) {     // acc + sum_j A[32 h + j][c] * nuL[32 h + j]  (pairwise)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *(const f32x4*)(nuL + 32 * h + 4 * q);
            acc = __builtin_elementwise_fma(Atc[2 * q], (f32x2){v[0], v[1]}, acc);
            acc = __builtin_elementwise_fma(Atc[2 * q + 1], (f32x2){v[2], v[3]}, acc);
        }
        return acc;
    };
    auto half16 = [&](const f32x2 (&M)[8], const float* vec, f32x2 acc) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4*)(vec + 16 * h + 4 * q);
            acc = __builtin_elementwise_fma(M[2 * q], (f32x2){v[0], v[1]}, acc);
            acc = __builtin_elementwise_fma(M[2 * q + 1], (f32x2){v[2], v[3]}, acc);
        }
        return acc;
    };
    auto a_times = [&]() __attribute__((always_inline)) {               // sum_c A[r][c] * dxL[c]
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < WNC / 4; ++q) {
            const f32x4 v = *(const f32x4*)(dxL + 4 * q);
            acc = __builtin_elementwise_fma(Ar[2 * q], (f32x2){v[0], v[1]}, acc);
            acc = __builtin_elementwise_fma(Ar[2 * q + 1], (f32x2){v[2], v[3]}, acc);
        }
        return acc[0] + acc[1];
    };
    const f32x2 zero2 = {0.f, 0.f};
    auto fold = [&](f32x2 v) __attribute__((always_inline)) { return v[0] + v[1]; };
    auto both_halves = [&](float v) __attribute__((always_inline)) { return v + __shfl_xor(v, 32, 64); };

    // (by-column LDS vectors are written by BOTH halves of a column -- same value, same address: no exec-masked region
    //  in the loop.  With the writes under `if (h == 0)` hipcc 7.2 moved the following full-wave ds_reads of the same
    //  array into the masked block, leaving lanes 32..63 with stale registers.)
    // A x of the incoming state
    dxL[c] = (float)x;
    xL[c] = (float)x;
    double zt = (double)a_times();

    bool converged = false;
    int iters = 0;
    float pri = 0.f, dua = 0.f, hx = 0.f;
    float rho_est = (float)a.rhos[ri];                                  // :211
    const float tolT = (float)a.tol;
    const int kmax = a.max_iter;

    // compute_residuals (:307-318) on the current state; leaves H x of the column in hx
    auto residuals = [&](float rho_carry, float& o_pri, float& o_dua) -> float {
        nuL[r] = (float)lam;
        const float t3 = both_halves(fold(at_times(zero2)));            // A' lam
        hx = both_halves(fold(half16(Hc, xL, zero2)));                  // H x
        const float v0 = wave_tmax(fabsf((float)(zt - z)));
        const float v1 = wave_tmax(fabsf((float)zt));
        const float v2 = wave_tmax(fabsf((float)z));
        const float v3 = wave_tmax(fabsf(hx + t3 + gc));
        const float v4 = wave_tmax(fabsf(hx));
        const float v5 = wave_tmax(fabsf(t3));
        const float v6 = wave_tmax(fabsf(gc));
        o_pri = v0;
        o_dua = v3;
        const float num = v0 / wtmax(v1, v2);                           // :315
        const float den = v3 / wtmax(wtmax(v4, v5), v6);                // :316
        float est = rho_carry * sqrtf(num / den);                       // :317
        if (est < (float)a.rho_min) est = (float)a.rho_min;             // torch.clamp: NaN stays NaN
        if (est > (float)a.rho_max) est = (float)a.rho_max;
        return est;
    };

    for (int k = 1; k <= kmax; ++k) {
        {                                                               // row role: lam_hat, nu
            const double p = zt - z;
            const double lh = lam + (double)rv * p;
            lam = lh;
            nuL[r] = (float)(lh + (double)rv * p);
        }
        {                                                               // column role: d = H x + g + A' nu ; dx = -K d
            const float d = both_halves(fold(half16(Hc, xL, at_times(zero2)))) + gc;
            dL[c] = d;
            const float dx = -both_halves(fold(half16(Kc, dL, zero2)));
            x += (double)dx;
            dxL[c] = dx;
            xL[c] = (float)x;
        }
        {                                                               // row role: A x, z
            zt += (double)a_times();
            const double v = zt + lam * inv;
            double zn = v;                                              // torch.clamp: NaN stays NaN
            if (v < (double)lr) zn = (double)lr;
            if (v > (double)ur) zn = (double)ur;
            z = zn;
        }
        iters = k;
        if ((k % a.check_interval) == 0) {                              // :218 (Q3 fixed: always check)
            const int ri_before = ri;
            rho_est = residuals(rho_est, pri, dua);                     // :220 (Q4: estimate is carried)
            if (rho_est > (float)a.rhos[ri] * tolT && ri < a.nrho - 1)            // :223
                ri += 1;
            else if (rho_est < (float)a.rhos[ri] / tolT && ri > 0)                // :226
                ri -= 1;
            if (a.info.trace && (k / a.check_interval) <= a.info.trace_cap && lane == 0) {
                double* tr = a.info.trace + ((size_t)b * a.info.trace_cap + (k / a.check_interval - 1)) * 4;
                tr[0] = (double)pri; tr[1] = (double)dua; tr[2] = (double)rho_est; tr[3] = (double)ri_before;
            }
            if (ri != ri_before) {                                      // "re-factor" = table lookup
                load_K();
                rv = (float)a.rhos[ri] * cr;
                inv = 1.0 / (double)rv;
            }
            if (pri < (float)a.thr_p && dua < (float)a.thr_d) {         // :233
                converged = true;
                break;
            }
        }
    }
    if (!converged) rho_est = residuals(rho_est, pri, dua);             // :243 (Q11 fixed: fresh state)

    // objective 1/2 x'Hx + g'x (compute_J :320-322): hx holds H x of the final state
    double jp = (h == 0 && cok) ? (double)((float)x * (0.5f * hx + gc)) : 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) jp += __shfl_xor(jp, off, 64);

    // ---- update_results (:278-305) and the persistent state
    if (a.out_x && h == 0 && cok) ((float*)a.out_x)[(size_t)b * n + c] = (float)x;
    if (a.out_z && rok) ((float*)a.out_z)[(size_t)b * m + r] = (float)z;
    if (a.out_lam && rok) ((float*)a.out_lam)[(size_t)b * m + r] = (float)lam;
    if (lane == 0) {
        if (a.info.iter) a.info.iter[b] = converged ? iters : a.max_iter;
        if (a.info.status) a.info.status[b] = converged ? RQP_STATUS_SOLVED : RQP_STATUS_MAX_ITER;
        if (a.info.rho_ind) a.info.rho_ind[b] = ri;
        if (a.info.pri_res) a.info.pri_res[b] = (double)pri;
        if (a.info.dua_res) a.info.dua_res[b] = (double)dua;
        if (a.info.rho_estimate) a.info.rho_estimate[b] = (double)rho_est;
        if (a.info.obj_val) a.info.obj_val[b] = jp;
        a.rho_ind[b] = a.warm_starting ? ri : a.rho_ind0;
    }
    if (h == 0 && cok) a.x[(size_t)b * n + c] = a.warm_starting ? x : 0.0;           // state persists (:304) or is cleared (:324-333)
    if (rok) {
        a.z[(size_t)b * m + r] = a.warm_starting ? z : 0.0;
        a.lam[(size_t)b * m + r] = a.warm_starting ? lam : 0.0;
    }
}

bool rqp_wave_fits(const rqp_handle* h) { return h->esz == 4 && h->n <= WNC && h->m <= WMC; }

hipError_t rqp_launch_solve_wave(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    k_admm_wave<<<h->B, 64, 0, s>>>(a);
    return hipGetLastError();
}
