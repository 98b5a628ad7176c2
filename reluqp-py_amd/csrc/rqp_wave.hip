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
// (rqp_admm.hip; reference line citations there); products in T (float: packed FMAs; double: the reference's default
// precision, same layout with twice the registers), float64 state.
#include "rqp_common.h"

namespace {

constexpr int WNC = 32, WMC = 64;      // caps: n <= 32, m <= 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// 16-byte chunks (global and LDS accesses) and element pairs (register storage, fma granularity)
template <typename T>
struct WV;
template <>
struct WV<float> {
    typedef f32x4 chunk;
    typedef f32x2 pair;
    static constexpr int W = 4, OCC = 3;      // elements per chunk; waves per SIMD (<= 168 VGPRs)
};
template <>
struct WV<double> {
    typedef f64x2 chunk;
    typedef f64x2 pair;
    static constexpr int W = 2, OCC = 1;      // matrices alone are 192 VGPRs
};

template <typename T>
__device__ __forceinline__ T wtmax(T a, T b) {                        // torch.max / norm(inf): NaN propagates
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
template <typename T>
__device__ __forceinline__ T wave_tmax(T v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = wtmax(v, (T)__shfl_xor(v, off, 64));
    return v;
}

}   // namespace

template <typename T>
__global__ void __launch_bounds__(64, WV<T>::OCC) k_admm_wave(SolveArgs a) {
    typedef typename WV<T>::chunk chunk;
    typedef typename WV<T>::pair pair;
    constexpr int W = WV<T>::W, PW = W / 2;                            // elements / pairs per 16-byte chunk
    __shared__ __attribute__((aligned(16))) T nuL[WMC];                // nu (lam at a check) by row
    __shared__ __attribute__((aligned(16))) T xL[WNC];                 // x by column
    __shared__ __attribute__((aligned(16))) T dL[WNC];                 // d by column
    __shared__ __attribute__((aligned(16))) T dxL[WNC];                // dx by column
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = a.n, m = a.m, ldn = a.ldn, ldm = a.ldm;
    const int r = lane, c = lane & 31, h = lane >> 5;
    const bool rok = r < m, cok = c < n;
    const T* A = (const T*)a.A + (size_t)b * a.sA;
    const T* At = (const T*)a.At + (size_t)b * a.sAt;
    const T* Ht = (const T*)a.Ht + (size_t)b * a.sH;
    const T* Kb = (const T*)a.K + (size_t)b * a.sK;

    // ---- matrices into registers (leading dimensions are multiples of 16 bytes, padding is zero); kept as element pairs:
    //      every product runs on pair FMAs (v_pk_fma_f32 for float) -- two accumulators, even / odd elements, added at the end
    pair Ar[WNC / 2], Atc[16], Hc[8], Kc[8];
    auto load_row = [&](pair* dst, int len, const T* src, bool ok, int ld_left) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < len / W; ++q) {
            chunk v;
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = (T)0;
            if (ok && W * q < ld_left) v = *(const chunk*)(src + W * q);
#pragma unroll
            for (int e = 0; e < PW; ++e) dst[q * PW + e] = (pair){v[2 * e], v[2 * e + 1]};
        }
    };
    load_row(Ar, WNC, A + (size_t)r * ldn, rok, ldn);                        // row r of A
    load_row(Atc, 32, At + (size_t)c * ldm + 32 * h, cok, ldm - 32 * h);     // rows 32 h .. of column c
    load_row(Hc, 16, Ht + (size_t)c * ldn + 16 * h, cok, ldn - 16 * h);      // sym(H): row c = column c
    int ri = a.rho_ind[b];
    auto load_K = [&]() __attribute__((always_inline)) {
        load_row(Kc, 16, Kb + ((size_t)ri * n + c) * ldn + 16 * h, cok, ldn - 16 * h);
    };
    load_K();

    // ---- vectors and state
    const T gc = cok ? ((const T*)a.g)[(size_t)b * n + c] : (T)0;
    const T lr = rok ? ((const T*)a.l)[(size_t)b * m + r] : (T)0;
    const T ur = rok ? ((const T*)a.u)[(size_t)b * m + r] : (T)0;
    const T cr = rok ? ((const T*)a.c)[(size_t)b * m + r] : (T)1;
    double x = cok ? a.x[(size_t)b * n + c] : 0.0;
    double z = rok ? a.z[(size_t)b * m + r] : 0.0;
    double lam = rok ? a.lam[(size_t)b * m + r] : 0.0;
    T rv = (T)a.rhos[ri] * cr;
    double inv = 1.0 / (double)rv;

    // products: column role reads by-row vectors (nuL) / by-column vectors (xL, dL); row role reads dxL
    auto dot = [&](const pair* M, int len, const T* vec, pair acc) __attribute__((always_inline)) {   // acc + sum M[j] vec[j], pairwise
#pragma unroll
        for (int q = 0; q < len / W; ++q) {
            const chunk v = *(const chunk*)(vec + W * q);
#pragma unroll
            for (int e = 0; e < PW; ++e) acc = __builtin_elementwise_fma(M[q * PW + e], (pair){v[2 * e], v[2 * e + 1]}, acc);
        }
        return acc;
    };
    const pair zero2 = {(T)0, (T)0};
    auto at_times = [&](pair acc) __attribute__((always_inline)) { return dot(Atc, 32, nuL + 32 * h, acc); };   // + A[32 h..][c]' nu[32 h..]
    auto h_times = [&](pair acc) __attribute__((always_inline)) { return dot(Hc, 16, xL + 16 * h, acc); };     // + H[c][16 h..] x[16 h..]
    auto k_times = [&]() __attribute__((always_inline)) { return dot(Kc, 16, dL + 16 * h, zero2); };           // K[c][16 h..] d[16 h..]
    auto a_times = [&]() __attribute__((always_inline)) {                                                      // A[r][:] dx
        const pair acc = dot(Ar, WNC, dxL, zero2);
        return acc[0] + acc[1];
    };
    auto fold = [&](pair v) __attribute__((always_inline)) { return v[0] + v[1]; };
    // lanes hand vectors to each other through LDS inside the wave: a wavefront-scope release + wave barrier orders the
    // hand-off for the compiler (the hardware already executes a wave's LDS operations in order)
    auto handoff = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto both_halves = [&](T v) __attribute__((always_inline)) { return v + (T)__shfl_xor(v, 32, 64); };

    // (by-column LDS vectors are written by BOTH halves of a column -- same value, same address: no exec-masked region
    //  in the loop.  With the writes under `if (h == 0)` hipcc 7.2 moved the following full-wave ds_reads of the same
    //  array into the masked block, leaving lanes 32..63 with stale registers.)
    // A x of the incoming state
    dxL[c] = (T)x;
    xL[c] = (T)x;
    handoff();
    double zt = (double)a_times();

    bool converged = false;
    int iters = 0;
    T pri = (T)0, dua = (T)0, hx = (T)0;
    T rho_est = (T)a.rhos[ri];                                          // :211
    const T tolT = (T)a.tol;
    const int kmax = a.max_iter;

    // compute_residuals (:307-318) on the current state; leaves H x of the column in hx
    auto residuals = [&](T rho_carry, T& o_pri, T& o_dua) -> T {
        nuL[r] = (T)lam;
        handoff();
        const T t3 = both_halves(fold(at_times(zero2)));                // A' lam
        hx = both_halves(fold(h_times(zero2)));                         // H x
        const T v0 = wave_tmax((T)fabs((T)(zt - z)));
        const T v1 = wave_tmax((T)fabs((T)zt));
        const T v2 = wave_tmax((T)fabs((T)z));
        const T v3 = wave_tmax((T)fabs(hx + t3 + gc));
        const T v4 = wave_tmax((T)fabs(hx));
        const T v5 = wave_tmax((T)fabs(t3));
        const T v6 = wave_tmax((T)fabs(gc));
        o_pri = v0;
        o_dua = v3;
        const T num = v0 / wtmax(v1, v2);                               // :315
        const T den = v3 / wtmax(wtmax(v4, v5), v6);                    // :316
        T est = rho_carry * (T)sqrt(num / den);                         // :317
        if (est < (T)a.rho_min) est = (T)a.rho_min;                     // torch.clamp: NaN stays NaN
        if (est > (T)a.rho_max) est = (T)a.rho_max;
        return est;
    };

    for (int k = 1; k <= kmax; ++k) {
        {                                                               // row role: lam_hat, nu
            const double p = zt - z;
            const double lh = lam + (double)rv * p;
            lam = lh;
            nuL[r] = (T)(lh + (double)rv * p);
        }
        handoff();
        {                                                               // column role: d = H x + g + A' nu ; dx = -K d
            const T d = both_halves(fold(h_times(at_times(zero2)))) + gc;
            dL[c] = d;
            handoff();
            const T dx = -both_halves(fold(k_times()));
            x += (double)dx;
            dxL[c] = dx;
            xL[c] = (T)x;
        }
        handoff();
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
            if (rho_est > (T)a.rhos[ri] * tolT && ri < a.nrho - 1)                // :223
                ri += 1;
            else if (rho_est < (T)a.rhos[ri] / tolT && ri > 0)                    // :226
                ri -= 1;
            if (a.info.trace && (k / a.check_interval) <= a.info.trace_cap && lane == 0) {
                double* tr = a.info.trace + ((size_t)b * a.info.trace_cap + (k / a.check_interval - 1)) * 4;
                tr[0] = (double)pri; tr[1] = (double)dua; tr[2] = (double)rho_est; tr[3] = (double)ri_before;
            }
            if (ri != ri_before) {                                      // "re-factor" = table lookup
                load_K();
                rv = (T)a.rhos[ri] * cr;
                inv = 1.0 / (double)rv;
            }
            if (pri < (T)a.thr_p && dua < (T)a.thr_d) {                 // :233
                converged = true;
                break;
            }
        }
    }
    if (!converged) rho_est = residuals(rho_est, pri, dua);             // :243 (Q11 fixed: fresh state)

    // objective 1/2 x'Hx + g'x (compute_J :320-322): hx holds H x of the final state
    double jp = (h == 0 && cok) ? (double)((T)x * ((T)0.5 * hx + gc)) : 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) jp += __shfl_xor(jp, off, 64);

    // ---- update_results (:278-305) and the persistent state
    if (a.out_x && h == 0 && cok) ((T*)a.out_x)[(size_t)b * n + c] = (T)x;
    if (a.out_z && rok) ((T*)a.out_z)[(size_t)b * m + r] = (T)z;
    if (a.out_lam && rok) ((T*)a.out_lam)[(size_t)b * m + r] = (T)lam;
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

bool rqp_wave_fits(const rqp_handle* h) { return h->n <= WNC && h->m <= WMC; }

hipError_t rqp_launch_solve_wave(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    if (h->esz == 4)
        k_admm_wave<float><<<h->B, 64, 0, s>>>(a);
    else
        k_admm_wave<double><<<h->B, 64, 0, s>>>(a);
    return hipGetLastError();
}
