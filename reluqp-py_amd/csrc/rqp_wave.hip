// rqp_wave.hip -- ADMM hot loop for SMALL per-instance problems: ONE WAVEFRONT = ONE QP.
//   k_admm_wave<T, 32, 64>       n <= 32, m <= 64    (BASELINE config 4 shape), float and double
//   k_admm_wave<float, 32, 128>  n <= 32, m <= 128   (two rows per lane: short-horizon condensed MPC, m = N (nx + nu) > 64)
//   k_admm_wave<float, 32, 64, 2>  n <= 64, m <= 128  (TWO wavefronts per QP: each owns 32 columns and 64 rows; the three
//                                  vector hand-offs of an iteration become workgroup barriers of a 128-thread workgroup)
// A 256-thread workgroup per instance is latency- and barrier-bound at these sizes (3072 MACs per iteration at n=32,
// m=64); here the whole solve of an instance runs inside one wave with every matrix in its registers, no barrier at all
// (LDS operations of a wave execute in order), and the CU keeps several instances in flight.
//
// Lane l of wave w plays two roles (NC / MC = columns / rows per wave, NWV waves per QP: NCT = NC NWV columns and
// MCT = MC NWV rows in all; HALF = 64 / NC lanes per column, RL = MC / 64 rows per lane):
//   row role     rows r_q = MC w + l + 64 q, q < RL:  Ar[q][:] = A[r_q][:], the float64 row state z, lam, A x, and l, u, rho
//   column role  c = NC w + l % NC, h = l / NC:  part h of column c:  Atc[j] = A[(MCT/HALF) h + j][c],
//                Kc[j] = K[c][(NCT/HALF) h + j],  Hc[j] = H[c][(NCT/HALF) h + j];  x[c] (float64) and g[c] on every part
// so A is held twice (row-major for A dx, column-major for A' nu).  Vectors cross between the roles through LDS read with
// wave-uniform-per-part addresses (broadcast ds_read_b128); with HALF = 2 the two parts of a column meet with one
// cross-half shuffle.
// Same recurrence (oracle/reluqp_oracle.py: forward_refine), check logic and quirk dispositions as k_admm_generic
// (rqp_admm.hip; reference line citations there); products in T (float: packed FMAs; double: the reference's default
// precision, same layout with twice the registers), float64 state.
#include "rqp_common.h"
#include "rqp_lanes.h"

namespace {

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
    static constexpr int W = 4;               // elements per chunk
};
template <>
struct WV<double> {
    typedef f64x2 chunk;
    typedef f64x2 pair;
    static constexpr int W = 2;
};

// waves per SIMD the register budget allows (matrix registers: A by row + A by column + K + H)
template <typename T, int NC, int MC, int NWV>
struct WOcc {
    static constexpr int regs = (int)(sizeof(T) / 4) * (NC * NWV * (MC / 64) + MC * NWV / (64 / NC) + 2 * NC * NWV / (64 / NC));
    // waves per SIMD (VALU operands must sit in the 256 arch VGPRs); __launch_bounds__ counts workgroups of NWV waves
    static constexpr int waves = regs <= 96 ? 3 : (regs <= 160 ? 2 : 1);
    static constexpr int value = waves;
};

template <typename T>
__device__ __forceinline__ T wtmax(T a, T b) {                        // torch.max / norm(inf): NaN propagates
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}

}   // namespace

template <typename T, int NC, int MC, int NWV = 1>
__global__ void __launch_bounds__(64 * NWV, (WOcc<T, NC, MC, NWV>::value)) k_admm_wave(SolveArgs a) {
    typedef typename WV<T>::chunk chunk;
    typedef typename WV<T>::pair pair;
    constexpr int W = WV<T>::W, PW = W / 2;                            // elements / pairs per 16-byte chunk
    constexpr int HALF = 64 / NC, RL = MC / 64;                        // lanes per column; rows per lane
    constexpr int NCT = NC * NWV, MCT = MC * NWV;                      // columns / rows of the whole instance
    constexpr int AT = MCT / HALF, KH = NCT / HALF;                    // rows of A / columns of K, H per column part
    static_assert(NC * HALF == 64 && RL * 64 == MC && (HALF == 1 || HALF == 2) && (NWV == 1 || NWV == 2), "lane mapping");
    __shared__ __attribute__((aligned(16))) T nuL[MCT];                // nu (lam at a check) by row
    __shared__ __attribute__((aligned(16))) T xL[NCT];                 // x by column
    __shared__ __attribute__((aligned(16))) T dL[NCT];                 // d by column
    __shared__ __attribute__((aligned(16))) T dxL[NCT];                // dx by column
    __shared__ T redL[NWV][8];                                         // cross-wave maxima / sums (NWV > 1)
    __shared__ T rhoL[32];                                             // the rho ladder: read at every check (LDS, not a dependent global load)
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x & 63;   // dispatch order: longest solve first
    const int wv = (NWV > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int n = a.n, m = a.m, ldn = a.ldn, ldm = a.ldm;
    const int c = NC * wv + lane % NC, h = lane / NC;                  // global column, column part
    const int r0 = MC * wv + lane;                                     // first row of this lane
    const bool cok = c < n;
    // rho-ladder window (rqp_common.h; same protocol as k_admm_res2): K slot s of this instance holds ladder index wb + s.
    // SolveArgs.cont = 2: only the instances that left their window (cstat = 1), resumed EXACTLY behind the check they left at.
    int ri = a.rho_ind[b];
    int k0 = 0;
    bool exact = false;
    if (a.cont == 2) {
        if (a.cstat[b] == 0) return;
        const int ci = a.cont_iter[b];
        if (ci >= 0) {
            k0 = ci;
            exact = true;
        }
    }
    const int wb = a.wbase ? a.wbase[(a.sK == 0) ? 0 : b] : 0;
    if (a.cstat) {
        const int sl = ri - wb;
        if (sl < 0 || sl >= a.kwin) {                                   // (uniform) the incoming index lies outside the window: leave
            if (threadIdx.x == 0) {                                     // untouched; rqp_solve re-centres the window and restarts it
                a.cstat[b] = 1;
                if (!exact) a.cont_iter[b] = -1;
                atomicAdd(a.ncont, 1);
            }
            return;
        }
    }
    const T* A = (const T*)a.A + (size_t)b * a.sA;
    const T* At = (const T*)a.At + (size_t)b * a.sAt;
    const T* Ht = (const T*)a.Ht + (size_t)b * a.sH;
    const T* Kb = (const T*)a.K + (size_t)b * a.sK;

    // ---- matrices into registers (leading dimensions are multiples of 16 bytes, padding is zero); kept as element pairs:
    //      every product runs on pair FMAs (v_pk_fma_f32 for float) -- two accumulators, even / odd elements, added at the end
    pair Ar[RL][NCT / 2], Atc[AT / 2], Hc[KH / 2], Kc[KH / 2];
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
#pragma unroll
    for (int q = 0; q < RL; ++q) load_row(Ar[q], NCT, A + (size_t)(r0 + 64 * q) * ldn, r0 + 64 * q < m, ldn);   // row r_q of A
    load_row(Atc, AT, At + (size_t)c * ldm + AT * h, cok, ldm - AT * h);     // rows AT h .. of column c
    load_row(Hc, KH, Ht + (size_t)c * ldn + KH * h, cok, ldn - KH * h);      // sym(H): row c = column c
    if (threadIdx.x < 32) rhoL[threadIdx.x] = ((int)threadIdx.x < a.nrho) ? (T)a.rhos[threadIdx.x] : (T)0;   // (nrho <= 32: rqp_wave_fits)
    auto load_K = [&]() __attribute__((always_inline)) {
        load_row(Kc, KH, Kb + ((size_t)(ri - wb) * n + c) * ldn + KH * h, cok, ldn - KH * h);
    };
    load_K();

    // ---- vectors and state
    const T gc = cok ? ((const T*)a.g)[(size_t)b * n + c] : (T)0;
    double x = cok ? a.x[(size_t)b * n + c] : 0.0;
    T lr[RL], ur[RL], cr[RL], rv[RL];
    double z[RL], lam[RL], zt[RL], inv[RL];
#pragma unroll
    for (int q = 0; q < RL; ++q) {
        const int r = r0 + 64 * q;
        const bool rok = r < m;
        lr[q] = rok ? ((const T*)a.l)[(size_t)b * m + r] : (T)0;
        ur[q] = rok ? ((const T*)a.u)[(size_t)b * m + r] : (T)0;
        cr[q] = rok ? ((const T*)a.c)[(size_t)b * m + r] : (T)1;
        z[q] = rok ? a.z[(size_t)b * m + r] : 0.0;
        lam[q] = rok ? a.lam[(size_t)b * m + r] : 0.0;
        rv[q] = (T)a.rhos[ri] * cr[q];
        inv[q] = 1.0 / (double)rv[q];
    }

    // products: column role reads by-row vectors (nuL) / by-column vectors (xL, dL); row role reads dxL
    auto dot = [&](const pair* M, int len, const T* vec, pair acc) __attribute__((always_inline)) {   // acc + sum M[j] vec[j], pairwise
        // two accumulator pairs, alternating: one chain left a hazard s_nop between every two dependent v_pk_fma_f32
        pair acc1 = {(T)0, (T)0};
#pragma unroll
        for (int q = 0; q < len / W; ++q) {
            const chunk v = *(const chunk*)(vec + W * q);
#pragma unroll
            for (int e = 0; e < PW; ++e) {
                const pair ve = {v[2 * e], v[2 * e + 1]};
                if ((q * PW + e) & 1) acc1 = __builtin_elementwise_fma(M[q * PW + e], ve, acc1);
                else acc = __builtin_elementwise_fma(M[q * PW + e], ve, acc);
            }
        }
        return acc + acc1;
    };
    const pair zero2 = {(T)0, (T)0};
    auto at_times = [&](pair acc) __attribute__((always_inline)) { return dot(Atc, AT, nuL + AT * h, acc); };   // + A[AT h..][c]' nu[AT h..]
    auto h_times = [&](pair acc) __attribute__((always_inline)) { return dot(Hc, KH, xL + KH * h, acc); };     // + H[c][KH h..] x[KH h..]
    auto k_times = [&]() __attribute__((always_inline)) { return dot(Kc, KH, dL + KH * h, zero2); };           // K[c][KH h..] d[KH h..]
    auto a_times = [&](int q) __attribute__((always_inline)) {                                                 // A[r_q][:] dx
        const pair acc = dot(Ar[q], NCT, dxL, zero2);
        return acc[0] + acc[1];
    };
    auto fold = [&](pair v) __attribute__((always_inline)) { return v[0] + v[1]; };
    // lanes hand vectors to each other through LDS inside the wave: a wavefront-scope release + wave barrier orders the
    // hand-off for the compiler (the hardware already executes a wave's LDS operations in order)
    auto handoff = [&]() __attribute__((always_inline)) {
        if constexpr (NWV > 1) {
            __syncthreads();                                            // the vectors cross between the two waves
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    // The 7 maxima of a check over the whole instance (all waves), NaN-propagating like torch.max / norm(inf); every lane gets
    // them.  In registers (rqp_lanes.h): the NaN flags travel as a bit mask, the values through v_max -- the shuffle version
    // (6 ds_bpermute levels per value, the NaN selects as exec-masked branches) was a fifth of the kernel's instructions.
    auto inst_tmax7 = [&](T (&v)[7]) __attribute__((always_inline)) {
        int nanm = 0;
#pragma unroll
        for (int e = 0; e < 7; ++e) nanm |= (v[e] != v[e]) ? (1 << e) : 0;
#pragma unroll
        for (int e = 0; e < 7; ++e) v[e] = lanes_max(v[e]);
        nanm = wave_or_i(nanm);
        if constexpr (NWV > 1) {
            if (lane == 0) {
#pragma unroll
                for (int e = 0; e < 7; ++e) redL[wv][e] = v[e];
                redL[wv][7] = (T)nanm;                                   // (a small integer: exact in T)
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                const T o0 = redL[0][e], o1 = redL[1][e];
                v[e] = o0 > o1 ? o0 : o1;                                 // (NaN-free: a wave maximum that was all NaN is in the mask)
                if (o1 != o1) v[e] = o0;
            }
            nanm = (int)redL[0][7] | (int)redL[1][7];
        }
#pragma unroll
        for (int e = 0; e < 7; ++e)
            if ((nanm >> e) & 1) v[e] = (T)__builtin_nan("");
    };
    auto all_parts = [&](T v) __attribute__((always_inline)) {          // sum over the HALF parts of a column
        if constexpr (HALF == 2) v = xor32_sum(v);                      // (lanes 32 apart hold the two parts: a + b, either order)
        return v;
    };

    // (by-column LDS vectors are written by EVERY part of a column -- same value, same address: no exec-masked region
    //  in the loop.  With the writes under `if (h == 0)` hipcc 7.2 moved the following full-wave ds_reads of the same
    //  array into the masked block, leaving lanes 32..63 with stale registers.)
    // A x of the incoming state
    dxL[c] = (T)x;
    xL[c] = (T)x;
    handoff();
#pragma unroll
    for (int q = 0; q < RL; ++q)                                        // (an exact continuation brought A x along)
        zt[q] = exact ? ((r0 + 64 * q < m) ? a.ax[(size_t)b * m + r0 + 64 * q] : 0.0) : (double)a_times(q);

    bool converged = false;
    int iters = 0;
    T pri = (T)0, dua = (T)0, hx = (T)0;
    T rho_est = exact ? (T)a.cont_rho[b] : (T)a.rhos[ri];               // :211
    const T tolT = (T)a.tol;
    const int kmax = a.max_iter;

    // compute_residuals (:307-318) on the current state; leaves H x of the column in hx
    T scl_p = (T)0, scl_d = (T)0;                                       // residual scales of the last check (eps_rel)
    auto residuals = [&](T rho_carry, T& o_pri, T& o_dua) -> T {
        T w0 = (T)0, w1 = (T)0, w2 = (T)0;
        const size_t smat = (a.sA == 0) ? 0 : (size_t)b;                // (Ruiz scaling: caller-space norms, SolveArgs.scE)
#pragma unroll
        for (int q = 0; q < RL; ++q) {
            const T we = (a.scE && r0 + 64 * q < m) ? (T)(1.0 / a.scE[smat * m + r0 + 64 * q]) : (T)1;
            nuL[r0 + 64 * q] = (T)lam[q];
            w0 = wtmax(w0, (T)fabs((T)(zt[q] - z[q])) * we);
            w1 = wtmax(w1, (T)fabs((T)zt[q]) * we);
            w2 = wtmax(w2, (T)fabs((T)z[q]) * we);
        }
        handoff();
        const T t3 = all_parts(fold(at_times(zero2)));                  // A' lam
        hx = all_parts(fold(h_times(zero2)));                           // H x
        const T wd = (a.scD && cok) ? (T)(1.0 / (a.scC[smat] * a.scD[smat * n + c])) : (T)1;
        T v[7] = {w0, w1, w2, (T)fabs(hx + t3 + gc) * wd, (T)fabs(hx) * wd, (T)fabs(t3) * wd, (T)fabs(gc) * wd};
        inst_tmax7(v);
        const T v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4], v5 = v[5], v6 = v[6];
        if constexpr (NWV > 1) __syncthreads();                         // redL is reused by the next check
        o_pri = v0;
        o_dua = v3;
        scl_p = wtmax(v1, v2);
        scl_d = wtmax(wtmax(v4, v5), v6);
        const T num = v0 / scl_p;                                       // :315
        const T den = v3 / scl_d;                                       // :316
        T est = rho_carry * (T)sqrt(num / den);                         // :317
        if (est < (T)a.rho_min) est = (T)a.rho_min;                     // torch.clamp: NaN stays NaN
        if (est > (T)a.rho_max) est = (T)a.rho_max;
        return est;
    };

    int to_chk = a.check_interval;                                      // iterations until k % check_interval == 0
    for (int k = k0 + 1; k <= kmax; ++k) {                              // (k0 > 0: an exact continuation, on the check grid)
#pragma unroll
        for (int q = 0; q < RL; ++q) {                                  // row role: lam_hat, nu
            const double p = zt[q] - z[q];
            const double lh = lam[q] + (double)rv[q] * p;
            lam[q] = lh;
            nuL[r0 + 64 * q] = (T)(lh + (double)rv[q] * p);
        }
        handoff();
        {                                                               // column role: d = H x + g + A' nu ; dx = -K d
            const T d = all_parts(fold(h_times(at_times(zero2)))) + gc;
            dL[c] = d;
            handoff();
            const T dx = -all_parts(fold(k_times()));
            x += (double)dx;
            dxL[c] = dx;
            xL[c] = (T)x;
        }
        handoff();
#pragma unroll
        for (int q = 0; q < RL; ++q) {                                  // row role: A x, z
            zt[q] += (double)a_times(q);
            const double v = zt[q] + lam[q] * inv[q];
            double zn = v;                                              // torch.clamp: NaN stays NaN
            if (v < (double)lr[q]) zn = (double)lr[q];
            if (v > (double)ur[q]) zn = (double)ur[q];
            z[q] = zn;
        }
        iters = k;
        if (--to_chk == 0) {                                            // k % check_interval == 0, :218 (Q3 fixed: always check)
            to_chk = a.check_interval;
            const int ri_before = ri;
            rho_est = residuals(rho_est, pri, dua);                     // :220 (Q4: estimate is carried)
            if (rho_est > rhoL[ri] * tolT && ri < a.nrho - 1)                     // :223
                ri += 1;
            else if (rho_est < rhoL[ri] / tolT && ri > 0)                         // :226
                ri -= 1;
            if (a.info.trace && (k / a.check_interval) <= a.info.trace_cap && threadIdx.x == 0) {
                double* tr = a.info.trace + ((size_t)b * a.info.trace_cap + (k / a.check_interval - 1)) * 4;
                tr[0] = (double)pri; tr[1] = (double)dua; tr[2] = (double)rho_est; tr[3] = (double)ri_before;
            }
            const T tp = a.eps_rel > 0 ? (T)a.thr_p + (T)a.eps_rel * scl_p : (T)a.thr_p;      // :233 (+ relative term, 8(f)-3)
            const T td = a.eps_rel > 0 ? (T)a.thr_d + (T)a.eps_rel * scl_d : (T)a.thr_d;
            if (pri < tp && dua < td) {
                converged = true;
                break;
            }
            if (ri != ri_before) {                                      // "re-factor" = table lookup
                if (a.cstat && k < kmax && (ri < wb || ri >= wb + a.kwin)) {
                    // the new index has no K in this instance's window: leave with the exact state; rqp_solve re-factors a
                    // window around ri and continues the instance behind this check (SolveArgs.cont = 2)
                    if (h == 0 && cok) a.x[(size_t)b * n + c] = x;
#pragma unroll
                    for (int q = 0; q < RL; ++q) {
                        const int r = r0 + 64 * q;
                        if (r < m) {
                            a.z[(size_t)b * m + r] = z[q];
                            a.lam[(size_t)b * m + r] = lam[q];
                            a.ax[(size_t)b * m + r] = zt[q];
                        }
                    }
                    if (threadIdx.x == 0) {
                        a.rho_ind[b] = ri;
                        a.cont_iter[b] = k;
                        a.cont_rho[b] = (double)rho_est;
                        a.cstat[b] = 1;
                        atomicAdd(a.ncont, 1);
                    }
                    return;
                }
                load_K();
#pragma unroll
                for (int q = 0; q < RL; ++q) {
                    rv[q] = (T)a.rhos[ri] * cr[q];
                    inv[q] = 1.0 / (double)rv[q];
                }
            }
        }
    }
    if (!converged) rho_est = residuals(rho_est, pri, dua);             // :243 (Q11 fixed: fresh state)

    // objective 1/2 x'Hx + g'x (compute_J :320-322): hx holds H x of the final state
    double jp = (h == 0 && cok) ? (double)((T)x * ((T)0.5 * hx + gc)) : 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) jp += __shfl_xor(jp, off, 64);
    if constexpr (NWV > 1) {
        __shared__ double jpL[NWV];
        if (lane == 0) jpL[wv] = jp;
        __syncthreads();
        jp = jpL[0] + jpL[1];
    }

    // ---- update_results (:278-305) and the persistent state
    if (a.out_x && h == 0 && cok) ((T*)a.out_x)[(size_t)b * n + c] = (T)x;
    if (threadIdx.x == 0) {
        if (a.info.iter) a.info.iter[b] = converged ? iters : a.max_iter;
        if (a.last_iter) a.last_iter[b] = converged ? iters : a.max_iter;
        if (a.info.status) a.info.status[b] = converged ? RQP_STATUS_SOLVED : ((pri != pri || dua != dua) ? RQP_STATUS_NAN : RQP_STATUS_MAX_ITER);
        if (a.info.rho_ind) a.info.rho_ind[b] = ri;
        if (a.info.pri_res) a.info.pri_res[b] = (double)pri;
        if (a.info.dua_res) a.info.dua_res[b] = (double)dua;
        if (a.info.rho_estimate) a.info.rho_estimate[b] = (double)rho_est;
        if (a.info.obj_val) a.info.obj_val[b] = jp;
        a.rho_ind[b] = (a.warm_starting || a.keep_state) ? ri : a.rho_ind0;
        if (a.cstat) a.cstat[b] = 0;
    }
    if (h == 0 && cok) a.x[(size_t)b * n + c] = (a.warm_starting || a.keep_state) ? x : 0.0;           // state persists (:304) or is cleared (:324-333)
#pragma unroll
    for (int q = 0; q < RL; ++q) {
        const int r = r0 + 64 * q;
        if (r < m) {
            if (a.out_z) ((T*)a.out_z)[(size_t)b * m + r] = (T)z[q];
            if (a.out_lam) ((T*)a.out_lam)[(size_t)b * m + r] = (T)lam[q];
            a.z[(size_t)b * m + r] = (a.warm_starting || a.keep_state) ? z[q] : 0.0;
            a.lam[(size_t)b * m + r] = (a.warm_starting || a.keep_state) ? lam[q] : 0.0;
        }
    }
}

// 0: does not fit; 1: <T, 32, 64>; 2: <float, 32, 128>; 3: <float, 32, 64, 2> (two wavefronts per instance)
static int wave_class(const rqp_handle* h) {
    if (h->nrho > 32) return 0;                     // (the ladder sits in a 32-entry LDS array)
    if (h->n <= 32 && h->m <= 64) return 1;
    if (h->esz == 4 && h->n <= 32 && h->m <= 128) return 2;
    // two wavefronts per instance: measured 0.72x the mid resident tile (n <= 56, m <= 128) but 1.96x the big tile, so it
    // only takes the sizes the mid tile cannot hold
    if (h->esz == 4 && h->n > 56 && h->n <= 64 && h->m <= 128) return 3;
    return 0;
}

bool rqp_wave_fits(const rqp_handle* h) { return wave_class(h) != 0; }

hipError_t rqp_launch_solve_wave(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    const int wc = wave_class(h);
    if (wc == 3)
        k_admm_wave<float, 32, 64, 2><<<h->B, 128, 0, s>>>(a);
    else if (wc == 2)
        k_admm_wave<float, 32, 128><<<h->B, 64, 0, s>>>(a);
    else if (h->esz == 4)
        k_admm_wave<float, 32, 64><<<h->B, 64, 0, s>>>(a);
    else
        k_admm_wave<double, 32, 64><<<h->B, 64, 0, s>>>(a);
    return hipGetLastError();
}
