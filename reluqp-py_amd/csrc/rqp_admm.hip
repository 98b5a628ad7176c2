// rqp_admm.hip -- the ADMM hot loop of ReLU_QP.solve (reference
// ReLU-QP-py/reluqp/reluqpth.py:201-249), GENERIC variant: any (n, m), float or double,
// matrices streamed from global memory (L2 / Infinity Cache / HBM) every iteration.
//
// One workgroup (256 threads = 4 wavefronts) owns one QP instance for its whole solve:
// iterate (jit_forward :84-89), every check_interval iterations the residuals
// (compute_residuals :307-318), the rho-index move (:223-227) and the termination test
// (:233), then update_results (:278-305) -- one launch, no host round trip.
//
// The iterate is the residual-correction statement of SURVEY.md Appendix A.2
// (oracle/reluqp_oracle.py: forward_refine): with K = (H + sigma I + A' rho A)^-1,
//     p = A x - z ; lam_hat = lam + rho p ; nu = lam_hat + rho p
//     d = H x + g + A' nu ; dx = -K d ; x += dx ; (A x) += A dx
//     z = clamp(A x + lam_hat / rho, l, u) ; lam = lam_hat
// Vector state (x, z, lam, A x) is accumulated in float64 in LDS; the four matrix-vector
// products run in T.  Algebraically this is exactly W s + b of reluqpth.py:71-77.
#include "rqp_common.h"

template <typename T>
struct VecT;
template <>
struct VecT<float> {
    typedef float4 type;
    static constexpr int W = 4;
};
template <>
struct VecT<double> {
    typedef double2 type;
    static constexpr int W = 2;
};

__device__ __forceinline__ void vload(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void vload(const double* p, double (&v)[2]) {
    double2 t = *reinterpret_cast<const double2*>(p);
    v[0] = t.x; v[1] = t.y;
}

// torch.max / vector_norm(inf) semantics: NaN propagates
template <typename T>
__device__ __forceinline__ T tmax(T a, T b) {
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}

// out[c] = sum_r Mat[r][c] * w[r]   (Mat row-major [R][ld], ld % W == 0, pad columns zero)
// Threads = (column group of W columns) x (row slice); partial sums meet in LDS `part`.
// `w`, `out`, `part` are LDS; ends with a barrier (out is complete and `w` may be reused).
template <typename T>
__device__ void colmv(const T* __restrict__ Mat, int ld, int R, int C, const T* w, T* out, T* part) {
    constexpr int W = VecT<T>::W;
    const int tid = threadIdx.x;
    const int CG = (C + W - 1) / W;
    for (int cg0 = 0; cg0 < CG; cg0 += RQP_NT) {
        const int CGc = min(RQP_NT, CG - cg0);
        const int RS = RQP_NT / CGc;
        const int cg = tid % CGc, rs = tid / CGc;
        T acc[W];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] = T(0);
        if (rs < RS) {
            const T* base = Mat + (size_t)(cg0 + cg) * W;
#pragma unroll 4
            for (int r = rs; r < R; r += RS) {
                T v[W];
                vload(base + (size_t)r * ld, v);
                const T wr = w[r];
#pragma unroll
                for (int e = 0; e < W; ++e) acc[e] = fma(v[e], wr, acc[e]);
            }
#pragma unroll
            for (int e = 0; e < W; ++e) part[(size_t)rs * (CGc * W) + cg * W + e] = acc[e];
        }
        __syncthreads();
        for (int c = tid; c < CGc * W; c += RQP_NT) {
            T s = T(0);
            for (int q = 0; q < RS; ++q) s += part[(size_t)q * (CGc * W) + c];
            if (cg0 * W + c < C) out[cg0 * W + c] = s;
        }
        __syncthreads();
    }
}

// NaN-propagating max over the workgroup of NV values per thread; result broadcast to all.
template <typename T, int NV>
__device__ void block_max(T (&v)[NV], T* red /* LDS [NV * 4] */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] = tmax(v[e], (T)__shfl_xor(v[e], off, RQP_WAVE));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
#pragma unroll
        for (int e = 0; e < NV; ++e) red[wave * NV + e] = v[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        T r = red[e];
        for (int w = 1; w < RQP_NT / RQP_WAVE; ++w) r = tmax(r, red[w * NV + e]);
        v[e] = r;
    }
    __syncthreads();
}

// sum over the workgroup (fixed order: wave shuffles, then the 4 wave sums), broadcast to all; scratch >= 4 doubles of LDS
__device__ double block_sum(double v, double* scratch) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, RQP_WAVE);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
    __syncthreads();
    return r;
}

template <typename T>
__global__ void __launch_bounds__(RQP_NT) k_admm_generic(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int W = VecT<T>::W;
    const int n = a.n, m = a.m, ldn = a.ldn, ldm = a.ldm;
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;   // dispatch order: longest solve first
    const int maxd = max(ldn, ldm);

    // ---- LDS carve-up (host computes the same size: rqp_launch_solve_generic)
    double* xs = (double*)smem_raw;       // [n]   x          (float64 accumulator)
    double* zs = xs + n;                  // [m]   z
    double* ls = zs + m;                  // [m]   lam
    double* zts = ls + m;                 // [m]   A x        (float64 accumulator)
    T* gT = (T*)(zts + m);                // [ldn]
    T* lT = gT + ldn;                     // [ldm]
    T* uT = lT + ldm;                     // [ldm]
    T* rvT = uT + ldm;                    // [ldm] rho vector of the current index
    T* cT = rvT + ldm;                    // [ldm] 1 / 1e3 equality scale
    T* vin = cT + ldm;                    // [maxd] product input
    T* hx = vin + maxd;                   // [ldn] H x
    T* vn = hx + ldn;                     // [ldn] n-sized product output
    T* dv = vn + ldn;                     // [ldn] d, then dx
    T* vm = dv + ldn;                     // [ldm] m-sized product output
    T* part = vm + ldm;                   // [RQP_NT * W]
    T* red = part + RQP_NT * W;           // [8 * 4]

    const T* Ht = (const T*)a.Ht + (size_t)b * a.sH;
    const T* A = (const T*)a.A + (size_t)b * a.sA;
    const T* At = (const T*)a.At + (size_t)b * a.sAt;
    const T* Kb = (const T*)a.K + (size_t)b * a.sK;
    const int wb = a.wbase ? a.wbase[(a.sK == 0) ? 0 : b] : 0;     // rho window: K slot s holds ladder index wb + s

    // exit-and-continue of a windowed handle (SolveArgs.cont = 2, see rqp_resident2.hip): only the instances that left their
    // window, resumed exactly where they stopped
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
    if (a.cstat && a.mode == 0) {
        const int sl = a.rho_ind[b] - wb;
        if (sl < 0 || sl >= a.kwin) {                              // incoming index outside the window: leave untouched
            if (tid == 0) {
                a.cstat[b] = 1;
                if (!exact) a.cont_iter[b] = -1;
                atomicAdd(a.ncont, 1);
            }
            return;
        }
    }

    if (a.mode == 3) {                       // certificate pass behind another solve kernel (rqp_solve): only the instances that
        const bool todo = a.info.status[b] == RQP_STATUS_MAX_ITER || a.info.status[b] == RQP_STATUS_NAN;   // spent their iterations are examined; the state the
        if (!todo) {                                                    // solve kernel was asked to keep is cleared here
            if (!a.warm_starting) {
                for (int i = tid; i < n; i += RQP_NT) a.x[(size_t)b * n + i] = 0.0;
                for (int i = tid; i < m; i += RQP_NT) {
                    a.z[(size_t)b * m + i] = 0.0;
                    a.lam[(size_t)b * m + i] = 0.0;
                }
                if (tid == 0) a.rho_ind[b] = a.rho_ind0;
            }
            return;
        }
    }
    for (int i = tid; i < n; i += RQP_NT) {
        xs[i] = a.x[(size_t)b * n + i];
        gT[i] = ((const T*)a.g)[(size_t)b * n + i];
    }
    for (int i = tid; i < m; i += RQP_NT) {
        zs[i] = a.z[(size_t)b * m + i];
        ls[i] = a.lam[(size_t)b * m + i];
        lT[i] = ((const T*)a.l)[(size_t)b * m + i];
        uT[i] = ((const T*)a.u)[(size_t)b * m + i];
        cT[i] = ((const T*)a.c)[(size_t)b * m + i];
    }
    int ri = a.rho_ind[b];
    T rho_est = (a.mode == 2) ? (T)a.rho_in : (exact ? (T)a.cont_rho[b] : (T)a.rhos[ri]);      // reluqpth.py:211
    for (int i = tid; i < m; i += RQP_NT) rvT[i] = (T)a.rhos[ri] * cT[i];
    for (int i = tid; i < n; i += RQP_NT) vin[i] = (T)xs[i];
    __syncthreads();
    if (!exact) {
        colmv<T>(At, ldm, n, m, vin, vm, part);                    // A x of the incoming state
        for (int i = tid; i < m; i += RQP_NT) zts[i] = (double)vm[i];
    } else {                                                       // (an exact continuation brought it along)
        for (int i = tid; i < m; i += RQP_NT) zts[i] = a.ax[(size_t)b * m + i];
    }
    __syncthreads();

    bool hx_valid = false, converged = false;
    int iters = k0;
    T pri = T(0), dua = T(0);
    const T tolT = (T)a.tol;
    const int kmax = (a.mode == 2) ? 0 : ((a.mode == 3) ? 1 : a.max_iter);
    int cert = 0;                             // RQP_STATUS_PRIMAL_INFEASIBLE / RQP_STATUS_DUAL_INFEASIBLE once a certificate holds

    // ---- infeasibility certificates (the reference has none: reluqpth.py:233 only tests residuals).
    // OSQP's tests on the directions of the last iteration: dy = rho (A x - z), the dual increment the next iteration will
    // apply (projected on the polar of the recession cone of [l, u]), and dx, the x step just taken (dv), with A dx in vm.
    //   primal infeasible:  u' max(dy,0) + l' min(dy,0) < -eps |dy|  and  |A' dy| < eps |dy|
    //   dual infeasible:    g' dx < -eps |dx|,  |H dx| < eps |dx|,  (A dx)_i <= eps |dx| where u_i is finite, >= -eps |dx| where l_i is
    // Uses vin, vn, part as scratch (hx, the state and dv / vm are left intact).
    auto certificates = [&]() -> int {
        const T inf = (T)INFINITY;
        T ndy = T(0);
        double lhs = 0.0;
        for (int i = tid; i < m; i += RQP_NT) {
            T dy = (T)((double)rvT[i] * (zts[i] - zs[i]));
            const bool uinf = uT[i] == inf, linf = lT[i] == -inf;
            if (uinf && linf) dy = T(0);
            else if (uinf) dy = dy < T(0) ? dy : T(0);
            else if (linf) dy = dy > T(0) ? dy : T(0);
            vin[i] = dy;
            ndy = tmax(ndy, (T)fabs(dy));
            if (dy > T(0)) lhs += (double)uT[i] * (double)dy;
            if (dy < T(0)) lhs += (double)lT[i] * (double)dy;
        }
        __syncthreads();
        colmv<T>(A, ldn, m, n, vin, vn, part);                     // A' dy
        T nat = T(0), ndx = T(0);
        double gdx = 0.0;
        for (int i = tid; i < n; i += RQP_NT) {
            nat = tmax(nat, (T)fabs(vn[i]));
            ndx = tmax(ndx, (T)fabs(dv[i]));
            gdx += (double)gT[i] * (double)dv[i];
        }
        __syncthreads();
        colmv<T>(Ht, ldn, n, n, dv, vn, part);                     // H dx
        T nhdx = T(0);
        for (int i = tid; i < n; i += RQP_NT) nhdx = tmax(nhdx, (T)fabs(vn[i]));
        T mx[4] = {ndy, nat, ndx, nhdx};
        block_max<T, 4>(mx, red);
        lhs = block_sum(lhs, (double*)part);
        gdx = block_sum(gdx, (double*)part);
        const T ep = (T)a.eps_pinf, ed = (T)a.eps_dinf;
        if (mx[0] > T(0) && lhs < -(double)(ep * mx[0]) && mx[1] < ep * mx[0]) return RQP_STATUS_PRIMAL_INFEASIBLE;
        double bad = 0.0;
        const T tol = ed * mx[2];
        for (int i = tid; i < m; i += RQP_NT)
            if ((uT[i] < inf && vm[i] > tol) || (lT[i] > -inf && vm[i] < -tol)) bad = 1.0;
        bad = block_sum(bad, (double*)part);
        if (mx[2] > T(0) && gdx < -(double)(ed * mx[2]) && mx[3] < ed * mx[2] && bad == 0.0) return RQP_STATUS_DUAL_INFEASIBLE;
        return 0;
    };

    // ---- residuals of the current state (compute_residuals, reluqpth.py:307-318) ----------
    // leaves H x in hx (hx_valid), returns pri/dua and the new carried rho estimate
    T scl_p = T(0), scl_d = T(0);          // max(|Ax|,|z|) and max(|Hx|,|A'lam|,|g|) of the last check (eps_rel)
    auto residuals = [&](T rho_carry, T& o_pri, T& o_dua) -> T {
        for (int i = tid; i < n; i += RQP_NT) vin[i] = (T)xs[i];
        __syncthreads();
        colmv<T>(Ht, ldn, n, n, vin, hx, part);                    // t2 = H x
        for (int i = tid; i < m; i += RQP_NT) vin[i] = (T)ls[i];
        __syncthreads();
        colmv<T>(A, ldn, m, n, vin, vn, part);                     // t3 = A' lam
        T v[7];
#pragma unroll
        for (int e = 0; e < 7; ++e) v[e] = T(0);
        // (Ruiz scaling: every term goes back to the caller's space before its norm -- SolveArgs.scE)
        const size_t smat = (a.sA == 0) ? 0 : (size_t)b;
        const double* sE = a.scE ? a.scE + smat * m : nullptr;
        const double* sD = a.scD ? a.scD + smat * n : nullptr;
        const double sc = a.scC ? a.scC[smat] : 1.0;
        for (int i = tid; i < m; i += RQP_NT) {
            const T we = sE ? (T)(1.0 / sE[i]) : T(1);
            const T t1 = (T)zts[i], zi = (T)zs[i];
            v[0] = tmax(v[0], (T)fabs((T)(zts[i] - zs[i])) * we);  // |A x - z|
            v[1] = tmax(v[1], (T)fabs(t1) * we);
            v[2] = tmax(v[2], (T)fabs(zi) * we);
        }
        for (int i = tid; i < n; i += RQP_NT) {
            const T wd = sD ? (T)(1.0 / (sc * sD[i])) : T(1);
            v[3] = tmax(v[3], (T)fabs(hx[i] + vn[i] + gT[i]) * wd);   // |H x + A' lam + g|
            v[4] = tmax(v[4], (T)fabs(hx[i]) * wd);
            v[5] = tmax(v[5], (T)fabs(vn[i]) * wd);
            v[6] = tmax(v[6], (T)fabs(gT[i]) * wd);
        }
        block_max<T, 7>(v, red);
        o_pri = v[0];
        o_dua = v[3];
        scl_p = tmax(v[1], v[2]);
        scl_d = tmax(tmax(v[4], v[5]), v[6]);
        const T num = v[0] / scl_p;                                 // :315
        const T den = v[3] / scl_d;                                 // :316
        T est = rho_carry * (T)sqrt(num / den);                    // :317
        if (est < (T)a.rho_min) est = (T)a.rho_min;                // torch.clamp: NaN stays NaN
        if (est > (T)a.rho_max) est = (T)a.rho_max;
        return est;
    };

    for (int k = k0 + 1; k <= kmax; ++k) {
        const int slot = min(max(ri - wb, 0), a.kwin - 1);         // (in range in mode 0; modes 1 / 3 re-centre the windows first)
        const T* Kj = Kb + (size_t)slot * n * ldn;
        if (!hx_valid) {
            for (int i = tid; i < n; i += RQP_NT) vin[i] = (T)xs[i];
            __syncthreads();
            colmv<T>(Ht, ldn, n, n, vin, hx, part);                // H x
        }
        for (int i = tid; i < m; i += RQP_NT) {
            const double rv = (double)rvT[i];
            const double p = zts[i] - zs[i];
            const double lh = ls[i] + rv * p;                      // lam_hat
            ls[i] = lh;
            vin[i] = (T)(lh + rv * p);                             // nu
        }
        __syncthreads();
        colmv<T>(A, ldn, m, n, vin, vn, part);                     // A' nu
        for (int i = tid; i < n; i += RQP_NT) dv[i] = hx[i] + gT[i] + vn[i];   // d
        __syncthreads();
        colmv<T>(Kj, ldn, n, n, dv, vn, part);                     // K d
        for (int i = tid; i < n; i += RQP_NT) {
            const T dx = -vn[i];
            xs[i] += (double)dx;
            dv[i] = dx;
        }
        __syncthreads();
        colmv<T>(At, ldm, n, m, dv, vm, part);                     // A dx
        for (int i = tid; i < m; i += RQP_NT) {
            const double zt = zts[i] + (double)vm[i];
            zts[i] = zt;
            const double v = zt + ls[i] / (double)rvT[i];
            double zn = v;                                          // torch.clamp: NaN stays NaN
            if (v < (double)lT[i]) zn = (double)lT[i];
            if (v > (double)uT[i]) zn = (double)uT[i];
            zs[i] = zn;
        }
        hx_valid = false;
        iters = k;
        __syncthreads();

        if (a.mode == 0 && (k % a.check_interval) == 0) {          // :218 (Q3 fixed: always check)
            const int ri_before = ri;
            rho_est = residuals(rho_est, pri, dua);                // :220 (Q4: estimate is carried)
            hx_valid = true;
            if (rho_est > (T)a.rhos[ri] * tolT && ri < a.nrho - 1)           // :223
                ri += 1;
            else if (rho_est < (T)a.rhos[ri] / tolT && ri > 0)               // :226
                ri -= 1;
            if (a.info.trace && (k / a.check_interval) <= a.info.trace_cap && tid == 0) {
                double* tr = a.info.trace + ((size_t)b * a.info.trace_cap + (k / a.check_interval - 1)) * 4;
                tr[0] = (double)pri; tr[1] = (double)dua; tr[2] = (double)rho_est; tr[3] = (double)ri_before;
            }
            if (ri != ri_before) {
                for (int i = tid; i < m; i += RQP_NT) rvT[i] = (T)a.rhos[ri] * cT[i];
                __syncthreads();
            }
            // :233, plus the OSQP-style relative term when eps_rel > 0 (0 = the reference's test)
            const T tp = a.eps_rel > 0 ? (T)a.thr_p + (T)a.eps_rel * scl_p : (T)a.thr_p;
            const T td = a.eps_rel > 0 ? (T)a.thr_d + (T)a.eps_rel * scl_d : (T)a.thr_d;
            if (pri < tp && dua < td) {
                converged = true;
                break;
            }
            if (a.check_infeas) {
                cert = certificates();
                if (cert) break;
            }
            if (a.cstat && k < kmax && (ri < wb || ri >= wb + a.kwin)) {
                // the new index has no K in this instance's window: leave with the exact state (see rqp_resident2.hip)
                for (int i = tid; i < n; i += RQP_NT) a.x[(size_t)b * n + i] = xs[i];
                for (int i = tid; i < m; i += RQP_NT) {
                    a.z[(size_t)b * m + i] = zs[i];
                    a.lam[(size_t)b * m + i] = ls[i];
                    a.ax[(size_t)b * m + i] = zts[i];
                }
                if (tid == 0) {
                    a.rho_ind[b] = ri;
                    a.cont_iter[b] = k;
                    a.cont_rho[b] = (double)rho_est;
                    a.cstat[b] = 1;
                    atomicAdd(a.ncont, 1);
                }
                return;
            }
        }
    }
    if (a.mode == 3) {                                             // one iteration taken from the persisted state: its directions
        cert = certificates();
        if (tid == 0 && cert) a.info.status[b] = cert;
        if (!a.warm_starting) {                                    // the solve kernel kept the state for this pass only
            for (int i = tid; i < n; i += RQP_NT) a.x[(size_t)b * n + i] = 0.0;
            for (int i = tid; i < m; i += RQP_NT) {
                a.z[(size_t)b * m + i] = 0.0;
                a.lam[(size_t)b * m + i] = 0.0;
            }
            if (tid == 0) a.rho_ind[b] = a.rho_ind0;
        }
        return;
    }

    if (a.mode == 1) {                                             // iterate-only: keep the state
        for (int i = tid; i < n; i += RQP_NT) a.x[(size_t)b * n + i] = xs[i];
        for (int i = tid; i < m; i += RQP_NT) {
            a.z[(size_t)b * m + i] = zs[i];
            a.lam[(size_t)b * m + i] = ls[i];
        }
        return;
    }
    if (!converged && !cert) rho_est = residuals(rho_est, pri, dua);   // :243 (Q11 fixed: fresh state)

    // objective 1/2 x'Hx + g'x (compute_J :320-322): hx holds H x of the final state
    double jp = 0.0;
    for (int i = tid; i < n; i += RQP_NT) jp += (double)((T)xs[i] * (T)(T(0.5) * hx[i] + gT[i]));
    for (int off = 32; off >= 1; off >>= 1) jp += __shfl_xor(jp, off, RQP_WAVE);
    double* redd = (double*)part;
    if ((tid & 63) == 0) redd[tid >> 6] = jp;
    __syncthreads();
    const double obj = redd[0] + redd[1] + redd[2] + redd[3];

    if (a.mode == 2) {
        if (tid == 0) {
            if (a.r_pri) a.r_pri[b] = (double)pri;
            if (a.r_dua) a.r_dua[b] = (double)dua;
            if (a.r_rho) a.r_rho[b] = (double)rho_est;
            if (a.r_obj) a.r_obj[b] = obj;
        }
        return;
    }

    // ---- update_results (reluqpth.py:278-305)
    if (a.out_x) for (int i = tid; i < n; i += RQP_NT) ((T*)a.out_x)[(size_t)b * n + i] = (T)xs[i];
    if (a.out_z) for (int i = tid; i < m; i += RQP_NT) ((T*)a.out_z)[(size_t)b * m + i] = (T)zs[i];
    if (a.out_lam) for (int i = tid; i < m; i += RQP_NT) ((T*)a.out_lam)[(size_t)b * m + i] = (T)ls[i];
    if (tid == 0) {
        if (a.cstat) a.cstat[b] = 0;
        if (a.info.iter) a.info.iter[b] = (converged || cert) ? iters : a.max_iter;
        if (a.last_iter) a.last_iter[b] = (converged || cert) ? iters : a.max_iter;
        if (a.info.status)
            a.info.status[b] = converged ? RQP_STATUS_SOLVED
                                         : (cert ? cert : ((pri != pri || dua != dua) ? RQP_STATUS_NAN : RQP_STATUS_MAX_ITER));
        if (a.info.rho_ind) a.info.rho_ind[b] = ri;
        if (a.info.pri_res) a.info.pri_res[b] = (double)pri;
        if (a.info.dua_res) a.info.dua_res[b] = (double)dua;
        if (a.info.rho_estimate) a.info.rho_estimate[b] = (double)rho_est;
        if (a.info.obj_val) a.info.obj_val[b] = obj;
    }
    if (a.warm_starting || a.keep_state) {                         // state + rho index persist (:304)
        for (int i = tid; i < n; i += RQP_NT) a.x[(size_t)b * n + i] = xs[i];
        for (int i = tid; i < m; i += RQP_NT) {
            a.z[(size_t)b * m + i] = zs[i];
            a.lam[(size_t)b * m + i] = ls[i];
        }
        if (tid == 0) a.rho_ind[b] = ri;
    } else {                                                       // clear_primal_dual (:324-333)
        for (int i = tid; i < n; i += RQP_NT) a.x[(size_t)b * n + i] = 0.0;
        for (int i = tid; i < m; i += RQP_NT) {
            a.z[(size_t)b * m + i] = 0.0;
            a.lam[(size_t)b * m + i] = 0.0;
        }
        if (tid == 0) a.rho_ind[b] = a.rho_ind0;
    }
}

size_t rqp_generic_lds_bytes(const rqp_handle* h) {
    const size_t W = (h->esz == 4) ? 4 : 2;
    const size_t maxd = (size_t)(h->ldn > h->ldm ? h->ldn : h->ldm);
    size_t dbl = (size_t)h->n + 3 * (size_t)h->m;
    size_t t = 4 * (size_t)h->ldn /*gT hx vn dv*/ + 5 * (size_t)h->ldm /*lT uT rvT cT vm*/ + maxd + RQP_NT * W + 32;
    return dbl * sizeof(double) + t * h->esz;
}

// once per handle (rqp_setup): raise the dynamic-LDS limit of the instantiation this handle launches
hipError_t rqp_prepare_generic(const rqp_handle* h) {
    const size_t lds = rqp_generic_lds_bytes(h);
    if (lds <= 48 * 1024) return hipSuccess;
    return h->esz == 4 ? rqp_raise_lds_limit((const void*)k_admm_generic<float>, (size_t)lds)
                       : rqp_raise_lds_limit((const void*)k_admm_generic<double>, (size_t)lds);
}

hipError_t rqp_launch_solve_generic(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    const size_t lds = rqp_generic_lds_bytes(h);
    if (h->esz == 4)
        k_admm_generic<float><<<h->B, RQP_NT, lds, s>>>(a);
    else
        k_admm_generic<double><<<h->B, RQP_NT, lds, s>>>(a);
    return hipGetLastError();
}
