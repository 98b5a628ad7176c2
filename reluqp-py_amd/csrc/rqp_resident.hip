// rqp_resident.hip -- the ADMM hot loop of ReLU_QP.solve (reference
// ReLU-QP-py/reluqp/reluqpth.py:201-249), RESIDENT variant for float32 problems whose
// factored KKT tile fits on one CU:  n <= 8*CB, n <= 32*KR, m <= 32*RB
// (RB=10, CB=13, KR=4  ->  n <= 104, m <= 320: BASELINE config n=100, m=300).
//
// One 256-thread workgroup (4 wavefronts, one per SIMD) owns one QP for its whole solve and
// two workgroups share a CU, so one QP's barrier bubbles are filled by the other's math.
// The matrices are read from HBM ONCE per solve:
//   A  (m x n)  lives in VGPRs : thread (p = tid>>3, q = tid&7) holds the RB x CB block
//               rows RB*p.., cols CB*q..  -- used for BOTH A dx (reduce over q: DPP) and
//               A' nu (reduce over p: permlane swaps + one LDS hop across the 4 waves)
//   K_j (n x n) lives in VGPRs : KR x CB block per thread, reloaded only when the rho index moves
//   H  (n x n)  lives in LDS   : lane-linear 16-byte image, read with ds_read_b128
// All vectors (x, z, lam, A x in float64; g, l, u, rho, nu, d, dx in float32) live in LDS.
//
// Same recurrence and same check logic as k_admm_generic (rqp_admm.hip); see that file and
// oracle/reluqp_oracle.py:forward_refine for the statement and the reference line citations.
#include <cstdio>
#include <cstdlib>

#include "rqp_common.h"

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 8 lanes that share tid>>3 (q = lane bits 0..2); every lane gets the total
__device__ __forceinline__ float allsum_q(float v) {
    v += dppf<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
    v += dppf<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
    v += dppf<0x141>(v);   // row_half_mirror      (i <-> 7-i)
    return v;
}
// NOTE (hipcc / ROCm 7.2): `r = __builtin_amdgcn_permlane32_swap(a, b, ..); r.x + r.y` is miscompiled
// to `v_add v, v1, v1` (both extracts read the first result), which silently doubles instead of
// summing partner lanes.  The inline-asm form below is what the builtin should have produced; the
// leading s_nop 1 covers the "VALU write -> v_permlane*_swap read" hazard (2 wait states).
// lanes i and i+32 exchange: returns (lower half: a.lo+a.hi, upper half: b.lo+b.hi)
__device__ __forceinline__ float swapsum32(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// even/odd 16-lane rows exchange: returns (even rows: a.even+a.odd, odd rows: b.even+b.odd)
__device__ __forceinline__ float swapsum16(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

template <typename T>
__device__ __forceinline__ T tmaxr(T a, T b) {   // NaN-propagating max (torch semantics)
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}

template <int RB_, int CB_, int KR_>
struct ResCfg {
    static constexpr int RB = RB_, CB = CB_, KR = KR_;
    static constexpr int NT = 256, Q = 8, P = 32, NW = 4;
    static constexpr int M = RB * P;            // padded rows of A
    static constexpr int N = CB * Q;            // padded cols of A / H / K
    static constexpr int KN = KR * P;           // padded rows of H / K  (>= N)
    static constexpr int AE = RB * CB;          // A elements per thread
    static constexpr int KE = KR * CB;          // K (and H) elements per thread
    static constexpr int KE4 = (KE + 3) / 4;    // H 16-byte units per thread
    static constexpr int H1 = (CB + 1) / 2;     // reduce-scatter widths of the A' product
    static constexpr int H2 = (H1 + 1) / 2;
    static constexpr size_t lds_bytes() {
        return (size_t)KE4 * NT * 16                       // Hs
               + (size_t)M * 8 * 4                         // zt64 lam64 z64 inv64
               + (size_t)M * 4 * 5                         // lT uT rv32 cT nu
               + (size_t)KN * 8                            // x64
               + (size_t)KN * 4 * 5                        // xin dxv hx g d
               + (size_t)NW * N * 4                        // part
               + 64 * 4 + 16 * 8;                          // red, redd
    }
};

template <class C>
__global__ void __launch_bounds__(256, 2) k_admm_resident(SolveArgs a, const float* __restrict__ Apack,
                                                          const float* __restrict__ Kpack,
                                                          const float* __restrict__ Hpack) {
    constexpr int RB = C::RB, CB = C::CB, KR = C::KR, NT = C::NT, M = C::M, N = C::N, KN = C::KN;
    constexpr int AE = C::AE, KE = C::KE, KE4 = C::KE4, H1 = C::H1, H2 = C::H2, NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // vectors first (every offset < 64 KB, so ds_* immediates reach them from one base VGPR), H last
    double* zt64 = (double*)smem_raw;                     // [M] A x
    double* lam64 = zt64 + M;                             // [M]
    double* z64 = lam64 + M;                              // [M]
    double* inv64 = z64 + M;                              // [M] 1 / rho_i
    double* x64 = inv64 + M;                              // [KN]
    double* redd = x64 + KN;                              // [16]
    float* lT = (float*)(redd + 16);                      // [M]
    float* uT = lT + M;
    float* rv32 = uT + M;                                 // rho vector of the current index
    float* cT = rv32 + M;                                 // 1 / 1e3 equality scale
    float* nu = cT + M;                                   // [M] nu (or lam at a check)
    float* xin = nu + M;                                  // [KN] float(x)
    float* dxv = xin + KN;                                // [KN] dx
    float* hx = dxv + KN;                                 // [KN] H x
    float* gT = hx + KN;                                  // [KN]
    float* dvec = gT + KN;                                // [KN] d
    float* part = dvec + KN;                              // [NW][N]
    float* red = part + NW * N;                           // [64]
    float* Hs = red + 64;                                 // [KE4][NT][4]  (16-byte aligned: see static_assert)
    static_assert(((size_t)M * 8 * 4 + KN * 8 + 16 * 8 + (size_t)M * 4 * 5 + KN * 4 * 5 + NW * N * 4 + 64 * 4) % 16 == 0,
                  "H image must start 16-byte aligned");

    const int n = a.n, m = a.m;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int q = tid & 7, p = tid >> 3, wave = tid >> 6, lane = tid & 63;
    const size_t mat = (a.sA == 0) ? 0 : (size_t)b;

    // ---- matrices: A and K_j to registers, H to LDS (each element read from HBM once per solve)
    float ar[RB][CB];
    {
        const float* Ap = Apack + mat * (size_t)AE * NT + tid;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) ar[r][c] = Ap[(size_t)(r * CB + c) * NT];
        const float4* Hp = (const float4*)(Hpack + mat * (size_t)KE4 * NT * 4);
#pragma unroll
        for (int u = 0; u < KE4; ++u) ((float4*)Hs)[u * NT + tid] = Hp[u * NT + tid];
    }
    int ri = a.rho_ind[b];
    float kr[KR][CB];
    auto load_K = [&](int j) {
        const float* Kp = Kpack + (mat * a.nrho + j) * (size_t)KE * NT + tid;
#pragma unroll
        for (int r = 0; r < KR; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) kr[r][c] = Kp[(size_t)(r * CB + c) * NT];
    };
    load_K(ri);

    // ---- vectors
    for (int i = tid; i < M; i += NT) {
        const bool in = i < m;
        zt64[i] = 0.0;
        z64[i] = in ? a.z[(size_t)b * m + i] : 0.0;
        lam64[i] = in ? a.lam[(size_t)b * m + i] : 0.0;
        lT[i] = in ? ((const float*)a.l)[(size_t)b * m + i] : 0.f;
        uT[i] = in ? ((const float*)a.u)[(size_t)b * m + i] : 0.f;
        cT[i] = in ? ((const float*)a.c)[(size_t)b * m + i] : 1.f;
    }
    for (int i = tid; i < KN; i += NT) {
        const bool in = i < n;
        const double xv = in ? a.x[(size_t)b * n + i] : 0.0;
        x64[i] = xv;
        xin[i] = (float)xv;
        gT[i] = in ? ((const float*)a.g)[(size_t)b * n + i] : 0.f;
        hx[i] = 0.f;
        dxv[i] = 0.f;
        dvec[i] = 0.f;
    }
    auto set_rho_rows = [&](int j) {
        const float rho = (float)a.rhos[j];
        for (int i = tid; i < M; i += NT) {
            const float rv = rho * cT[i];
            rv32[i] = rv;
            inv64[i] = 1.0 / (double)rv;
        }
    };
    __syncthreads();
    set_rho_rows(ri);
    __syncthreads();

    // where this lane deposits its share of the A' reduce-scatter (see prod_At): ONE address VGPR
    float* pw_ptr;
    int pw_n;
    {
        const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1;
        const int cbase = H2 * b4 + H1 * b5;
        int nval = b4 ? (H1 - H2) : H2;                 // entries of s2[] that are not step-B duplicates
        if (b5 && cbase + nval > CB) nval = CB - cbase; // ... nor step-A duplicates
        pw_n = (b3 == 0) ? nval : 0;
        pw_ptr = part + wave * N + CB * q + cbase;
    }

    // ---- products -----------------------------------------------------------------------
    // y[RB*p + r] = sum_c A[..][CB*q + c] * v[CB*q + c], summed over q: every q-lane gets all RB sums
    auto prod_A = [&](const float* v, float (&acc)[RB]) {
        float vc[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) vc[c] = v[CB * q + c];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c) s = fmaf(ar[r][c], vc[c], s);
            acc[r] = allsum_q(s);
        }
    };
    // the value of acc[] that belongs to the row this lane owns in pass `ps` (row RB*p + 8*ps + q)
    auto pick = [&](const float (&acc)[RB], int ps) -> float {
        float v = acc[8 * ps];
#pragma unroll
        for (int r = 1; r < 8; ++r)
            if (8 * ps + r < RB) v = (q == r) ? acc[8 * ps + r] : v;
        return v;
    };
    // wave-partial of A' w: part[wave][CB*q + c] = sum over this wave's rows of A[row][.] * w[row]
    auto prod_At = [&](const float* w) {
        float wr[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) wr[r] = w[RB * p + r];
        float acc[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < RB; ++r) s = fmaf(ar[r][c], wr[r], s);
            acc[c] = s;
        }
        // reduce-scatter over the 8 row groups of the wave (lane bits 5, 4, 3)
        float s1[H1];
#pragma unroll
        for (int i = 0; i < H1; ++i) s1[i] = (i + H1 < CB) ? swapsum32(acc[i], acc[i + H1]) : swapsum32(acc[i], acc[i]);
        float s2[H2];
#pragma unroll
        for (int i = 0; i < H2; ++i) s2[i] = (i + H2 < H1) ? swapsum16(s1[i], s1[i + H2]) : swapsum16(s1[i], s1[i]);
#pragma unroll
        for (int i = 0; i < H2; ++i) s2[i] += dppf<0x128>(s2[i]);       // row_ror:8 (xor 8)
        // lane class (b5, b4) now holds the wave totals of columns cbase .. cbase + nval - 1:
        //   s2[i] <-> column i + H2*b4 + H1*b5 ; the tail entries of the odd classes are duplicates
        if (pw_n > 0) {
#pragma unroll
            for (int i = 0; i < H2; ++i)
                if (i < pw_n) pw_ptr[i] = s2[i];
        }
    };
    // H x -> hx[KR*p + r]   (H from LDS, lane-linear b128 image)
    auto prod_H = [&]() {
        float vc[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) vc[c] = xin[CB * q + c];
        float acc[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) acc[r] = 0.f;
#pragma unroll
        for (int u = 0; u < KE4; ++u) {                            // 4 elements per ds_read_b128
            const float4 t = ((const float4*)Hs)[u * NT + tid];
            const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int e = 4 * u + v;
                if (e < KE) acc[e / CB] = fmaf(tv[v], vc[e % CB], acc[e / CB]);
            }
        }
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const float s = allsum_q(acc[r]);
            if (q == 0) hx[KR * p + r] = s;
        }
    };

    // ---- A x of the incoming state, H x
    {
        float acc[RB];
        prod_A(xin, acc);
#pragma unroll
        for (int ps = 0; ps * 8 < RB; ++ps)
            if (8 * ps + q < RB) zt64[RB * p + 8 * ps + q] = (double)pick(acc, ps);
        prod_H();
    }

    float rho_est = (a.mode == 2) ? (float)a.rho_in : (float)a.rhos[ri];     // reluqpth.py:211
    float pri = 0.f, dua = 0.f;
    bool converged = false;
    int iters = 0;
    const float tolT = (float)a.tol;
    const int kmax = (a.mode == 2) ? 0 : a.max_iter;

    // ---- compute_residuals (reluqpth.py:307-318) on the current state; needs hx = H x valid
    auto residuals = [&](float rho_carry, float& o_pri, float& o_dua) -> float {
        float v[7];
#pragma unroll
        for (int e = 0; e < 7; ++e) v[e] = 0.f;
#pragma unroll
        for (int ps = 0; ps * 8 < RB; ++ps)
            if (8 * ps + q < RB) {
                const int i = RB * p + 8 * ps + q;
                nu[i] = (float)lam64[i];
                v[0] = tmaxr(v[0], fabsf((float)(zt64[i] - z64[i])));
                v[1] = tmaxr(v[1], fabsf((float)zt64[i]));
                v[2] = tmaxr(v[2], fabsf((float)z64[i]));
            }
        __syncthreads();
        prod_At(nu);                                               // t3 = A' lam (wave partials)
        __syncthreads();
        if (tid < N) {
            float t3 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t3 += part[w * N + tid];
            v[3] = fabsf(hx[tid] + t3 + gT[tid]);
            v[4] = fabsf(hx[tid]);
            v[5] = fabsf(t3);
            v[6] = fabsf(gT[tid]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int e = 0; e < 7; ++e) v[e] = tmaxr(v[e], __shfl_xor(v[e], off, 64));
        if (lane == 0)
#pragma unroll
            for (int e = 0; e < 7; ++e) red[wave * 8 + e] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            float r = red[e];
#pragma unroll
            for (int w = 1; w < NW; ++w) r = tmaxr(r, red[w * 8 + e]);
            v[e] = r;
        }
        __syncthreads();
        o_pri = v[0];
        o_dua = v[3];
        const float num = v[0] / tmaxr(v[1], v[2]);
        const float den = v[3] / tmaxr(tmaxr(v[4], v[5]), v[6]);
        float est = rho_carry * sqrtf(num / den);
        if (est < (float)a.rho_min) est = (float)a.rho_min;         // torch.clamp: NaN stays NaN
        if (est > (float)a.rho_max) est = (float)a.rho_max;
        return est;
    };

    for (int k = 1; k <= kmax; ++k) {
        // P2: rows owned by this lane: p = A x - z ; lam_hat ; nu
#pragma unroll
        for (int ps = 0; ps * 8 < RB; ++ps)
            if (8 * ps + q < RB) {
                const int i = RB * p + 8 * ps + q;
                const double rv = (double)rv32[i];
                const double pr = zt64[i] - z64[i];
                const double lh = lam64[i] + rv * pr;
                lam64[i] = lh;
                nu[i] = (float)(lh + rv * pr);
            }
        __syncthreads();                                           // S1: nu (and hx) visible
        prod_At(nu);                                               // P3: wave partials of A' nu
        __syncthreads();                                           // S2
        if (tid < N) {                                             // P4: d = H x + g + A' nu
            float s = hx[tid] + gT[tid];
#pragma unroll
            for (int w = 0; w < NW; ++w) s += part[w * N + tid];
            dvec[tid] = s;
        }
        __syncthreads();                                           // S3
        {                                                          // P5: dx = -K d ; x += dx
            float vc[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) vc[c] = dvec[CB * q + c];
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CB; ++c) s = fmaf(kr[r][c], vc[c], s);
                s = allsum_q(s);
                if (q == 0) {
                    const int i = KR * p + r;
                    const float dx = -s;
                    const double xn = x64[i] + (double)dx;
                    x64[i] = xn;
                    dxv[i] = dx;
                    xin[i] = (float)xn;
                }
            }
        }
        __syncthreads();                                           // S4: dx, xin visible
        {                                                          // P6: A x += A dx ; z = clamp(..)
            float acc[RB];
            prod_A(dxv, acc);
#pragma unroll
            for (int ps = 0; ps * 8 < RB; ++ps)
                if (8 * ps + q < RB) {
                    const int i = RB * p + 8 * ps + q;
                    const double zt = zt64[i] + (double)pick(acc, ps);
                    zt64[i] = zt;
                    const double v = zt + lam64[i] * inv64[i];
                    double zn = v;                                  // torch.clamp: NaN stays NaN
                    if (v < (double)lT[i]) zn = (double)lT[i];
                    if (v > (double)uT[i]) zn = (double)uT[i];
                    z64[i] = zn;
                }
        }
        prod_H();                                                  // P1: H x of the new x
        iters = k;

        if (a.mode == 0 && (k % a.check_interval) == 0) {          // reluqpth.py:218 (Q3 fixed)
            const int ri_before = ri;
            rho_est = residuals(rho_est, pri, dua);                // :220 (Q4: carried estimate)
            if (rho_est > (float)a.rhos[ri] * tolT && ri < a.nrho - 1)          // :223
                ri += 1;
            else if (rho_est < (float)a.rhos[ri] / tolT && ri > 0)              // :226
                ri -= 1;
            if (a.info.trace && (k / a.check_interval) <= a.info.trace_cap && tid == 0) {
                double* tr = a.info.trace + ((size_t)b * a.info.trace_cap + (k / a.check_interval - 1)) * 4;
                tr[0] = (double)pri; tr[1] = (double)dua; tr[2] = (double)rho_est; tr[3] = (double)ri_before;
            }
            if (pri < (float)a.thr_p && dua < (float)a.thr_d) {    // :233
                converged = true;
                break;
            }
            if (ri != ri_before) {                                 // adaptive-rho "re-factor": table lookup
                load_K(ri);
                set_rho_rows(ri);
                __syncthreads();
            }
        }
    }

    if (a.mode == 1) {                                             // iterate-only: keep the state
        __syncthreads();
        for (int i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = x64[i];
        for (int i = tid; i < m; i += NT) {
            a.z[(size_t)b * m + i] = z64[i];
            a.lam[(size_t)b * m + i] = lam64[i];
        }
        return;
    }
    __syncthreads();
    if (!converged) rho_est = residuals(rho_est, pri, dua);        // :243 (Q11 fixed: fresh state)

    // objective 1/2 x'Hx + g'x (compute_J :320-322)
    double jp = 0.0;
    if (tid < N) jp = (double)(xin[tid] * (0.5f * hx[tid] + gT[tid]));
    for (int off = 32; off >= 1; off >>= 1) jp += __shfl_xor(jp, off, 64);
    if (lane == 0) redd[wave] = jp;
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
    if (a.out_x) for (int i = tid; i < n; i += NT) ((float*)a.out_x)[(size_t)b * n + i] = (float)x64[i];
    if (a.out_z) for (int i = tid; i < m; i += NT) ((float*)a.out_z)[(size_t)b * m + i] = (float)z64[i];
    if (a.out_lam) for (int i = tid; i < m; i += NT) ((float*)a.out_lam)[(size_t)b * m + i] = (float)lam64[i];
    if (tid == 0) {
        if (a.info.iter) a.info.iter[b] = converged ? iters : a.max_iter;
        if (a.info.status) a.info.status[b] = converged ? RQP_STATUS_SOLVED : RQP_STATUS_MAX_ITER;
        if (a.info.rho_ind) a.info.rho_ind[b] = ri;
        if (a.info.pri_res) a.info.pri_res[b] = (double)pri;
        if (a.info.dua_res) a.info.dua_res[b] = (double)dua;
        if (a.info.rho_estimate) a.info.rho_estimate[b] = (double)rho_est;
        if (a.info.obj_val) a.info.obj_val[b] = obj;
    }
    if (a.warm_starting) {                                         // state + rho index persist (:304)
        for (int i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = x64[i];
        for (int i = tid; i < m; i += NT) {
            a.z[(size_t)b * m + i] = z64[i];
            a.lam[(size_t)b * m + i] = lam64[i];
        }
        if (tid == 0) a.rho_ind[b] = ri;
    } else {                                                       // clear_primal_dual (:324-333)
        for (int i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = 0.0;
        for (int i = tid; i < m; i += NT) {
            a.z[(size_t)b * m + i] = 0.0;
            a.lam[(size_t)b * m + i] = 0.0;
        }
        if (tid == 0) a.rho_ind[b] = a.rho_ind0;
    }
}

// ---------------------------------------------------------------------------- packing
// Register/LDS images of the matrices, written once at setup so that the solve kernel's loads
// are lane-linear: Apack[mat][e][t], Kpack[mat][j][e][t] (e = r*CB + c), Hpack[mat][u][t][4].
template <class C>
__global__ void k_pack_resident(int n, int m, int ldn, int nrho, const float* __restrict__ A, const float* __restrict__ Ht,
                                const float* __restrict__ K, float* __restrict__ Apack, float* __restrict__ Kpack,
                                float* __restrict__ Hpack) {
    constexpr int RB = C::RB, CB = C::CB, KR = C::KR, NT = C::NT, AE = C::AE, KE = C::KE, KE4 = C::KE4;
    const int mat = blockIdx.y;
    const int t = threadIdx.x, q = t & 7, p = t >> 3;
    const float* Am = A + (size_t)mat * m * ldn;
    const float* Hm = Ht + (size_t)mat * n * ldn;
    if (blockIdx.x == 0) {
        float* Ap = Apack + (size_t)mat * AE * NT;
        for (int e = 0; e < AE; ++e) {
            const int r = RB * p + e / CB, c = CB * q + e % CB;
            Ap[(size_t)e * NT + t] = (r < m && c < n) ? Am[(size_t)r * ldn + c] : 0.f;
        }
        float* Hp = Hpack + (size_t)mat * KE4 * NT * 4;
        for (int e = 0; e < KE4 * 4; ++e) {
            const int r = KR * p + e / CB, c = CB * q + e % CB;
            // H[r][c] = Ht[c][r]
            Hp[((size_t)(e >> 2) * NT + t) * 4 + (e & 3)] = (e < KE && r < n && c < n) ? Hm[(size_t)c * ldn + r] : 0.f;
        }
    } else {
        const int j = blockIdx.x - 1;
        const float* Kj = K + ((size_t)mat * nrho + j) * n * ldn;
        float* Kp = Kpack + ((size_t)mat * nrho + j) * KE * NT;
        for (int e = 0; e < KE; ++e) {
            const int r = KR * p + e / CB, c = CB * q + e % CB;
            Kp[(size_t)e * NT + t] = (r < n && c < n) ? Kj[(size_t)r * ldn + c] : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------ host side
typedef ResCfg<10, 13, 4> CfgC2;     // n <= 104, m <= 320

bool rqp_resident_fits(const rqp_handle* h) {
    return h->esz == 4 && h->n <= CfgC2::N && h->n <= CfgC2::KN && h->m <= CfgC2::M;
}

size_t rqp_resident_pack_elems(const rqp_handle* h, size_t* a_elems, size_t* k_elems, size_t* h_elems) {
    *a_elems = (size_t)h->nmat * CfgC2::AE * CfgC2::NT;
    *k_elems = (size_t)h->nmat * h->nrho * CfgC2::KE * CfgC2::NT;
    *h_elems = (size_t)h->nmat * CfgC2::KE4 * CfgC2::NT * 4;
    return *a_elems + *k_elems + *h_elems;
}

hipError_t rqp_launch_pack_resident(const rqp_handle* h, hipStream_t s) {
    dim3 grid(1 + h->nrho, h->nmat);
    k_pack_resident<CfgC2><<<grid, CfgC2::NT, 0, s>>>(h->n, h->m, h->ldn, h->nrho, (const float*)h->A, (const float*)h->Ht,
                                                        (const float*)h->K, h->Apack, h->Kpack, h->Hpack);
    return hipGetLastError();
}

hipError_t rqp_launch_solve_resident(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    const size_t lds = CfgC2::lds_bytes();
    hipError_t e = hipFuncSetAttribute((const void*)k_admm_resident<CfgC2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (const char* dbg = getenv("RQP_DEBUG")) {
        if (dbg[0] == '1') {
            int nb = -1;
            hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_admm_resident<CfgC2>, CfgC2::NT, lds);
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, (const void*)k_admm_resident<CfgC2>);
            fprintf(stderr, "[rqp] k_admm_resident: blocks/CU=%d (err %d) lds=%zu B static_lds=%zu regs=%d scratch=%zu B\n", nb, (int)oe,
                    lds, (size_t)fa.sharedSizeBytes, fa.numRegs, (size_t)fa.localSizeBytes);
        }
    }
    k_admm_resident<CfgC2><<<h->B, CfgC2::NT, lds, s>>>(a, h->Apack, h->Kpack, h->Hpack);
    return hipGetLastError();
}
