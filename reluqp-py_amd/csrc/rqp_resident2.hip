// rqp_resident2.hip -- RESIDENT ADMM kernel, second layout ("column block per wave").
// Same recurrence, checks and semantics as k_admm_generic (rqp_admm.hip; reference
// ReLU-QP-py/reluqp/reluqpth.py:201-249, statement in oracle/reluqp_oracle.py:forward_refine).
//
// One 256-thread workgroup = one QP; 2 workgroups per CU.  Wave w (of 4) owns the column block
// [CW*w, CW*w + CW) of A (CW = 26) for ALL rows, and the matching CW rows of K and H:
//
//   A' nu  : thread (pl = lane>>1, q = lane&1) holds rows RB*pl.., cols CW*w + CQ*q..  (RB x CQ = 10 x 13,
//            65 float2 VGPR pairs).  Reducing over pl stays INSIDE the wave (permlane32/16 swaps + DPP), so the
//            wave produces its 26 entries of d = H x + g + A' nu with no cross-wave step.
//   K d    : thread (rr = lane>>3, cc = lane&7) holds rows CW*w + KR*rr.., cols KC*cc..  (4 x 13); reduce over cc by
//            DPP; the wave ends up owning dx and x for its own 26 columns.
//   A dx   : needs only the wave's own 26 dx values (wave-local LDS hop, no barrier); the 4 per-wave partial row
//            sums meet in LDS and are added, in fixed order, by the row's owner thread (row i <-> thread i % 256).
//   H x    : H is symmetric, so d = H x + g + A' nu = [A; H]' [nu; x] + g: the H rows (4 per lane, from a lane-linear
//            LDS image) are simply appended to the A rows of the transposed product -- no separate H phase.
//
// Barriers per iteration: 3 (nu visible / d visible / partials + x visible) instead of 4 + a d-assembly phase in
// the first resident kernel; all FMAs are v_pk_fma_f32 on row pairs (the same register pairing serves A dx and A' nu).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "rqp_common.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// acc + float(half of k) * v in one v_fma_mix_f32 (op_sel picks the low / high half of the dword; float32 multiply-add).
// Inline asm: left to the compiler the conversion is hoisted out of the solve loop and K sits in registers as float32 again.
__device__ __forceinline__ float fma_mix_lo(h2 k, float v, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(k), "v"(v));
    return acc;
}
__device__ __forceinline__ float fma_mix_hi(h2 k, float v, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(k), "v"(v));
    return acc;
}
template <int CTRL>
__device__ __forceinline__ float dpp2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// hipcc (ROCm 7.2) miscompiles `r = __builtin_amdgcn_permlane32_swap(..); r.x + r.y` (both extracts read the
// first result): inline asm, with the 2 wait states a VALU-written operand needs before a permlane swap.
__device__ __forceinline__ float swsum32(float a, float b) {      // lower half: a.lo+a.hi ; upper half: b.lo+b.hi
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swsum16(float a, float b) {      // even rows: a.even+a.odd ; odd rows: b.even+b.odd
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float wave_max(float v) {              // max over the 64 lanes, every lane gets it (NaN-free inputs)
    v = fmaxf(v, dpp2<0xB1>(v));
    v = fmaxf(v, dpp2<0x4E>(v));
    v = fmaxf(v, dpp2<0x141>(v));
    v = fmaxf(v, dpp2<0x128>(v));
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = fmaxf(a, b);
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ unsigned wave_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);
    unsigned a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = a | b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a | b;
}
template <typename T>
__device__ __forceinline__ T tmax2(T a, T b) {                    // NaN-propagating max (torch semantics)
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}

template <int RB_, int CQ_, int KR_, int KC_>
struct Res2Cfg {
    static constexpr int RB = RB_, CQ = CQ_, KR = KR_, KC = KC_;
    static constexpr int NT = 256, NW = 4, NQ = 2, PL = 32;
    static constexpr int RP = RB / 2, KP = KR / 2;
    static constexpr int CW = NQ * CQ;           // columns per wave
    static constexpr int N = NW * CW;            // padded columns
    static constexpr int M = PL * RB;            // padded rows
    // vector slots: column j lives at SW*(j/CW) + j%CW.  Each wave uses 8*KR >= CW slots; the pitch SW adds 4 so that
    // the 8 column groups a wave reads together (SW*(cc>>1) + CQ*(cc&1)) fall in distinct LDS banks
    static constexpr int SW = 8 * KR + 4;
    static constexpr int ND = NW * SW;           // padded vector length
    static constexpr int AE2 = RP * CQ;          // A float2 pairs per thread
    static constexpr int KE2 = KP * KC;          // K / H float2 pairs per thread
    static constexpr int HR = 4, HP = HR / 2;    // H rows per row group in the transposed product (PL*HR >= N)
    static constexpr int HE2 = HP * CQ;          // H float2 pairs per thread
    static constexpr int HU = HE2 / 2;           // H 16-byte units per thread
    static constexpr int H1 = (CQ + 1) / 2, H2 = (H1 + 1) / 2;
    // workgroups per CU the register budget is shaped for (one wave per SIMD each): 2 for the big tile, up to 8 otherwise
    static constexpr int MATREGS = 2 * (RP * CQ + KP * KC);        // VGPRs pinned by A and K; ~90 more are working registers
    static constexpr int WAVES_PER_SIMD = (MATREGS > 100) ? 2 : ((MATREGS > 30) ? 3 : 4);
    static_assert(RB % 2 == 0 && KR % 2 == 0 && HP == 2 && HU == CQ && PL * HR >= N, "row pairs");
    static_assert(8 * KR >= CW && 8 * KC >= N && M <= 2 * NT && KC == CQ && NQ == 2, "tile shape");
    static constexpr size_t lds_bytes() {
        return (size_t)M * 8 * 4 + ND * 8 + 16 * 8       // zt64 lam64 z64 inv64 | x64 | redd
               + (size_t)M * 4 * 4                       // lT uT rv32 nu
               + (size_t)ND * 4 * 6                      // xin xnat dxv hx gT dvec
               + (size_t)NW * M * 4 + 64 * 4             // part, red
               + (size_t)HU * NT * 16;                   // Hs
    }
};

// DIAG = true is a separate diagnostic build (RQP_DIAG=1): s_memtime stamps accumulate the cycles wave 0 spends
// in each segment of the iteration into `dbg` (never read by the kernel; never timed as the product).
// KH = true: the K(rho) tile is stored as fp16 row pairs (rqp_dims.tile_dtype = RQP_TILE_F16, BASELINE config 5): half the
// registers and half the reload bytes; products accumulate in float32 (v_fma_mix_f32 reads the half operand directly), the
// per-(matrix, rho) power-of-two scale Kscale keeps the entries inside the fp16 range.  K only preconditions dx = -K d.
template <class C, bool DIAG, bool KH, bool KD = false>
__global__ void __launch_bounds__(256, C::WAVES_PER_SIMD) k_admm_res2(SolveArgs a, const float* __restrict__ Apack,
                                                      const float* __restrict__ Kpack,
                                                      const float* __restrict__ Hpack, unsigned long long* dbg,
                                                      const float* __restrict__ Kscale) {
    constexpr int RB = C::RB, CQ = C::CQ, KR = C::KR, KC = C::KC, NT = C::NT, NW = C::NW, RP = C::RP, KP = C::KP;
    constexpr int CW = C::CW, M = C::M, ND = C::ND, SW = C::SW, AE2 = C::AE2, KE2 = C::KE2, HU = C::HU, HP = C::HP, HR = C::HR, H1 = C::H1, H2 = C::H2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* zt64 = (double*)smem_raw;                    // [M] A x        (row i owned by thread i % 256)
    double* lam64 = zt64 + M;
    double* z64 = lam64 + M;
    double* inv64 = z64 + M;                              // 1 / rho_i
    double* x64 = inv64 + M;                              // [ND]
    double* redd = x64 + ND;                              // [16]
    float* lT = (float*)(redd + 16);                      // [M]
    float* uT = lT + M;
    float* rv32 = uT + M;
    float* nu = rv32 + M;                                 // [M] nu (lam at a check)
    float* xin = nu + M;                                  // [ND] float(x), slot order (SW per wave)
    float* xnat = xin + ND;                               // [ND] float(x), natural order (rows of H)
    float* dxv = xnat + ND;                               // [ND] dx
    float* hx = dxv + ND;                                 // [ND] H x
    float* gT = hx + ND;                                  // [ND]
    float* dvec = gT + ND;                                // [ND] d (A' lam at a check)
    float* part = dvec + ND;                              // [NW][M] per-wave partial row sums of A dx
    float* red = part + NW * M;                           // [64]
    float* Hs = red + 64;                                 // [HU][NT][4]
    static_assert(((size_t)M * 8 * 4 + ND * 8 + 16 * 8 + (size_t)M * 4 * 4 + ND * 4 * 6 + NW * M * 4 + 64 * 4) % 16 == 0,
                  "H image must start 16-byte aligned");

    const int n = a.n, m = a.m;
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;   // dispatch order: longest solve first
    const int wave = tid >> 6, lane = tid & 63;
    const int q = lane & 1, pl = lane >> 1, rr = lane >> 3, cc = lane & 7;
    const size_t mat = (a.sA == 0) ? 0 : (size_t)b;
    const float* cg = (const float*)a.c + (size_t)b * m;
    // continue modes (SolveArgs.cont).  1: only the instances the MFMA kernel handed over, resumed at their iteration count
    // with A x recomputed.  2: only the instances that left their rho window (cstat = 1), resumed EXACTLY where they stopped.
    int k0 = 0;
    bool exact = false;                                            // resume behind the check of iteration k0 with A x = ax
    if (a.cont == 1) {
        if (a.info.status[b] != RQP_STATUS_CONTINUE) {             // (uniform per workgroup)
            if (!a.warm_starting) {                                // the first kernel kept every state for this pass: clear it
                for (int i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = 0.0;
                for (int i = tid; i < m; i += NT) {
                    a.z[(size_t)b * m + i] = 0.0;
                    a.lam[(size_t)b * m + i] = 0.0;
                }
                if (tid == 0) a.rho_ind[b] = a.rho_ind0;
            }
            return;
        }
        k0 = a.cont_iter[b];
    } else if (a.cont == 2) {
        if (a.cstat[b] == 0) return;
        const int ci = a.cont_iter[b];
        if (ci >= 0) {
            k0 = ci;
            exact = true;
        }
    }
    // rho window: K slot s of this matrix holds ladder index wb + s
    const int wb = a.wbase ? a.wbase[mat] : 0;
    if (a.cstat && a.mode == 0) {
        const int sl = a.rho_ind[b] - wb;
        if (sl < 0 || sl >= a.kwin) {                              // (uniform) the incoming index lies outside the window: leave
            if (tid == 0) {                                        // untouched; rqp_solve re-centres the window and restarts it
                a.cstat[b] = 1;
                if (!exact) a.cont_iter[b] = -1;
                atomicAdd(a.ncont, 1);
            }
            return;
        }
    }

    // ---- matrices: A, K_j -> VGPR pairs ; H -> LDS  (each element read from HBM once per solve)
    f2 ar[RP][CQ];
    {
        const f2* Ap = (const f2*)(Apack + mat * (size_t)AE2 * NT * 2) + tid;
#pragma unroll
        for (int rp = 0; rp < RP; ++rp)
#pragma unroll
            for (int c = 0; c < CQ; ++c) ar[rp][c] = Ap[(size_t)(rp * CQ + c) * NT];
        const float4* Hp = (const float4*)(Hpack + mat * (size_t)HU * NT * 4);
#pragma unroll
        for (int u = 0; u < HU; ++u) ((float4*)Hs)[u * NT + tid] = Hp[u * NT + tid];
    }
    int ri = a.rho_ind[b];
    float* rhoL = red + 32;                               // [32] the rho ladder: read at every check (LDS, not a dependent global load)
    if (tid < 32) rhoL[tid] = (tid < a.nrho) ? (float)a.rhos[tid] : 0.f;     // (nrho <= 32: res2_pick)
    typedef typename std::conditional<KH, h2, f2>::type kpair_t;      // a row pair of K: two floats, or two halves in one dword
    kpair_t kr[KP][KC];
    float kscale = 1.f;
    // KD (RQP_FLAG_LOW_MEMORY, float32 only): K straight from the row-major K(rho) table of the factor kernel (Kpack = NULL) --
    // lane (rr, cc) takes the row pairs of its rows, columns KC cc .. KC cc + KC - 1: the 8 lanes of a row group read one
    // whole row between them, so every cache line is fetched once.  No packed copy of the table (3.9 GB and 1.3 ms of setup
    // at B = 4096); the guarded 4-byte loads make a K load ~2x slower, +0.9 % per solve (A/B, tools/ab_bench.sh).
    auto load_K = [&](int jl) {
        int j = jl - wb;                                            // K slot (modes 1 / 2 of a windowed handle: rqp_iterate /
        j = j < 0 ? 0 : (j >= a.kwin ? a.kwin - 1 : j);             // rqp_compute_residuals re-centre the windows first)
        if constexpr (!KD) {
            const kpair_t* Kp = (const kpair_t*)Kpack + (mat * a.kwin + j) * (size_t)KE2 * NT + tid;
#pragma unroll
            for (int kp = 0; kp < KP; ++kp)
#pragma unroll
                for (int c = 0; c < KC; ++c) kr[kp][c] = Kp[(size_t)(kp * KC + c) * NT];
            if constexpr (KH) kscale = Kscale[mat * a.kwin + j];
        } else {
            // (lane-derived offsets through an opaque copy: otherwise the addresses and predicates of this rare path are
            //  hoisted out of the solve loop and cost it registers)
            int lane_o = lane, wave_o = wave;
            asm volatile("" : "+v"(lane_o), "+v"(wave_o));
            const int rr = lane_o >> 3, cc = lane_o & 7, wave = wave_o;
            const float* Kj = (const float*)a.K + ((a.sK == 0) ? (size_t)0 : (size_t)b * a.sK) + (size_t)j * n * a.ldn;     // (j: slot)
            // Unguarded vector loads (4-byte aligned float4: the rows start at multiples of ldn floats, the lane's run at KC cc):
            //  * columns >= n over-read into the next row (the table is allocated with a zeroed tail): those K entries meet
            //    d = 0 exactly -- the A and H images are zero-padded, so the padding columns of d are 0 -- and K is finite;
            //  * rows >= n (and the rows past the wave's CW) read row n - 1 instead and are multiplied by a 0 / 1 row mask.
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                const int lr = KR * rr + 2 * kp, r = CW * wave + lr;
                const bool ok0 = lr < CW && r < n, ok1 = lr + 1 < CW && r + 1 < n;
                const float* row0 = Kj + (size_t)(r < n ? r : n - 1) * a.ldn + KC * cc;
                const float* row1 = Kj + (size_t)(r + 1 < n ? r + 1 : n - 1) * a.ldn + KC * cc;
                float t0[KC], t1[KC];
#pragma unroll
                for (int c = 0; c + 4 <= KC; c += 4) {
                    const f4u u0 = *(const f4u*)(row0 + c), u1 = *(const f4u*)(row1 + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { t0[c + e] = u0[e]; t1[c + e] = u1[e]; }
                }
#pragma unroll
                for (int c = KC - KC % 4; c < KC; ++c) { t0[c] = row0[c]; t1[c] = row1[c]; }
                const f2 msk = {ok0 ? 1.f : 0.f, ok1 ? 1.f : 0.f};
#pragma unroll
                for (int c = 0; c < KC; ++c) kr[kp][c] = (f2){t0[c], t1[c]} * msk;
            }
        }
    };
    load_K(ri);

    // ---- vectors.  Row i is initialised, updated and re-scaled by the SAME thread (i % 256): no barriers needed
    auto set_rho_rows = [&](int j) {
        const float rho = (float)a.rhos[j];
        for (int i = tid; i < M; i += NT) {
            const float rv = rho * ((i < m) ? cg[i] : 1.f);
            rv32[i] = rv;
            inv64[i] = 1.0 / (double)rv;
        }
    };
    for (int i = tid; i < M; i += NT) {
        const bool in = i < m;
        zt64[i] = (exact && in) ? a.ax[(size_t)b * m + i] : 0.0;
        z64[i] = in ? a.z[(size_t)b * m + i] : 0.0;
        lam64[i] = in ? a.lam[(size_t)b * m + i] : 0.0;
        lT[i] = in ? ((const float*)a.l)[(size_t)b * m + i] : 0.f;
        uT[i] = in ? ((const float*)a.u)[(size_t)b * m + i] : 0.f;
        nu[i] = 0.f;
    }
    set_rho_rows(ri);
    for (int i = tid; i < ND; i += NT) {                  // slot i <-> column SW-block (i / SW), offset (i % SW)
        const int col = CW * (i / SW) + (i % SW);
        const bool in = (i % SW) < CW && col < n;
        const double xv = in ? a.x[(size_t)b * n + col] : 0.0;
        x64[i] = xv;
        xin[i] = (float)xv;
        xnat[i] = (i < n) ? (float)a.x[(size_t)b * n + i] : 0.f;
        gT[i] = in ? ((const float*)a.g)[(size_t)b * n + col] : 0.f;
        hx[i] = 0.f;
        dxv[i] = 0.f;
        dvec[i] = 0.f;
    }

    // where this lane deposits its share of the A' reduce-scatter: columns colw .. colw + ncolw - 1
    int colw, ncolw;
    {
        const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1;
        const int cbase = H2 * b4 + H1 * b5;
        int nval = b4 ? (H1 - H2) : H2;                   // entries that are not step-B duplicates
        if (b5 && cbase + nval > CQ) nval = CQ - cbase;   // ... nor step-A duplicates
        ncolw = ((lane & 0xE) == 0) ? nval : 0;           // one writer per (q, class): lane bits 1..3 == 0
        colw = SW * wave + CQ * q + cbase;
    }
    __syncthreads();

    // ---- products ------------------------------------------------------------------------------------
    // wave partial of A v over the wave's CW columns -> part[wave][row]   (v: this wave's CW entries)
    auto load_vc = [&](const float* v, float (&vc)[CQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CQ; ++c) vc[c] = v[SW * wave + CQ * q + c];
    };
    auto prod_A = [&](const float (&vc)[CQ]) __attribute__((always_inline)) {
        f2 acc[RP];
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            f2 s = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CQ; ++c) s = __builtin_elementwise_fma(ar[rp][c], (f2){vc[c], vc[c]}, s);
            s.x += dpp2<0xB1>(s.x);                                  // the other q (lane ^ 1)
            s.y += dpp2<0xB1>(s.y);
            acc[rp] = s;
        }
        if (q == 0) {
            f2* dst = (f2*)(part + wave * M + RB * pl);
#pragma unroll
            for (int rp = 0; rp < RP; ++rp) dst[rp] = acc[rp];
        }
    };
    // Transposed product for the wave's own columns, reduced over the wave's 32 row groups:
    //   USE_A: + sum_rows A[row][col] * w[row]          (w = nu, or lam at a check)
    //   USE_H: + sum_rows H[row][col] * xnat[row]       (= (H x)[col], H symmetric)
    //   out[slot(col)] = sum (+ gT when addg)
    auto prod_At = [&](auto use_a, auto use_h, const float* w, float* out, bool addg) {
        constexpr bool USE_A = decltype(use_a)::value, USE_H = decltype(use_h)::value;
        f2 wr[RP];
        f2 wx[HP];
        if constexpr (USE_A) {
#pragma unroll
            for (int rp = 0; rp < RP; ++rp) wr[rp] = ((const f2*)(w + RB * pl))[rp];
        }
        if constexpr (USE_H) {
            const float4 xv = *(const float4*)(xnat + HR * pl);        // x of this lane group's 4 rows of H
            wx[0] = (f2){xv.x, xv.y};
            wx[1] = (f2){xv.z, xv.w};
        }
        // column sum over this lane's rows: A rows from VGPRs first, then the 4 H rows of one ds_read_b128 that was
        // issued one column earlier (software pipeline of depth 1: its latency hides under the A-row FMAs)
        auto hload = [&](int c) -> float4 { return ((const float4*)Hs)[c * NT + tid]; };
        auto colsum = [&](int c, const float4& hv) -> float {
            f2 t2 = {0.f, 0.f};
            if constexpr (USE_A) {
#pragma unroll
                for (int rp = 0; rp < RP; ++rp) t2 = __builtin_elementwise_fma(ar[rp][c], wr[rp], t2);
            }
            if constexpr (USE_H) {
                t2 = __builtin_elementwise_fma((f2){hv.x, hv.y}, wx[0], t2);
                t2 = __builtin_elementwise_fma((f2){hv.z, hv.w}, wx[1], t2);
            }
            return t2.x + t2.y;
        };
        // reduce-scatter over the 32 row groups of the wave (lane bits 5,4 by swaps; 3,2,1 by DPP); columns are
        // produced in the order the first swap level consumes them (0, H1, 1, H1+1, ...), so few sums are live
        // The two columns of a swap pair are summed TOGETHER, their FMAs alternating: each column is one dependent chain, and
        // two dependent v_pk_fma_f32 back to back cost a hazard s_nop (4 issue cycles) -- the H tail of a lone column had two
        float s1[H1];
        auto colsum2 = [&](int c0, int c1, const float4& h0, const float4& h1, float& a0, float& a1) __attribute__((always_inline)) {
            f2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};
            if constexpr (USE_A) {
#pragma unroll
                for (int rp = 0; rp < RP; ++rp) {
                    t0 = __builtin_elementwise_fma(ar[rp][c0], wr[rp], t0);
                    t1 = __builtin_elementwise_fma(ar[rp][c1], wr[rp], t1);
                }
            }
            if constexpr (USE_H) {
                t0 = __builtin_elementwise_fma((f2){h0.x, h0.y}, wx[0], t0);
                t1 = __builtin_elementwise_fma((f2){h1.x, h1.y}, wx[0], t1);
                t0 = __builtin_elementwise_fma((f2){h0.z, h0.w}, wx[1], t0);
                t1 = __builtin_elementwise_fma((f2){h1.z, h1.w}, wx[1], t1);
            }
            a0 = t0.x + t0.y;
            a1 = t1.x + t1.y;
        };
        float4 hcur = {0.f, 0.f, 0.f, 0.f}, hnext = {0.f, 0.f, 0.f, 0.f};
        if constexpr (USE_H) {
            hcur = hload(0);
            if (H1 < CQ) hnext = hload(H1);
        }
#pragma unroll
        for (int i = 0; i < H1; ++i) {
            const bool pair = (i + H1 < CQ);
            if (pair) {
                float4 hc2 = hcur, hn2 = hnext;                          // next pair's H rows: requested before this pair's FMAs
                if constexpr (USE_H) {
                    if (i + 1 < H1) hc2 = hload(i + 1);
                    if (i + 1 + H1 < CQ) hn2 = hload(i + 1 + H1);
                }
                float a0, a1;
                colsum2(i, i + H1, hcur, hnext, a0, a1);
                s1[i] = swsum32(a0, a1);
                hcur = hc2;
                hnext = hn2;
            } else {
                const float a0 = colsum(i, hcur);
                s1[i] = swsum32(a0, a0);
            }
        }
        float gpre[H2];
#pragma unroll
        for (int i = 0; i < H2; ++i) gpre[i] = addg ? gT[colw + i] : 0.f;     // lands while the reduction runs
        float s2[H2];
#pragma unroll
        for (int i = 0; i < H2; ++i) s2[i] = (i + H2 < H1) ? swsum16(s1[i], s1[i + H2]) : swsum16(s1[i], s1[i]);
#pragma unroll
        for (int i = 0; i < H2; ++i) {
            float v = s2[i];
            v += dpp2<0x4E>(v);       // quad_perm [2,3,0,1]  (lane ^ 2)
            v += dpp2<0x124>(v);      // row_ror:4
            v += dpp2<0x128>(v);      // row_ror:8   -> sum over lane bits 2,3
            s2[i] = v;
        }
        if (ncolw > 0) {
#pragma unroll
            for (int i = 0; i < H2; ++i)
                if (i < ncolw) out[colw + i] = s2[i] + gpre[i];
        }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    // y[CW*wave + KR*rr + r] = sum_c Mat[..][KC*cc + c] * v[KC*cc + c] summed over cc; lanes cc == 0 get the sums
    auto prod_K = [&](const float* v, float (&s)[KR]) {
        float vc[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) vc[c] = v[SW * (cc >> 1) + CQ * (cc & 1) + c];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            f2 t = {0.f, 0.f};
            if constexpr (KH) {
#pragma unroll
                for (int c = 0; c < KC; ++c) {                       // v_fma_mix_f32: fp16 operand, float32 multiply-add
                    t.x = fma_mix_lo(kr[kp][c], vc[c], t.x);
                    t.y = fma_mix_hi(kr[kp][c], vc[c], t.y);
                }
                t.x *= kscale;
                t.y *= kscale;
            } else {
#pragma unroll
                for (int c = 0; c < KC; ++c) t = __builtin_elementwise_fma(kr[kp][c], (f2){vc[c], vc[c]}, t);
            }
            s[2 * kp] = t.x;
            s[2 * kp + 1] = t.y;
        }
        // (level by level over all rows: a DPP read of a register the previous VALU instruction wrote costs two wait states)
#pragma unroll
        for (int r = 0; r < KR; ++r) s[r] += dpp2<0xB1>(s[r]);
#pragma unroll
        for (int r = 0; r < KR; ++r) s[r] += dpp2<0x4E>(s[r]);
#pragma unroll
        for (int r = 0; r < KR; ++r) s[r] += dpp2<0x141>(s[r]);         // row_half_mirror
    };
    // rows owned by this thread: tid (all waves) and tid + 256 (wave 0 only)
    // do_a: A x += sum of the 4 wave partials ; z = clamp(A x + lam/rho)      (completes state k)
    // do_b: lam_hat, nu of the NEXT iteration                                  (advances lam)
    // Rows i = tid (+ NT for the lanes that own a second row: with M = NT + 64 that is exactly wave 0).  The two rows of a
    // lane are processed in ONE pass with their loads issued together and their (latency-bound, float64) chains
    // interleaved, instead of two passes back to back -- wave 0 was 520 cycles behind the others at the next barrier.
    auto row_body = [&](auto nr, bool init, bool do_a, bool do_b) __attribute__((always_inline)) {
        constexpr int R = decltype(nr)::value;
        // every LDS operand of the pass is requested up front (one latency, not one per dependent step: left to itself the
        // compiler interleaves reads and waits -- five serial LDS round trips in the one-row path), then the float64 chain
        double zt[R], z[R], lmv[R], iv[R];
        float adx[R][NW], lo[R], hi[R], rvf[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = tid + q * NT;
            zt[q] = zt64[i];
            lmv[q] = lam64[i];
            if (do_a) {
                iv[q] = inv64[i];
                lo[q] = lT[i];
                hi[q] = uT[i];
            } else {
                z[q] = z64[i];
            }
            rvf[q] = rv32[i];                                          // (unconditional: do_b is a run-time flag at one call site)
            if (init || do_a) {
#pragma unroll
                for (int w = 0; w < NW; ++w) adx[q][w] = part[w * M + i];
            }
        }
        // one wait for all of them; the empty-asm operand lists pin every request above it and every use below it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < R; ++q) {
            asm volatile("" : "+v"(zt[q]), "+v"(lmv[q]), "+v"(rvf[q]));
            if (do_a) asm volatile("" : "+v"(iv[q]), "+v"(lo[q]), "+v"(hi[q]));
            else asm volatile("" : "+v"(z[q]));
            if (init || do_a) asm volatile("" : "+v"(adx[q][0]), "+v"(adx[q][1]), "+v"(adx[q][2]), "+v"(adx[q][3]));
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = tid + q * NT;
            if (init || do_a) {
                const float ad = ((adx[q][0] + adx[q][1]) + adx[q][2]) + adx[q][3];
                zt[q] = (init ? 0.0 : zt[q]) + (double)ad;
                zt64[i] = zt[q];
            }
            if (do_a) {
                const double v = zt[q] + lmv[q] * iv[q];
                z[q] = v;                                              // torch.clamp: NaN stays NaN
                if (v < (double)lo[q]) z[q] = (double)lo[q];
                if (v > (double)hi[q]) z[q] = (double)hi[q];
                z64[i] = z[q];
            }
            if (do_b) {
                const double rv = (double)rvf[q];
                const double pr = zt[q] - z[q];
                const double lh = lmv[q] + rv * pr;
                lam64[i] = lh;
                nu[i] = (float)(lh + rv * pr);
            }
        }
    };
    auto row_pass = [&](bool init, bool do_a, bool do_b) __attribute__((always_inline)) {
        if constexpr (M > NT) {
            static_assert(M <= 2 * NT, "at most two rows per lane");
            if (tid + NT < M) {
                row_body(std::integral_constant<int, 2>{}, init, do_a, do_b);
                return;
            }
        }
        if (tid < M) row_body(std::integral_constant<int, 1>{}, init, do_a, do_b);
    };

    __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): A, K and the vectors have landed (see the rho move)
    unsigned long long t_last = 0, t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int seg) {
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (seg >= 0) t_acc[seg] += t - t_last;
            t_last = t;
        }
    };

    float rho_est = (a.mode == 2) ? (float)a.rho_in : ((a.cont == 1 || exact) ? (float)a.cont_rho[b] : (float)a.rhos[ri]);   // reluqpth.py:211
    float pri = 0.f, dua = 0.f;
    bool converged = false;
    int iters = k0;
    const float tolT = (float)a.tol;
    const int kmax = (a.mode == 2) ? 0 : a.max_iter;

    // ---- A x of the incoming state (an exact continuation brought it along: only lam_hat / nu of the next iteration are due)
    if (!exact) {
        float vc0[CQ];
        load_vc(xin, vc0);
        prod_A(vc0);
        __syncthreads();
        row_pass(true, false, kmax > k0);
    } else {
        row_pass(false, false, kmax > k0);
    }

    // ---- compute_residuals (reluqpth.py:307-318) on the current state (hx = H x valid)
    float scl_p = 0.f, scl_d = 0.f;                                    // residual scales of the last check (eps_rel)
    auto residuals = [&](float rho_carry, float& o_pri, float& o_dua) -> float {
        float v[7];
#pragma unroll
        for (int e = 0; e < 7; ++e) v[e] = 0.f;
        // (Ruiz scaling: every term goes back to the caller's space before its norm -- SolveArgs.scE; the weights are read
        //  here, at the check, from an opaque copy of tid: no registers of the solve loop)
        int tid_w = tid;
        asm volatile("" : "+v"(tid_w));
        for (int i = tid_w; i < M; i += NT) {
            const float we = (a.scE && i < m) ? (float)(1.0 / a.scE[mat * m + i]) : 1.f;
            nu[i] = (float)lam64[i];
            v[0] = tmax2(v[0], fabsf((float)(zt64[i] - z64[i])) * we);
            v[1] = tmax2(v[1], fabsf((float)zt64[i]) * we);
            v[2] = tmax2(v[2], fabsf((float)z64[i]) * we);
        }
        __syncthreads();
        if (tid < ND && (tid % SW) >= CW) {                              // padding slots: exact zeros
            dxv[tid] = 0.f;
            hx[tid] = 0.f;
        }
        prod_At(YES, NO, nu, dxv, false);                              // t3 = A' lam  (dxv is dead here: scratch)
        prod_At(NO, YES, nu, hx, false);                               // t2 = H x
        __syncthreads();
        if (tid < ND) {                                                 // padding slots hold zeros
            float wd = 1.f;
            if (a.scD) {
                const int col = CW * (tid_w / SW) + (tid_w % SW);
                if ((tid_w % SW) < CW && col < n) wd = (float)(1.0 / (a.scC[mat] * a.scD[mat * n + col]));
            }
            const float t3 = dxv[tid];
            v[3] = fabsf(hx[tid] + t3 + gT[tid]) * wd;
            v[4] = fabsf(hx[tid]) * wd;
            v[5] = fabsf(t3) * wd;
            v[6] = fabsf(gT[tid]) * wd;
        }
        // wave max of 7 values + NaN mask (torch max/norm propagate NaN): v_max (IEEE maxNum) on DPP / permlane swaps
        unsigned nanm = 0;
#pragma unroll
        for (int e = 0; e < 7; ++e) nanm |= (v[e] != v[e]) ? (1u << e) : 0u;
        float nm = __builtin_bit_cast(float, nanm);                    // bit pattern; OR-reduced with integer ops below
#pragma unroll
        for (int e = 0; e < 7; ++e) v[e] = wave_max(v[e]);
        nanm = wave_or(__builtin_bit_cast(unsigned, nm));
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < 7; ++e) red[wave * 8 + e] = v[e];
            red[wave * 8 + 7] = __builtin_bit_cast(float, nanm);
        }
        __syncthreads();
        nanm = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) nanm |= __builtin_bit_cast(unsigned, red[w * 8 + 7]);
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            float r = red[e];
#pragma unroll
            for (int w = 1; w < NW; ++w) r = fmaxf(r, red[w * 8 + e]);
            v[e] = ((nanm >> e) & 1u) ? __builtin_nanf("") : r;
        }
        __syncthreads();
        o_pri = v[0];
        o_dua = v[3];
        scl_p = tmax2(v[1], v[2]);
        scl_d = tmax2(tmax2(v[4], v[5]), v[6]);
        const float num = v[0] / scl_p;
        const float den = v[3] / scl_d;
        float est = rho_carry * sqrtf(num / den);
        if (est < (float)a.rho_min) est = (float)a.rho_min;             // torch.clamp: NaN stays NaN
        if (est > (float)a.rho_max) est = (float)a.rho_max;
        return est;
    };

    stamp(-1);
    int to_chk = a.check_interval - (k0 % a.check_interval);           // iterations until k % check_interval == 0
    for (int k = k0 + 1; k <= kmax; ++k) {
        __syncthreads();                                               // B3: nu (and hg) visible
        stamp(0);
        prod_At(YES, YES, nu, dvec, true);                             // d = H x + g + A' nu   (own columns)
        stamp(1);
        __syncthreads();                                               // B1: d visible
        stamp(2);
        {
            float s[KR];
            static_assert(KR == 4 || KR == 2, "vectorised x update");
            const int j = SW * wave + KR * rr;
            double2* xp = (double2*)(x64 + j);
            double2 xold[KR / 2];                                      // x of this lane's slots: requested before the products
#pragma unroll
            for (int hlf = 0; hlf < KR / 2; ++hlf) xold[hlf] = xp[hlf];
            prod_K(dvec, s);                                           // K d
            // dx goes to LDS first and the A dx operands are requested right behind it: the float64 x update below runs
            // while that same-wave LDS hop is in flight (LDS operations of a wave execute in order: behind the x stores the
            // reads would wait for them too)
            if (cc == 0) {                 // this lane owns slots j..j+3 (rows >= CW of the group: zero rows of K, padding slots)
#pragma unroll
                for (int hlf = 0; hlf < KR / 2; ++hlf) ((f2*)(dxv + j))[hlf] = (f2){-s[2 * hlf], -s[2 * hlf + 1]};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // dx of this wave's columns: same-wave LDS hop
            __builtin_amdgcn_wave_barrier();
            float vc[CQ];
            load_vc(dxv, vc);
            if (cc == 0) {
                f2* xn2 = (f2*)(xnat + CW * wave + KR * rr);                   // natural order: real columns only
#pragma unroll
                for (int hlf = 0; hlf < KR / 2; ++hlf) {
                    double2 xa = xold[hlf];
                    const f2 dx = {-s[2 * hlf], -s[2 * hlf + 1]};
                    xa.x += (double)dx.x;
                    xa.y += (double)dx.y;
                    xp[hlf] = xa;
                    const f2 xf = {(float)xa.x, (float)xa.y};
                    ((f2*)(xin + j))[hlf] = xf;
                    if (KR * rr + 2 * hlf + 1 < CW) xn2[hlf] = xf;
                }
            }
            stamp(3);
            prod_A(vc);                                                // partial A dx
        }
        stamp(4);
        __syncthreads();                                               // B2: partials and x visible
        stamp(5);
        stamp(6);
        iters = k;
        const bool on_grid = (--to_chk == 0);
        if (on_grid) to_chk = a.check_interval;
        const bool check = (a.mode == 0) && on_grid;                   // k % check_interval == 0: reluqpth.py:218 (Q3 fixed)
        if (!check) {
            row_pass(false, true, k < kmax);
            stamp(7);
        } else {
            row_pass(false, true, false);
            const int ri_before = ri;
            rho_est = residuals(rho_est, pri, dua);                    // :220 (Q4: carried estimate)
            rho_est = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rho_est)));
            pri = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pri)));
            dua = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dua)));
            if (rho_est > rhoL[ri] * tolT && ri < a.nrho - 1)                   // :223
                ri += 1;
            else if (rho_est < rhoL[ri] / tolT && ri > 0)                       // :226
                ri -= 1;
            if (a.info.trace && (k / a.check_interval) <= a.info.trace_cap && tid == 0) {
                double* tr = a.info.trace + ((size_t)b * a.info.trace_cap + (k / a.check_interval - 1)) * 4;
                tr[0] = (double)pri; tr[1] = (double)dua; tr[2] = (double)rho_est; tr[3] = (double)ri_before;
            }
            float tp = (float)a.thr_p, td = (float)a.thr_d;            // :233 (+ relative term when eps_rel > 0, 8(f)-3)
            if (a.eps_rel > 0) {
                tp += (float)a.eps_rel * __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scl_p)));
                td += (float)a.eps_rel * __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scl_d)));
            }
            if (pri < tp && dua < td) {
                converged = true;
                break;
            }
            if (ri != ri_before) {                                     // adaptive-rho "re-factor": table lookup
                if (a.cstat && k < kmax && (ri < wb || ri >= wb + a.kwin)) {
                    // the new index has no K in this instance's window: leave with the exact state; rqp_solve re-factors a
                    // window around ri and continues the instance behind this check (SolveArgs.cont = 2)
                    // (indices from an opaque copy of tid: this rare path must not cost the solve loop hoisted address registers)
                    int tid_o = tid;
                    asm volatile("" : "+v"(tid_o));
                    for (int i = tid_o; i < n; i += NT) a.x[(size_t)b * n + i] = x64[SW * (i / CW) + i % CW];
                    for (int i = tid_o; i < m; i += NT) {
                        a.z[(size_t)b * m + i] = z64[i];
                        a.lam[(size_t)b * m + i] = lam64[i];
                        a.ax[(size_t)b * m + i] = zt64[i];
                    }
                    if (tid_o == 0) {
                        a.rho_ind[b] = ri;
                        a.cont_iter[b] = k;
                        a.cont_rho[b] = (double)rho_est;
                        a.cstat[b] = 1;
                        atomicAdd(a.ncont, 1);
                    }
                    return;
                }
                load_K(ri);
                set_rho_rows(ri);
                // wait for the K loads HERE: left pending, the compiler guards every first use of a K register in the solve
                // loop with its own s_waitcnt vmcnt(N) -- 14 wait instructions per iteration that almost never wait
                __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
            }
            if (k < kmax) row_pass(false, false, true);
            stamp(8);
        }
    }
    if constexpr (DIAG) {
        if (lane == 0)
            for (int e = 0; e < 10; ++e) dbg[((size_t)b * 4 + wave) * 10 + e] = (e == 9) ? (unsigned long long)iters : t_acc[e];
    }

    __syncthreads();
    if (a.mode == 1) {                                                 // iterate-only: keep the state
        for (int i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = x64[SW * (i / CW) + i % CW];
        for (int i = tid; i < m; i += NT) {
            a.z[(size_t)b * m + i] = z64[i];
            a.lam[(size_t)b * m + i] = lam64[i];
        }
        return;
    }
    if (!converged) rho_est = residuals(rho_est, pri, dua);            // :243 (Q11 fixed: fresh state)

    // objective 1/2 x'Hx + g'x (compute_J :320-322)
    double jp = 0.0;
    if (tid < ND) jp = (double)(xin[tid] * (0.5f * hx[tid] + gT[tid]));
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
    if (a.out_x) for (int i = tid; i < n; i += NT) ((float*)a.out_x)[(size_t)b * n + i] = (float)x64[SW * (i / CW) + i % CW];
    if (a.out_z) for (int i = tid; i < m; i += NT) ((float*)a.out_z)[(size_t)b * m + i] = (float)z64[i];
    if (a.out_lam) for (int i = tid; i < m; i += NT) ((float*)a.out_lam)[(size_t)b * m + i] = (float)lam64[i];
    if (tid == 0) {
        if (a.cstat) a.cstat[b] = 0;
        if (a.info.iter) a.info.iter[b] = converged ? iters : a.max_iter;
        if (a.last_iter) a.last_iter[b] = converged ? iters : a.max_iter;
        if (a.info.status) a.info.status[b] = converged ? RQP_STATUS_SOLVED : ((pri != pri || dua != dua) ? RQP_STATUS_NAN : RQP_STATUS_MAX_ITER);
        if (a.info.rho_ind) a.info.rho_ind[b] = ri;
        if (a.info.pri_res) a.info.pri_res[b] = (double)pri;
        if (a.info.dua_res) a.info.dua_res[b] = (double)dua;
        if (a.info.rho_estimate) a.info.rho_estimate[b] = (double)rho_est;
        if (a.info.obj_val) a.info.obj_val[b] = obj;
    }
    if ((a.warm_starting || a.keep_state)) {                                             // state + rho index persist (:304)
        for (int i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = x64[SW * (i / CW) + i % CW];
        for (int i = tid; i < m; i += NT) {
            a.z[(size_t)b * m + i] = z64[i];
            a.lam[(size_t)b * m + i] = lam64[i];
        }
        if (tid == 0) a.rho_ind[b] = ri;
    } else {                                                           // clear_primal_dual (:324-333)
        for (int i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = 0.0;
        for (int i = tid; i < m; i += NT) {
            a.z[(size_t)b * m + i] = 0.0;
            a.lam[(size_t)b * m + i] = 0.0;
        }
        if (tid == 0) a.rho_ind[b] = a.rho_ind0;
    }
}

// ---------------------------------------------------------------------------- packing
// Lane-linear images written once at setup:
//   Apack[mat][pair = rp*CQ + c][t][2]      = A[RB*pl + 2rp + h][CW*w + CQ*q + c]          (k_pack_res2_ah)
//   Kpack[mat][j][pair = kp*KC + c][t][2]   = K_j[CW*w + KR*rr + 2kp + h][KC*cc + c]   (0 when KR*rr + 2kp + h >= CW)
//   Hpack[mat][c][t][4]                     = H[HR*pl + 0..3][CW*w + CQ*q + c]
//   KH: Kpack holds fp16 pairs (one dword per row pair) of K_j / Kscale[mat][j], Kscale = 2^e with max|K_j| / Kscale <= 2^14
template <class C, bool KH>
__global__ void k_pack_res2(int n, int ldn, int nrho, const float* __restrict__ K, float* __restrict__ Kpack, float* __restrict__ Kscale,
                            const int32_t* __restrict__ only) {
    constexpr int KR = C::KR, KC = C::KC, NT = C::NT, CW = C::CW, KE2 = C::KE2;
    const int mat = blockIdx.y;                                    // (nrho here = K slots per matrix, rqp_handle.kwin)
    if (only && !only[mat]) return;                                // re-pack of moved windows only
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int rr = lane >> 3, cc = lane & 7;
    // The K_j images from the row-major table (full-ladder, fp16-tile and non-windowed handles; a windowed float32 handle gets them
    // from the factor kernel: rqp_handle.kpack_direct).  The source passes through LDS: coalesced row reads in, the lane-linear
    // gather runs on LDS (the direct gather from global memory was 4-byte accesses 52 B or a whole row apart: 5 ms per 4096
    // instances); stage = n * ldn floats, 3 workgroups per CU.  The (A, H) images: k_pack_res2_ah below.
    extern __shared__ __attribute__((aligned(16))) float stage[];
    {
        const int j = blockIdx.x;                                  // K slot
        const float* Kj = K + ((size_t)mat * nrho + j) * n * ldn;
        for (int i = t; i < n * ldn; i += NT) stage[i] = Kj[i];
        __syncthreads();
        float inv_scale = 1.f;
        if constexpr (KH) {                                          // block max |K_j| -> power-of-two scale
            __shared__ float smax[NT];
            float mx = 0.f;
            for (int i = t; i < n * ldn; i += NT) mx = fmaxf(mx, fabsf(stage[i]));
            smax[t] = mx;
            __syncthreads();
            for (int off = NT / 2; off >= 1; off >>= 1) {
                if (t < off) smax[t] = fmaxf(smax[t], smax[t + off]);
                __syncthreads();
            }
            int e2 = 0;
            (void)frexpf(fmaxf(smax[0], 1e-30f), &e2);               // max = f * 2^e2, f in [0.5, 1)
            const float sc = ldexpf(1.f, e2 - 14);
            inv_scale = ldexpf(1.f, 14 - e2);
            if (t == 0) Kscale[(size_t)mat * nrho + j] = sc;
        }
        float* Kp = Kpack + ((size_t)mat * nrho + j) * KE2 * NT * (KH ? 1 : 2);
        for (int pair = 0; pair < KE2; ++pair) {
            const int lr = KR * rr + 2 * (pair / KC);
            const int r = CW * w + lr, c = KC * cc + pair % KC;
            f2 v;
            v.x = (lr < CW && r < n && c < n) ? stage[r * ldn + c] : 0.f;
            v.y = (lr + 1 < CW && r + 1 < n && c < n) ? stage[(r + 1) * ldn + c] : 0.f;
            if constexpr (KH)
                ((h2*)Kp)[(size_t)pair * NT + t] = (h2){(_Float16)(v.x * inv_scale), (_Float16)(v.y * inv_scale)};
            else
                ((f2*)Kp)[(size_t)pair * NT + t] = v;
        }
    }
}


// The (A, H) images, round 3 late: one workgroup per QUARTER of the row groups of A (pl in [8 ch, 8 ch + 8): rows [RB 8 ch, RB 8 (ch + 1)),
// i.e. lanes [16 ch, 16 ch + 16) of every wave of the solve kernel) plus one for H -- stages of M / 4 x ldn and n x ldn floats
// (32 / 40 KB on the big tile: 3-4 workgroups per CU) instead of one workgroup per matrix with the whole of A staged (125 KB, one
// per CU: 0.42 ms for the headline batch).  Thread t of an A workgroup = (part = t >> 6, w = (t >> 4) & 3, l16 = t & 15): pairs part,
// part + 4, ... of lane 16 ch + l16 of wave w; 128 contiguous bytes per 16 threads.
template <class C>
__global__ void k_pack_res2_ah(int n, int m, int ldn, const float* __restrict__ A, const float* __restrict__ Ht,
                               float* __restrict__ Apack, float* __restrict__ Hpack) {
    constexpr int RB = C::RB, CQ = C::CQ, NT = C::NT, CW = C::CW, AE2 = C::AE2, HU = C::HU, HR = C::HR, RCH = C::M / 4;
    const int mat = blockIdx.y, ch = blockIdx.x, t = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) float stage[];
    if (ch < 4) {
        if (!A) return;                                            // (rqp_update_mats with a new H only -- Apack stands)
        const int r0 = RCH * ch;
        const float* Am = A + (size_t)mat * m * ldn + (size_t)r0 * ldn;
        const int avail = (m - r0) * ldn;                          // floats of A from row r0 on (<= 0: a chunk of padding rows)
        if ((((size_t)Am) & 15) == 0) {                             // 16-byte loads (ldn % 4 == 0; 4-byte loads ran this kernel at 3 TB/s)
            for (int i = 4 * t; i < RCH * ldn; i += 4 * NT)
                *(float4*)(stage + i) = (i < avail) ? *(const float4*)(Am + i) : (float4){0.f, 0.f, 0.f, 0.f};     // (avail % 4 == 0)
        } else {
            for (int i = t; i < RCH * ldn; i += NT) stage[i] = (i < avail) ? Am[i] : 0.f;
        }
        __syncthreads();
        const int l16 = t & 15, w = (t >> 4) & 3, part = t >> 6;
        const int lane = 16 * ch + l16, q = lane & 1, pl = lane >> 1, tk = 64 * w + lane;
        f2* Ap = (f2*)(Apack + (size_t)mat * AE2 * NT * 2);
        for (int pair = part; pair < AE2; pair += 4) {
            const int lr = RB * pl + 2 * (pair / CQ) - r0, c = CW * w + CQ * q + pair % CQ;      // rows >= m hold zeros in the stage
            f2 v;
            v.x = (c < n) ? stage[lr * ldn + c] : 0.f;
            v.y = (c < n) ? stage[(lr + 1) * ldn + c] : 0.f;
            Ap[(size_t)pair * NT + tk] = v;
        }
    } else {
        const int w = t >> 6, lane = t & 63, q = lane & 1, pl = lane >> 1;
        const float* Hm = Ht + (size_t)mat * n * ldn;
        if ((((size_t)Hm) & 15) == 0) {
            for (int i = 4 * t; i < n * ldn; i += 4 * NT) *(float4*)(stage + i) = *(const float4*)(Hm + i);
        } else {
            for (int i = t; i < n * ldn; i += NT) stage[i] = Hm[i];
        }
        __syncthreads();
        float4* Hp = (float4*)(Hpack + (size_t)mat * HU * NT * 4);
        for (int c0 = 0; c0 < CQ; ++c0) {                              // unit c0 = (H[HR*pl + 0..3][col(c0)]) : one b128 per column
            const int c = CW * w + CQ * q + c0;
            float hv[HR];
            for (int h = 0; h < HR; ++h) {
                const int r = HR * pl + h;
                hv[h] = (r < n && c < n) ? stage[c * ldn + r] : 0.f;   // H[r][c] = Ht[c][r]
            }
            Hp[(size_t)c0 * NT + t] = (float4){hv[0], hv[1], hv[2], hv[3]};
        }
    }
}

// ------------------------------------------------------------------------------ host side
typedef Res2Cfg<10, 13, 4, 13> Cfg2C2;     // n <= 104, m <= 320   (BASELINE: n = 100, m = 300)
typedef Res2Cfg<2, 4, 2, 4> Cfg2C4;        // n <= 32,  m <= 64    (BASELINE config 4: n = 32, m = 64)
typedef Res2Cfg<4, 7, 2, 7> Cfg2M;         // n <= 56,  m <= 128
typedef Res2Cfg<10, 10, 4, 10> Cfg2N;      // n <= 80,  m <= 320   (condensed linear MPC, N = 20, nu = 4: n = 80, m = 320)

static int res2_pick(const rqp_handle* h) {            // smallest tile that holds the problem; -1: none
    if (h->esz != 4 || h->nrho > 32) return -1;
    if (h->n <= Cfg2C4::N && h->m <= Cfg2C4::M) return 0;
    if (h->n <= Cfg2M::N && h->m <= Cfg2M::M) return 1;
    if (h->n <= Cfg2N::N && h->m <= Cfg2N::M) return 3;
    if (h->n <= Cfg2C2::N && h->m <= Cfg2C2::M) return 2;
    return -1;
}

bool rqp_res2_fits(const rqp_handle* h) { return res2_pick(h) >= 0; }

template <class C>
static void pack_elems_t(const rqp_handle* h, size_t* a_elems, size_t* k_elems, size_t* h_elems) {
    *a_elems = (size_t)h->nmat * C::AE2 * C::NT * 2;
    *k_elems = h->k_direct ? 0 : (size_t)h->nmat * h->kwin * C::KE2 * C::NT * (h->dims.tile_dtype == RQP_TILE_F16 ? 1 : 2);
    *h_elems = (size_t)h->nmat * C::HU * C::NT * 4;
}
void rqp_res2_pack_elems(const rqp_handle* h, size_t* a_elems, size_t* k_elems, size_t* h_elems) {
    switch (res2_pick(h)) {
        case 0: pack_elems_t<Cfg2C4>(h, a_elems, k_elems, h_elems); break;
        case 1: pack_elems_t<Cfg2M>(h, a_elems, k_elems, h_elems); break;
        case 3: pack_elems_t<Cfg2N>(h, a_elems, k_elems, h_elems); break;
        default: pack_elems_t<Cfg2C2>(h, a_elems, k_elems, h_elems); break;
    }
}

void rqp_res2_kp_layout(const rqp_handle* h, int* cw, int* kr, int* kc) {
    switch (res2_pick(h)) {
        case 0: *cw = Cfg2C4::CW; *kr = Cfg2C4::KR; *kc = Cfg2C4::KC; break;
        case 1: *cw = Cfg2M::CW; *kr = Cfg2M::KR; *kc = Cfg2M::KC; break;
        case 3: *cw = Cfg2N::CW; *kr = Cfg2N::KR; *kc = Cfg2N::KC; break;
        default: *cw = Cfg2C2::CW; *kr = Cfg2C2::KR; *kc = Cfg2C2::KC; break;
    }
}

template <class C>
static hipError_t pack_t(const rqp_handle* h, const void* A_src, const int32_t* only, hipStream_t s) {
    const size_t stage_ah = (size_t)(C::M / 4 > h->n ? C::M / 4 : h->n) * h->ldn * sizeof(float);
    const size_t stage_k = (size_t)h->n * h->ldn * sizeof(float);
    // only != NULL: the K blocks of the matrices whose window moved (A and H have not changed)
    if (!only)
        k_pack_res2_ah<C><<<dim3(5, h->nmat), C::NT, stage_ah, s>>>(h->n, h->m, h->ldn, (const float*)A_src, (const float*)h->Ht, h->Apack, h->Hpack);
    if (h->dims.tile_dtype == RQP_TILE_F16) {
        k_pack_res2<C, true><<<dim3(h->kwin, h->nmat), C::NT, stage_k, s>>>(h->n, h->ldn, h->kwin, (const float*)h->K, h->Kpack, h->Kscale, only);
    } else {
        if (!h->k_direct && !h->kpack_direct)       // (low-memory handles read K from the row-major table; kpack_direct: the factor kernel wrote the image)
            k_pack_res2<C, false><<<dim3(h->kwin, h->nmat), C::NT, stage_k, s>>>(h->n, h->ldn, h->kwin, (const float*)h->K, h->Kpack, nullptr, only);
    }
    return hipGetLastError();
}
hipError_t rqp_launch_pack_res2(const rqp_handle* h, const void* A_src, const int32_t* only, hipStream_t s) {
    switch (res2_pick(h)) {
        case 0: return pack_t<Cfg2C4>(h, A_src, only, s);
        case 1: return pack_t<Cfg2M>(h, A_src, only, s);
        case 3: return pack_t<Cfg2N>(h, A_src, only, s);
        default: return pack_t<Cfg2C2>(h, A_src, only, s);
    }
}

template <class C>
static hipError_t prepare_t(const rqp_handle* h) {
    const size_t lds = C::lds_bytes();
    {
        const size_t stage_ah = (size_t)(C::M / 4 > h->n ? C::M / 4 : h->n) * h->ldn * sizeof(float);
        hipError_t ae = rqp_raise_lds_limit((const void*)k_pack_res2_ah<C>, stage_ah);
        if (ae != hipSuccess) return ae;
    }
    {   // K pack kernel: its LDS stage holds one K_j (n x ldn floats)
        const size_t stage = (size_t)h->n * h->ldn * sizeof(float);
        hipError_t pe = (h->dims.tile_dtype == RQP_TILE_F16)
                            ? rqp_raise_lds_limit((const void*)k_pack_res2<C, true>, (size_t)stage)
                            : rqp_raise_lds_limit((const void*)k_pack_res2<C, false>, (size_t)stage);
        if (pe != hipSuccess) return pe;
    }
    if (h->dims.tile_dtype == RQP_TILE_F16)
        return rqp_raise_lds_limit((const void*)k_admm_res2<C, false, true>, (size_t)lds);
    hipError_t e = rqp_raise_lds_limit((const void*)k_admm_res2<C, false, false>, (size_t)lds);
    if (e != hipSuccess) return e;
    if (h->k_direct) {
        e = rqp_raise_lds_limit((const void*)k_admm_res2<C, false, false, true>, (size_t)lds);
        if (e != hipSuccess) return e;
    }
    if (h->debug & 2) e = rqp_raise_lds_limit((const void*)k_admm_res2<C, true, false>, (size_t)lds);
    if (h->debug & 1) {
        int nb = -1;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_admm_res2<C, false, false>, C::NT, lds);
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, (const void*)k_admm_res2<C, false, false>);
        fprintf(stderr, "[rqp] k_admm_res2<%d,%d,%d,%d>: blocks/CU=%d (err %d) lds=%zu B regs=%d scratch=%zu B\n", C::RB, C::CQ,
                C::KR, C::KC, nb, (int)oe, lds, fa.numRegs, (size_t)fa.localSizeBytes);
    }
    return e;
}
hipError_t rqp_prepare_res2(const rqp_handle* h) {
    switch (res2_pick(h)) {
        case 0: return prepare_t<Cfg2C4>(h);
        case 1: return prepare_t<Cfg2M>(h);
        case 3: return prepare_t<Cfg2N>(h);
        default: return prepare_t<Cfg2C2>(h);
    }
}

template <class C>
static hipError_t solve_t(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    const size_t lds = C::lds_bytes();
    if (h->dims.tile_dtype == RQP_TILE_F16) {
        k_admm_res2<C, false, true><<<h->B, C::NT, lds, s>>>(a, h->Apack, h->Kpack, h->Hpack, nullptr, h->Kscale);
        return hipGetLastError();
    }
    if ((h->debug & 2) && !h->k_direct) {   // diagnostic build: per-segment cycle shares of the iteration (synchronous, debug only)
        unsigned long long* dbg = nullptr;
        const size_t cnt = (size_t)h->B * 4 * 10;
        if (hipMalloc((void**)&dbg, cnt * 8) != hipSuccess) return hipErrorOutOfMemory;
        k_admm_res2<C, true, false><<<h->B, C::NT, lds, s>>>(a, h->Apack, h->Kpack, h->Hpack, dbg, nullptr);
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> hbuf(cnt);
        (void)hipMemcpy(hbuf.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        static const char* names[9] = {"B3 wait", "A'nu+Hx", "B1 wait", "Kd+x", "A dx", "B2 wait", "-", "rows", "check"};
        for (int w = 0; w < 4; ++w) {
            double tot[9] = {0}, its = 0;
            for (int b = 0; b < h->B; ++b) {
                for (int e = 0; e < 9; ++e) tot[e] += (double)hbuf[((size_t)b * 4 + w) * 10 + e];
                its += (double)hbuf[((size_t)b * 4 + w) * 10 + 9];
            }
            fprintf(stderr, "[rqp diag] wave %d cycles/iteration:", w);
            double sum = 0;
            for (int e = 0; e < 9; ++e) { fprintf(stderr, " %s=%.0f", names[e], tot[e] / its); sum += tot[e] / its; }
            fprintf(stderr, " | total=%.0f\n", sum);
        }
        return hipGetLastError();
    }
    if (h->k_direct)
        k_admm_res2<C, false, false, true><<<h->B, C::NT, lds, s>>>(a, h->Apack, nullptr, h->Hpack, nullptr, nullptr);
    else
        k_admm_res2<C, false, false><<<h->B, C::NT, lds, s>>>(a, h->Apack, h->Kpack, h->Hpack, nullptr, nullptr);
    return hipGetLastError();
}
hipError_t rqp_launch_solve_res2(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    switch (res2_pick(h)) {
        case 0: return solve_t<Cfg2C4>(h, a, s);
        case 1: return solve_t<Cfg2M>(h, a, s);
        case 3: return solve_t<Cfg2N>(h, a, s);
        default: return solve_t<Cfg2C2>(h, a, s);
    }
}
