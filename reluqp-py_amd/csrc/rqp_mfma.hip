// rqp_mfma.hip -- ADMM hot loop for batches that SHARE (H, A) (linear MPC): the batch is the N
// dimension of fp32 MFMA GEMMs (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains at the fp32 vector rate, one
// instruction = 1024 MACs instead of 64 -- the instruction overhead of the per-instance kernels disappears).
//
// One 256-thread workgroup (4 waves, one per SIMD, up to 512 VGPRs each) iterates a TILE of 16 instance SLOTS:
//   GEMM1  d  = [A; H']' [nu; x] + g     (n x 16)   K split over the 4 waves (k-steps dealt round-robin), partials meet in LDS
//   GEMM2  dx = -K_j d                   (n x 16)   K split; per-column choice of K_j (each instance has its own
//                                                   rho index): one masked pass per distinct index in the tile
//   GEMM3  A dx                          (m x 16)   M split: row tiles dealt in snake order; a wave owns the row state of its tiles
// The matrices are MFMA A-operands.  [A; H'] (GEMM1) and A (GEMM3) stay in registers for the whole solve: 225 per lane,
// in the accumulator half of the unified register file (AGPRs, pinned with empty asm constraints -- left alone hipcc
// keeps MFMA sources in arch VGPRs and spills them).  K_j blocks (25 per lane) sit in two tagged VGPR blocks that
// refill from L2.  All operands are read from lane-linear images packed at setup (k_pack_mfma), so every load is one
// coalesced 256 B row per wave; groups of all-zero operands (block-triangular MPC matrices, padding) are skipped
// (k_meta_mfma).  B-operands (nu, x, dx) and the K-split partials pass through LDS in [k][16] layout.  Per-instance
// state lives in registers in the MFMA D layout (lane = slot column, 4 rows per 16x16 tile): z, lam in float32 and the
// high word of A x; its low word (A x is a float-float pair -- the role float64 plays in the other kernels: only A x
// needs the extra bits), l, u and g wait in LDS.
// Large batches run a persistent grid: a slot whose instance exits at a check takes the next unsolved instance.
// Register-allocation notes (hipcc 7.2): rare-path sizes/offsets go through opaque copies, otherwise LICM hoists
// hundreds of loop-invariant addresses and predicates out of the solve loop and the kernel spills.
// Same recurrence, check logic and quirk dispositions as k_admm_generic (rqp_admm.hip); reference line citations there.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rqp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int NB_, int MBW_>
struct MfmaCfg {
    static constexpr int NB = NB_, MBW = MBW_;        // n tiles of 16 ; m tiles of 16 per wave
    static constexpr int NT = 256, NW = 4, TB = 16;
    static constexpr int MB = NW * MBW;
    static constexpr int NP = 16 * NB, MP = 16 * MB, KT = MP + NP;
    static constexpr int KS1 = KT / 4 / NW;           // k-steps of GEMM1 per wave
    static constexpr int KS2 = NB;                    // k-steps of GEMM2 per wave   (NP/4/NW)
    static constexpr int KS3 = NP / 4;                // k-steps of GEMM3 (every wave runs all of them)
    // Ownership (chosen so that block-triangular A -- condensed MPC -- spreads evenly over the waves and its all-zero
    // operand tiles can be skipped in groups):
    //   GEMM1 k-steps are dealt round-robin: local step s of wave w is global k-step NW s + w;
    //   GEMM3 row tiles are dealt in snake order: local tile tl of wave w is global tile NW tl + (tl odd ? NW-1-w : w).
    static constexpr int G1 = 5, NG1 = KS1 / G1;      // GEMM1: k-steps per skip group; groups
    static constexpr int G3 = 5, NG3 = KS3 / G3;      // GEMM3
    static_assert(KS1 % G1 == 0 && KS3 % G3 == 0, "skip groups");
    // lane-linear operand images (floats): W1 [NW][KS1][NB][64] | W3 [NW][MBW][KS3][64] | K [nrho][NW][KS2][NB][64]
    // followed by the skip table (ints): g1_start [NW][NB] | g3_count [NW][MBW]
    static constexpr size_t W1_ELEMS = (size_t)NW * KS1 * NB * 64, W3_ELEMS = (size_t)NW * MBW * KS3 * 64;
    static constexpr size_t KJ_ELEMS = (size_t)NW * KS2 * NB * 64;
    static constexpr size_t META_ELEMS = (size_t)NW * NB + NW * MBW;
    __host__ __device__ static constexpr int tile_of(int w, int tl) { return NW * tl + ((tl & 1) ? NW - 1 - w : w); }
    static constexpr size_t lds_floats() {
        return (size_t)KT * 16 + NP * 16 + 2 * NW * NP * 16 + 3 * MP * 16 + NB * NT + NP * 16 + NW * 16 * 4 + 16 * 16 * 8 + 64 + 8 * 16;
    }
};

__device__ __forceinline__ float nanmaxf(float a, float b) {          // NaN-propagating max (torch semantics)
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}

// Swizzle of the [row][16] LDS arrays that are written in the MFMA D layout (lane = column, rows 16 T + 4 kq + r) and read
// in the MFMA B layout (lane kq reads row 4 S + kq) -- V1, part, part2: each group of 4 rows (64 floats = one pass over
// the banks) is rotated by 16 * (group & 3):
//      addr(row, col) = (row >> 2) * 64 + ((16 * (row & 3) + col + 16 * ((row >> 2) & 3)) & 63)
// which makes BOTH access patterns conflict-free (unrotated, the four kq of a D-layout write hit the same bank).
// The kernel forms these addresses as base + compile-time offset (dl_base / bl_base / co_base below).
//
// DIAG = true is a separate diagnostic build (RQP_DIAG=1): s_memtime stamps accumulate the cycles each wave spends per segment.
template <class C, bool DIAG>
__global__ void __launch_bounds__(256, 1) k_admm_mfma(SolveArgs a, const float* __restrict__ img, int* __restrict__ queue, unsigned long long* __restrict__ dbg) {
    constexpr int NB = C::NB, MBW = C::MBW, NT = C::NT, NW = C::NW, NP = C::NP, MP = C::MP, KT = C::KT;
    constexpr int KS1 = C::KS1, KS2 = C::KS2, KS3 = C::KS3;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* V1 = sm;                          // [KT][16] (swizzled rows)  rows 0..MP-1: nu (lam / 0 at a check); rows MP..: x (0 at a check)
    float* V3 = V1 + KT * 16;                // [NP][16]  dx (x at the start)
    float* part = V3 + NP * 16;              // [NW][NP][16] wave partials of GEMM1 (swizzled rows)
    float* part2 = part + NW * NP * 16;      // [NW][NP][16] wave partials of GEMM2
    float* LB = part2 + NW * NP * 16;        // [MBW][NT][4] l of the lane's own rows (owner-only, 128-bit per lane and tile)
    float* UB = LB + MP * 16;                // [MBW][NT][4] u
    float* ZL = UB + MP * 16;                // [MBW][NT][4] low word of the float-float A x
    float* T3 = ZL + MP * 16;                // [NB][NT]    A' lam of the pending check
    float* GV = T3 + NB * NT;                // [NP][16] g
    float* red = GV + NP * 16;               // [NW][16][4] row-side maxima per (wave, instance)
    float* rr = red + NW * 16 * 4;           // [16 rowgroups][16][8] column-side maxima per (row group, instance)
    float* rhosf = rr + 16 * 16 * 8;         // [64] rho ladder
    float* inst = rhosf + 64;                // [8][16]: 0 rho_est, 1 settings, 2 instance id, 3 k at its start, 4 rho index,
                                             //          5 done, 6 column to (re)load, 7 id + 1 of an instance that just exited
    int* inst_i = (int*)inst;

    const int n = a.n, m = a.m;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i16 = lane & 15, kq = lane >> 4;       // MFMA lane coordinates
    const int cj = tid & 15, rg = tid >> 4;          // column-owner coordinates: instance cj, rows rg + 16 e
    // Columns are SLOTS: slot c of workgroup b starts with instance 16 b + c; with a queue (persistent grid, large batches)
    // a slot whose instance has exited takes the next unsolved instance at the same check, so no MFMA column idles while
    // there is work left and the tile does not wait for its slowest member.
    unsigned long long t_begin = 0;
    if constexpr (DIAG) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");
    const int kmax = a.max_iter;
    const bool refill = queue != nullptr && kmax > 0 && (kmax % a.check_interval) == 0;

    // ---- resident A-operands (images: see MfmaCfg) -------------------------------------------------------------------
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // Swizzled addresses as (per-phase base) + (compile-time offset).  The bases are recomputed from an opaque copy of
    // the lane id at the top of each phase (OPQ): a handful of VALU ops, instead of dozens of loop-invariant address
    // registers that LICM would otherwise keep alive across the whole solve loop.
#define OPQ(v) asm volatile("" : "+v"(v))
    // D-layout element (row 16 T + 4 kq + r, column i16) -> base(r) + 256 T      [T counted inside the array]
    auto dl_base = [&](int lp, int r) __attribute__((always_inline)) { return (lp >> 4) * 64 + ((lp + 16 * r) & 63); };
    // B-layout element (row 4 (S0 + s) + kq, column i16) -> base(s & 3) + 64 s   [S0 = first k-step of this wave]
    auto bl_base = [&](int lp, int S0, int c) __attribute__((always_inline)) { return S0 * 64 + ((lp + 16 * ((S0 + c) & 3)) & 63); };
    // column-owner element (row rg + 16 e, column cj), tp = thread id -> base + 256 e
    auto co_base = [&](int tp) __attribute__((always_inline)) {
        const int cjp = tp & 15, rgp = tp >> 4;
        return (rgp >> 2) * 64 + ((16 * (rgp & 3) + cjp + 16 * ((rgp >> 2) & 3)) & 63);
    };
    // (all loads of an image are issued before the first value is pinned: the pin is a use, and a use directly behind
    //  its load would serialise 225 L2 round trips -- 55 us of prologue, measured)
    float aw1[KS1][NB];              // GEMM1: S[4 (NW s + wave) + kq][16 t + i16],  S = [A (MP rows); H' (NP rows)]
    {
        const float* w1 = img + (size_t)wave_u * KS1 * NB * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
            for (int t = 0; t < NB; ++t) aw1[s][t] = w1[(s * NB + t) * 64];
    }
    float a3[MBW][KS3];              // GEMM3: A[16 tile_of(wave, tl) + i16][4 s + kq]
    {
        const float* w3 = img + C::W1_ELEMS + (size_t)wave_u * MBW * KS3 * 64 + lane;
#pragma unroll
        for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
            for (int s = 0; s < KS3; ++s) a3[tl][s] = w3[(tl * KS3 + s) * 64];
    }
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int t = 0; t < NB; ++t) asm volatile("" : "+a"(aw1[s][t]));   // resident MFMA A-operands live in the accumulator half (AGPRs)
#pragma unroll
    for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
        for (int s = 0; s < KS3; ++s) asm volatile("" : "+a"(a3[tl][s]));
    // GEMM2: K_j[16 t + i16][4 (KS2 wave + s) + kq] in two tagged VGPR blocks
    const float* kimg = img + C::W1_ELEMS + C::W3_ELEMS + (size_t)wave_u * KS2 * NB * 64;
    float kb0[KS2][NB], kb1[KS2][NB];
    int ktag0 = -1, ktag1 = -1;
    // skip table of this wave (uniform): first k-step group of GEMM1 with a non-zero operand per n tile; number of
    // leading k-step groups of GEMM3 to run per row tile
    int g1s[NB], g3c[MBW];
    {
        const int* meta = (const int*)(img + C::W1_ELEMS + C::W3_ELEMS + (size_t)a.nrho * C::KJ_ELEMS);
#pragma unroll
        for (int t = 0; t < NB; ++t) g1s[t] = __builtin_amdgcn_readfirstlane(meta[wave_u * NB + t]);
#pragma unroll
        for (int tl = 0; tl < MBW; ++tl) g3c[tl] = __builtin_amdgcn_readfirstlane(meta[NW * NB + wave_u * MBW + tl]);
    }

    // ---- per-instance scalars and vectors -----------------------------------------------------------------------
    for (int i = tid; i < a.nrho && i < 64; i += NT) rhosf[i] = (float)a.rhos[i];
    if (tid < 16) {
        // slot -> instance: SolveArgs.order groups the instances by the rho index they START at (every distinct index among a
        // tile's columns is one more K pass; warm-started batches arrive with their persisted indices)
        const int st = blockIdx.x * 16 + tid;
        const bool ok = st < a.B;
        const int s0 = ok ? st : blockIdx.x * 16;                  // padding columns mirror the tile's first instance (no extra K block)
        const int id = ok ? (a.order ? a.order[st] : st) : a.B;
        const int ri = a.rho_ind[a.order ? a.order[s0] : s0];
        inst_i[2 * 16 + tid] = id;
        inst_i[3 * 16 + tid] = 0;
        inst_i[4 * 16 + tid] = ri;
        inst_i[5 * 16 + tid] = ok ? 0 : 1;                         // padding columns start "done"
        inst_i[6 * 16 + tid] = 1;                                  // every column loads its instance
        inst_i[7 * 16 + tid] = 0;
        inst[0 * 16 + tid] = (float)a.rhos[ri];                    // rho_est = rhos[rho_ind]  (:211)
    }
    if (tid == 0) {                                       // float copies of the scalar settings, read back in the decision block
        inst[16 + 0] = (float)a.tol;
        inst[16 + 1] = (float)a.thr_p;
        inst[16 + 2] = (float)a.thr_d;
        inst[16 + 3] = (float)a.rho_min;
        inst[16 + 4] = (float)a.rho_max;
        inst[16 + 5] = (float)a.eps_rel;
    }
    // row state (wave w owns rows [16 MBW w, 16 MBW (w+1)); lane: slot i16, rows 16 T + 4 kq + r); the low word of A x lives in ZL
    float zh[MBW][4], zz[MBW][4], lm[MBW][4];
    unsigned eqmask = 0;                                  // bit (4 tl + r): equality row (rho * 1e3)
    float xr[NB];                                         // column state: x of rows rg + 16 e of slot cj (g in LDS)
#pragma unroll
    for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) zh[tl][r] = zz[tl][r] = lm[tl][r] = 0.f;
#pragma unroll
    for (int e = 0; e < NB; ++e) xr[e] = 0.f;
    int ri_l = 0;                                         // rho index of this lane's MFMA column
    float rho_ne = 1.f, rho_eq = 1.f, inv_ne = 1.f, inv_eq = 1.f;   // rho of this lane's instance (plain / equality rows), inverses
    auto set_rho = [&]() __attribute__((always_inline)) {
        rho_ne = rhosf[ri_l];
        rho_eq = rho_ne * 1e3f;
        inv_ne = 1.0f / rho_ne;
        inv_eq = 1.0f / rho_eq;
    };

    // lam_hat and nu of the next iteration from the current state (p = A x - z)
    auto make_nu = [&]() __attribute__((always_inline)) {
        int lp = lane;
        OPQ(lp);
        int nb[2][4];                                     // rows of tile_of(w, tl): base[tl & 1][r] + 1024 tl
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            nb[0][r] = 256 * wave_u + dl_base(lp, r);
            nb[1][r] = 256 * (NW - 1 - wave_u) + dl_base(lp, r);
        }
#pragma unroll
        for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                const float p = (zh[tl][r] - zz[tl][r]) + ZL[(tl * NT + tid) * 4 + r];
                const float lh = lm[tl][r] + rho * p;
                lm[tl][r] = lh;
                V1[nb[tl & 1][r] + 256 * NW * tl] = lh + rho * p;
            }
    };

    // The loop is a small state machine so that each GEMM has ONE call site (one copy of its operands' live ranges):
    //   ph 4  load the state of the flagged slots (all of them at the start; refilled ones later)
    //   ph 0  start: GEMM3 on x of the loaded slots (0 elsewhere) -> A x of the incoming state
    //   ph 1  iterate: GEMM1 -> d -> GEMM2 -> dx, x -> GEMM3 -> row update
    //   ph 2  check, part 1: GEMM1 on [lam; 0] -> t3 = A' lam
    //   ph 3  check, part 2: GEMM1 on [0; x]   -> t2 = H x ; residuals, rho moves, exits (compute_residuals :307-318)
    int ph = 4, k = 0, to_chk = a.check_interval;         // to_chk: iterations until k is a multiple of check_interval
    bool final_chk = false;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;                  // row-side maxima of the pending check

    unsigned long long t_last = 0, t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int seg) __attribute__((always_inline)) {
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (seg >= 0) t_acc[seg] += t - t_last;
            t_last = t;
        }
    };
    stamp(-1);
    if constexpr (DIAG) t_acc[3] = t_last - t_begin;      // prologue: operand images, settings

    while (true) {
        __syncthreads();
        if (ph == 4) {
            // (sizes and ids pass through opaque copies so that none of this rare path's predicates and addresses are hoisted
            //  out of the solve loop as live registers)
            int n_o = n, m_o = m, tp = tid;
            asm volatile("" : "+s"(n_o), "+s"(m_o), "+v"(tp));
            const int c_o = tp & 15, rg_o = tp >> 4, kq_o = (tp >> 4) & 3, wave_o = tp >> 6;
            const int id = inst_i[2 * 16 + c_o];
            const bool fresh = inst_i[6 * 16 + c_o] != 0, real = id < a.B;
            if (fresh) {
                unsigned em = 0;
                if ((m_o & 3) == 0) {                        // the lane's 4 rows of a tile are contiguous and 16 B aligned:
#pragma unroll                                               // one sector per (instance, tile, float array)
                    for (int tl = 0; tl < MBW; ++tl) {
                        const int row0 = 16 * C::tile_of(wave_o, tl) + 4 * kq_o;
                        const bool ok = real && row0 < m_o;
                        const size_t o = ok ? (size_t)id * m_o + row0 : 0;
                        const f32x4 l4 = *(const f32x4*)((const float*)a.l + o), u4 = *(const f32x4*)((const float*)a.u + o);
                        const f32x4 c4 = *(const f32x4*)((const float*)a.c + o);
                        const f64x2 z01 = *(const f64x2*)(a.z + o), z23 = *(const f64x2*)(a.z + o + 2);
                        const f64x2 y01 = *(const f64x2*)(a.lam + o), y23 = *(const f64x2*)(a.lam + o + 2);
                        const float zv[4] = {(float)z01[0], (float)z01[1], (float)z23[0], (float)z23[1]};
                        const float yv[4] = {(float)y01[0], (float)y01[1], (float)y23[0], (float)y23[1]};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            zh[tl][r] = 0.f;
                            ZL[(tl * NT + tp) * 4 + r] = 0.f;
                            zz[tl][r] = ok ? zv[r] : 0.f;
                            lm[tl][r] = ok ? yv[r] : 0.f;
                            LB[(tl * NT + tp) * 4 + r] = ok ? l4[r] : 0.f;
                            UB[(tl * NT + tp) * 4 + r] = ok ? u4[r] : 0.f;
                            if (ok && c4[r] > 1.f) em |= 1u << (4 * tl + r);
                        }
                    }
                } else {
#pragma unroll
                    for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * C::tile_of(wave_o, tl) + 4 * kq_o + r;
                            const bool ok = real && row < m_o;
                            const size_t o = (size_t)(real ? id : 0) * m_o + (row < m_o ? row : 0);
                            zh[tl][r] = 0.f;
                            ZL[(tl * NT + tp) * 4 + r] = 0.f;
                            zz[tl][r] = ok ? (float)a.z[o] : 0.f;
                            lm[tl][r] = ok ? (float)a.lam[o] : 0.f;
                            LB[(tl * NT + tp) * 4 + r] = ok ? ((const float*)a.l)[o] : 0.f;
                            UB[(tl * NT + tp) * 4 + r] = ok ? ((const float*)a.u)[o] : 0.f;
                            const float cv = (row < m_o) ? ((const float*)a.c)[o] : 1.f;
                            if (cv > 1.f) em |= 1u << (4 * tl + r);
                        }
                }
                eqmask = em;
            }
            const int xb = co_base(tp);
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int row = rg_o + 16 * e;
                if (fresh) {
                    const bool ok = real && row < n_o;
                    const size_t o = (size_t)(real ? id : 0) * n_o + (row < n_o ? row : 0);
                    xr[e] = ok ? (float)a.x[o] : 0.f;
                    GV[tp + 256 * e] = ok ? ((const float*)a.g)[o] : 0.f;      // [rg + 16 e][cj]
                }
                V3[tp + 256 * e] = fresh ? xr[e] : 0.f;          // GEMM3 on x: A x of the loaded slots, + 0 for the others
                V1[MP * 16 + xb + 256 * e] = xr[e];
            }
            __syncthreads();
            if (tid < 16) inst_i[6 * 16 + tid] = 0;
            ri_l = inst_i[4 * 16 + i16];
            set_rho();
            ph = 0;
            __syncthreads();
        }
        stamp(0);
        if (ph != 0) {                                   // ---------------- GEMM1: wave partial of S' V1
            float bv[KS1];                                           // all B operands first: no LDS latency inside the MFMA stream
            int lp = lane;
            OPQ(lp);
            {
                const int rb = wave_u * 64 + ((lp + 16 * wave_u) & 63);      // row 4 (NW s + w) + kq: group NW s + w, rotation w
#pragma unroll
                for (int s = 0; s < KS1; ++s) bv[s] = V1[rb + 64 * NW * s];
            }
            int pb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[r] = wave_u * NP * 16 + dl_base(lp, r);
#pragma unroll
            for (int t = 0; t < NB; ++t) {                            // one n tile after the other; its leading all-zero groups are skipped
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};              // (one accumulator live across the group branches: five of them
                int g0 = g1s[t];                                      //  with the branches in between spill)
                asm volatile("" : "+s"(g0));                          // opaque: the group tests stay scalar compares in place
#pragma unroll
                for (int g = 0; g < C::NG1; ++g) {
                    if (g >= g0) {
#pragma unroll
                        for (int s = g * C::G1; s < (g + 1) * C::G1; ++s) {
                            asm volatile("" ::"a"(aw1[s][t]));          // keep the operand in its AGPR: the MFMA reads it from there
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw1[s][t], bv[s], acc, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) part[pb[r] + 256 * t] = acc[r];
            }
            stamp(1);
            __syncthreads();
            stamp(2);
        }
        bool run_g3 = (ph == 0);
        if (ph == 1) {
            {                                                        // dx partial = K_j d, K_j chosen per column
                f32x4 sel[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) sel[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                float bv[KS2];                                       // B operands: d = g + sum of the 4 partials (fixed order)
                int lp = lane;
                OPQ(lp);
                {
                    int gb[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) gb[c] = bl_base(lp, KS2 * wave_u, c);
                    float gk[KS2], pk[KS2][NW];                      // all LDS operands first (one latency, not KS2 of them)
#pragma unroll
                    for (int s = 0; s < KS2; ++s) {
                        gk[s] = GV[(KS2 * wave_u + s) * 64 + lp];        // g[4 S + kq][i16]
#pragma unroll
                        for (int w = 0; w < NW; ++w) pk[s][w] = part[w * NP * 16 + gb[s & 3] + 64 * s];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < KS2; ++s) {
                        float d = gk[s];
#pragma unroll
                        for (int w = 0; w < NW; ++w) d += pk[s][w];
                        bv[s] = d;
                    }
                }
                unsigned long long todo = __ballot(lane < 16);       // one representative lane per instance column
                bool first_pass = true;
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    const int j = __builtin_amdgcn_readlane(ri_l, src);
                    if (j != ktag0 && j != ktag1) {                  // miss: refill a block from the L2 image of K_j
                        // block 0 is sticky while any column of the tile still uses its K; block 1 takes the rest
                        const bool b0_used = __ballot(ri_l == ktag0) != 0ull;
                        const bool to0 = ktag0 < 0 || (!b0_used && ktag1 >= 0);
                        unsigned loff = lane;
                        asm volatile("" : "+v"(loff));               // opaque: keeps this rare path's addresses out of the loop header
                        const float* kp = kimg + (size_t)j * C::KJ_ELEMS + loff;
                        if (to0) {
#pragma unroll
                            for (int s = 0; s < KS2; ++s)
#pragma unroll
                                for (int t = 0; t < NB; ++t) kb0[s][t] = kp[(s * NB + t) * 64];
                            ktag0 = j;
                        } else {
#pragma unroll
                            for (int s = 0; s < KS2; ++s)
#pragma unroll
                                for (int t = 0; t < NB; ++t) kb1[s][t] = kp[(s * NB + t) * 64];
                            ktag1 = j;
                        }
                    }
                    f32x4 acc[NB];
#pragma unroll
                    for (int t = 0; t < NB; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (j == ktag0) {
#pragma unroll
                        for (int s = 0; s < KS2; ++s)
#pragma unroll
                            for (int t = 0; t < NB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb0[s][t], bv[s], acc[t], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int s = 0; s < KS2; ++s)
#pragma unroll
                            for (int t = 0; t < NB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb1[s][t], bv[s], acc[t], 0, 0, 0);
                    }
                    if (first_pass) {                                // (usually the only one: every column shares one K_j)
#pragma unroll
                        for (int t = 0; t < NB; ++t) sel[t] = acc[t];
                        first_pass = false;
                    } else {
                        const bool mine = (ri_l == j);
#pragma unroll
                        for (int t = 0; t < NB; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) sel[t][r] = mine ? acc[t][r] : sel[t][r];
                    }
                    todo &= ~__ballot(lane < 16 && ri_l == j);
                }
#pragma unroll
                for (int t = 0; t < NB; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part2[wave_u * NP * 16 + dl_base(lp, r) + 256 * t] = sel[t][r];
            }
            stamp(5);
            __syncthreads();
            stamp(6);
            {
                int tp = tid;
                OPQ(tp);
                const int xb = co_base(tp);
                float pk[NB][NW];                                     // all partials first (one LDS latency, not NB of them)
#pragma unroll
                for (int e = 0; e < NB; ++e)
#pragma unroll
                    for (int w = 0; w < NW; ++w) pk[e][w] = part2[w * NP * 16 + xb + 256 * e];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < NB; ++e) {                        // dx = -K d ; x += dx
                    float kd = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) kd += pk[e][w];
                    const float dx = -kd;
                    xr[e] += dx;
                    V3[tp + 256 * e] = dx;                            // [rg + 16 e][cj]
                    V1[MP * 16 + xb + 256 * e] = xr[e];
                }
            }
            stamp(7);
            __syncthreads();
            stamp(8);
            if constexpr (DIAG) t_acc[11] += 1;
            run_g3 = true;
        }
        bool nu_done = false;
        if (run_g3) {                                    // ---------------- GEMM3: (A V3)[rows of this wave][16]
            // Row tiles one after the other: the row update of tile tl (VALU + LDS) runs in the shadow of tile tl+1's MFMAs.
            // start (ph 0): V3 = x, A x accumulates from 0, z and lam stay; otherwise A x += A dx, z = clamp(A x + lam_hat / rho)
            // and, when no check follows, lam_hat / nu of the next iteration right away (what make_nu does after a check).
            const bool upd = (ph == 1);
            const bool fin_next = upd && !refill && (k + 1 >= kmax) && (to_chk != 1);
            const bool with_nu = upd && to_chk != 1 && !fin_next;
            int lp = lane;
            OPQ(lp);
            int nb[2][4];                                 // rows of tile_of(w, tl): base[tl & 1][r] + 1024 tl
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nb[0][r] = 256 * wave_u + dl_base(lp, r);
                nb[1][r] = 256 * (NW - 1 - wave_u) + dl_base(lp, r);
            }
            float bv[KS3];
#pragma unroll
            for (int s = 0; s < KS3; ++s) bv[s] = V3[64 * s + lp];      // dx[4 s + kq][i16]
            // One row tile after the other: its LDS operands (A x low word, l, u) are requested first and arrive while the
            // tile's KS3-long MFMA chain runs.  (fp32 MFMA and VALU do not overlap on this hardware -- tools/mfma_rate.hip --
            // so the row update itself is simply kept short: uniform branches instead of per-row selects.)
#pragma unroll
            for (int tl = 0; tl < MBW; ++tl) {
                float pz[4], pl[4], pu[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pz[r] = ZL[(tl * NT + tid) * 4 + r];
                    pl[r] = LB[(tl * NT + tid) * 4 + r];         // (4 consecutive floats per lane: one ds_read_b128 per array)
                    pu[r] = UB[(tl * NT + tid) * 4 + r];
                }
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                int gc = g3c[tl];
                asm volatile("" : "+s"(gc));                  // opaque: the group tests stay scalar compares in place
#pragma unroll
                for (int g = 0; g < C::NG3; ++g) {           // trailing all-zero k-step groups of the tile are skipped
                    if (g < gc) {
#pragma unroll
                        for (int s = g * C::G3; s < (g + 1) * C::G3; ++s) {
                            asm volatile("" ::"a"(a3[tl][s]));  // keep the operand in its AGPR: the MFMA reads it from there
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[tl][s], bv[s], acc, 0, 0, 0);
                        }
                    }
                }
                {                                            // (start: upd false -- A x accumulates from 0, z stays)
                    float pp[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float adx = acc[r];            // float-float accumulation of A x (two-sum + renormalisation)
                        const float sgm = zh[tl][r] + adx;
                        const float bb = sgm - zh[tl][r];
                        const float err = (zh[tl][r] - (sgm - bb)) + (adx - bb);
                        const float lo = pz[r] + err;
                        const float hi = sgm + lo;
                        const float zlo = lo - (hi - sgm);
                        ZL[(tl * NT + tid) * 4 + r] = zlo;
                        zh[tl][r] = hi;
                        const bool eq = (eqmask >> (4 * tl + r)) & 1u;
                        const float v = hi + (zlo + lm[tl][r] * (eq ? inv_eq : inv_ne));
                        float zn = v;                        // torch.clamp: NaN stays NaN
                        if (v < pl[r]) zn = pl[r];
                        if (v > pu[r]) zn = pu[r];
                        zn = upd ? zn : zz[tl][r];
                        zz[tl][r] = zn;
                        pp[r] = (hi - zn) + zlo;             // p = A x - z of the new state
                    }
                    if (with_nu) {                           // lam_hat, nu of the next iteration
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                            const float lh = lm[tl][r] + rho * pp[r];
                            lm[tl][r] = lh;
                            V1[nb[tl & 1][r] + 256 * NW * tl] = lh + rho * pp[r];
                        }
                    }
                }
            }
            if (upd) {
                k += 1;
                to_chk -= 1;
            }
            nu_done = with_nu;
            stamp(9);
        }
        // ---------------------------------------------------------------------------------- what comes next
        if (ph == 0 || ph == 1) {
            final_chk = !refill && ((ph == 0) ? (kmax == 0) : (k >= kmax && to_chk != 0));
            const bool chk = (ph == 1 && to_chk == 0) || final_chk;                       // :218 (Q3 fixed) / :243
            if (to_chk == 0) to_chk = a.check_interval;
            if (!chk) {
                if (!nu_done) make_nu();                 // (only after the start pass; iterations fold it into the row update)
                ph = 1;
            } else {                                     // check part 1: V1 = [lam; 0], row-side maxima
                v0 = 0.f; v1 = 0.f; v2 = 0.f;
                int lp = lane, tp = tid;
                OPQ(lp);
                OPQ(tp);
                const int xb = co_base(tp);
#pragma unroll
                for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (Ruiz scaling: caller-space norms, SolveArgs.scE; shared matrices: one set of factors)
                        const int row = 16 * C::tile_of(wave_u, tl) + 4 * (lp >> 4) + r;
                        const float we = (a.scE && row < m) ? (float)(1.0 / a.scE[row]) : 1.f;
                        const float zlo = ZL[(tl * NT + tid) * 4 + r];
                        v0 = nanmaxf(v0, fabsf((zh[tl][r] - zz[tl][r]) + zlo) * we);
                        v1 = nanmaxf(v1, fabsf(zh[tl][r] + zlo) * we);
                        v2 = nanmaxf(v2, fabsf(zz[tl][r]) * we);
                        V1[256 * C::tile_of(wave_u, tl) + dl_base(lp, r)] = lm[tl][r];
                    }
#pragma unroll
                for (int e = 0; e < NB; ++e) V1[MP * 16 + xb + 256 * e] = 0.f;
                ph = 2;
            }
        } else if (ph == 2) {                            // t3 = A' lam ; then V1 = [0; x]
            int lp = lane, tp = tid;
            OPQ(lp);
            OPQ(tp);
            const int xb = co_base(tp);
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += part[w * NP * 16 + xb + 256 * e];
                T3[e * NT + tid] = t;
                V1[MP * 16 + xb + 256 * e] = xr[e];
            }
#pragma unroll
            for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
                for (int r = 0; r < 4; ++r) V1[256 * C::tile_of(wave_u, tl) + dl_base(lp, r)] = 0.f;
            ph = 3;
        } else {                                         // ph == 3: t2 = H x ; residuals and decisions
            v0 = nanmaxf(v0, __shfl_xor(v0, 16, 64)); v0 = nanmaxf(v0, __shfl_xor(v0, 32, 64));
            v1 = nanmaxf(v1, __shfl_xor(v1, 16, 64)); v1 = nanmaxf(v1, __shfl_xor(v1, 32, 64));
            v2 = nanmaxf(v2, __shfl_xor(v2, 16, 64)); v2 = nanmaxf(v2, __shfl_xor(v2, 32, 64));
            if (kq == 0) {
                red[(wave * 16 + i16) * 4 + 0] = v0;
                red[(wave * 16 + i16) * 4 + 1] = v1;
                red[(wave * 16 + i16) * 4 + 2] = v2;
            }
            float w3 = 0.f, w4 = 0.f, w5 = 0.f, w6 = 0.f, jp = 0.f;
            int tp3 = tid;
            OPQ(tp3);
            const int xb3 = co_base(tp3);
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int row = rg + 16 * e;
                float t2 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t2 += part[w * NP * 16 + xb3 + 256 * e];
                const float ge = GV[row * 16 + cj], t3 = T3[e * NT + tid];
                const float wd = (a.scD && row < n) ? (float)(1.0 / (a.scC[0] * a.scD[row])) : 1.f;
                w3 = nanmaxf(w3, fabsf(t2 + t3 + ge) * wd);
                w4 = nanmaxf(w4, fabsf(t2) * wd);
                w5 = nanmaxf(w5, fabsf(t3) * wd);
                w6 = nanmaxf(w6, fabsf(ge) * wd);
                jp += xr[e] * (0.5f * t2 + ge);                      // compute_J :320-322
            }
            rr[(rg * 16 + cj) * 8 + 3] = w3;
            rr[(rg * 16 + cj) * 8 + 4] = w4;
            rr[(rg * 16 + cj) * 8 + 5] = w5;
            rr[(rg * 16 + cj) * 8 + 6] = w6;
            rr[(rg * 16 + cj) * 8 + 7] = jp;
            __syncthreads();
            if (tid < 16) {                                          // one thread per instance decides
                int j = tid;
                asm volatile("" : "+v"(j));              // opaque: none of this block's addresses are hoisted out of the solve loop
                const float tolT = inst[16 + 0], thr_p = inst[16 + 1], thr_d = inst[16 + 2], rmin = inst[16 + 3], rmax = inst[16 + 4];
                float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f, q6 = 0.f, obj = 0.f;
                for (int w = 0; w < NW; ++w) {
                    q0 = nanmaxf(q0, red[(w * 16 + j) * 4 + 0]);
                    q1 = nanmaxf(q1, red[(w * 16 + j) * 4 + 1]);
                    q2 = nanmaxf(q2, red[(w * 16 + j) * 4 + 2]);
                }
                for (int g2 = 0; g2 < 16; ++g2) {
                    q3 = nanmaxf(q3, rr[(g2 * 16 + j) * 8 + 3]);
                    q4 = nanmaxf(q4, rr[(g2 * 16 + j) * 8 + 4]);
                    q5 = nanmaxf(q5, rr[(g2 * 16 + j) * 8 + 5]);
                    q6 = nanmaxf(q6, rr[(g2 * 16 + j) * 8 + 6]);
                    obj += rr[(g2 * 16 + j) * 8 + 7];
                }
                int newly = 0;
                const bool alive = inst_i[5 * 16 + j] == 0;
                float est = 0.f, num = 0.f, den = 1.f;
                int ri = 0, id = 0, kc = 0;
                bool conv = false, last = false;
                if (alive) {
                    num = q0 / nanmaxf(q1, q2);                                           // :315
                    den = q3 / nanmaxf(nanmaxf(q4, q5), q6);                              // :316
                    est = inst[0 * 16 + j] * sqrtf(num / den);                            // :317 (Q4: carried)
                    if (est < rmin) est = rmin;                                           // torch.clamp: NaN stays NaN
                    if (est > rmax) est = rmax;
                    ri = inst_i[4 * 16 + j];
                    const int ri_before = ri;
                    if (!final_chk) {
                        if (est > rhosf[ri] * tolT && ri < a.nrho - 1) ri += 1;           // :223
                        else if (est < rhosf[ri] / tolT && ri > 0) ri -= 1;               // :226
                    }
                    inst[0 * 16 + j] = est;
                    inst_i[4 * 16 + j] = ri;
                    id = inst_i[2 * 16 + j];
                    kc = k - inst_i[3 * 16 + j];                                          // this instance's own iteration count
                    const int chk_no = kc / a.check_interval;
                    if (!final_chk && a.info.trace && chk_no <= a.info.trace_cap) {
                        double* tr = a.info.trace + ((size_t)id * a.info.trace_cap + (chk_no - 1)) * 4;
                        tr[0] = (double)q0; tr[1] = (double)q3; tr[2] = (double)est; tr[3] = (double)ri_before;
                    }
                    const float er = inst[16 + 5];                                        // eps_rel (0: the reference's absolute test)
                    const float tp = er > 0.f ? thr_p + er * nanmaxf(q1, q2) : thr_p;
                    const float td = er > 0.f ? thr_d + er * nanmaxf(nanmaxf(q4, q5), q6) : thr_d;
                    conv = !final_chk && (q0 < tp && q3 < td);                            // :233
                    last = final_chk || kc >= kmax;                                        // :243 max-iter fallthrough
                }
                // straggler hand-off: few columns of the tile still iterate (and no queue refills them) -> they leave with
                // status CONTINUE and finish on the per-instance kernel; the tile ends here
                const bool still = alive && !conv && !last;
                const int nact = __popcll(__ballot(still));
                const bool hand = still && !refill && a.handoff_cols > 0 && nact <= a.handoff_cols && a.info.status != nullptr;
                if (alive && (conv || last || hand)) {
                        float est_out = est;
                        if (!conv && !final_chk && !hand) {   // max_iter is a multiple of check_interval: the reference runs
                            // compute_residuals once more on the same state (:243), compounding the estimate again
                            est_out = est * sqrtf(num / den);
                            if (est_out < rmin) est_out = rmin;
                            if (est_out > rmax) est_out = rmax;
                        }
                        newly = id + 1;
                        inst_i[5 * 16 + j] = 1;
                        const size_t bj = (size_t)id;
                        if (hand) {
                            a.info.status[bj] = RQP_STATUS_CONTINUE;
                            a.cont_iter[bj] = kc;
                            a.cont_rho[bj] = (double)est;
                        } else {
                        if (a.info.iter) a.info.iter[bj] = conv ? kc : a.max_iter;
                        if (a.info.status) a.info.status[bj] = conv ? RQP_STATUS_SOLVED : ((q0 != q0 || q3 != q3) ? RQP_STATUS_NAN : RQP_STATUS_MAX_ITER);
                        if (a.info.rho_ind) a.info.rho_ind[bj] = ri;
                        if (a.info.pri_res) a.info.pri_res[bj] = (double)q0;
                        if (a.info.dua_res) a.info.dua_res[bj] = (double)q3;
                        if (a.info.rho_estimate) a.info.rho_estimate[bj] = (double)est_out;
                        if (a.info.obj_val) a.info.obj_val[bj] = (double)obj;
                        }
                        a.rho_ind[bj] = (a.warm_starting || a.keep_state) ? ri : a.rho_ind0;
                        if (refill) {                    // this slot takes the next unsolved instance
                            const int nsl = (int)gridDim.x * 16 + atomicAdd(queue, 1);
                            if (nsl < a.B) {
                                const int nxt = a.order ? a.order[nsl] : nsl;
                                const int rn = a.rho_ind[nxt];
                                inst_i[2 * 16 + j] = nxt;
                                inst_i[3 * 16 + j] = k;
                                inst_i[4 * 16 + j] = rn;
                                inst_i[5 * 16 + j] = 0;
                                inst_i[6 * 16 + j] = 1;
                                inst[0 * 16 + j] = rhosf[rn];
                            }
                        }
                }
                inst_i[7 * 16 + j] = newly;
                // Columns without a live instance keep iterating on garbage; give them the rho index of a live column so that
                // they never cost a GEMM2 pass or a K refill of their own (their results are already out).
                const bool live = inst_i[5 * 16 + j] == 0;
                const int ri_now = inst_i[4 * 16 + j];
                const unsigned long long lm16 = __ballot(live);      // lanes 0..15 = the 16 columns
                if (lm16) {
                    const int ri_live = __shfl(ri_now, __ffsll((long long)lm16) - 1, 64);
                    if (!live) inst_i[4 * 16 + j] = ri_live;
                }
            }
            __syncthreads();
            // instances that just finished: x, z, lam out (update_results :278-305) and the persistent state
            // (sizes and offsets pass through opaque copies so that none of this rare path's predicates and addresses
            //  are hoisted out of the solve loop as live registers)
            int n_o = n, m_o = m, tid_o = tid;
            asm volatile("" : "+s"(n_o), "+s"(m_o), "+v"(tid_o));
            const int c_o = tid_o & 15, rg_o = tid_o >> 4, kq_o = (tid_o >> 4) & 3, wave_o = tid_o >> 6;
            const int oid = inst_i[7 * 16 + c_o] - 1;    // instance that just left this thread's column (-1: none)
            if (oid >= 0) {
#pragma unroll
                for (int e = 0; e < NB; ++e) {
                    const int row = rg_o + 16 * e;
                    if (row < n_o) {
                        const size_t o = (size_t)oid * n_o + row;
                        if (a.out_x) ((float*)a.out_x)[o] = xr[e];
                        a.x[o] = (a.warm_starting || a.keep_state) ? (double)xr[e] : 0.0;
                    }
                }
                if ((m_o & 3) == 0) {                        // 16 B / 32 B stores: the lane's 4 rows of a tile are contiguous
#pragma unroll
                    for (int tl = 0; tl < MBW; ++tl) {
                        const int row0 = 16 * C::tile_of(wave_o, tl) + 4 * kq_o;
                        if (row0 < m_o) {
                            const size_t o = (size_t)oid * m_o + row0;
                            const f32x4 z4 = {zz[tl][0], zz[tl][1], zz[tl][2], zz[tl][3]};
                            const f32x4 y4 = {lm[tl][0], lm[tl][1], lm[tl][2], lm[tl][3]};
                            if (a.out_z) *(f32x4*)((float*)a.out_z + o) = z4;
                            if (a.out_lam) *(f32x4*)((float*)a.out_lam + o) = y4;
                            const bool ws = (a.warm_starting || a.keep_state) != 0;
                            *(f64x2*)(a.z + o) = (f64x2){ws ? (double)z4[0] : 0.0, ws ? (double)z4[1] : 0.0};
                            *(f64x2*)(a.z + o + 2) = (f64x2){ws ? (double)z4[2] : 0.0, ws ? (double)z4[3] : 0.0};
                            *(f64x2*)(a.lam + o) = (f64x2){ws ? (double)y4[0] : 0.0, ws ? (double)y4[1] : 0.0};
                            *(f64x2*)(a.lam + o + 2) = (f64x2){ws ? (double)y4[2] : 0.0, ws ? (double)y4[3] : 0.0};
                        }
                    }
                } else {
#pragma unroll
                    for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * C::tile_of(wave_o, tl) + 4 * kq_o + r;
                            if (row < m_o) {
                                const size_t o = (size_t)oid * m_o + row;
                                if (a.out_z) ((float*)a.out_z)[o] = zz[tl][r];
                                if (a.out_lam) ((float*)a.out_lam)[o] = lm[tl][r];
                                a.z[o] = (a.warm_starting || a.keep_state) ? (double)zz[tl][r] : 0.0;
                                a.lam[o] = (a.warm_starting || a.keep_state) ? (double)lm[tl][r] : 0.0;
                            }
                        }
                }
            }
            ri_l = inst_i[4 * 16 + i16];
            set_rho();
            int nd = 0, nf = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                nd += inst_i[5 * 16 + j];
                nf += inst_i[6 * 16 + j];
            }
            if (nd == 16) break;                         // every slot has exited and the queue is empty
            stamp(-1);
            if (nf) {
                ph = 4;                                  // load the refilled slots, then A x of their x, then nu for everyone
            } else {
#pragma unroll
                for (int e = 0; e < NB; ++e) V1[MP * 16 + xb3 + 256 * e] = xr[e];    // x rows are already there; keep explicit
                make_nu();
                ph = 1;
            }
        }
        if (ph == 1) stamp(10); else stamp(-1);
    }
    if constexpr (DIAG) {
        stamp(-1);
        t_acc[4] = t_last - t_begin;                      // whole workgroup
        if (lane == 0)
            for (int e = 0; e < 12; ++e) dbg[((size_t)blockIdx.x * 4 + wave) * 12 + e] = t_acc[e];
    }
}

#undef OPQ

// ---------------------------------------------------------------------------- packing
// Lane-linear operand images (lane l of an MFMA A-operand holds element [l & 15][l >> 4] of its 16 x 4 tile):
//   W1[w][s][t][l] = S[4 (KS1 w + s) + (l >> 4)][16 t + (l & 15)],  S = [A (MP rows, zero padded); H' (NP rows)]
//   W3[w][tl][s][l] = A[16 (MBW w + tl) + (l & 15)][4 s + (l >> 4)]
//   K[j][w][s][t][l] = K_j[16 t + (l & 15)][4 (KS2 w + s) + (l >> 4)]
template <class C>
// Kscale != NULL (rqp_dims.tile_dtype = RQP_TILE_F16): K_j enters the image rounded to the fp16 tile of the resident kernel --
// (half)(K / scale_j) * scale_j with the power-of-two scale k_pack_res2 chose -- so the MFMA phase and the resident
// continuation of a solve see the SAME K.  The MFMA operands are float32 registers either way; K only preconditions.
__global__ void k_pack_mfma(int n, int m, int ldn, int nrho, const float* __restrict__ A, const float* __restrict__ Ht,
                            const float* __restrict__ K, float* __restrict__ img, const float* __restrict__ Kscale) {
    constexpr int NB = C::NB, MBW = C::MBW, MP = C::MP, KS1 = C::KS1, KS2 = C::KS2, KS3 = C::KS3, NW = C::NW;
    const size_t total = C::W1_ELEMS + C::W3_ELEMS + (size_t)nrho * C::KJ_ELEMS;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx & 63), i16 = l & 15, kq = l >> 4;
        float v = 0.f;
        if (idx < C::W1_ELEMS) {
            const int q = (int)(idx >> 6), t = q % NB, s = (q / NB) % KS1, w = q / (NB * KS1);
            const int k = 4 * (NW * s + w) + kq, i = 16 * t + i16;
            if (i < n) {
                if (k < MP) { if (k < m) v = A[(size_t)k * ldn + i]; }
                else if (k - MP < n) v = Ht[(size_t)(k - MP) * ldn + i];
            }
        } else if (idx < C::W1_ELEMS + C::W3_ELEMS) {
            const int q = (int)((idx - C::W1_ELEMS) >> 6), s = q % KS3, tl = (q / KS3) % MBW, w = q / (KS3 * MBW);
            const int r = 16 * C::tile_of(w, tl) + i16, c = 4 * s + kq;
            if (r < m && c < n) v = A[(size_t)r * ldn + c];
        } else {
            const size_t o = idx - C::W1_ELEMS - C::W3_ELEMS;
            const int j = (int)(o / C::KJ_ELEMS), q = (int)((o % C::KJ_ELEMS) >> 6), t = q % NB, s = (q / NB) % KS2, w = q / (NB * KS2);
            const int r = 16 * t + i16, c = 4 * (KS2 * w + s) + kq;
            if (r < n && c < n) v = K[((size_t)j * n + r) * ldn + c];
            if (Kscale) {
                const float sc = Kscale[j];
                v = (float)(_Float16)(v * (1.f / sc)) * sc;               // (sc is a power of two: both scalings are exact)
            }
        }
        img[idx] = v;
    }
}

// Skip table from the packed images (one thread per entry): g1_start[w][t] = first k-step group of GEMM1 whose operands
// for n tile t are not all zero (NG1 if none); g3_count[w][tl] = 1 + last k-step group of GEMM3 with a non-zero operand.
template <class C>
__global__ void k_meta_mfma(const float* __restrict__ img, int* __restrict__ meta) {
    constexpr int NB = C::NB, MBW = C::MBW, KS1 = C::KS1, KS3 = C::KS3, NW = C::NW;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < NW * NB) {
        const int w = e / NB, t = e % NB;
        int first = KS1;
        for (int s = KS1 - 1; s >= 0; --s)
            for (int l = 0; l < 64; ++l)
                if (img[((size_t)(w * KS1 + s) * NB + t) * 64 + l] != 0.f) first = s;
        meta[e] = first / C::G1;
    } else if (e < NW * NB + NW * MBW) {
        const int q = e - NW * NB, w = q / MBW, tl = q % MBW;
        int last = -1;
        for (int s = 0; s < KS3; ++s)
            for (int l = 0; l < 64; ++l)
                if (img[C::W1_ELEMS + ((size_t)(w * MBW + tl) * KS3 + s) * 64 + l] != 0.f) last = s;
        meta[e] = (last < 0) ? 0 : last / C::G3 + 1;
    }
}

// ------------------------------------------------------------------------------ host side
typedef MfmaCfg<5, 5> CfgM55;        // n <= 80, m <= 320   (linear MPC, N=20, nx=12, nu=4 condensed)

bool rqp_mfma_fits(const rqp_handle* h) {
    return h->esz == 4 && h->dims.shared_mats && h->n <= CfgM55::NP && h->m <= CfgM55::MP && h->nrho <= 64;
}

size_t rqp_mfma_img_elems(const rqp_handle* h) {
    return CfgM55::W1_ELEMS + CfgM55::W3_ELEMS + (size_t)h->nrho * CfgM55::KJ_ELEMS + CfgM55::META_ELEMS;
}

hipError_t rqp_launch_pack_mfma(const rqp_handle* h, hipStream_t s) {
    k_pack_mfma<CfgM55><<<256, 256, 0, s>>>(h->n, h->m, h->ldn, h->nrho, (const float*)h->A, (const float*)h->Ht, (const float*)h->K, h->W1img,
                                        h->dims.tile_dtype == RQP_TILE_F16 ? h->Kscale : nullptr);
    int* meta = (int*)(h->W1img + CfgM55::W1_ELEMS + CfgM55::W3_ELEMS + (size_t)h->nrho * CfgM55::KJ_ELEMS);
    k_meta_mfma<CfgM55><<<1, 64, 0, s>>>(h->W1img, meta);
    return hipGetLastError();
}
hipError_t rqp_prepare_mfma(const rqp_handle* h) {
    const size_t lds = CfgM55::lds_floats() * sizeof(float);
    hipError_t e = rqp_raise_lds_limit((const void*)k_admm_mfma<CfgM55, false>, (size_t)lds);
    if (e == hipSuccess && (h->debug & 2))
        e = rqp_raise_lds_limit((const void*)k_admm_mfma<CfgM55, true>, (size_t)lds);
    return e;
}
hipError_t rqp_launch_solve_mfma(const rqp_handle* h, const SolveArgs& a0, hipStream_t s) {
    const size_t lds = CfgM55::lds_floats() * sizeof(float);
    hipError_t e;
    const int tiles = (h->B + 15) / 16;
    // (slots could be grouped by starting rho index through SolveArgs.order as in rqp_mfmal.hip -- the kernel follows it -- but with
    //  K in registers a pass per distinct index is cheap here: the closed loop measured 1.4 % SLOWER with the extra sort launch)
    const SolveArgs& a = a0;
    // More tiles than CUs (one workgroup per CU: 150 KB of LDS, up to 512 registers per lane): persistent grid, slots
    // refill from a queue.  (The refill needs every exit to fall on a check, i.e. max_iter on the check grid.)
    const int ncu = h->ncu;
    int grid = tiles;
    int* queue = nullptr;
    if (tiles > ncu && h->queue && a.max_iter > 0 && a.check_interval > 0 && a.max_iter % a.check_interval == 0) {
        grid = ncu;
        queue = h->queue;
        e = hipMemsetAsync(queue, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    {
        if (h->debug & 2) {      // diagnostic build: per-segment cycle shares of the iteration (synchronous, debug only)
            unsigned long long* dbg = nullptr;
            const size_t cnt = (size_t)grid * 4 * 12;
            if (hipMalloc((void**)&dbg, cnt * 8) != hipSuccess) return hipErrorOutOfMemory;
            k_admm_mfma<CfgM55, true><<<grid, CfgM55::NT, lds, s>>>(a, h->W1img, queue, dbg);
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> hb(cnt);
            (void)hipMemcpy(hb.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
            (void)hipFree(dbg);
            static const char* names[11] = {"top wait", "GEMM1", "wait", "d", "wait", "GEMM2", "wait", "x", "wait", "GEMM3+rows", "next"};
            for (int w = 0; w < 4; ++w) {
                double tot[12] = {0};
                for (int t = 0; t < grid; ++t)
                    for (int e2 = 0; e2 < 12; ++e2) tot[e2] += (double)hb[((size_t)t * 4 + w) * 12 + e2];
                fprintf(stderr, "[rqp diag mfma] wave %d, %.1f iterations/workgroup, s_memtime ticks per iteration:", w, tot[11] / grid);
                for (int e2 = 0; e2 < 11; ++e2)
                    if (e2 != 3 && e2 != 4) fprintf(stderr, "  %s %.1f", names[e2], tot[e2] / tot[11]);
                double it = 0;
                for (int e2 = 0; e2 < 11; ++e2)
                    if (e2 != 3 && e2 != 4) it += tot[e2];
                fprintf(stderr, "\n[rqp diag mfma] wave %d per workgroup: total %.0f ticks = prologue %.0f + iterations %.0f + load/start/check/exit %.0f\n", w,
                        tot[4] / grid, tot[3] / grid, it / grid, (tot[4] - tot[3] - it) / grid);
            }
            return hipGetLastError();
        }
    }
    k_admm_mfma<CfgM55, false><<<grid, CfgM55::NT, lds, s>>>(a, h->W1img, queue, nullptr);
    return hipGetLastError();
}
