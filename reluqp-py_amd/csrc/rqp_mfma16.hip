// rqp_mfma16.hip -- the shared-(H, A) ADMM kernel of rqp_mfma.hip on the 16-BIT matrix pipe (rqp_dims.tile_dtype =
// RQP_TILE_BF16; BASELINE config 5 "16-bit tile, float32 residual", SURVEY.md 7.3 "MFMA f16").
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate: 64 FLOP/clk/SIMD, 1/16 of the bf16 rate
// (MI355X_MICROARCH.md, Matrix cores): k_admm_mfma spends 250 x 32 = 8 000 of its 15 000 cycles per tile-iteration in
// it.  Here every matrix operand is stored as TWO bf16 planes -- hi = bf16(v), mid = bf16(v - hi): 16 significant bits --
// and so is every vector operand; a product is three v_mfma_f32_16x16x32_bf16 (hi hi + hi mid + mid hi, float32
// accumulate; the dropped mid mid term is 2^-16 relative): per tile-iteration 3 x (65 + 15 + 60) MFMAs of 16 cycles
// instead of 250 + 75 of 32.  The recurrence, the checks and the state (float32, A x float-float) are EXACTLY those of
// k_admm_mfma: there is no increment form and no refresh, because 2^-16 operand error moves the fixed point of
// d = H x + g + A' nu by ~1.5e-5 |H| |x| -- two orders below the eps_abs = 1e-3 thresholds (stated tolerance:
// eps_abs >= 1e-5; measured on the config-3 batch: identical iteration counts at 1e-3 and 1e-5).  A single bf16 plane
// (8 bits) does NOT work in this role: the solve stalls or, for K with rho x 1e3 equality rows, diverges (oracle
// experiments); a plain fp16 plane needs per-instance dynamic scaling of nu (up to 1e5) -- bf16 planes need none.
//
// Layout differences from rqp_mfma.hip (everything else -- slots, persistent grid, refill queue, straggler hand-off,
// decision block -- is the same code):
//   * MFMA K-step = 32.  GEMM1 (K = MP + NP = 400 -> 13 steps): steps 0..11 dealt round-robin over the 4 waves (3 each,
//     all NB n-tiles); step 12 (half padding) is dealt BY TILE (tile t to wave t & 3).  GEMM2 (K = NP = 80 -> 3 steps): one step
//     each for waves 0..2.  GEMM3 (K = 80 -> 3 steps): every wave, its own MBW row tiles.
//   * A operand of one (tile, step) = 8 dwords per lane: [hi plane: 8 bf16][mid plane: 8 bf16], lane (i = l & 15, kq = l >> 4)
//     holds k = 32 s + 8 kq .. + 7 of row i.  GEMM1 (136) + GEMM3 (120) operands fill the 256 AGPRs; two tagged K_j blocks
//     (2 x 40) sit in VGPRs.
//   * B operands live in LDS as bf16 planes [16 instances][K] (pitch 16 B x odd: conflict-free ds_read_b128): a lane reads its
//     8 consecutive k of one instance with one 16-byte load per plane.  Producers write them in that form: the row owners 4
//     consecutive rows (one ds_write_b64 per plane), the column owners single elements.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "rqp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int NB_, int MBW_>
struct Mfma16Cfg {
    static constexpr int NB = NB_, MBW = MBW_;        // n tiles of 16 ; m tiles of 16 per wave
    static constexpr int NT = 256, NW = 4, TB = 16;
    static constexpr int MB = NW * MBW;
    static constexpr int NP = 16 * NB, MP = 16 * MB, KT = MP + NP;
    static constexpr int KS1 = (KT + 31) / 32;        // k-steps of GEMM1 (13): 0 .. KS1-2 round-robin, the last one by tile
    static constexpr int LS1 = (KS1 - 1) / NW;        // regular k-steps per wave (3)
    static constexpr int KS2 = (NP + 31) / 32;        // k-steps of GEMM2 and GEMM3 (3)
    static constexpr int XT = (NB + NW - 1) / NW;     // extra-step operand slots per wave (2: tiles w and w + 4)
    // The LAST k-step of each GEMM is a HALF step (16 k: v_mfma_f32_16x16x16_bf16, operands of 2 dwords per plane): K = 400 =
    // 12 x 32 + 16 and K = 80 = 2 x 32 + 16 exactly -- no padded half tiles in the register file (28 AGPRs less, which is what
    // keeps the resident operands out of scratch).
    static constexpr int P1 = KT + 8;                 // bf16 pitch of the [16][K] planes of GEMM1's B operand (816 B = 16 x 51)
    static constexpr int P2 = NP + 8;                 // ... of GEMM2 / GEMM3's (176 B = 16 x 11)
    static_assert((KS1 - 1) % NW == 0 && KS2 <= NW && KT % 32 == 16 && NP % 32 == 16 && ((P1 / 8) & 1) == 1 && ((P2 / 8) & 1) == 1,
                  "k-step dealing / half steps / pitches");
    // lane-linear operand images (dwords, 8 per lane and operand): W1 [NW][LS1 * NB + XT][64][8] | W3 [NW][MBW][KS2][64][8] |
    // K [nrho][KS2][NB][64][8]
    static constexpr int O1 = LS1 * NB + XT;          // GEMM1 operands per wave (17)
    static constexpr size_t W1_ELEMS = (size_t)NW * O1 * 64 * 8, W3_ELEMS = (size_t)NW * MBW * KS2 * 64 * 8;
    static constexpr size_t KJ_ELEMS = (size_t)KS2 * NB * 64 * 8;
    __host__ __device__ static constexpr int tile_of(int w, int tl) { return NW * tl + ((tl & 1) ? NW - 1 - w : w); }
    static constexpr size_t lds_bytes() {
        return (size_t)2 * 16 * P1 * 2                // V1h V1m
               + (size_t)4 * 16 * P2 * 2              // V3h V3m Dh Dm
               + (size_t)NW * NP * 16 * 4             // part
               + (size_t)(NW - 1) * NP * 16 * 4       // part2
               + (size_t)3 * MP * 16 * 4              // LB UB ZL
               + (size_t)NB * NT * 4 + NP * 16 * 4    // T3 GV
               + (size_t)(NW * 16 * 4 + 16 * 16 * 8 + 64 + 8 * 16) * 4;
    }
};

#ifndef MFMA_PIN
#define MFMA_PIN(acc)
#endif

namespace {

__device__ __forceinline__ float nanmaxf16(float a, float b) {        // NaN-propagating max (torch semantics)
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}

// two float32 -> their bf16 hi and mid planes, packed as (a | b << 16): hi = bf16(v), mid = bf16(v - hi), round-to-nearest-even
// (v_cvt_pk_bf16_f32; NaN stays NaN, inf stays inf with a NaN mid -- the products then propagate NaN like the fp32 kernel)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid) {
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    const bf16x2 m = {(__bf16)(a - (float)h[0]), (__bf16)(b - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
}
__device__ __forceinline__ void split_one(float a, unsigned short& hi, unsigned short& mid) {
    const __bf16 h = (__bf16)a;
    const __bf16 m = (__bf16)(a - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    mid = __builtin_bit_cast(unsigned short, m);
}
// acc += A B with A = ah + am, B = bh + bm (bf16 planes): the three leading terms
// (hipcc selects the AGPR-destination MFMA forms in a kernel that pins operands with "a" constraints: the accumulators need
//  AGPRs of their own next to the resident operands -- 20 for GEMM1's five -- which the half steps leave free; with all 256
//  AGPRs pinned it spilled 18 operands to scratch and reloaded them inside GEMM1 at every iteration)
__device__ __forceinline__ f32x4 mfma3(const u32x4& ah, const u32x4& am, const u32x4& bh, const u32x4& bm, f32x4 acc) {
    MFMA_PIN(acc);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, am), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    MFMA_PIN(acc);
    return acc;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
// the same for a half step (16 k): operands of 4 bf16 per lane and plane, lane (i, kq) holds k = 4 kq .. + 3
__device__ __forceinline__ f32x4 mfma3h(const u32x2& ah, const u32x2& am, const u32x2& bh, const u32x2& bm, f32x4 acc) {
    MFMA_PIN(acc);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, am), __builtin_bit_cast(s16x4, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bh), acc, 0, 0, 0);
    MFMA_PIN(acc);
    // hipcc 7.2 inserts no wait between a v_mfma_f32_16x16x16_bf16 and a DEPENDENT v_mfma_f32_16x16x32_bf16 (SrcC = the result):
    // the 16x16x32 then accumulates onto a stale register (measured: 53 % of the config-3 batch ended at other iteration
    // counts, 4 % with the half step first; exact again with this drain).  24 wait states cover the 4-pass producer.
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc));
    return acc;
}

}   // namespace

// DIAG = true is a separate diagnostic build (RQP_DIAG=1): s_memtime stamps accumulate the cycles each wave spends per segment.
template <class C, bool DIAG>
__global__ void __launch_bounds__(256, 1) k_admm_mfma16(SolveArgs a, const unsigned* __restrict__ img, int* __restrict__ queue,
                                                        unsigned long long* __restrict__ dbg) {
    constexpr int NB = C::NB, MBW = C::MBW, NT = C::NT, NW = C::NW, NP = C::NP, MP = C::MP;
    constexpr int KS1 = C::KS1, LS1 = C::LS1, KS2 = C::KS2, P1 = C::P1, P2 = C::P2, O1 = C::O1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    unsigned short* V1h = (unsigned short*)smraw;           // [16][P1] bf16 hi plane of GEMM1's B operand: k < MP: nu (lam / 0 at a check), MP <= k: x (0)
    unsigned short* V1m = V1h + 16 * P1;                    //          mid plane
    unsigned short* V3h = V1m + 16 * P1;                    // [16][P2] dx (x at the start)
    unsigned short* V3m = V3h + 16 * P2;
    unsigned short* Dh = V3m + 16 * P2;                     // [16][P2] d = H x + g + A' nu: GEMM2's B operand
    unsigned short* Dm = Dh + 16 * P2;
    float* part = (float*)(Dm + 16 * P2);                  // [NW][NP][16] wave partials of GEMM1 (swizzled rows, as in rqp_mfma.hip)
    float* part2 = part + NW * NP * 16;                     // [NW-1][NP][16] wave partials of GEMM2
    float* LB = part2 + (NW - 1) * NP * 16;                 // [MBW][NT][4] l of the lane's own rows
    float* UB = LB + MP * 16;                               // u
    float* ZL = UB + MP * 16;                               // low word of the float-float A x
    float* T3 = ZL + MP * 16;                               // [NB][NT] A' lam of the pending check
    float* GV = T3 + NB * NT;                               // [NP][16] g
    float* red = GV + NP * 16;                              // [NW][16][4]
    float* rr = red + NW * 16 * 4;                          // [16][16][8]
    float* rhosf = rr + 16 * 16 * 8;                        // [64]
    float* inst = rhosf + 64;                               // [8][16] (rows as in rqp_mfma.hip)
    int* inst_i = (int*)inst;
    static_assert((2 * 16 * P1 * 2 + 4 * 16 * P2 * 2) % 16 == 0, "float arrays start 16-byte aligned");

    const int n = a.n, m = a.m;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i16 = lane & 15, kq = lane >> 4;       // MFMA lane coordinates
    const int cj = tid & 15, rg = tid >> 4;          // column-owner coordinates: instance cj, rows rg + 16 e
    const int kmax = a.max_iter;
    const bool refill = queue != nullptr && kmax > 0 && (kmax % a.check_interval) == 0;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define OPQ(v) asm volatile("" : "+v"(v))
    auto dl_base = [&](int lp, int r) __attribute__((always_inline)) { return (lp >> 4) * 64 + ((lp + 16 * r) & 63); };
    auto co_base = [&](int tp) __attribute__((always_inline)) {
        const int cjp = tp & 15, rgp = tp >> 4;
        return (rgp >> 2) * 64 + ((16 * (rgp & 3) + cjp + 16 * ((rgp >> 2) & 3)) & 63);
    };

    // ---- resident A-operands: [hi 4 dwords][mid 4 dwords] per operand and lane, all loads issued before the first pin
    constexpr int R1 = LS1 * NB, XT = O1 - R1;       // regular (full-step) operands of GEMM1 per wave; half-step slots
    u32x4 w1h[R1], w1m[R1];          // GEMM1: o = ls * NB + t: S[32 (NW ls + wave) + 8 kq ..][16 t + i16]
    u32x2 x1h[XT], x1m[XT];          //        half step KS1-1 (k = 32 (KS1-1) + 4 kq ..), tile wave + NW e
    {
        const u32x4* p = (const u32x4*)(img + ((size_t)wave_u * O1 * 64 + lane) * 8);
#pragma unroll
        for (int o = 0; o < R1; ++o) {
            w1h[o] = p[(size_t)o * 64 * 2];
            w1m[o] = p[(size_t)o * 64 * 2 + 1];
        }
#pragma unroll
        for (int e = 0; e < XT; ++e) {
            const u32x4 v = p[(size_t)(R1 + e) * 64 * 2];
            x1h[e] = (u32x2){v[0], v[1]};
            x1m[e] = (u32x2){v[2], v[3]};
        }
    }
    u32x4 w3h[MBW][KS2 - 1], w3m[MBW][KS2 - 1];   // GEMM3: A[16 tile_of(wave, tl) + i16][32 s + 8 kq ..]
    u32x2 x3h[MBW], x3m[MBW];                     //        half step: columns 32 (KS2-1) + 4 kq ..
    {
        const u32x4* p = (const u32x4*)(img + C::W1_ELEMS + ((size_t)wave_u * MBW * KS2 * 64 + lane) * 8);
#pragma unroll
        for (int tl = 0; tl < MBW; ++tl) {
#pragma unroll
            for (int s = 0; s < KS2 - 1; ++s) {
                w3h[tl][s] = p[(size_t)(tl * KS2 + s) * 64 * 2];
                w3m[tl][s] = p[(size_t)(tl * KS2 + s) * 64 * 2 + 1];
            }
            const u32x4 v = p[(size_t)(tl * KS2 + KS2 - 1) * 64 * 2];
            x3h[tl] = (u32x2){v[0], v[1]};
            x3m[tl] = (u32x2){v[2], v[3]};
        }
    }
#pragma unroll
    for (int o = 0; o < R1; ++o) asm volatile("" : "+a"(w1h[o]), "+a"(w1m[o]));     // resident MFMA A-operands live in the AGPRs
#pragma unroll
    for (int e = 0; e < XT; ++e) asm volatile("" : "+a"(x1h[e]), "+a"(x1m[e]));
#pragma unroll
    for (int tl = 0; tl < MBW; ++tl) {
#pragma unroll
        for (int s = 0; s < KS2 - 1; ++s) asm volatile("" : "+a"(w3h[tl][s]), "+a"(w3m[tl][s]));
        asm volatile("" : "+a"(x3h[tl]), "+a"(x3m[tl]));
    }
    // GEMM2: K_j[16 t + i16][32 wave + 8 kq ..] (waves 0 .. KS2-1) in two tagged VGPR blocks
    const unsigned* kimg = img + C::W1_ELEMS + C::W3_ELEMS + (size_t)(wave_u < KS2 ? wave_u : 0) * NB * 64 * 8;
    u32x4 k0h[NB], k0m[NB], k1h[NB], k1m[NB];
    int ktag0 = -1, ktag1 = -1;

    // ---- per-instance scalars and vectors (rqp_mfma.hip) ----------------------------------------------------------
    for (int i = tid; i < a.nrho && i < 64; i += NT) rhosf[i] = (float)a.rhos[i];
    if (tid < 16) {
        const int st = blockIdx.x * 16 + tid;                      // slot -> instance through SolveArgs.order (rqp_mfma.hip)
        const bool ok = st < a.B;
        const int s0 = ok ? st : blockIdx.x * 16;
        const int id = ok ? (a.order ? a.order[st] : st) : a.B;
        const int ri = a.rho_ind[a.order ? a.order[s0] : s0];
        inst_i[2 * 16 + tid] = id;
        inst_i[3 * 16 + tid] = 0;
        inst_i[4 * 16 + tid] = ri;
        inst_i[5 * 16 + tid] = ok ? 0 : 1;
        inst_i[6 * 16 + tid] = 1;
        inst_i[7 * 16 + tid] = 0;
        inst[0 * 16 + tid] = (float)a.rhos[ri];
    }
    if (tid == 0) {
        inst[16 + 0] = (float)a.tol;
        inst[16 + 1] = (float)a.thr_p;
        inst[16 + 2] = (float)a.thr_d;
        inst[16 + 3] = (float)a.rho_min;
        inst[16 + 4] = (float)a.rho_max;
        inst[16 + 5] = (float)a.eps_rel;
    }
    // zero the B-operand planes once (their pitch padding is never read)
    for (int i = tid; i < 16 * P1; i += NT) { V1h[i] = 0; V1m[i] = 0; }
    for (int i = tid; i < 16 * P2; i += NT) { V3h[i] = 0; V3m[i] = 0; Dh[i] = 0; Dm[i] = 0; }
    float zh[MBW][4], zz[MBW][4], lm[MBW][4];
    unsigned eqmask = 0;
    float xr[NB];
#pragma unroll
    for (int tl = 0; tl < MBW; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) zh[tl][r] = zz[tl][r] = lm[tl][r] = 0.f;
#pragma unroll
    for (int e = 0; e < NB; ++e) xr[e] = 0.f;
    int ri_l = 0;
    float rho_ne = 1.f, rho_eq = 1.f, inv_ne = 1.f, inv_eq = 1.f;
    auto set_rho = [&]() __attribute__((always_inline)) {
        rho_ne = rhosf[ri_l];
        rho_eq = rho_ne * 1e3f;
        inv_ne = 1.0f / rho_ne;
        inv_eq = 1.0f / rho_eq;
    };
    // rows 16 T + 4 kq + 0..3 of instance i16 -> the two planes of V1 (one 8-byte store each)
    auto put_rows = [&](int lp, int T, float v0, float v1, float v2, float v3) __attribute__((always_inline)) {
        unsigned h0, m0, h1, m1;
        split_pair(v0, v1, h0, m0);
        split_pair(v2, v3, h1, m1);
        const int o = (lp & 15) * P1 + 16 * T + 4 * (lp >> 4);
        *(u32x2*)(V1h + o) = (u32x2){h0, h1};
        *(u32x2*)(V1m + o) = (u32x2){m0, m1};
    };
    // x (or 0) of the column owner's rows -> V1 rows MP + row ; v (or 0) -> V3 rows
    auto put_x = [&](int tp, int e, float v) __attribute__((always_inline)) {
        unsigned short h, mm;
        split_one(v, h, mm);
        const int o = (tp & 15) * P1 + MP + (tp >> 4) + 16 * e;
        V1h[o] = h;
        V1m[o] = mm;
    };
    auto put_v3 = [&](int tp, int e, float v) __attribute__((always_inline)) {
        unsigned short h, mm;
        split_one(v, h, mm);
        const int o = (tp & 15) * P2 + (tp >> 4) + 16 * e;
        V3h[o] = h;
        V3m[o] = mm;
    };

    auto make_nu = [&]() __attribute__((always_inline)) {
        int lp = lane;
        OPQ(lp);
#pragma unroll
        for (int tl = 0; tl < MBW; ++tl) {
            float nv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                const float p = (zh[tl][r] - zz[tl][r]) + ZL[(tl * NT + tid) * 4 + r];
                const float lh = lm[tl][r] + rho * p;
                lm[tl][r] = lh;
                nv[r] = lh + rho * p;
            }
            put_rows(lp, C::tile_of(wave_u, tl), nv[0], nv[1], nv[2], nv[3]);
        }
    };

    int ph = 4, k = 0, to_chk = a.check_interval;
    bool final_chk = false;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;

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

    while (true) {
        __syncthreads();
        if (ph == 4) {
            int n_o = n, m_o = m, tp = tid;
            asm volatile("" : "+s"(n_o), "+s"(m_o), "+v"(tp));
            const int c_o = tp & 15, rg_o = tp >> 4, kq_o = (tp >> 4) & 3, wave_o = tp >> 6;
            const int id = inst_i[2 * 16 + c_o];
            const bool fresh = inst_i[6 * 16 + c_o] != 0, real = id < a.B;
            if (fresh) {
                unsigned em = 0;
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
                eqmask = em;
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int row = rg_o + 16 * e;
                if (fresh) {
                    const bool ok = real && row < n_o;
                    const size_t o = (size_t)(real ? id : 0) * n_o + (row < n_o ? row : 0);
                    xr[e] = ok ? (float)a.x[o] : 0.f;
                    GV[tp + 256 * e] = ok ? ((const float*)a.g)[o] : 0.f;      // [rg + 16 e][cj]
                }
                put_v3(tp, e, fresh ? xr[e] : 0.f);          // GEMM3 on x: A x of the loaded slots, + 0 for the others
                put_x(tp, e, xr[e]);
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
            int lp = lane;
            OPQ(lp);
            const int bo = (lp & 15) * P1 + 8 * (lp >> 4);           // this lane's 8 k of k-step 0
            u32x4 bh[LS1], bm[LS1];                                  // all B operands first: no LDS latency inside the MFMA stream
#pragma unroll
            for (int ls = 0; ls < LS1; ++ls) {
                bh[ls] = *(const u32x4*)(V1h + bo + 32 * (NW * ls + wave_u));
                bm[ls] = *(const u32x4*)(V1m + bo + 32 * (NW * ls + wave_u));
            }
            const int bx = (lp & 15) * P1 + 32 * (KS1 - 1) + 4 * (lp >> 4);     // half step: k = 32 (KS1-1) + 4 kq ..
            const u32x2 bxh = *(const u32x2*)(V1h + bx), bxm = *(const u32x2*)(V1m + bx);
            int pb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[r] = wave_u * NP * 16 + dl_base(lp, r);
            // One n tile after the other, each ONE accumulation chain (half step of the owner wave first, then the wave's LS1 full
            // steps) whose result is pinned to arch VGPRs once, at its end, and stored at once.  hipcc uses the AGPR-destination
            // MFMA forms in this kernel (operands pinned with "a" constraints): five accumulators live across the tile loop made
            // it shuffle resident operands between the two register files (and miscompute); pins around every product cost 8
            // copies + a drain per product.
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (wave_u == (t & (NW - 1))) {                       // (uniform branch)
                    asm volatile("" ::"a"(x1h[t / NW]), "a"(x1m[t / NW]));
                    acc = mfma3h(x1h[t / NW], x1m[t / NW], bxh, bxm, acc);
                }
#pragma unroll
                for (int ls = 0; ls < LS1; ++ls) {
                    asm volatile("" ::"a"(w1h[ls * NB + t]), "a"(w1m[ls * NB + t]));
                    acc = mfma3(w1h[ls * NB + t], w1m[ls * NB + t], bh[ls], bm[ls], acc);
                }
                asm volatile("" : "+v"(acc));
#pragma unroll
                for (int r = 0; r < 4; ++r) part[pb[r] + 256 * t] = acc[r];
            }
            stamp(1);
            __syncthreads();
            stamp(2);
        }
        bool run_g3 = (ph == 0);
        if (ph == 1) {
            {   // d = g + sum of the 4 partials (fixed order) by the column owners (all 256 threads) -> bf16 planes [16][P2]: the
                // D-layout -> B-layout transposition of d costs 40 strided LDS reads per lane when the three GEMM2 waves do it
                int tp = tid;
                OPQ(tp);
                const int xb = co_base(tp);
                float gk[NB], pk[NB][NW];
#pragma unroll
                for (int e = 0; e < NB; ++e) {
                    gk[e] = GV[tp + 256 * e];
#pragma unroll
                    for (int w = 0; w < NW; ++w) pk[e][w] = part[w * NP * 16 + xb + 256 * e];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < NB; ++e) {
                    float d = gk[e];
#pragma unroll
                    for (int w = 0; w < NW; ++w) d += pk[e][w];
                    unsigned short h, mm;
                    split_one(d, h, mm);
                    const int o = (tp & 15) * P2 + (tp >> 4) + 16 * e;
                    Dh[o] = h;
                    Dm[o] = mm;
                }
            }
            stamp(3);
            __syncthreads();
            stamp(4);
            if (wave_u < KS2) {                                      // dx partial = K_j d over this wave's k-step, K_j chosen per column
                int lp = lane;
                OPQ(lp);
                // B operand of this wave's k-step: full step (waves < KS2-1): d rows 32 wave + 8 kq .., half step (wave KS2-1): rows
                // 32 wave + 4 kq .. (in dh[0..1] / dm[0..1])
                const bool half = wave_u == KS2 - 1;
                u32x4 dh, dm;
                {
                    const int o = (lp & 15) * P2 + 32 * wave_u + (half ? 4 : 8) * (lp >> 4);
                    if (!half) {
                        dh = *(const u32x4*)(Dh + o);
                        dm = *(const u32x4*)(Dm + o);
                    } else {
                        const u32x2 a2 = *(const u32x2*)(Dh + o), b2 = *(const u32x2*)(Dm + o);
                        dh = (u32x4){a2[0], a2[1], 0u, 0u};
                        dm = (u32x4){b2[0], b2[1], 0u, 0u};
                    }
                }
                f32x4 sel[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) sel[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                unsigned long long todo = __ballot(lane < 16);
                bool first_pass = true;
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    const int j = __builtin_amdgcn_readlane(ri_l, src);
                    if (j != ktag0 && j != ktag1) {                  // miss: refill a block from the L2 image of K_j
                        const bool b0_used = __ballot(ri_l == ktag0) != 0ull;
                        const bool to0 = ktag0 < 0 || (!b0_used && ktag1 >= 0);
                        unsigned loff = lane;
                        asm volatile("" : "+v"(loff));
                        const u32x4* kp = (const u32x4*)(kimg + (size_t)j * C::KJ_ELEMS + (size_t)loff * 8);
                        if (to0) {
#pragma unroll
                            for (int t = 0; t < NB; ++t) {
                                k0h[t] = kp[(size_t)t * 64 * 2];
                                k0m[t] = kp[(size_t)t * 64 * 2 + 1];
                            }
                            ktag0 = j;
                        } else {
#pragma unroll
                            for (int t = 0; t < NB; ++t) {
                                k1h[t] = kp[(size_t)t * 64 * 2];
                                k1m[t] = kp[(size_t)t * 64 * 2 + 1];
                            }
                            ktag1 = j;
                        }
                    }
                    f32x4 acc[NB];
#pragma unroll
                    for (int t = 0; t < NB; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    // (a half-step operand sits in the first 4 dwords of its image slot: hi = k?h.xy, mid = k?h.zw)
                    if (j == ktag0) {
                        if (!half) {
#pragma unroll
                            for (int t = 0; t < NB; ++t) acc[t] = mfma3(k0h[t], k0m[t], dh, dm, acc[t]);
                        } else {
#pragma unroll
                            for (int t = 0; t < NB; ++t)
                                acc[t] = mfma3h((u32x2){k0h[t][0], k0h[t][1]}, (u32x2){k0h[t][2], k0h[t][3]}, (u32x2){dh[0], dh[1]}, (u32x2){dm[0], dm[1]}, acc[t]);
                        }
                    } else {
                        if (!half) {
#pragma unroll
                            for (int t = 0; t < NB; ++t) acc[t] = mfma3(k1h[t], k1m[t], dh, dm, acc[t]);
                        } else {
#pragma unroll
                            for (int t = 0; t < NB; ++t)
                                acc[t] = mfma3h((u32x2){k1h[t][0], k1h[t][1]}, (u32x2){k1h[t][2], k1h[t][3]}, (u32x2){dh[0], dh[1]}, (u32x2){dm[0], dm[1]}, acc[t]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NB; ++t) asm volatile("" : "+v"(acc[t]));
                    if (first_pass) {
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
                float pk[NB][KS2];
#pragma unroll
                for (int e = 0; e < NB; ++e)
#pragma unroll
                    for (int w = 0; w < KS2; ++w) pk[e][w] = part2[w * NP * 16 + xb + 256 * e];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < NB; ++e) {                        // dx = -K d ; x += dx
                    float kd = 0.f;
#pragma unroll
                    for (int w = 0; w < KS2; ++w) kd += pk[e][w];
                    const float dx = -kd;
                    xr[e] += dx;
                    put_v3(tp, e, dx);
                    put_x(tp, e, xr[e]);
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
            const bool upd = (ph == 1);
            const bool fin_next = upd && !refill && (k + 1 >= kmax) && (to_chk != 1);
            const bool with_nu = upd && to_chk != 1 && !fin_next;
            int lp = lane;
            OPQ(lp);
            u32x4 bh[KS2 - 1], bm[KS2 - 1];
            u32x2 bxh, bxm;
            {
                const int bo = (lp & 15) * P2 + 8 * (lp >> 4);
#pragma unroll
                for (int s = 0; s < KS2 - 1; ++s) {
                    bh[s] = *(const u32x4*)(V3h + bo + 32 * s);
                    bm[s] = *(const u32x4*)(V3m + bo + 32 * s);
                }
                const int bx = (lp & 15) * P2 + 32 * (KS2 - 1) + 4 * (lp >> 4);
                bxh = *(const u32x2*)(V3h + bx);
                bxm = *(const u32x2*)(V3m + bx);
            }
#pragma unroll
            for (int tl = 0; tl < MBW; ++tl) {
                float pz[4], pl[4], pu[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pz[r] = ZL[(tl * NT + tid) * 4 + r];
                    pl[r] = LB[(tl * NT + tid) * 4 + r];
                    pu[r] = UB[(tl * NT + tid) * 4 + r];
                }
                // (issuing the chain of tile tl + 1 before this tile's row update -- bf16 MFMAs run beside the VALU -- measured
                //  2 % SLOWER: the dependent chain blocks the wave's issue unless the scheduler interleaves it, and it does not)
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                asm volatile("" ::"a"(x3h[tl]), "a"(x3m[tl]));
                acc = mfma3h(x3h[tl], x3m[tl], bxh, bxm, acc);
#pragma unroll
                for (int s = 0; s < KS2 - 1; ++s) {
                    asm volatile("" ::"a"(w3h[tl][s]), "a"(w3m[tl][s]));
                    acc = mfma3(w3h[tl][s], w3m[tl][s], bh[s], bm[s], acc);
                }
                asm volatile("" : "+v"(acc));                        // (one chain per row tile, pinned once at its end: GEMM1)
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
                    float nv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                        const float lh = lm[tl][r] + rho * pp[r];
                        lm[tl][r] = lh;
                        nv[r] = lh + rho * pp[r];
                    }
                    put_rows(lp, C::tile_of(wave_u, tl), nv[0], nv[1], nv[2], nv[3]);
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
                if (!nu_done) make_nu();
                ph = 1;
            } else {                                     // check part 1: V1 = [lam; 0], row-side maxima
                v0 = 0.f; v1 = 0.f; v2 = 0.f;
                int lp = lane, tp = tid;
                OPQ(lp);
                OPQ(tp);
#pragma unroll
                for (int tl = 0; tl < MBW; ++tl) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (Ruiz scaling: caller-space norms, SolveArgs.scE; shared matrices: one set of factors)
                        const int row = 16 * C::tile_of(wave_u, tl) + 4 * (lp >> 4) + r;
                        const float we = (a.scE && row < m) ? (float)(1.0 / a.scE[row]) : 1.f;
                        const float zlo = ZL[(tl * NT + tid) * 4 + r];
                        v0 = nanmaxf16(v0, fabsf((zh[tl][r] - zz[tl][r]) + zlo) * we);
                        v1 = nanmaxf16(v1, fabsf(zh[tl][r] + zlo) * we);
                        v2 = nanmaxf16(v2, fabsf(zz[tl][r]) * we);
                    }
                    put_rows(lp, C::tile_of(wave_u, tl), lm[tl][0], lm[tl][1], lm[tl][2], lm[tl][3]);
                }
#pragma unroll
                for (int e = 0; e < NB; ++e) put_x(tp, e, 0.f);
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
                put_x(tp, e, xr[e]);
            }
#pragma unroll
            for (int tl = 0; tl < MBW; ++tl) put_rows(lp, C::tile_of(wave_u, tl), 0.f, 0.f, 0.f, 0.f);
            ph = 3;
        } else {                                         // ph == 3: t2 = H x ; residuals and decisions
            v0 = nanmaxf16(v0, __shfl_xor(v0, 16, 64)); v0 = nanmaxf16(v0, __shfl_xor(v0, 32, 64));
            v1 = nanmaxf16(v1, __shfl_xor(v1, 16, 64)); v1 = nanmaxf16(v1, __shfl_xor(v1, 32, 64));
            v2 = nanmaxf16(v2, __shfl_xor(v2, 16, 64)); v2 = nanmaxf16(v2, __shfl_xor(v2, 32, 64));
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
                w3 = nanmaxf16(w3, fabsf(t2 + t3 + ge) * wd);
                w4 = nanmaxf16(w4, fabsf(t2) * wd);
                w5 = nanmaxf16(w5, fabsf(t3) * wd);
                w6 = nanmaxf16(w6, fabsf(ge) * wd);
                jp += xr[e] * (0.5f * t2 + ge);                      // compute_J :320-322
            }
            rr[(rg * 16 + cj) * 8 + 3] = w3;
            rr[(rg * 16 + cj) * 8 + 4] = w4;
            rr[(rg * 16 + cj) * 8 + 5] = w5;
            rr[(rg * 16 + cj) * 8 + 6] = w6;
            rr[(rg * 16 + cj) * 8 + 7] = jp;
            __syncthreads();
            if (tid < 16) {                                          // one thread per instance decides (rqp_mfma.hip, same block)
                int j = tid;
                asm volatile("" : "+v"(j));
                const float tolT = inst[16 + 0], thr_p = inst[16 + 1], thr_d = inst[16 + 2], rmin = inst[16 + 3], rmax = inst[16 + 4];
                float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f, q6 = 0.f, obj = 0.f;
                for (int w = 0; w < NW; ++w) {
                    q0 = nanmaxf16(q0, red[(w * 16 + j) * 4 + 0]);
                    q1 = nanmaxf16(q1, red[(w * 16 + j) * 4 + 1]);
                    q2 = nanmaxf16(q2, red[(w * 16 + j) * 4 + 2]);
                }
                for (int g2 = 0; g2 < 16; ++g2) {
                    q3 = nanmaxf16(q3, rr[(g2 * 16 + j) * 8 + 3]);
                    q4 = nanmaxf16(q4, rr[(g2 * 16 + j) * 8 + 4]);
                    q5 = nanmaxf16(q5, rr[(g2 * 16 + j) * 8 + 5]);
                    q6 = nanmaxf16(q6, rr[(g2 * 16 + j) * 8 + 6]);
                    obj += rr[(g2 * 16 + j) * 8 + 7];
                }
                int newly = 0;
                const bool alive = inst_i[5 * 16 + j] == 0;
                float est = 0.f, num = 0.f, den = 1.f;
                int ri = 0, id = 0, kc = 0;
                bool conv = false, last = false;
                if (alive) {
                    num = q0 / nanmaxf16(q1, q2);                                         // :315
                    den = q3 / nanmaxf16(nanmaxf16(q4, q5), q6);                          // :316
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
                    kc = k - inst_i[3 * 16 + j];
                    const int chk_no = kc / a.check_interval;
                    if (!final_chk && a.info.trace && chk_no <= a.info.trace_cap) {
                        double* tr = a.info.trace + ((size_t)id * a.info.trace_cap + (chk_no - 1)) * 4;
                        tr[0] = (double)q0; tr[1] = (double)q3; tr[2] = (double)est; tr[3] = (double)ri_before;
                    }
                    const float er = inst[16 + 5];
                    const float tp = er > 0.f ? thr_p + er * nanmaxf16(q1, q2) : thr_p;
                    const float td = er > 0.f ? thr_d + er * nanmaxf16(nanmaxf16(q4, q5), q6) : thr_d;
                    conv = !final_chk && (q0 < tp && q3 < td);                            // :233
                    last = final_chk || kc >= kmax;                                        // :243 max-iter fallthrough
                }
                const bool still = alive && !conv && !last;
                const int nact = __popcll(__ballot(still));
                const bool hand = still && !refill && a.handoff_cols > 0 && nact <= a.handoff_cols && a.info.status != nullptr;
                if (alive && (conv || last || hand)) {
                    float est_out = est;
                    if (!conv && !final_chk && !hand) {
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
                    if (refill) {
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
                const bool live = inst_i[5 * 16 + j] == 0;
                const int ri_now = inst_i[4 * 16 + j];
                const unsigned long long lm16 = __ballot(live);
                if (lm16) {
                    const int ri_live = __shfl(ri_now, __ffsll((long long)lm16) - 1, 64);
                    if (!live) inst_i[4 * 16 + j] = ri_live;
                }
            }
            __syncthreads();
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
                for (int e = 0; e < NB; ++e) put_x(tp3, e, xr[e]);
                make_nu();
                ph = 1;
            }
        }
        if (ph == 1) stamp(10); else stamp(-1);
    }
    if constexpr (DIAG) {
        if (lane == 0)
            for (int e = 0; e < 12; ++e) dbg[((size_t)blockIdx.x * 4 + wave) * 12 + e] = t_acc[e];
    }
}

#undef OPQ

// ---------------------------------------------------------------------------- packing
// Operand o of lane l = [hi plane: 4 dwords][mid plane: 4 dwords]; dword q of a plane = elements j = 2q (low half), 2q + 1 of
// the lane's 8 consecutive k (k = 32 s + 8 (l >> 4) + j), row / column i = l & 15 of the tile:
//   W1[w][o]: o = ls NB + t (ls < LS1): S[32 (NW ls + w) + 8 kq + j][16 t + i] ; o = LS1 NB + e: step KS1-1, tile t = w + NW e (< NB)
//             S = [A (MP rows, zero padded); H' (NP rows)]
//   W3[w][tl][s]: A[16 tile_of(w, tl) + i][32 s + 8 kq + j]
//   K[jr][s][t] : K_jr[16 t + i][32 s + 8 kq + j]
template <class C>
__global__ void k_pack_mfma16(int n, int m, int ldn, int nrho, const float* __restrict__ A, const float* __restrict__ Ht,
                              const float* __restrict__ K, unsigned* __restrict__ img) {
    constexpr int NB = C::NB, MBW = C::MBW, MP = C::MP, KS1 = C::KS1, LS1 = C::LS1, KS2 = C::KS2, NW = C::NW, O1 = C::O1;
    const size_t n_ops = (size_t)NW * O1 + (size_t)NW * MBW * KS2 + (size_t)nrho * KS2 * NB;      // operands of 64 lanes x 8 dwords
    const size_t total = n_ops * 64 * 4;                                                         // one thread per (operand, lane, dword pair q)
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx & 3), l = (int)((idx >> 2) & 63), i = l & 15, kq = l >> 4;
        const size_t op = idx >> 8;
        // which matrix element block this operand is: (kind, s = k-step, tile / row-tile coordinates)
        int kind, s, t = 0, w = 0, tl = 0, jr = 0;
        if (op < (size_t)NW * O1) {
            kind = 0;
            w = (int)(op / O1);
            const int o = (int)(op % O1);
            if (o < LS1 * NB) { s = NW * (o / NB) + w; t = o % NB; }
            else { s = KS1 - 1; t = w + NW * (o - LS1 * NB); }
        } else if (op < (size_t)NW * O1 + (size_t)NW * MBW * KS2) {
            kind = 1;
            const int o = (int)(op - (size_t)NW * O1);
            s = o % KS2; tl = (o / KS2) % MBW; w = o / (KS2 * MBW);
        } else {
            kind = 2;
            const size_t o = op - (size_t)NW * O1 - (size_t)NW * MBW * KS2;
            t = (int)(o % NB); s = (int)((o / NB) % KS2); jr = (int)(o / ((size_t)NB * KS2));
        }
        const bool half = s == ((kind == 0) ? KS1 - 1 : KS2 - 1);      // half step: 4 k per lane, [hi 2 dwords][mid 2 dwords][4 unused]
        float v[2] = {0.f, 0.f};
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * (half ? (q & 1) : q) + h;
            const int k = 32 * s + (half ? 4 : 8) * kq + j;
            if (kind == 0) {
                const int col = 16 * t + i;
                if (t < NB && col < n) {
                    if (k < MP) { if (k < m) v[h] = A[(size_t)k * ldn + col]; }
                    else if (k - MP < n) v[h] = Ht[(size_t)(k - MP) * ldn + col];
                }
            } else if (kind == 1) {
                const int r = 16 * C::tile_of(w, tl) + i;
                if (r < m && k < n) v[h] = A[(size_t)r * ldn + k];
            } else {
                const int r = 16 * t + i;
                if (r < n && k < n) v[h] = K[((size_t)jr * n + r) * ldn + k];
            }
        }
        unsigned hi, mid;
        split_pair(v[0], v[1], hi, mid);
        unsigned* dst = img + (op * 64 + l) * 8;
        if (!half) {
            dst[q] = hi;
            dst[4 + q] = mid;
        } else if (q < 2) {
            dst[q] = hi;
            dst[2 + q] = mid;
        } else {
            dst[2 + q] = 0u;                                            // dwords 4..7 (q = 2, 3 -> 4, 5) ...
            dst[4 + q] = 0u;                                            // ... and 6, 7
        }
    }
}

// ------------------------------------------------------------------------------ host side
typedef Mfma16Cfg<5, 5> Cfg16M55;        // n <= 80, m <= 320   (linear MPC, N=20, nx=12, nu=4 condensed)

size_t rqp_mfma16_img_elems(const rqp_handle* h) {
    return Cfg16M55::W1_ELEMS + Cfg16M55::W3_ELEMS + (size_t)h->nrho * Cfg16M55::KJ_ELEMS;
}

hipError_t rqp_launch_pack_mfma16(const rqp_handle* h, hipStream_t s) {
    k_pack_mfma16<Cfg16M55><<<256, 256, 0, s>>>(h->n, h->m, h->ldn, h->nrho, (const float*)h->A, (const float*)h->Ht, (const float*)h->K,
                                              (unsigned*)h->W1img);
    return hipGetLastError();
}
hipError_t rqp_prepare_mfma16(const rqp_handle* h) {
    hipError_t e = rqp_raise_lds_limit((const void*)k_admm_mfma16<Cfg16M55, false>, Cfg16M55::lds_bytes());
    if (e == hipSuccess && (h->debug & 2)) e = rqp_raise_lds_limit((const void*)k_admm_mfma16<Cfg16M55, true>, Cfg16M55::lds_bytes());
    return e;
}
hipError_t rqp_launch_solve_mfma16(const rqp_handle* h, const SolveArgs& a0, hipStream_t s) {
    const size_t lds = Cfg16M55::lds_bytes();
    const int tiles = (h->B + 15) / 16;
    // (slots could be grouped by starting rho index through SolveArgs.order as in rqp_mfmal.hip -- the kernel follows it -- but with
    //  K in registers a pass per distinct index is cheap here: the closed loop measured 1.4 % SLOWER with the extra sort launch)
    const SolveArgs& a = a0;
    int grid = tiles;
    int* queue = nullptr;
    if (tiles > h->ncu && h->queue && a.max_iter > 0 && a.check_interval > 0 && a.max_iter % a.check_interval == 0) {
        grid = h->ncu;            // persistent grid, slots refill from a queue (rqp_mfma.hip)
        queue = h->queue;
        hipError_t e = hipMemsetAsync(queue, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    if (h->debug & 2) {          // diagnostic build: per-segment cycle shares of the iteration (synchronous, debug only)
        unsigned long long* dbg = nullptr;
        const size_t cnt = (size_t)grid * 4 * 12;
        if (hipMalloc((void**)&dbg, cnt * 8) != hipSuccess) return hipErrorOutOfMemory;
        k_admm_mfma16<Cfg16M55, true><<<grid, Cfg16M55::NT, lds, s>>>(a, (const unsigned*)h->W1img, queue, dbg);
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> hb(cnt);
        (void)hipMemcpy(hb.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        static const char* names[11] = {"top wait", "GEMM1", "wait", "d", "wait", "GEMM2", "wait", "x", "wait", "GEMM3+rows", "next"};
        for (int w = 0; w < 4; ++w) {
            double tot[12] = {0};
            for (int t = 0; t < grid; ++t)
                for (int e2 = 0; e2 < 12; ++e2) tot[e2] += (double)hb[((size_t)t * 4 + w) * 12 + e2];
            fprintf(stderr, "[rqp diag mfma16] wave %d, %.1f iterations/workgroup, cycles per iteration:", w, tot[11] / grid);
            double it = 0;
            for (int e2 = 0; e2 < 11; ++e2) { fprintf(stderr, "  %s %.0f", names[e2], tot[e2] / tot[11]); it += tot[e2] / tot[11]; }
            fprintf(stderr, "  | total %.0f\n", it);
        }
        return hipGetLastError();
    }
    k_admm_mfma16<Cfg16M55, false><<<grid, Cfg16M55::NT, lds, s>>>(a, (const unsigned*)h->W1img, queue, nullptr);
    return hipGetLastError();
}
