// rqp_mfmad.hip -- the streamed-operand MFMA kernel of rqp_mfmal.hip in FLOAT64 (the reference's default -- and only working --
// precision) for shared-(H, A) batches: n <= 160, m <= 320 (the condensed linear-MPC form n = 80, m = 320).  The batch is
// the N = 16 axis of v_mfma_f64_16x16x4_f64 (exact float64 FMA chains); state, residuals, checks in float64.
// Same structure as k_admm_mfmad (groups = non-zero 16 x 16 blocks streamed from L2 through a 5-slot register ring, one stream
// per wave and GEMM, vectors in LDS as [block][lane] 4-vectors, no vector ALU in the visit loops, static tiles); what differs:
//   * the D layout of the float64 MFMA: register r of lane (kq, i16) is row kq + 4 r of the 16-row tile (float32: 4 kq + r), so
//     the k index of MFMA j of a group is 16 blk + kq + 4 j and the operand images are packed accordingly;
//   * a group is 2 KB (two buffer_load_dwordx4 per lane), a vector element 32 bytes (two ds_read_b128);
//   * A x is a plain float64 accumulator (no float-float pair); 10 n tiles and 20 m tiles (2 / 3 per wave).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rqp_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MD_NW = 8, MD_NT = 512, MD_NB = 10, MD_MB = 20;      // waves, threads, n tiles, m tiles
constexpr int MD_NP = 16 * MD_NB, MD_MP = 16 * MD_MB;
constexpr int MD_KB1 = MD_MB + MD_NB, MD_KB2 = MD_NB;              // k blocks of GEMM1 (30: A's rows, then H's) / GEMM2, GEMM3 (10)
constexpr int MD_TN = (MD_NB + MD_NW - 1) / MD_NW;                // n tiles per wave (2)
constexpr int MD_TM = (MD_MB + MD_NW - 1) / MD_NW;                // m tiles per wave (3; tiles >= MB do not exist)
constexpr int MD_D = 5;                                            // groups in flight per wave
constexpr int md_up(int v) { return (v + MD_D - 1) / MD_D * MD_D; }
constexpr int MD_CAP1 = md_up(MD_TN * MD_KB1), MD_CAP3 = md_up(MD_TM * MD_KB2), MD_CAP2 = md_up(MD_TN * MD_KB2);   // stream capacities per wave (groups)
constexpr int MD_LAST = 1 << 16, MD_NULL = 1 << 17;               // stream entry flags

// image layout (4-byte words; a float64 operand is 2 words).  A stream entry: blk | tile << 8 | MD_LAST (last group of the tile);
// streams are padded to multiples of MD_D groups with MD_NULL entries.
//   meta  int [64]              : ng1[8], ng3[8], ng2[8] (padded group counts of wave w's streams), [24] blocks per tile of stream 2,
//                                 [32 + w TN + e] the e-th n tile of wave w (99: none) -- tiles are dealt longest-first over the SIMDs
//   kb    int [8][TB]           : what the solve kernel reads (copied to LDS): one byte per group, blk | 0x40 last | 0x80 null
//   kx1   int [8][CAP1]   kx3 int [8][CAP3]   kx2 int [8][CAP2]   (setup only: the entries with their tiles, for k_pack_mfmad)
//   nzf   int [NB * KB1 + MB * KB2]   non-zero flags of the 16 x 16 blocks (setup scratch)
//   W1    f64 [8][CAP1][64][4]  S'[16 t + i16][16 blk + kq + 4 j],  S = [A (MP rows); H' (NP rows)]
//   W3    f64 [8][CAP3][64][4]  A[16 T + i16][16 blk + kq + 4 j]
//   K     f64 [nrho][8][CAP2][64][4]   K_j[16 t + i16][16 blk + kq + 4 j]
constexpr int MD_TB1 = 2 * (MD_CAP1 / MD_D) + 4, MD_TB3 = 2 * (MD_CAP3 / MD_D) + 4, MD_TB = MD_TB1 + MD_TB3;   // byte tables (dwords)
constexpr size_t MD_OFF_KB = 64;                                   // kb [8][TB]: stream 1 | stream 3 of wave w: 8 BYTES per block of 5 groups
constexpr size_t MD_OFF_KX1 = MD_OFF_KB + 8 * MD_TB, MD_OFF_KX3 = MD_OFF_KX1 + 8 * MD_CAP1, MD_OFF_KX2 = MD_OFF_KX3 + 8 * MD_CAP3;
constexpr size_t MD_OFF_NZ = MD_OFF_KX2 + 8 * MD_CAP2;
constexpr size_t MD_NNZ = (size_t)MD_NB * MD_KB1 + (size_t)MD_MB * MD_KB2;
constexpr size_t MD_OFF_W1 = (MD_OFF_NZ + MD_NNZ + 7) / 8 * 8;      // (32-byte aligned)
constexpr size_t MD_N1 = (size_t)8 * MD_CAP1 * 512, MD_N3 = (size_t)8 * MD_CAP3 * 512, MD_KJ = (size_t)8 * MD_CAP2 * 512;   // words
constexpr size_t MD_OFF_W3 = MD_OFF_W1 + MD_N1, MD_OFF_K = MD_OFF_W3 + MD_N3;
static_assert(MD_OFF_W1 % 8 == 0 && MD_OFF_W3 % 8 == 0 && MD_OFF_K % 8 == 0, "32-byte alignment");

constexpr size_t md_lds_bytes() {
    return (size_t)(MD_KB1 + 2 * MD_NB + MD_MB) * 64 * 32   // V1 | V3 | DV | AD (red, rr of a check alias AD)
           + 64 * 8 + 16 * 8 + 8 * 16 * 4 + 8 * MD_TB * 4;   // rho ladder | rho estimates | inst | stream tables
}

__device__ __forceinline__ double nanmaxd(double a, double b) {          // NaN-propagating max (torch semantics)
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
__device__ __forceinline__ void lds_barrier() {          // orders LDS only: the operand ring's global loads stay in flight
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}   // namespace

// DIAG = true is a separate diagnostic build (RQP_DIAG=1): s_memtime stamps accumulate the ticks each wave spends per segment.
template <bool DIAG>
__global__ void __launch_bounds__(512, 2) k_admm_mfmad(SolveArgs a, const int* __restrict__ img, unsigned long long* __restrict__ dbg) {
    constexpr int NW = MD_NW, NT = MD_NT, NB = MD_NB, MB = MD_MB, TN = MD_TN, TM = MD_TM, D = MD_D;
    extern __shared__ __attribute__((aligned(32))) unsigned char smd[];
    f64x4* V1 = (f64x4*)smd;                 // [KB1][64]: blocks < MB: nu (lam / 0 at a check); blocks MB..: x (0 at a check)
    f64x4* V3 = V1 + MD_KB1 * 64;            // [NB][64] dx (x at the start; H x in the second pass of a check)
    f64x4* DV = V3 + NB * 64;                // [NB][64] d (A' lam in a check)
    f64x4* AD = DV + NB * 64;                // [MB][64] raw GEMM3 results of this wave's m tiles
    double* red = (double*)AD;               // [NW][16][4] row-side maxima per (wave, instance)   (phase 3 runs no GEMM3: AD is free)
    double* rr = red + NW * 16 * 4;          // [NW * 4][16][8] column-side maxima per (wave, kq, instance)
    double* rhosd = (double*)(AD + MB * 64); // [64]
    double* estd = rhosd + 64;               // [16] carried rho estimates
    int* inst_i = (int*)(estd + 16);         // [8][16]: 2 instance, 4 rho index, 5 done
    int* tabs = inst_i + 8 * 16;             // [NW][TB] this wave's stream tables (bytes)
    static_assert((size_t)NW * 16 * 4 * 8 + (size_t)NW * 4 * 16 * 8 * 8 <= (size_t)MB * 64 * 32, "check scratch aliases AD");

    const int n = a.n, m = a.m;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // slot -> instance: SolveArgs.order groups the instances by the rho index they START at (every distinct index among a tile's
    // columns is one more pass over the dense K stream; warm-started batches arrive with their persisted indices)
    const int slot = blockIdx.x * 16 + i16;                  // this lane's slot (MFMA column)
    const bool real = slot < a.B;
    const int id = real ? (a.order ? a.order[slot] : slot) : 0;
    const int kmax = a.max_iter;
    const int* meta = img;
    const int ng1 = __builtin_amdgcn_readfirstlane(meta[wave_u]), ng3 = __builtin_amdgcn_readfirstlane(meta[8 + wave_u]);
    const int ng2 = __builtin_amdgcn_readfirstlane(meta[16 + wave_u]);   // (tiles of the dense stream x nb5)
    // streams: buffer loads -- scalar byte offset from the image + 32 lane (no vector ALU in the address)
    const __amdgpu_buffer_rsrc_t imgr = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, 0x7fffffff, 0x00020000);
    const unsigned oW1 = (unsigned)(MD_OFF_W1 + (size_t)wave_u * MD_CAP1 * 512) * 4u;
    const unsigned oW3 = (unsigned)(MD_OFF_W3 + (size_t)wave_u * MD_CAP3 * 512) * 4u;
    const unsigned oK0 = (unsigned)(MD_OFF_K + (size_t)wave_u * MD_CAP2 * 512) * 4u;              // + j * KJ * 4
    const int* tab1 = tabs + wave_u * MD_TB;
    const int* tab3 = tab1 + MD_TB1;
    const int nb5 = __builtin_amdgcn_readfirstlane(meta[24]);       // blocks per tile of the dense stream (a multiple of D)
    int tN[TN];                                                      // this wave's n tiles (99: none): dealt longest-first at setup
#pragma unroll
    for (int e = 0; e < TN; ++e) tN[e] = __builtin_amdgcn_readfirstlane(meta[32 + wave_u * TN + e]);
    for (int i = lane; i < MD_TB; i += 64) tabs[wave_u * MD_TB + i] = meta[MD_OFF_KB + wave_u * MD_TB + i];
    // (lv: the lane number through an opaque copy per loop iteration -- hipcc otherwise hoists every (array, tile) address of
    //  the unrolled state code out of the solve loop and spills them)
    int lv = lane;

    // ---- scalars
    for (int i = tid; i < a.nrho && i < 64; i += NT) rhosd[i] = a.rhos[i];
    if (tid < 16) {
        const int st = blockIdx.x * 16 + tid;
        const bool ok = st < a.B;
        const int s0 = ok ? st : blockIdx.x * 16;                  // padding columns mirror the tile's first instance
        const int idt = a.order ? a.order[s0] : s0;
        const int ri = a.rho_ind[idt];
        inst_i[2 * 16 + tid] = idt;
        inst_i[4 * 16 + tid] = ri;
        inst_i[5 * 16 + tid] = ok ? 0 : 1;                         // padding columns start "done"
        estd[tid] = a.rhos[ri];                                    // rho_est = rhos[rho_ind]  (:211)
    }
    // ---- state.  Rows: m tiles T = wave + NW tl (< MB), rows 16 T + kq + 4 r of instance i16 (the D layout of the float64 MFMA).
    //      Columns: n tiles t = wave + NW e (< NB).
    double zt[TM][4], zz[TM][4], lm[TM][4], lb[TM][4], ub[TM][4];
    unsigned eqmask = 0;
    double xs[TN][4], gs[TN][4];
#pragma unroll
    for (int tl = 0; tl < TM; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (wave_u + NW * tl) + kq + 4 * r;
            const bool ok = real && row < m;
            const size_t o = (size_t)(real ? id : 0) * m + (row < m ? row : 0);
            zt[tl][r] = 0.0;
            zz[tl][r] = ok ? a.z[o] : 0.0;
            lm[tl][r] = ok ? a.lam[o] : 0.0;
            lb[tl][r] = ok ? ((const double*)a.l)[o] : 0.0;
            ub[tl][r] = ok ? ((const double*)a.u)[o] : 0.0;
            const double cv = (row < m) ? ((const double*)a.c)[o] : 1.0;
            if (cv > 1.0) eqmask |= 1u << (4 * tl + r);
        }
#pragma unroll
    for (int e = 0; e < TN; ++e) {
        const int t = tN[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + kq + 4 * r;
            const bool ok = real && t < NB && row < n;
            const size_t o = (size_t)(real ? id : 0) * n + (row < n ? row : 0);
            xs[e][r] = ok ? a.x[o] : 0.0;
            gs[e][r] = ok ? ((const double*)a.g)[o] : 0.0;
        }
        if (t < NB) {
            const f64x4 xv = {xs[e][0], xs[e][1], xs[e][2], xs[e][3]};
            V3[t * 64 + lane] = xv;                                  // start pass: GEMM3 on x
            V1[(MB + t) * 64 + lane] = xv;
        }
    }

    // ---- the operand ring (rqp_mfmal.hip): D groups of 2 KB in flight, one request per visit into the slot of the visit before,
    //      periodic across the streams; no vector ALU inside the visit loops
    f64x4 o_[D];
    auto fetch = [&](int j, unsigned base) __attribute__((always_inline)) {             // base: uniform byte offset of slot 0's group
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(imgr, (unsigned)lv * 32u, base + 2048u * (unsigned)j, 0);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(imgr, (unsigned)lv * 32u + 16u, base + 2048u * (unsigned)j, 0);
        struct { u32x4 l, h; } pr = {lo, hi};
        o_[j] = __builtin_bit_cast(f64x4, pr);
    };
    auto pro = [&](unsigned off) __attribute__((always_inline)) {      // (slot D - 1 is requested by the first visit)
#pragma unroll
        for (int j = 0; j < D - 1; ++j) fetch(j, off);
    };
    auto body = [&](unsigned offX, int ng, unsigned offY, const int* tab, const f64x4* Bv, f64x4* Out, int ta, int tb, int tc) __attribute__((always_inline)) {
        f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0}, acc2 = acc;
        int kA = __builtin_amdgcn_readfirstlane(tab[0]), kB = __builtin_amdgcn_readfirstlane(tab[1]), kN = __builtin_amdgcn_readfirstlane(tab[2]);
        int tile = ta;                                               // (the stream's tiles: ta, tb, tc)
        f64x4 bn = Bv[(kA & 63) * 64 + lv];                          // the vector operand is read one visit ahead
        auto block = [&](int g0, bool own) __attribute__((always_inline)) {
            const int q = (g0 / D) * 2;
            const int nB = tab[q + 3], nN = tab[q + 4];              // the next block's entries (VGPR copies until the block's end)
            const unsigned pr = own ? offX + 2048u * (unsigned)(g0 + D) : offY;
            const unsigned p0 = offX + 2048u * (unsigned)g0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int e = (j < 4 ? (kA >> (8 * j)) : kB) & 0xff;
                const int en = j < 3 ? (kA >> (8 * (j + 1))) : (j == 3 ? kB : kN);
                const f64x4 b = bn;
                const f64x4 av = o_[j];
                if (j == 0) fetch(D - 1, p0); else fetch(j - 1, pr);  // (fetch adds 2048 x slot)
                bn = Bv[(en & 63) * 64 + lv];                        // (past the stream's end: a valid LDS address, value unused)
                __builtin_amdgcn_sched_barrier(0);                   // (the requests stay ahead of this visit's MFMAs)
                if (!(e & 0x80)) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], b[0], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], b[1], acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], b[2], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], b[3], acc2, 0, 0, 0);
                }
                if (e & 0x40) {                                      // (uniform) last group of a tile
                    Out[tile * 64 + lv] = acc + acc2;
                    acc = (f64x4){0.0, 0.0, 0.0, 0.0};
                    acc2 = acc;
                    tile = (tile == ta) ? tb : tc;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            kA = kN;
            kB = __builtin_amdgcn_readfirstlane(nB);
            kN = __builtin_amdgcn_readfirstlane(nN);
        };
        int g0 = 0;
        for (; g0 + D < ng; g0 += D) block(g0, true);
        block(g0, false);
    };
    // DENSE stream (K_j): ntw tiles of nb5 blocks 0 .. nb5 - 1 each (nb5 = 5 or 10), every LDS offset an immediate; the first operand
    // of a chunk of 5 is read at the end of the chunk before it -- both candidates (the next chunk's, and block 0 for a new tile)
    auto body_dense = [&](unsigned offX, int ntw, unsigned offY, f64x4* Out) __attribute__((always_inline)) {
        f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0}, acc2 = acc;
        int tile = tN[0];
        int left = ntw * (nb5 / D);                                  // chunks to go
        unsigned px = offX;
        f64x4 bw = DV[lv], bc = bw;
        for (int tl = 0; tl < ntw; ++tl) {
#pragma unroll
            for (int c = 0; c < NB / D; ++c) {
                if (c * D < nb5) {                                   // (uniform)
                    left -= 1;
                    const unsigned pr = left ? px + 2048u * D : offY;
                    f64x4 bn = c == 0 ? bw : bc;
#pragma unroll
                    for (int j = 0; j < D; ++j) {                    // visit: MFMA, vector read, MFMA, request, MFMA, MFMA
                        const int pb = c * D + j;
                        const f64x4 b = bn;
                        const f64x4 av = o_[j];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], b[0], acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        bn = DV[(pb + 1) * 64 + lv];                 // (block NB: the array behind DV, unused)
                        if (j == D - 1) { bc = bn; bw = DV[lv]; }
                        __builtin_amdgcn_sched_barrier(0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], b[1], acc2, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (j == 0) fetch(D - 1, px); else fetch(j - 1, pr);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], b[2], acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], b[3], acc2, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    px += 2048u * D;
                }
            }
            Out[tile * 64 + lv] = acc + acc2;
            acc = (f64x4){0.0, 0.0, 0.0, 0.0};
            acc2 = acc;
            tile = tN[TN - 1];
        }
    };
    const int ntw2 = ng2 / nb5;
    pro(oW3);
    __syncthreads();
    int ri_l = inst_i[4 * 16 + i16];
    double rho_ne = 1.0, rho_eq = 1.0, inv_ne = 1.0, inv_eq = 1.0;
    auto set_rho = [&]() __attribute__((always_inline)) {
        rho_ne = rhosd[ri_l];
        rho_eq = rho_ne * 1e3;
        inv_ne = 1.0 / rho_ne;
        inv_eq = 1.0 / rho_eq;
    };
    set_rho();

    // lam_hat and nu of the next iteration from the current state (p = A x - z) -> V1
    auto make_nu = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tl = 0; tl < TM; ++tl) {
            if (wave_u + NW * tl < MB) {
                f64x4 nu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                    const double p = zt[tl][r] - zz[tl][r];
                    const double lh = lm[tl][r] + rho * p;
                    lm[tl][r] = lh;
                    nu[r] = lh + rho * p;
                }
                V1[(wave_u + NW * tl) * 64 + lv] = nu;
            }
        }
    };

    // phases as in rqp_mfmal.hip: 0 start (GEMM3 on x -> A x), 1 iterate, 2 check part 1 (A' lam), 3 check part 2 (H x, decisions)
    int ph = 0, k = 0, to_chk = a.check_interval;
    bool final_chk = false;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;

    unsigned long long t_last = 0, t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int seg) __attribute__((always_inline)) {
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (seg >= 0 && ph == 1) t_acc[seg] += t - t_last;
            t_last = t;
        }
    };
    stamp(-1);
    while (true) {
        asm volatile("" : "+v"(lv));
        lds_barrier();
        stamp(0);
        unsigned long long todo = 0;
        int jr = 0;
        if (ph != 0) {                                   // ---------------- GEMM1: S' V1 for this wave's n tiles
            if (ph == 1) {                                           // (GEMM2's first groups are requested during GEMM1)
                todo = __ballot(lane < 16);
                jr = __builtin_amdgcn_readlane(ri_l, __ffsll((long long)todo) - 1);
            }
            body(oW1, ng1, ph == 1 ? (ntw2 ? oK0 + (unsigned)jr * (unsigned)(MD_KJ * 4) : oW3) : oW1, tab1, V1, ph == 3 ? V3 : DV, tN[0], tN[TN - 1], 0);
            stamp(1);
            if (ph == 1) {
#pragma unroll
                for (int e = 0; e < TN; ++e)                         // d = H x + g + A' nu
                    if (tN[e] < NB) {
                        f64x4 d = DV[tN[e] * 64 + lv];
                        d[0] += gs[e][0]; d[1] += gs[e][1]; d[2] += gs[e][2]; d[3] += gs[e][3];
                        DV[tN[e] * 64 + lv] = d;
                    }
            }
            stamp(2);
            lds_barrier();                                           // V1 is free again (ph 2 rewrites it); d is visible (ph 1)
            stamp(3);
        }
        bool run_g3 = (ph == 0);
        if (ph == 1) {                                   // ---------------- GEMM2: dx = -K_j d for this wave's n tiles, K_j per column
            double sel[TN][4];
            bool first_pass = true;
            while (true) {                                           // one pass per distinct rho index of the tile (usually one)
                const bool mine = (ri_l == jr);
                const int jc = jr;
                todo &= ~__ballot(lane < 16 && ri_l == jr);
                if (todo) jr = __builtin_amdgcn_readlane(ri_l, __ffsll((long long)todo) - 1);
                body_dense(oK0 + (unsigned)jc * (unsigned)(MD_KJ * 4), ntw2, todo ? oK0 + (unsigned)jr * (unsigned)(MD_KJ * 4) : oW3, V3);
                if constexpr (DIAG) t_acc[10] += 1;                  // K passes (one per distinct rho index of the tile)
                stamp(4);
#pragma unroll
                for (int e = 0; e < TN; ++e)
                    if (tN[e] < NB) {
                        const f64x4 v = V3[tN[e] * 64 + lv];
#pragma unroll
                        for (int r = 0; r < 4; ++r) sel[e][r] = (first_pass || mine) ? v[r] : sel[e][r];
                    }
                first_pass = false;
                if (!todo) break;
            }
#pragma unroll
            for (int e = 0; e < TN; ++e) {
                const int t = tN[e];
                if (t < NB) {
                    f64x4 dx, xv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dx[r] = -sel[e][r];
                        xs[e][r] += dx[r];
                        xv[r] = xs[e][r];
                    }
                    V3[t * 64 + lv] = dx;
                    V1[(MB + t) * 64 + lv] = xv;
                }
            }
            stamp(5);
            lds_barrier();
            stamp(6);
            run_g3 = true;
        }
        bool nu_done = false;
        if (run_g3) {                                    // ---------------- GEMM3: A V3 for this wave's m tiles + row update
            const bool upd = (ph == 1);
            const bool fin_next = upd && (k + 1 >= kmax) && (to_chk != 1);
            const bool with_nu = upd && to_chk != 1 && !fin_next;
            body(oW3, ng3, oW1, tab3, V3, AD, wave_u, wave_u + NW, wave_u + 2 * NW);
            stamp(7);
#pragma unroll
            for (int tl = 0; tl < TM; ++tl) {
                const int T = wave_u + NW * tl;
                if (T < MB) {
                    const f64x4 acc = AD[T * 64 + lv];
                    double pp[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        zt[tl][r] += acc[r];                 // A x (float64 accumulator)
                        const bool eq = (eqmask >> (4 * tl + r)) & 1u;
                        const double v = zt[tl][r] + lm[tl][r] * (eq ? inv_eq : inv_ne);
                        double zn = v;                       // torch.clamp: NaN stays NaN
                        if (v < lb[tl][r]) zn = lb[tl][r];
                        if (v > ub[tl][r]) zn = ub[tl][r];
                        zn = upd ? zn : zz[tl][r];
                        zz[tl][r] = zn;
                        pp[r] = zt[tl][r] - zn;              // p = A x - z of the new state
                    }
                    if (with_nu) {                           // lam_hat, nu of the next iteration
                        f64x4 nu;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                            const double lh = lm[tl][r] + rho * pp[r];
                            lm[tl][r] = lh;
                            nu[r] = lh + rho * pp[r];
                        }
                        V1[T * 64 + lv] = nu;
                    }
                }
            }
            stamp(8);
            if (upd) {
                if constexpr (DIAG) t_acc[11] += 1;
                k += 1;
                to_chk -= 1;
            }
            nu_done = with_nu;
        }
        // ---------------------------------------------------------------------------------- what comes next
        if (ph == 0 || ph == 1) {
            final_chk = (ph == 0) ? (kmax == 0) : (k >= kmax && to_chk != 0);
            const bool chk = (ph == 1 && to_chk == 0) || final_chk;                       // :218 (Q3 fixed) / :243
            if (to_chk == 0) to_chk = a.check_interval;
            if (!chk) {
                if (!nu_done) make_nu();
                ph = 1;
            } else {                                     // check part 1: V1 = [lam; 0], row-side maxima
                v0 = 0.0; v1 = 0.0; v2 = 0.0;
                int kq_o = kq;
                asm volatile("" : "+v"(kq_o));
#pragma unroll
                for (int tl = 0; tl < TM; ++tl) {
                    if (wave_u + NW * tl < MB) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * (wave_u + NW * tl) + kq_o + 4 * r;
                            const double we = (a.scE && row < m) ? 1.0 / a.scE[row] : 1.0;   // (Ruiz: caller-space norms)
                            v0 = nanmaxd(v0, fabs(zt[tl][r] - zz[tl][r]) * we);
                            v1 = nanmaxd(v1, fabs(zt[tl][r]) * we);
                            v2 = nanmaxd(v2, fabs(zz[tl][r]) * we);
                        }
                        V1[(wave_u + NW * tl) * 64 + lv] = (f64x4){lm[tl][0], lm[tl][1], lm[tl][2], lm[tl][3]};
                    }
                }
#pragma unroll
                for (int e = 0; e < TN; ++e)
                    if (tN[e] < NB) V1[(MB + tN[e]) * 64 + lv] = (f64x4){0.0, 0.0, 0.0, 0.0};
                ph = 2;
            }
        } else if (ph == 2) {                            // (A' lam is in DV) ; V1 = [0; x]
#pragma unroll
            for (int tl = 0; tl < TM; ++tl)
                if (wave_u + NW * tl < MB) V1[(wave_u + NW * tl) * 64 + lv] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < TN; ++e)
                if (tN[e] < NB) V1[(MB + tN[e]) * 64 + lv] = (f64x4){xs[e][0], xs[e][1], xs[e][2], xs[e][3]};
            ph = 3;
        } else {                                         // ph == 3: residuals and decisions (H x is in V3, A' lam in DV)
            double w3 = 0.0, w4 = 0.0, w5 = 0.0, w6 = 0.0, jp = 0.0;
            {
                int kq_o = kq;
                asm volatile("" : "+v"(kq_o));                       // (keeps the rare per-row address arithmetic inside the branch)
#pragma unroll
                for (int e = 0; e < TN; ++e) {
                    const int t = tN[e];
                    if (t < NB) {
                        const f64x4 t2v = V3[t * 64 + lv], t3v = DV[t * 64 + lv];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                // column-side maxima of compute_residuals (:313-316), objective
                            const int row = 16 * t + kq_o + 4 * r;
                            const double wd = (a.scD && row < n) ? 1.0 / (a.scC[0] * a.scD[row]) : 1.0;   // (Ruiz: caller-space norms)
                            const double t2 = t2v[r], t3 = t3v[r], ge = gs[e][r];
                            w3 = nanmaxd(w3, fabs(t2 + t3 + ge) * wd);
                            w4 = nanmaxd(w4, fabs(t2) * wd);
                            w5 = nanmaxd(w5, fabs(t3) * wd);
                            w6 = nanmaxd(w6, fabs(ge) * wd);
                            jp += xs[e][r] * (0.5 * t2 + ge);        // compute_J :320-322
                        }
                    }
                }
            }
            v0 = nanmaxd(v0, __shfl_xor(v0, 16, 64)); v0 = nanmaxd(v0, __shfl_xor(v0, 32, 64));
            v1 = nanmaxd(v1, __shfl_xor(v1, 16, 64)); v1 = nanmaxd(v1, __shfl_xor(v1, 32, 64));
            v2 = nanmaxd(v2, __shfl_xor(v2, 16, 64)); v2 = nanmaxd(v2, __shfl_xor(v2, 32, 64));
            if (kq == 0) {
                red[(wave * 16 + i16) * 4 + 0] = v0;
                red[(wave * 16 + i16) * 4 + 1] = v1;
                red[(wave * 16 + i16) * 4 + 2] = v2;
            }
            {
                double* q = rr + ((wave * 4 + kq) * 16 + i16) * 8;
                q[3] = w3; q[4] = w4; q[5] = w5; q[6] = w6; q[7] = jp;
            }
            lds_barrier();
            if (tid < 16) {                                          // one thread per instance decides
                const int j = tid;
                const double tolT = a.tol, thr_p = a.thr_p, thr_d = a.thr_d, rmin = a.rho_min, rmax = a.rho_max;
                double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0, q4 = 0.0, q5 = 0.0, q6 = 0.0, obj = 0.0;
                for (int w = 0; w < NW; ++w) {
                    q0 = nanmaxd(q0, red[(w * 16 + j) * 4 + 0]);
                    q1 = nanmaxd(q1, red[(w * 16 + j) * 4 + 1]);
                    q2 = nanmaxd(q2, red[(w * 16 + j) * 4 + 2]);
                }
                for (int g2 = 0; g2 < NW * 4; ++g2) {
                    q3 = nanmaxd(q3, rr[(g2 * 16 + j) * 8 + 3]);
                    q4 = nanmaxd(q4, rr[(g2 * 16 + j) * 8 + 4]);
                    q5 = nanmaxd(q5, rr[(g2 * 16 + j) * 8 + 5]);
                    q6 = nanmaxd(q6, rr[(g2 * 16 + j) * 8 + 6]);
                    obj += rr[(g2 * 16 + j) * 8 + 7];
                }
                const bool alive = inst_i[5 * 16 + j] == 0;
                if (alive) {
                    const double num = q0 / nanmaxd(q1, q2);                              // :315
                    const double den = q3 / nanmaxd(nanmaxd(q4, q5), q6);                 // :316
                    double est = estd[j] * sqrt(num / den);                               // :317 (Q4: carried)
                    if (est < rmin) est = rmin;                                           // torch.clamp: NaN stays NaN
                    if (est > rmax) est = rmax;
                    int ri = inst_i[4 * 16 + j];
                    const int ri_before = ri;
                    if (!final_chk) {
                        if (est > rhosd[ri] * tolT && ri < a.nrho - 1) ri += 1;           // :223
                        else if (est < rhosd[ri] / tolT && ri > 0) ri -= 1;               // :226
                    }
                    estd[j] = est;
                    inst_i[4 * 16 + j] = ri;
                    const int idj = inst_i[2 * 16 + j];
                    const int chk_no = k / a.check_interval;
                    if (!final_chk && a.info.trace && chk_no <= a.info.trace_cap) {
                        double* tr = a.info.trace + ((size_t)idj * a.info.trace_cap + (chk_no - 1)) * 4;
                        tr[0] = q0; tr[1] = q3; tr[2] = est; tr[3] = (double)ri_before;
                    }
                    const double er = a.eps_rel;
                    const double tp = er > 0.0 ? thr_p + er * nanmaxd(q1, q2) : thr_p;
                    const double td = er > 0.0 ? thr_d + er * nanmaxd(nanmaxd(q4, q5), q6) : thr_d;
                    const bool conv = !final_chk && (q0 < tp && q3 < td);                 // :233
                    const bool last = final_chk || k >= kmax;                             // :243 max-iter fallthrough
                    if (conv || last) {
                        double est_out = est;
                        if (!conv && !final_chk) {                   // max_iter on the check grid: the reference's extra compute_residuals (:243)
                            est_out = est * sqrt(num / den);
                            if (est_out < rmin) est_out = rmin;
                            if (est_out > rmax) est_out = rmax;
                        }
                        inst_i[5 * 16 + j] = 2;                      // exits now (2: outputs due; 1: gone)
                        const size_t bj = (size_t)idj;
                        if (a.info.iter) a.info.iter[bj] = conv ? k : a.max_iter;
                        if (a.info.status) a.info.status[bj] = conv ? RQP_STATUS_SOLVED : ((q0 != q0 || q3 != q3) ? RQP_STATUS_NAN : RQP_STATUS_MAX_ITER);
                        if (a.info.rho_ind) a.info.rho_ind[bj] = ri;
                        if (a.info.pri_res) a.info.pri_res[bj] = q0;
                        if (a.info.dua_res) a.info.dua_res[bj] = q3;
                        if (a.info.rho_estimate) a.info.rho_estimate[bj] = est_out;
                        if (a.info.obj_val) a.info.obj_val[bj] = obj;
                        a.rho_ind[bj] = (a.warm_starting || a.keep_state) ? ri : a.rho_ind0;
                    }
                }
                // columns without a live instance take the rho index of a live one (no K pass of their own)
                const bool live = inst_i[5 * 16 + j] == 0;
                const int ri_now = inst_i[4 * 16 + j];
                const unsigned long long lm16 = __ballot(live);
                if (lm16) {
                    const int ri_live = __shfl(ri_now, __ffsll((long long)lm16) - 1, 64);
                    if (!live) inst_i[4 * 16 + j] = ri_live;
                }
            }
            __syncthreads();
            if (inst_i[5 * 16 + i16] == 2) {             // this lane's instance just finished: x, z, lam out + the persistent state (:278-305)
                const bool ws = (a.warm_starting || a.keep_state) != 0;
                int kq_o = kq, id_o = id;
                asm volatile("" : "+v"(kq_o), "+v"(id_o));
#pragma unroll
                for (int e = 0; e < TN; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int t = tN[e], row = 16 * t + kq_o + 4 * r;
                        if (t < NB && row < n) {
                            const size_t o = (size_t)id_o * n + row;
                            if (a.out_x) ((double*)a.out_x)[o] = xs[e][r];
                            a.x[o] = ws ? xs[e][r] : 0.0;
                        }
                    }
#pragma unroll
                for (int tl = 0; tl < TM; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * (wave_u + NW * tl) + kq_o + 4 * r;
                        if (row < m) {
                            const size_t o = (size_t)id_o * m + row;
                            if (a.out_z) ((double*)a.out_z)[o] = zz[tl][r];
                            if (a.out_lam) ((double*)a.out_lam)[o] = lm[tl][r];
                            a.z[o] = ws ? zz[tl][r] : 0.0;
                            a.lam[o] = ws ? lm[tl][r] : 0.0;
                        }
                    }
            }
            __syncthreads();
            if (tid < 16 && inst_i[5 * 16 + tid] == 2) inst_i[5 * 16 + tid] = 1;
            __syncthreads();
            ri_l = inst_i[4 * 16 + i16];
            set_rho();
            int nd = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) nd += (inst_i[5 * 16 + j] != 0);
            if (nd == 16) break;                         // every instance of the tile has exited
            make_nu();                                   // (V1's x rows hold x again since ph 2)
            ph = 1;
        }
        if (ph == 1) stamp(9); else stamp(-1);
    }
    if constexpr (DIAG) {
        if (lane == 0)
            for (int e = 0; e < 12; ++e) dbg[((size_t)blockIdx.x * NW + wave) * 12 + e] = t_acc[e];
    }
}

// ---------------------------------------------------------------------------- setup: streams + operand images
__device__ __forceinline__ double md_s_elem(int n, int m, int ldn, const double* A, const double* Ht, int k, int col) {   // S[k][col]
    if (col >= n) return 0.0;
    if (k < MD_MP) return k < m ? A[(size_t)k * ldn + col] : 0.0;
    return (k - MD_MP < n) ? Ht[(size_t)(k - MD_MP) * ldn + col] : 0.0;
}
// one wave per 16 x 16 block: is any element non-zero?
__global__ void k_nz_mfmad(int n, int m, int ldn, const double* __restrict__ A, const double* __restrict__ Ht, int* __restrict__ img) {
    int* nzf = img + MD_OFF_NZ;
    const int b = blockIdx.x, lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    bool nz = false;
    if (b < MD_NB * MD_KB1) {
        const int t = b / MD_KB1, blk = b % MD_KB1;
        for (int j = 0; j < 4; ++j) nz |= md_s_elem(n, m, ldn, A, Ht, 16 * blk + 4 * q + j, 16 * t + i) != 0.0;
    } else {
        const int T = (b - MD_NB * MD_KB1) / MD_KB2, blk = (b - MD_NB * MD_KB1) % MD_KB2;
        const int r = 16 * T + i;
        for (int j = 0; j < 4; ++j) {
            const int c = 16 * blk + 4 * q + j;
            nz |= (r < m && c < n) && A[(size_t)r * ldn + c] != 0.0;
        }
    }
    const unsigned long long any = __ballot(nz);
    if (lane == 0) nzf[b] = any != 0ull;
}
// one thread per wave of the solve kernel: its three streams (tiles back to back; a tile without a non-zero block gets one
// group -- block 0, whose operand is then zero -- so that its result is written)
__global__ void k_meta_mfmad(int n, int m, int* __restrict__ img) {
    int* meta = img;
    const int* nzf = meta + MD_OFF_NZ;
    const int w = threadIdx.x;
    const int nbt = (n + 15) / 16;
    // n tiles are dealt to the waves longest-first (by their GEMM1 group counts) in snake order over the SIMDs -- waves w and w + 4
    // share a SIMD: ranks 0..3 -> waves 0..3, ranks 4..7 -> waves 7..4 (largest beside smallest), the next round reversed
    if (w == 0) {
        int cnt[MD_NB], ord[MD_NB];
        for (int t = 0; t < MD_NB; ++t) {
            int c = 0;
            for (int blk = 0; blk < MD_KB1; ++blk) c += nzf[t * MD_KB1 + blk] != 0;
            cnt[t] = c;
            ord[t] = t;
        }
        for (int i = 1; i < MD_NB; ++i)                              // (stable insertion sort, descending)
            for (int q = i; q > 0 && cnt[ord[q]] > cnt[ord[q - 1]]; --q) { const int tmp = ord[q]; ord[q] = ord[q - 1]; ord[q - 1] = tmp; }
        for (int i = 0; i < MD_NW * MD_TN; ++i) meta[32 + i] = 99;
        for (int r = 0; r < MD_NB; ++r) {
            const int round = r / MD_NW, pos2 = r % MD_NW;
            int wv = pos2 < 4 ? pos2 : 11 - pos2;                    // 0 1 2 3 7 6 5 4
            if (round & 1) wv = MD_NW - 1 - wv;
            meta[32 + wv * MD_TN + round] = ord[r];
        }
    }
    __syncthreads();
    if (w >= MD_NW) return;
    int* kx = meta + MD_OFF_KX1 + w * MD_CAP1;
    int pos = 0;
    for (int e = 0; e < MD_TN; ++e) {
        const int t = meta[32 + w * MD_TN + e];
        if (t >= MD_NB) break;
        const int first = pos;
        for (int blk = 0; blk < MD_KB1; ++blk)
            if (nzf[t * MD_KB1 + blk]) kx[pos++] = blk | (t << 8);
        if (pos == first) kx[pos++] = 0 | (t << 8);
        kx[pos - 1] |= MD_LAST;
    }
    while (pos % MD_D) kx[pos++] = MD_NULL;
    meta[w] = pos;
    kx = meta + MD_OFF_KX3 + w * MD_CAP3;
    pos = 0;
    for (int tl = 0; tl < MD_TM; ++tl) {
        const int T = w + MD_NW * tl;
        if (T >= MD_MB) break;
        const int first = pos;
        for (int blk = 0; blk < MD_KB2; ++blk)
            if (nzf[MD_NB * MD_KB1 + T * MD_KB2 + blk]) kx[pos++] = blk | (T << 8);
        if (pos == first) kx[pos++] = 0 | (T << 8);
        kx[pos - 1] |= MD_LAST;
    }
    while (pos % MD_D) kx[pos++] = MD_NULL;
    meta[8 + w] = pos;
    kx = meta + MD_OFF_KX2 + w * MD_CAP2;
    pos = 0;
    const int nb5 = (nbt + MD_D - 1) / MD_D * MD_D;
    for (int e = 0; e < MD_TN; ++e) {                                 // K_j is dense: nb5 blocks for each tile of the problem's own n
        const int t = meta[32 + w * MD_TN + e];                       // (real tiles rank before the all-zero ones: they come first)
        if (t >= nbt) break;
        for (int blk = 0; blk < nb5; ++blk) kx[pos++] = blk | (t << 8) | (blk >= nbt ? MD_NULL : 0);
    }
    meta[16 + w] = pos;
    if (w == 0) meta[24] = nb5;
    // the byte tables the solve kernel reads (streams 1 and 3): 8 bytes per block of MD_D groups
    unsigned char* kb = (unsigned char*)(meta + MD_OFF_KB + w * MD_TB);
    for (int st = 0; st < 2; ++st) {
        const int* src = meta + (st == 0 ? MD_OFF_KX1 + w * MD_CAP1 : MD_OFF_KX3 + w * MD_CAP3);
        const int cnt = meta[(st == 0 ? 0 : 8) + w];
        const int tbd = st == 0 ? MD_TB1 : MD_TB3;
        for (int i = 0; i < 4 * tbd; ++i) {
            const int blkno = i >> 3, j = i & 7, g = blkno * MD_D + j;
            unsigned char v = 0x80;
            if (j < MD_D && g < cnt) {
                const int kd = src[g];
                v = (unsigned char)((kd & 63) | ((kd & MD_LAST) ? 0x40 : 0) | ((kd & MD_NULL) ? 0x80 : 0));
            }
            kb[i] = v;
        }
        kb += 4 * tbd;
    }
}

// operand images of the streams (after k_meta_mfmad); groups past a stream's end are zero.  Element j of lane (kq, i16) of a group
// is column 16 blk + kq + 4 j of row 16 tile + i16 (the k order of the float64 MFMA's D layout)
__global__ void k_pack_mfmad(int n, int m, int ldn, int nrho, const double* __restrict__ A, const double* __restrict__ Ht,
                             const double* __restrict__ K, int* __restrict__ img) {
    const int* meta = img;
    double* W1 = (double*)(img + MD_OFF_W1);
    double* W3 = (double*)(img + MD_OFF_W3);
    double* Kimg = (double*)(img + MD_OFF_K);
    const size_t n1 = MD_N1 / 2, n3 = MD_N3 / 2, kj = MD_KJ / 2, nk = (size_t)nrho * kj;       // (elements)
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n1 + n3 + nk; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 3), l = (int)((idx >> 2) & 63), i16 = l & 15, kq = l >> 4;
        if (idx < n1) {
            const int grp = (int)(idx >> 8), w = grp / MD_CAP1, pos = grp % MD_CAP1;
            double v = 0.0;
            if (pos < meta[w]) {
                const int kd = meta[MD_OFF_KX1 + grp], blk = kd & 0xff, t = (kd >> 8) & 0xff;
                if (!(kd & MD_NULL)) v = md_s_elem(n, m, ldn, A, Ht, 16 * blk + kq + 4 * j, 16 * t + i16);
            }
            W1[idx] = v;
        } else if (idx < n1 + n3) {
            const size_t o = idx - n1;
            const int grp = (int)(o >> 8), w = grp / MD_CAP3, pos = grp % MD_CAP3;
            double v = 0.0;
            if (pos < meta[8 + w]) {
                const int kd = meta[MD_OFF_KX3 + grp], blk = kd & 0xff, T = (kd >> 8) & 0xff;
                const int r = 16 * T + i16, c = 16 * blk + kq + 4 * j;
                if (r < m && c < n && !(kd & MD_NULL)) v = A[(size_t)r * ldn + c];
            }
            W3[o] = v;
        } else {
            const size_t o = idx - n1 - n3;
            const int jr = (int)(o / kj), grp = (int)((o % kj) >> 8), w = grp / MD_CAP2, pos = grp % MD_CAP2;
            double v = 0.0;
            if (pos < meta[16 + w]) {
                const int kd = meta[MD_OFF_KX2 + grp], blk = kd & 0xff, t = (kd >> 8) & 0xff;
                const int r = 16 * t + i16, c = 16 * blk + kq + 4 * j;
                if (r < n && c < n && !(kd & MD_NULL)) v = K[((size_t)jr * n + r) * ldn + c];
            }
            Kimg[o] = v;
        }
    }
}

// ------------------------------------------------------------------------------ host side
bool rqp_mfmad_fits(const rqp_handle* h) {
    return h->esz == 8 && h->dims.shared_mats && h->n <= MD_NP && h->m <= MD_MP && h->nrho <= 64;
}
size_t rqp_mfmad_img_elems(const rqp_handle* h) { return MD_OFF_K + (size_t)h->nrho * MD_KJ; }      // (4-byte words)

hipError_t rqp_launch_pack_mfmad(const rqp_handle* h, hipStream_t s) {
    int* img = (int*)h->W1img;
    hipError_t e = hipMemsetAsync(img, 0, MD_OFF_W1 * sizeof(int), s);
    if (e != hipSuccess) return e;
    k_nz_mfmad<<<(unsigned)MD_NNZ, 64, 0, s>>>(h->n, h->m, h->ldn, (const double*)h->A, (const double*)h->Ht, img);
    k_meta_mfmad<<<1, 64, 0, s>>>(h->n, h->m, img);
    k_pack_mfmad<<<1024, 256, 0, s>>>(h->n, h->m, h->ldn, h->nrho, (const double*)h->A, (const double*)h->Ht, (const double*)h->K, img);
    return hipGetLastError();
}
hipError_t rqp_prepare_mfmad(const rqp_handle* h) {
    hipError_t e = rqp_raise_lds_limit((const void*)k_admm_mfmad<false>, md_lds_bytes());
    if (e == hipSuccess && (h->debug & 2)) e = rqp_raise_lds_limit((const void*)k_admm_mfmad<true>, md_lds_bytes());
    return e;
}
hipError_t rqp_launch_solve_mfmad(const rqp_handle* h, const SolveArgs& a0, hipStream_t s) {
    const int grid = (h->B + 15) / 16;
    SolveArgs a = a0;
    if (grid > 1 && h->order_d) {                // slots grouped by the rho index the instances start at (rqp_mfmal.hip)
        hipError_t e = rqp_launch_order_by(h, h->rho_ind, s);
        if (e != hipSuccess) return e;
        a.order = h->order_d;
    }
    const int* img = (const int*)h->W1img;
    if (h->debug & 2) {          // diagnostic build: per-segment tick shares of the iteration (synchronous, debug only)
        unsigned long long* dbg = nullptr;
        const size_t cnt = (size_t)grid * MD_NW * 12;
        if (hipMalloc((void**)&dbg, cnt * 8) != hipSuccess) return hipErrorOutOfMemory;
        k_admm_mfmad<true><<<grid, MD_NT, md_lds_bytes(), s>>>(a, img, dbg);
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> hb(cnt);
        (void)hipMemcpy(hb.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        static const char* names[10] = {"top wait", "GEMM1", "d", "wait", "GEMM2", "x", "wait", "GEMM3", "rows", "next"};
        for (int w = 0; w < MD_NW; ++w) {
            double tot[12] = {0};
            for (int t = 0; t < grid; ++t)
                for (int e2 = 0; e2 < 12; ++e2) tot[e2] += (double)hb[((size_t)t * MD_NW + w) * 12 + e2];
            fprintf(stderr, "[rqp diag mfmad] wave %d, %.1f iterations/workgroup, s_memtime ticks per iteration:", w, tot[11] / grid);
            double it = 0;
            for (int e2 = 0; e2 < 10; ++e2) {
                fprintf(stderr, "  %s %.1f", names[e2], tot[e2] / tot[11]);
                it += tot[e2];
            }
            fprintf(stderr, "  | sum %.1f  | K passes per iteration %.3f\n", it / tot[11], tot[10] / tot[11]);
        }
        return hipGetLastError();
    }
    k_admm_mfmad<false><<<grid, MD_NT, md_lds_bytes(), s>>>(a, img, nullptr);
    return hipGetLastError();
}
