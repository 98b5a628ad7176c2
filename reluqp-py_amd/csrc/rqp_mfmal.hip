// rqp_mfmal.hip -- shared-(H, A) ADMM kernel for LARGE / SPARSE problems: n <= 320, m <= 640 (the reference's own linear-MPC
// form is the sparse one, loose_code/RandomLinMPC.py:54-66: N = 20, nx = 12, nu = 4 -> n = 320, m = 560; 2.4 % of A and 0.4 % of H
// are non-zero).  k_admm_mfma pins its operands in registers (n <= 80, m <= 320) and such a batch fell to the streaming
// kernel: 196-240 k QP/s, every instance re-reading 2.2 MB of matrices per iteration.
//
// Same recurrence, checks, quirk dispositions and per-instance semantics as k_admm_mfma (rqp_mfma.hip; reference line citations
// in rqp_admm.hip); the batch is again the N = 16 axis of v_mfma_f32_16x16x4_f32 (exact float32 products).  What differs:
//   * Operands are STREAMED from L2.  The unit is a GROUP: one 16 x 16 block of a matrix = 4 MFMAs, one buffer_load_dwordx4 per
//     lane (1 KB per wave, lane-linear) and one ds_read_b128 of the vector operand.  Every wave walks ONE stream of groups per
//     GEMM (its tiles back to back; a flag on a tile's last group sends the accumulator to LDS and clears it), requested
//     ML_D - 1 = 4 groups ahead through a register ring that stays in flight across the LDS barriers (the last visits of a
//     GEMM request the first groups of the next one).  GEMM1 ([A; H']' [nu; x]) and GEMM3 (A dx) stream only the NON-ZERO blocks
//     of their matrices (k_nz_mfmal / k_meta_mfmal at setup); GEMM2 (K_j d) is dense over ceil(n / 16) blocks.
//   * No vector-ALU instruction inside the visit loops (the float32 MFMA shares its lanes with the VALU: one address v_add per
//     visit cost 38 of a visit's 128 MFMA cycles): scalar operand offsets, compile-time LDS offsets in the dense stream.
//   * The k index of MFMA j of a group is 16 blk + 4 kq + j (kq = lane / 16): the rows a lane supplies as the B operand are the
//     rows it holds of a D result (16 T + 4 kq + r).  Vectors live in LDS as [block][lane] float4 -- results are written and
//     operands read as lane-linear b128 accesses, no swizzle, no bank conflicts.
//   * No K split and no partial exchange: an n tile (GEMM1, GEMM2) or m tile (GEMM3) belongs to ONE wave.  The owner of an n
//     tile keeps its x, g in registers in the D layout and produces d, dx, x for it; the column-side maxima of a check are
//     formed in that layout too.  Three LDS barriers per iteration.
//   * 8 waves (512 threads, two per SIMD: the partner wave's MFMAs cover a wave's epilogue); n tiles t = wave + 8 e, m tiles
//     T = wave + 8 tl.  Row state (z, lam, float-float A x, l, u) of 5 m tiles per wave in registers.
//   * Static tiles: one 16-instance tile per workgroup, no refill queue and no hand-off (iteration counts of a batch of
//     MPC problems from one plant are within a check or two of each other).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rqp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ML_NW = 8, ML_NT = 512, ML_NB = 20, ML_MB = 40;      // waves, threads, n tiles, m tiles
constexpr int ML_NP = 16 * ML_NB, ML_MP = 16 * ML_MB;
constexpr int ML_KB1 = ML_MB + ML_NB, ML_KB2 = ML_NB;              // k blocks of GEMM1 (60: A's rows, then H's) / GEMM2, GEMM3 (20)
constexpr int ML_TN = (ML_NB + ML_NW - 1) / ML_NW;                // n tiles per wave (3)
constexpr int ML_TM = ML_MB / ML_NW;                               // m tiles per wave (5)
constexpr int ML_D = 5;                                            // groups in flight per wave
constexpr int ml_up(int v) { return (v + ML_D - 1) / ML_D * ML_D; }
constexpr int ML_CAP1 = ml_up(ML_TN * ML_KB1), ML_CAP3 = ml_up(ML_TM * ML_KB2), ML_CAP2 = ml_up(ML_TN * ML_KB2);   // stream capacities per wave (groups)
constexpr int ML_LAST = 1 << 16, ML_NULL = 1 << 17;               // stream entry flags
static_assert(ML_MB % ML_NW == 0, "tiling");

// image layout (4-byte words).  A stream entry: blk | tile << 8 | ML_LAST (last group of the tile); streams are padded to
// multiples of ML_D groups with ML_NULL entries (no MFMAs; they keep the ring's slot order across streams).
//   meta  int [32]              : ng1[8], ng3[8], ng2[8] (padded group counts of wave w's streams), [24] blocks per tile of stream 2
//   kb    int [8][TB]           : what the solve kernel reads (copied to LDS): one byte per group, blk | 0x40 last | 0x80 null;
//                                 stream 2 is dense and regular (tiles x nb5 blocks 0 .. nb5-1): it needs no table
//   kx1   int [8][CAP1]   kx3 int [8][CAP3]   kx2 int [8][CAP2]   (setup only: the entries with their tiles, for k_pack_mfmal)
//   nzf   int [NB * KB1 + MB * KB2]   non-zero flags of the 16 x 16 blocks (setup scratch)
//   W1    f32 [8][CAP1][64][4]  S'[16 t + i16][16 blk + 4 kq + j],  S = [A (MP rows); H' (NP rows)]
//   W3    f32 [8][CAP3][64][4]  A[16 T + i16][16 blk + 4 kq + j]
//   K     f32 [nrho][8][CAP2][64][4]   K_j[16 t + i16][16 blk + 4 kq + j]
constexpr int ML_TB1 = 2 * (ML_CAP1 / ML_D) + 4, ML_TB3 = 2 * (ML_CAP3 / ML_D) + 4, ML_TB = ML_TB1 + ML_TB3;   // byte tables (dwords)
constexpr size_t ML_OFF_KB = 32;                                   // kb [8][TB]: stream 1 | stream 3 of wave w: 8 BYTES per block of 5 groups
constexpr size_t ML_OFF_KX1 = ML_OFF_KB + 8 * ML_TB, ML_OFF_KX3 = ML_OFF_KX1 + 8 * ML_CAP1, ML_OFF_KX2 = ML_OFF_KX3 + 8 * ML_CAP3;
constexpr size_t ML_OFF_NZ = ML_OFF_KX2 + 8 * ML_CAP2;
constexpr size_t ML_NNZ = (size_t)ML_NB * ML_KB1 + (size_t)ML_MB * ML_KB2;
constexpr size_t ML_OFF_W1 = ML_OFF_NZ + ML_NNZ;
constexpr size_t ML_N1 = (size_t)8 * ML_CAP1 * 256, ML_N3 = (size_t)8 * ML_CAP3 * 256, ML_KJ = (size_t)8 * ML_CAP2 * 256;
constexpr size_t ML_OFF_W3 = ML_OFF_W1 + ML_N1, ML_OFF_K = ML_OFF_W3 + ML_N3;
static_assert(ML_OFF_W1 % 4 == 0 && ML_OFF_W3 % 4 == 0 && ML_OFF_K % 4 == 0, "float4 alignment");

constexpr size_t ml_lds_floats() {
    return (size_t)(ML_KB1 + 2 * ML_NB + ML_MB) * 256     // V1 | V3 | DV | AD (red, rr of a check alias AD)
           + 64 + 8 * 16 + 8 * ML_TB;                      // rho ladder | inst | stream tables
}

__device__ __forceinline__ float nanmaxl(float a, float b) {          // NaN-propagating max (torch semantics)
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
__device__ __forceinline__ void lds_barrier() {          // orders LDS only: the operand ring's global loads stay in flight
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}   // namespace

// DIAG = true is a separate diagnostic build (RQP_DIAG=1): s_memtime stamps accumulate the ticks each wave spends per segment.
template <bool DIAG, int EXP = 0>     // EXP != 0: timing experiments of the diagnostic build (wrong results): 1 no operand loads, 2 no MFMAs, 3 no vector reads
__global__ void __launch_bounds__(512, 2) k_admm_mfmal(SolveArgs a, const float* __restrict__ img, unsigned long long* __restrict__ dbg) {
    constexpr int NW = ML_NW, NT = ML_NT, NB = ML_NB, MB = ML_MB, TN = ML_TN, TM = ML_TM, D = ML_D;
    extern __shared__ __attribute__((aligned(16))) float sml[];
    f32x4* V1 = (f32x4*)sml;                 // [KB1][64]: blocks < MB: nu (lam / 0 at a check); blocks MB..: x (0 at a check)
    f32x4* V3 = V1 + ML_KB1 * 64;            // [NB][64] dx (x at the start; H x in the second pass of a check)
    f32x4* DV = V3 + NB * 64;                // [NB][64] d (A' lam in a check)
    f32x4* AD = DV + NB * 64;                // [MB][64] raw GEMM3 results of this wave's m tiles
    float* red = (float*)AD;                 // [NW][16][4] row-side maxima per (wave, instance)   (phase 3 runs no GEMM3: AD is free)
    float* rr = red + NW * 16 * 4;           // [NW * 4][16][8] column-side maxima per (wave, kq, instance)
    float* rhosf = (float*)(AD + MB * 64);   // [64]
    float* inst = rhosf + 64;                // [8][16]: 0 rho_est, 2 instance, 4 rho index, 5 done
    int* inst_i = (int*)inst;
    int* tabs = (int*)(inst + 8 * 16);       // [NW][TB] this wave's stream tables (bytes)

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
    const int* meta = (const int*)img;
    const int ng1 = __builtin_amdgcn_readfirstlane(meta[wave_u]), ng3 = __builtin_amdgcn_readfirstlane(meta[8 + wave_u]);
    const int ng2 = __builtin_amdgcn_readfirstlane(meta[16 + wave_u]);   // (tiles of the dense stream x nb5)
    // streams: buffer loads -- scalar byte offset from the image + 16 lane (no vector ALU in the address)
    const __amdgpu_buffer_rsrc_t imgr = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, 0x7fffffff, 0x00020000);
    const unsigned oW1 = (unsigned)(ML_OFF_W1 + (size_t)wave_u * ML_CAP1 * 256) * 4u;
    const unsigned oW3 = (unsigned)(ML_OFF_W3 + (size_t)wave_u * ML_CAP3 * 256) * 4u;
    const unsigned oK0 = (unsigned)(ML_OFF_K + (size_t)wave_u * ML_CAP2 * 256) * 4u;              // + j * KJ * 4
    const int* tab1 = tabs + wave_u * ML_TB;
    const int* tab3 = tab1 + ML_TB1;
    const int nb5 = __builtin_amdgcn_readfirstlane(meta[24]);       // blocks per tile of the dense stream (a multiple of D)
    for (int i = lane; i < ML_TB; i += 64) tabs[wave_u * ML_TB + i] = meta[ML_OFF_KB + wave_u * ML_TB + i];
    // (lv: the lane number through an opaque copy per loop iteration -- hipcc otherwise hoists every (array, tile) address of
    //  the unrolled state code out of the solve loop and spills them)
    int lv = lane;

    // ---- scalars
    for (int i = tid; i < a.nrho && i < 64; i += NT) rhosf[i] = (float)a.rhos[i];
    if (tid < 16) {
        const int st = blockIdx.x * 16 + tid;
        const bool ok = st < a.B;
        const int s0 = ok ? st : blockIdx.x * 16;                  // padding columns mirror the tile's first instance
        const int idt = a.order ? a.order[s0] : s0;
        const int ri = a.rho_ind[idt];
        // cont = 3 (second launch of a regrouped cold solve, below): only the instances that left at the first check take part
        const bool take = ok && (a.cont != 3 || a.info.status[idt] == RQP_STATUS_CONTINUE);
        inst_i[2 * 16 + tid] = idt;
        inst_i[4 * 16 + tid] = ri;
        inst_i[5 * 16 + tid] = take ? 0 : 1;                       // padding columns (and instances already out) start "done"
        inst[0 * 16 + tid] = (a.cont == 3 && take) ? (float)a.cont_rho[idt] : (float)a.rhos[ri];   // rho_est = rhos[rho_ind] (:211) / carried
    }
    if (a.cont == 3) {                                             // a tile without a continuing instance has nothing to do
        __syncthreads();
        int nd0 = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) nd0 += inst_i[5 * 16 + j];
        if (nd0 == 16) return;
    }
    // ---- state.  Rows: m tiles T = wave + NW tl, rows 16 T + 4 kq + r of instance i16.  Columns: n tiles t = wave + NW e.
    float zh[TM][4], zl[TM][4], zz[TM][4], lm[TM][4], lb[TM][4], ub[TM][4];
    unsigned eqmask = 0;
    float xs[TN][4], gs[TN][4];
#pragma unroll
    for (int tl = 0; tl < TM; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (wave_u + NW * tl) + 4 * kq + r;
            const bool ok = real && row < m;
            const size_t o = (size_t)(real ? id : 0) * m + (row < m ? row : 0);
            zh[tl][r] = 0.f;
            zl[tl][r] = 0.f;
            if (a.cont == 3 && ok) {                             // exact continuation: the float-float A x it left with
                const double ax = a.ax[o];
                zh[tl][r] = (float)ax;
                zl[tl][r] = (float)(ax - (double)zh[tl][r]);
            }
            zz[tl][r] = ok ? (float)a.z[o] : 0.f;
            lm[tl][r] = ok ? (float)a.lam[o] : 0.f;
            lb[tl][r] = ok ? ((const float*)a.l)[o] : 0.f;
            ub[tl][r] = ok ? ((const float*)a.u)[o] : 0.f;
            const float cv = (row < m) ? ((const float*)a.c)[o] : 1.f;
            if (cv > 1.f) eqmask |= 1u << (4 * tl + r);
        }
#pragma unroll
    for (int e = 0; e < TN; ++e) {
        const int t = wave_u + NW * e;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * kq + r;
            const bool ok = real && t < NB && row < n;
            const size_t o = (size_t)(real ? id : 0) * n + (row < n ? row : 0);
            xs[e][r] = ok ? (float)a.x[o] : 0.f;
            gs[e][r] = ok ? ((const float*)a.g)[o] : 0.f;
        }
        if (t < NB) {
            const f32x4 xv = {xs[e][0], xs[e][1], xs[e][2], xs[e][3]};
            V3[t * 64 + lane] = a.cont == 3 ? (f32x4){0.f, 0.f, 0.f, 0.f} : xv;   // start pass: GEMM3 on x (a continuation brings A x along: + 0)
            V1[(MB + t) * 64 + lane] = xv;
        }
    }

    // ---- the operand ring: D groups in flight, shared by the three GEMMs.  Every visit of a slot consumes it and requests one
    //      new group into the slot of the visit BEFORE it (whose registers are free), unconditionally: the slot order is
    //      periodic across the streams (whose lengths are multiples of D) and a request is always D - 1 visits ahead of its use.
    //      NO VECTOR-ALU INSTRUCTION inside the visit loops: the float32 MFMA shares its lanes with the VALU, and one v_add per
    //      visit (an address) cost 38 cycles of a 128-cycle visit (RQP_DIAG experiments) -- operand addresses are scalar bases
    //      + immediates, the dense stream's vector operands sit at compile-time LDS offsets.
    f32x4 o_[D];
    auto fetch = [&](int j, unsigned base) __attribute__((always_inline)) {             // base: uniform byte offset of slot 0's group
        if constexpr (EXP != 1)
            o_[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(imgr, (unsigned)lv * 16u, base + 1024u * (unsigned)j, 0));
    };
    auto pro = [&](unsigned off) __attribute__((always_inline)) {      // (slot D - 1 is requested by the first visit)
#pragma unroll
        for (int j = 0; j < D - 1; ++j) fetch(j, off);
    };
    // SPARSE stream X (ng groups, a multiple of D, entries tab[]: 8 bytes per block of D); at a tile's last group the sum goes to
    // Out[tile] (this lane's own float4 of the D layout; the tiles of a stream are wave, wave + NW, ...).  The slots are refilled
    // from X, in the last block from the head of the NEXT stream (offY).  (One v_add per visit remains: the LDS address.)
    auto body = [&](unsigned offX, int ng, unsigned offY, const int* tab, const f32x4* Bv, f32x4* Out) __attribute__((always_inline)) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = acc;          // two chains: a dependent MFMA waits 40 cycles, an independent one 32
        int kA = __builtin_amdgcn_readfirstlane(tab[0]), kB = __builtin_amdgcn_readfirstlane(tab[1]), kN = __builtin_amdgcn_readfirstlane(tab[2]);
        int tile = wave_u;
        f32x4 bn = Bv[(kA & 63) * 64 + lv];                          // the vector operand is read one visit ahead
        auto block = [&](int g0, bool own) __attribute__((always_inline)) {
            const int q = (g0 / D) * 2;
            const int nB = tab[q + 3], nN = tab[q + 4];              // the next block's entries (VGPR copies until the block's end)
            const unsigned pr = own ? offX + 1024u * (unsigned)(g0 + D) : offY;
            const unsigned p0 = offX + 1024u * (unsigned)g0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int e = (j < 4 ? (kA >> (8 * j)) : kB) & 0xff;
                const int en = j < 3 ? (kA >> (8 * (j + 1))) : (j == 3 ? kB : kN);
                const f32x4 b = bn;
                const f32x4 av = o_[j];
                if (j == 0) fetch(D - 1, p0); else fetch(j - 1, pr);  // (fetch adds 1024 x slot)
                if constexpr (EXP != 3) bn = Bv[(en & 63) * 64 + lv];   // (past the stream's end: a valid LDS address, value unused)
                __builtin_amdgcn_sched_barrier(0);                   // (the requests stay ahead of this visit's MFMAs)
                if (EXP != 2 && !(e & 0x80)) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b[0], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], b[1], acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], b[2], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], b[3], acc2, 0, 0, 0);
                }
                if (e & 0x40) {                                      // (uniform) last group of a tile
                    Out[tile * 64 + lv] = acc + acc2;
                    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    acc2 = acc;
                    tile += NW;
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
    // DENSE stream (K_j): ntw tiles of nb5 blocks 0 .. nb5 - 1 each (nb5 = 5, 10, 15 or 20; blocks past ceil(n / 16) have zero
    // operands).  Block p of a tile multiplies DV[p]: every LDS offset is an immediate.  The first operand of a chunk of 5 is read
    // at the end of the chunk before it -- both candidates (the next chunk's, and block 0 for a new tile).
    auto body_dense = [&](unsigned offX, int ntw, unsigned offY, f32x4* Out) __attribute__((always_inline)) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = acc;
        int tile = wave_u;
        int left = ntw * (nb5 / D);                                  // chunks to go
        unsigned px = offX;
        // vector operands: bq[j] serves visit j of a chunk; read TWO visits ahead (LDS returns queue behind the operand
        // stream's returns).  The reads for the first two visits of the next chunk are issued for both candidates: the next
        // chunk of the same tile (blocks pb + 2) and the first chunk of the next tile (blocks 0, 1).
        f32x4 bq[D], bw0 = DV[lv], bw1 = DV[64 + lv], bc0 = bw0, bc1 = bw1;
        for (int tl = 0; tl < ntw; ++tl) {
#pragma unroll
            for (int c = 0; c < NB / D; ++c) {
                if (c * D < nb5) {                                   // (uniform)
                    left -= 1;
                    const unsigned pr = left ? px + 1024u * D : offY;
                    bq[0] = c == 0 ? bw0 : bc0;
                    bq[1] = c == 0 ? bw1 : bc1;
#pragma unroll
                    for (int j = 0; j < D; ++j) {                    // visit: MFMA, vector read, MFMA, request, MFMA, MFMA
                        const int pb = c * D + j;
                        const f32x4 b = bq[j];
                        const f32x4 av = o_[j];
                        if constexpr (EXP != 2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b[0], acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (EXP != 3) {
                            if (j + 2 < D) bq[j + 2] = DV[(pb + 2) * 64 + lv];
                            else if (j + 2 == D) { bc0 = DV[(pb + 2) * 64 + lv]; bw0 = DV[lv]; }             // (blocks NB, NB + 1: behind DV, unused)
                            else { bc1 = DV[(pb + 2) * 64 + lv]; bw1 = DV[64 + lv]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (EXP != 2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], b[1], acc2, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (j == 0) fetch(D - 1, px); else fetch(j - 1, pr);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (EXP != 2) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], b[2], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], b[3], acc2, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    px += 1024u * D;
                }
            }
            Out[tile * 64 + lv] = acc + acc2;
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc2 = acc;
            tile += NW;
        }
    };
    const int ntw2 = ng2 / nb5;
    pro(oW3);
    __syncthreads();
    int ri_l = inst_i[4 * 16 + i16];
    float rho_ne = 1.f, rho_eq = 1.f, inv_ne = 1.f, inv_eq = 1.f;
    auto set_rho = [&]() __attribute__((always_inline)) {
        rho_ne = rhosf[ri_l];
        rho_eq = rho_ne * 1e3f;
        inv_ne = 1.0f / rho_ne;
        inv_eq = 1.0f / rho_eq;
    };
    set_rho();

    // lam_hat and nu of the next iteration from the current state (p = A x - z) -> V1
    auto make_nu = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tl = 0; tl < TM; ++tl) {
            f32x4 nu;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                const float p = (zh[tl][r] - zz[tl][r]) + zl[tl][r];
                const float lh = lm[tl][r] + rho * p;
                lm[tl][r] = lh;
                nu[r] = lh + rho * p;
            }
            V1[(wave_u + NW * tl) * 64 + lv] = nu;
        }
    };

    // phases as in rqp_mfma.hip: 0 start (GEMM3 on x -> A x), 1 iterate, 2 check part 1 (A' lam), 3 check part 2 (H x, decisions)
    int ph = 0, k = (a.cont == 3) ? a.k0 : 0, to_chk = a.check_interval;   // (k0: a multiple of check_interval)
    bool final_chk = false;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;

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
            // (the ring's next stream: K_j in an iteration -- unless this wave owns no tile of the dense stream -- else GEMM1 again)
            body(oW1, ng1, ph == 1 ? (ntw2 ? oK0 + (unsigned)jr * (unsigned)(ML_KJ * 4) : oW3) : oW1, tab1, V1, ph == 3 ? V3 : DV);
            stamp(1);
            if (ph == 1) {
#pragma unroll
                for (int e = 0; e < TN; ++e)                         // d = H x + g + A' nu
                    if (wave_u + NW * e < NB) {
                        f32x4 d = DV[(wave_u + NW * e) * 64 + lv];
                        d[0] += gs[e][0]; d[1] += gs[e][1]; d[2] += gs[e][2]; d[3] += gs[e][3];
                        DV[(wave_u + NW * e) * 64 + lv] = d;
                    }
            }
            stamp(2);
            lds_barrier();                                           // V1 is free again (ph 2 rewrites it); d is visible (ph 1)
            stamp(3);
        }
        bool run_g3 = (ph == 0);
        if (ph == 1) {                                   // ---------------- GEMM2: dx = -K_j d for this wave's n tiles, K_j per column
            float sel[TN][4];
            bool first_pass = true;
            while (true) {                                           // one pass per distinct rho index of the tile (usually one)
                const bool mine = (ri_l == jr);
                const int jc = jr;
                todo &= ~__ballot(lane < 16 && ri_l == jr);
                if (todo) jr = __builtin_amdgcn_readlane(ri_l, __ffsll((long long)todo) - 1);
                body_dense(oK0 + (unsigned)jc * (unsigned)(ML_KJ * 4), ntw2, todo ? oK0 + (unsigned)jr * (unsigned)(ML_KJ * 4) : oW3, V3);
                if constexpr (DIAG) t_acc[10] += 1;                  // K passes (one per distinct rho index of the tile)
                stamp(4);
#pragma unroll
                for (int e = 0; e < TN; ++e)
                    if (wave_u + NW * e < NB) {
                        const f32x4 v = V3[(wave_u + NW * e) * 64 + lv];
#pragma unroll
                        for (int r = 0; r < 4; ++r) sel[e][r] = (first_pass || mine) ? v[r] : sel[e][r];
                    }
                first_pass = false;
                if (!todo) break;
            }
#pragma unroll
            for (int e = 0; e < TN; ++e) {
                const int t = wave_u + NW * e;
                if (t < NB) {
                    f32x4 dx, xv;
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
            body(oW3, ng3, oW1, tab3, V3, AD);
            stamp(7);
#pragma unroll
            for (int tl = 0; tl < TM; ++tl) {
                const int T = wave_u + NW * tl;
                const f32x4 acc = AD[T * 64 + lv];
                float pp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float adx = acc[r];            // float-float accumulation of A x (two-sum + renormalisation)
                    const float sgm = zh[tl][r] + adx;
                    const float bb = sgm - zh[tl][r];
                    const float err = (zh[tl][r] - (sgm - bb)) + (adx - bb);
                    const float lo = zl[tl][r] + err;
                    const float hi = sgm + lo;
                    const float zlo = lo - (hi - sgm);
                    zl[tl][r] = zlo;
                    zh[tl][r] = hi;
                    const bool eq = (eqmask >> (4 * tl + r)) & 1u;
                    const float v = hi + (zlo + lm[tl][r] * (eq ? inv_eq : inv_ne));
                    float zn = v;                        // torch.clamp: NaN stays NaN
                    if (v < lb[tl][r]) zn = lb[tl][r];
                    if (v > ub[tl][r]) zn = ub[tl][r];
                    zn = upd ? zn : zz[tl][r];
                    zz[tl][r] = zn;
                    pp[r] = (hi - zn) + zlo;             // p = A x - z of the new state
                }
                if (with_nu) {                           // lam_hat, nu of the next iteration
                    f32x4 nu;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float rho = ((eqmask >> (4 * tl + r)) & 1u) ? rho_eq : rho_ne;
                        const float lh = lm[tl][r] + rho * pp[r];
                        lm[tl][r] = lh;
                        nu[r] = lh + rho * pp[r];
                    }
                    V1[T * 64 + lv] = nu;
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
                v0 = 0.f; v1 = 0.f; v2 = 0.f;
                int kq_o = kq;
                asm volatile("" : "+v"(kq_o));
#pragma unroll
                for (int tl = 0; tl < TM; ++tl) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * (wave_u + NW * tl) + 4 * kq_o + r;
                        const float we = (a.scE && row < m) ? (float)(1.0 / a.scE[row]) : 1.f;   // (Ruiz: caller-space norms)
                        v0 = nanmaxl(v0, fabsf((zh[tl][r] - zz[tl][r]) + zl[tl][r]) * we);
                        v1 = nanmaxl(v1, fabsf(zh[tl][r] + zl[tl][r]) * we);
                        v2 = nanmaxl(v2, fabsf(zz[tl][r]) * we);
                    }
                    V1[(wave_u + NW * tl) * 64 + lv] = (f32x4){lm[tl][0], lm[tl][1], lm[tl][2], lm[tl][3]};
                }
#pragma unroll
                for (int e = 0; e < TN; ++e)
                    if (wave_u + NW * e < NB) V1[(MB + wave_u + NW * e) * 64 + lv] = (f32x4){0.f, 0.f, 0.f, 0.f};
                ph = 2;
            }
        } else if (ph == 2) {                            // (A' lam is in DV) ; V1 = [0; x]
#pragma unroll
            for (int tl = 0; tl < TM; ++tl) V1[(wave_u + NW * tl) * 64 + lv] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < TN; ++e)
                if (wave_u + NW * e < NB) V1[(MB + wave_u + NW * e) * 64 + lv] = (f32x4){xs[e][0], xs[e][1], xs[e][2], xs[e][3]};
            ph = 3;
        } else {                                         // ph == 3: residuals and decisions (H x is in V3, A' lam in DV)
            float w3 = 0.f, w4 = 0.f, w5 = 0.f, w6 = 0.f, jp = 0.f;
            {
                int kq_o = kq;
                asm volatile("" : "+v"(kq_o));                       // (keeps the rare per-row address arithmetic inside the branch)
#pragma unroll
                for (int e = 0; e < TN; ++e) {
                    const int t = wave_u + NW * e;
                    if (t < NB) {
                        const f32x4 t2v = V3[t * 64 + lv], t3v = DV[t * 64 + lv];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                // column-side maxima of compute_residuals (:313-316), objective
                            const int row = 16 * t + 4 * kq_o + r;
                            const float wd = (a.scD && row < n) ? (float)(1.0 / (a.scC[0] * a.scD[row])) : 1.f;   // (Ruiz: caller-space norms)
                            const float t2 = t2v[r], t3 = t3v[r], ge = gs[e][r];
                            w3 = nanmaxl(w3, fabsf(t2 + t3 + ge) * wd);
                            w4 = nanmaxl(w4, fabsf(t2) * wd);
                            w5 = nanmaxl(w5, fabsf(t3) * wd);
                            w6 = nanmaxl(w6, fabsf(ge) * wd);
                            jp += xs[e][r] * (0.5f * t2 + ge);       // compute_J :320-322
                        }
                    }
                }
            }
            v0 = nanmaxl(v0, __shfl_xor(v0, 16, 64)); v0 = nanmaxl(v0, __shfl_xor(v0, 32, 64));
            v1 = nanmaxl(v1, __shfl_xor(v1, 16, 64)); v1 = nanmaxl(v1, __shfl_xor(v1, 32, 64));
            v2 = nanmaxl(v2, __shfl_xor(v2, 16, 64)); v2 = nanmaxl(v2, __shfl_xor(v2, 32, 64));
            if (kq == 0) {
                red[(wave * 16 + i16) * 4 + 0] = v0;
                red[(wave * 16 + i16) * 4 + 1] = v1;
                red[(wave * 16 + i16) * 4 + 2] = v2;
            }
            {
                float* q = rr + ((wave * 4 + kq) * 16 + i16) * 8;
                q[3] = w3; q[4] = w4; q[5] = w5; q[6] = w6; q[7] = jp;
            }
            lds_barrier();
            if (tid < 16) {                                          // one thread per instance decides
                const int j = tid;
                const float tolT = (float)a.tol, thr_p = (float)a.thr_p, thr_d = (float)a.thr_d, rmin = (float)a.rho_min, rmax = (float)a.rho_max;
                float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f, q6 = 0.f, obj = 0.f;
                for (int w = 0; w < NW; ++w) {
                    q0 = nanmaxl(q0, red[(w * 16 + j) * 4 + 0]);
                    q1 = nanmaxl(q1, red[(w * 16 + j) * 4 + 1]);
                    q2 = nanmaxl(q2, red[(w * 16 + j) * 4 + 2]);
                }
                for (int g2 = 0; g2 < NW * 4; ++g2) {
                    q3 = nanmaxl(q3, rr[(g2 * 16 + j) * 8 + 3]);
                    q4 = nanmaxl(q4, rr[(g2 * 16 + j) * 8 + 4]);
                    q5 = nanmaxl(q5, rr[(g2 * 16 + j) * 8 + 5]);
                    q6 = nanmaxl(q6, rr[(g2 * 16 + j) * 8 + 6]);
                    obj += rr[(g2 * 16 + j) * 8 + 7];
                }
                const bool alive = inst_i[5 * 16 + j] == 0;
                if (alive) {
                    const float num = q0 / nanmaxl(q1, q2);                               // :315
                    const float den = q3 / nanmaxl(nanmaxl(q4, q5), q6);                  // :316
                    float est = inst[0 * 16 + j] * sqrtf(num / den);                      // :317 (Q4: carried)
                    if (est < rmin) est = rmin;                                           // torch.clamp: NaN stays NaN
                    if (est > rmax) est = rmax;
                    int ri = inst_i[4 * 16 + j];
                    const int ri_before = ri;
                    if (!final_chk) {
                        if (est > rhosf[ri] * tolT && ri < a.nrho - 1) ri += 1;           // :223
                        else if (est < rhosf[ri] / tolT && ri > 0) ri -= 1;               // :226
                    }
                    inst[0 * 16 + j] = est;
                    inst_i[4 * 16 + j] = ri;
                    const int idj = inst_i[2 * 16 + j];
                    const int chk_no = k / a.check_interval;
                    if (!final_chk && a.info.trace && chk_no <= a.info.trace_cap) {
                        double* tr = a.info.trace + ((size_t)idj * a.info.trace_cap + (chk_no - 1)) * 4;
                        tr[0] = (double)q0; tr[1] = (double)q3; tr[2] = (double)est; tr[3] = (double)ri_before;
                    }
                    const float er = (float)a.eps_rel;
                    const float tp = er > 0.f ? thr_p + er * nanmaxl(q1, q2) : thr_p;
                    const float td = er > 0.f ? thr_d + er * nanmaxl(nanmaxl(q4, q5), q6) : thr_d;
                    const bool conv = !final_chk && (q0 < tp && q3 < td);                 // :233
                    const bool last = final_chk || k >= kmax;                             // :243 max-iter fallthrough
                    if (!conv && !last && a.leave_at > 0 && k == a.leave_at) {
                        // regrouped cold solve: leave behind the FIRST check with the exact state; the second launch continues the
                        // instance in a tile of instances at the same rho index (rqp_launch_solve_mfmal)
                        inst_i[5 * 16 + j] = 3;
                        a.info.status[idj] = RQP_STATUS_CONTINUE;
                        a.cont_rho[idj] = (double)est;
                        a.rho_ind[idj] = ri;
                    } else if (conv || last) {
                        float est_out = est;
                        if (!conv && !final_chk) {                   // max_iter on the check grid: the reference's extra compute_residuals (:243)
                            est_out = est * sqrtf(num / den);
                            if (est_out < rmin) est_out = rmin;
                            if (est_out > rmax) est_out = rmax;
                        }
                        inst_i[5 * 16 + j] = 2;                      // exits now (2: outputs due; 1: gone)
                        const size_t bj = (size_t)idj;
                        if (a.info.iter) a.info.iter[bj] = conv ? k : a.max_iter;
                        if (a.info.status) a.info.status[bj] = conv ? RQP_STATUS_SOLVED : ((q0 != q0 || q3 != q3) ? RQP_STATUS_NAN : RQP_STATUS_MAX_ITER);
                        if (a.info.rho_ind) a.info.rho_ind[bj] = ri;
                        if (a.info.pri_res) a.info.pri_res[bj] = (double)q0;
                        if (a.info.dua_res) a.info.dua_res[bj] = (double)q3;
                        if (a.info.rho_estimate) a.info.rho_estimate[bj] = (double)est_out;
                        if (a.info.obj_val) a.info.obj_val[bj] = (double)obj;
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
            if (inst_i[5 * 16 + i16] >= 2) {             // this lane's instance just finished: x, z, lam out + the persistent state (:278-305)
                const bool leaves = inst_i[5 * 16 + i16] == 3;               // (3: it continues in the second launch -- state kept, A x too)
                const bool ws = (a.warm_starting || a.keep_state) != 0 || leaves;
                int kq_o = kq, id_o = id;
                asm volatile("" : "+v"(kq_o), "+v"(id_o));
#pragma unroll
                for (int e = 0; e < TN; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int t = wave_u + NW * e, row = 16 * t + 4 * kq_o + r;
                        if (t < NB && row < n) {
                            const size_t o = (size_t)id_o * n + row;
                            if (a.out_x) ((float*)a.out_x)[o] = xs[e][r];
                            a.x[o] = ws ? (double)xs[e][r] : 0.0;
                        }
                    }
#pragma unroll
                for (int tl = 0; tl < TM; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * (wave_u + NW * tl) + 4 * kq_o + r;
                        if (row < m) {
                            const size_t o = (size_t)id_o * m + row;
                            if (a.out_z) ((float*)a.out_z)[o] = zz[tl][r];
                            if (a.out_lam) ((float*)a.out_lam)[o] = lm[tl][r];
                            a.z[o] = ws ? (double)zz[tl][r] : 0.0;
                            a.lam[o] = ws ? (double)lm[tl][r] : 0.0;
                            if (leaves) a.ax[o] = (double)zh[tl][r] + (double)zl[tl][r];
                        }
                    }
            }
            __syncthreads();
            if (tid < 16 && inst_i[5 * 16 + tid] >= 2) inst_i[5 * 16 + tid] = 1;
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
__device__ __forceinline__ float ml_s_elem(int n, int m, int ldn, const float* A, const float* Ht, int k, int col) {   // S[k][col]
    if (col >= n) return 0.f;
    if (k < ML_MP) return k < m ? A[(size_t)k * ldn + col] : 0.f;
    return (k - ML_MP < n) ? Ht[(size_t)(k - ML_MP) * ldn + col] : 0.f;
}
// one wave per 16 x 16 block: is any element non-zero?
__global__ void k_nz_mfmal(int n, int m, int ldn, const float* __restrict__ A, const float* __restrict__ Ht, float* __restrict__ img) {
    int* nzf = (int*)img + ML_OFF_NZ;
    const int b = blockIdx.x, lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    bool nz = false;
    if (b < ML_NB * ML_KB1) {
        const int t = b / ML_KB1, blk = b % ML_KB1;
        for (int j = 0; j < 4; ++j) nz |= ml_s_elem(n, m, ldn, A, Ht, 16 * blk + 4 * q + j, 16 * t + i) != 0.f;
    } else {
        const int T = (b - ML_NB * ML_KB1) / ML_KB2, blk = (b - ML_NB * ML_KB1) % ML_KB2;
        const int r = 16 * T + i;
        for (int j = 0; j < 4; ++j) {
            const int c = 16 * blk + 4 * q + j;
            nz |= (r < m && c < n) && A[(size_t)r * ldn + c] != 0.f;
        }
    }
    const unsigned long long any = __ballot(nz);
    if (lane == 0) nzf[b] = any != 0ull;
}
// one thread per wave of the solve kernel: its three streams (tiles back to back; a tile without a non-zero block gets one
// group -- block 0, whose operand is then zero -- so that its result is written)
__global__ void k_meta_mfmal(int n, int m, float* __restrict__ img) {
    int* meta = (int*)img;
    const int* nzf = meta + ML_OFF_NZ;
    const int w = threadIdx.x;
    if (w >= ML_NW) return;
    const int nbt = (n + 15) / 16;
    int* kx = meta + ML_OFF_KX1 + w * ML_CAP1;
    int pos = 0;
    for (int e = 0; e < ML_TN; ++e) {
        const int t = w + ML_NW * e;
        if (t >= ML_NB) break;
        const int first = pos;
        for (int blk = 0; blk < ML_KB1; ++blk)
            if (nzf[t * ML_KB1 + blk]) kx[pos++] = blk | (t << 8);
        if (pos == first) kx[pos++] = 0 | (t << 8);
        kx[pos - 1] |= ML_LAST;
    }
    while (pos % ML_D) kx[pos++] = ML_NULL;
    meta[w] = pos;
    kx = meta + ML_OFF_KX3 + w * ML_CAP3;
    pos = 0;
    for (int tl = 0; tl < ML_TM; ++tl) {
        const int T = w + ML_NW * tl;
        const int first = pos;
        for (int blk = 0; blk < ML_KB2; ++blk)
            if (nzf[ML_NB * ML_KB1 + T * ML_KB2 + blk]) kx[pos++] = blk | (T << 8);
        if (pos == first) kx[pos++] = 0 | (T << 8);
        kx[pos - 1] |= ML_LAST;
    }
    while (pos % ML_D) kx[pos++] = ML_NULL;
    meta[8 + w] = pos;
    kx = meta + ML_OFF_KX2 + w * ML_CAP2;
    pos = 0;
    const int nb5 = (nbt + ML_D - 1) / ML_D * ML_D;
    for (int e = 0; e < ML_TN; ++e) {                                 // K_j is dense: nb5 blocks for each tile of the problem's own n
        const int t = w + ML_NW * e;
        if (t >= nbt) break;
        for (int blk = 0; blk < nb5; ++blk) kx[pos++] = blk | (t << 8) | (blk >= nbt ? ML_NULL : 0);
    }
    meta[16 + w] = pos;
    if (w == 0) meta[24] = nb5;
    // the byte tables the solve kernel reads (streams 1 and 3): 8 bytes per block of ML_D groups
    unsigned char* kb = (unsigned char*)(meta + ML_OFF_KB + w * ML_TB);
    for (int st = 0; st < 2; ++st) {
        const int* src = meta + (st == 0 ? ML_OFF_KX1 + w * ML_CAP1 : ML_OFF_KX3 + w * ML_CAP3);
        const int cnt = meta[(st == 0 ? 0 : 8) + w];
        const int tbd = st == 0 ? ML_TB1 : ML_TB3;
        for (int i = 0; i < 4 * tbd; ++i) {
            const int blkno = i >> 3, j = i & 7, g = blkno * ML_D + j;
            unsigned char v = 0x80;
            if (j < ML_D && g < cnt) {
                const int kd = src[g];
                v = (unsigned char)((kd & 63) | ((kd & ML_LAST) ? 0x40 : 0) | ((kd & ML_NULL) ? 0x80 : 0));
            }
            kb[i] = v;
        }
        kb += 4 * tbd;
    }
}

// operand images of the streams (after k_meta_mfmal); groups past a stream's end are zero
__global__ void k_pack_mfmal(int n, int m, int ldn, int nrho, const float* __restrict__ A, const float* __restrict__ Ht,
                             const float* __restrict__ K, float* __restrict__ img) {
    const int* meta = (const int*)img;
    float* W1 = img + ML_OFF_W1;
    float* W3 = img + ML_OFF_W3;
    float* Kimg = img + ML_OFF_K;
    const size_t nk = (size_t)nrho * ML_KJ;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < ML_N1 + ML_N3 + nk; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 3), l = (int)((idx >> 2) & 63), i16 = l & 15, kq = l >> 4;
        if (idx < ML_N1) {
            const int grp = (int)(idx >> 8), w = grp / ML_CAP1, pos = grp % ML_CAP1;
            float v = 0.f;
            if (pos < meta[w]) {
                const int kd = meta[ML_OFF_KX1 + grp], blk = kd & 0xff, t = (kd >> 8) & 0xff;
                if (!(kd & ML_NULL)) v = ml_s_elem(n, m, ldn, A, Ht, 16 * blk + 4 * kq + j, 16 * t + i16);
            }
            W1[idx] = v;
        } else if (idx < ML_N1 + ML_N3) {
            const size_t o = idx - ML_N1;
            const int grp = (int)(o >> 8), w = grp / ML_CAP3, pos = grp % ML_CAP3;
            float v = 0.f;
            if (pos < meta[8 + w]) {
                const int kd = meta[ML_OFF_KX3 + grp], blk = kd & 0xff, T = (kd >> 8) & 0xff;
                const int r = 16 * T + i16, c = 16 * blk + 4 * kq + j;
                if (r < m && c < n && !(kd & ML_NULL)) v = A[(size_t)r * ldn + c];
            }
            W3[o] = v;
        } else {
            const size_t o = idx - ML_N1 - ML_N3;
            const int jr = (int)(o / ML_KJ), grp = (int)((o % ML_KJ) >> 8), w = grp / ML_CAP2, pos = grp % ML_CAP2;
            float v = 0.f;
            if (pos < meta[16 + w]) {
                const int kd = meta[ML_OFF_KX2 + grp], blk = kd & 0xff, t = (kd >> 8) & 0xff;
                const int r = 16 * t + i16, c = 16 * blk + 4 * kq + j;
                if (r < n && c < n && !(kd & ML_NULL)) v = K[((size_t)jr * n + r) * ldn + c];
            }
            Kimg[o] = v;
        }
    }
}

// ------------------------------------------------------------------------------ host side
bool rqp_mfmal_fits(const rqp_handle* h) {
    return h->esz == 4 && h->dims.shared_mats && h->n <= ML_NP && h->m <= ML_MP && h->nrho <= 64;
}
size_t rqp_mfmal_img_elems(const rqp_handle* h) { return ML_OFF_K + (size_t)h->nrho * ML_KJ; }

hipError_t rqp_launch_pack_mfmal(const rqp_handle* h, hipStream_t s) {
    hipError_t e = hipMemsetAsync(h->W1img, 0, ML_OFF_W1 * sizeof(int), s);
    if (e != hipSuccess) return e;
    k_nz_mfmal<<<(unsigned)ML_NNZ, 64, 0, s>>>(h->n, h->m, h->ldn, (const float*)h->A, (const float*)h->Ht, h->W1img);
    k_meta_mfmal<<<1, 64, 0, s>>>(h->n, h->m, h->W1img);
    k_pack_mfmal<<<1024, 256, 0, s>>>(h->n, h->m, h->ldn, h->nrho, (const float*)h->A, (const float*)h->Ht, (const float*)h->K, h->W1img);
    return hipGetLastError();
}
hipError_t rqp_prepare_mfmal(const rqp_handle* h) {
    hipError_t e = rqp_raise_lds_limit((const void*)k_admm_mfmal<false>, ml_lds_floats() * sizeof(float));
    if (e == hipSuccess && (h->debug & 2)) {
        (void)rqp_raise_lds_limit((const void*)k_admm_mfmal<true, 1>, ml_lds_floats() * sizeof(float));
        (void)rqp_raise_lds_limit((const void*)k_admm_mfmal<true, 2>, ml_lds_floats() * sizeof(float));
        (void)rqp_raise_lds_limit((const void*)k_admm_mfmal<true, 3>, ml_lds_floats() * sizeof(float));
        e = rqp_raise_lds_limit((const void*)k_admm_mfmal<true>, ml_lds_floats() * sizeof(float));
    }
    return e;
}
// Regrouped cold solve.  Every distinct rho index among a tile's live columns costs one pass over the dense K stream (the dominant
// GEMM): a cold batch starts at ONE index, but at the first check its instances split (79 % / 21 % on the sparse config-3 batch) and
// the tiles run 1.335 passes per iteration from then on.  So a cold solve (warm_starting = 0, or the first solve of a handle) runs as TWO
// launches: every instance leaves behind its first check with its exact state (x, z, lam, float-float A x, carried rho estimate: the
// continuation is bit-identical to an uninterrupted solve), the slots are re-sorted by the NEW indices (stable counting sort; the
// instances that converged at the first check go last), and the second launch continues in homogeneous tiles.  No host round trip.
__global__ void k_regroup_key(int B, const int32_t* __restrict__ status, const int32_t* __restrict__ rho_ind, int32_t* __restrict__ key) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) key[i] = status[i] == RQP_STATUS_CONTINUE ? rho_ind[i] + 1 : 0;
}

hipError_t rqp_launch_solve_mfmal(const rqp_handle* h, const SolveArgs& a0, hipStream_t s) {
    const int grid = (h->B + 15) / 16;
    SolveArgs a = a0;
    if (grid > 1 && h->order_d) {                // slots grouped by the rho index the instances start at (k_order_stable)
        hipError_t e = rqp_launch_order_by(h, h->rho_ind, s);
        if (e != hipSuccess) return e;
        a.order = h->order_d;
    }
    const bool two = grid > 1 && h->order_d && h->ax_d && h->cont_rho_d && h->key_d && (!a.warm_starting || a.cold) && !a.keep_state && a.cont == 0 &&
                     a.info.status && a.check_interval > 0 && a.max_iter > a.check_interval && h->nrho <= 62 && !(h->debug & 2);
    if (two) {
        const size_t lds = ml_lds_floats() * sizeof(float);
        SolveArgs p1 = a;
        p1.leave_at = a.check_interval;
        p1.ax = h->ax_d;
        p1.cont_rho = h->cont_rho_d;
        k_admm_mfmal<false><<<grid, ML_NT, lds, s>>>(p1, h->W1img, nullptr);
        k_regroup_key<<<(h->B + 255) / 256, 256, 0, s>>>(h->B, a.info.status, h->rho_ind, h->key_d);
        hipError_t e = rqp_launch_order_by(h, h->key_d, s);
        if (e != hipSuccess) return e;
        SolveArgs p2 = a;
        p2.cont = 3;
        p2.k0 = a.check_interval;
        p2.ax = h->ax_d;
        p2.cont_rho = h->cont_rho_d;
        p2.order = h->order_d;
        k_admm_mfmal<false><<<grid, ML_NT, lds, s>>>(p2, h->W1img, nullptr);
        return hipGetLastError();
    }
    if (h->debug & 2) {          // diagnostic build: per-segment tick shares of the iteration (synchronous, debug only)
        unsigned long long* dbg = nullptr;
        const size_t cnt = (size_t)grid * ML_NW * 12;
        if (hipMalloc((void**)&dbg, cnt * 8) != hipSuccess) return hipErrorOutOfMemory;
        const char* ex = getenv("RQP_MLEXP");
        const int exv = ex ? atoi(ex) : 0;
        if (exv == 1) k_admm_mfmal<true, 1><<<grid, ML_NT, ml_lds_floats() * sizeof(float), s>>>(a, h->W1img, dbg);
        else if (exv == 2) k_admm_mfmal<true, 2><<<grid, ML_NT, ml_lds_floats() * sizeof(float), s>>>(a, h->W1img, dbg);
        else if (exv == 3) k_admm_mfmal<true, 3><<<grid, ML_NT, ml_lds_floats() * sizeof(float), s>>>(a, h->W1img, dbg);
        else k_admm_mfmal<true><<<grid, ML_NT, ml_lds_floats() * sizeof(float), s>>>(a, h->W1img, dbg);
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> hb(cnt);
        (void)hipMemcpy(hb.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        static const char* names[10] = {"top wait", "GEMM1", "d", "wait", "GEMM2", "x", "wait", "GEMM3", "rows", "next"};
        for (int w = 0; w < ML_NW; ++w) {
            double tot[12] = {0};
            for (int t = 0; t < grid; ++t)
                for (int e2 = 0; e2 < 12; ++e2) tot[e2] += (double)hb[((size_t)t * ML_NW + w) * 12 + e2];
            fprintf(stderr, "[rqp diag mfmal] wave %d, %.1f iterations/workgroup, s_memtime ticks per iteration:", w, tot[11] / grid);
            double it = 0;
            for (int e2 = 0; e2 < 10; ++e2) {
                fprintf(stderr, "  %s %.1f", names[e2], tot[e2] / tot[11]);
                it += tot[e2];
            }
            fprintf(stderr, "  | sum %.1f  | K passes per iteration %.3f\n", it / tot[11], tot[10] / tot[11]);
        }
        return hipGetLastError();
    }
    k_admm_mfmal<false><<<grid, ML_NT, ml_lds_floats() * sizeof(float), s>>>(a, h->W1img, nullptr);
    return hipGetLastError();
}
