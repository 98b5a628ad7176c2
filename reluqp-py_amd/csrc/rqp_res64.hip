// rqp_res64.hip -- RESIDENT ADMM kernel in FLOAT64 (the reference's default -- and only working -- precision,
// SURVEY.md Q2) for n <= 104, m <= 320: the sizes of the headline workload (n = 100, m = 300).
// Same recurrence, checks and semantics as k_admm_generic<double> (rqp_admm.hip; reference
// ReLU-QP-py/reluqp/reluqpth.py:201-249, statement in oracle/reluqp_oracle.py:forward_refine); that kernel streams
// Ht, A, K_j, At from L2/HBM every iteration (640 KB per instance-iteration in float64: HBM-bound).  Here one
// 512-thread workgroup = one QP = one CU, and A (240 KB) and K_j (80 KB) live in the CU's 512 KB register file for the
// whole solve; H (84.5 KB) sits in LDS; HBM is touched once per solve and per rho move.
//
//   wave w (of 8) owns the column block [13 w, 13 w + 13) of A for ALL rows, and rows [13 w, 13 w + 13) of K:
//   A' nu : lane l holds rows 5 l .. 5 l + 4 of the block (5 x 13 doubles).  Its 13 partial column sums -- plus the H rows
//           l and l + 64 of the same columns read from LDS: d = [A; H]' [nu; x] + g, H symmetric -- are transposed through a
//           per-wave LDS slab and summed by 52 lanes (13 columns x 4 quarters), then two shuffle levels.
//   K d   : lane (rr = l >> 2, cc = l & 3) holds K[13 w + rr][26 cc .. 26 cc + 25]; reduce over cc by two shuffle levels;
//           lanes cc == 0, rr < 13 own x and dx of column 13 w + rr (float64 registers).
//   A dx  : needs only the wave's own 13 dx values; the 8 per-wave partial row sums meet in LDS and are added in fixed
//           order by the row's owner thread (row i <-> thread i), which keeps z, lam, A x of its row in registers.
// 3 barriers per iteration.  182 float64 FMAs per lane and iteration: float64 VALU-bound.
#include <cstdio>
#include <type_traits>
#include <vector>

#include "rqp_common.h"
#include "rqp_lanes.h"

namespace {

constexpr int R64_NT = 512, R64_NW = 8, R64_CW = 13, R64_RL = 5, R64_N = 104, R64_M = 320, R64_KC = 26;
constexpr int R64_PS = 10;                         // doubles per row of the A dx partials: 8 waves + 2 (16-byte rows, conflict-free b128 reads)
constexpr int R64_RS = 66;                         // stride (doubles) of a column's 64 lane partials in the reduce slab
constexpr size_t r64_lds_doubles() {
    return (size_t)R64_N * R64_N                   // Hs [col][row]
           + (size_t)R64_NW * R64_CW * R64_RS      // reduce slabs; the A dx partials part[8][320] alias their start
           + 5 * R64_M + 4 * R64_N + 64 + 16 + 32; // nu, l, u, rho, 1/rho | d | g | x | H x | reductions | check scalars | rho ladder
}

// v + (v of lane ^ 1) and v + (v of lane ^ 2): quad permutes on the two 32-bit halves (DPP, no LDS round trip --
// __shfl_xor on a double is two ds_bpermute)
template <int CTRL>
__device__ __forceinline__ double dpp_add64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double quad_sum(double v) {     // sum over the 4 lanes of a quad, every lane gets it
    v = dpp_add64<0xB1>(v);                                // quad_perm [1,0,3,2]
    return dpp_add64<0x4E>(v);                             // quad_perm [2,3,0,1]
}

template <typename U>
__device__ __forceinline__ U tmx(U a, U b) {       // torch.max / norm(inf): NaN propagates
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}

}   // namespace

// DIAG = true is a separate diagnostic build (RQP_DIAG=1): s_memtime stamps accumulate the cycles each wave spends in
// every segment of the iteration into `dbg` (never read by the kernel; never timed as the product).
template <bool DIAG>
__global__ void __launch_bounds__(512, 2) k_admm_res64(SolveArgs a, unsigned long long* dbg) {
    constexpr int NT = R64_NT, NW = R64_NW, CW = R64_CW, RL = R64_RL, N = R64_N, M = R64_M, KC = R64_KC, RS = R64_RS;
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    double* Hs = sm64;                             // [N][N]: Hs[col * N + row]
    double* slab = Hs + N * N;                     // [NW][CW][RS]
    double* part = slab;                           // [M][PS]: the 8 wave partials of a row side by side (alias: the slabs are dead
                                                   //          between the K product and the next A' nu)
    double* nuL = slab + NW * CW * RS;             // [M]
    double* loL = nuL + M;                         // [M] l  (row bounds wait in LDS: the register file is full of A and K)
    double* hiL = loL + M;                         // [M] u
    double* rvL = hiL + M;                         // [M] rho_i
    double* invL = rvL + M;                        // [M] 1 / rho_i
    double* dL = invL + M;                         // [N]
    double* gL = dL + N;                           // [N]
    double* xL = gL + N;                           // [N]
    double* hxL = xL + N;                          // [N] H x of the last check (objective at the exit)
    double* red = hxL + N;                         // [64]
    double* stat = red + 64;                       // [16] scalars of the last check: pri, dua, rho estimate, scales (they are
                                                   //      read again only at the next check / the exit: LDS, not registers)
    double* rhoL = stat + 16;                      // [32] the rho ladder (read at every check: LDS, not a global load)
    static_assert(M * R64_PS <= NW * CW * RS && NW <= R64_PS && N % 2 == 0, "part aliases the reduce slabs");

    const int n = a.n, m = a.m, ldn = a.ldn;
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rr = lane >> 2, cc = lane & 3;
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
    if (a.cstat && a.mode == 0) {
        const int sl = ri - wb;
        if (sl < 0 || sl >= a.kwin) {                                 // (uniform) the incoming index lies outside the window: leave
            if (tid == 0) {                                           // untouched; rqp_solve re-centres the window and restarts it
                a.cstat[b] = 1;
                if (!exact) a.cont_iter[b] = -1;
                atomicAdd(a.ncont, 1);
            }
            return;
        }
    }
    // Waves w and w + 4 share a SIMD; VALU issue is arbitrated by priority, then age, so the second-dispatched half loses every
    // segment (RQP_DIAG: A' nu + H x 2 624 vs 2 263 cycles) and the first half waits for it at each barrier.  One static
    // priority step for that half evens the pair out (MI355X_MICROARCH.md, "Two waves per SIMD", item 4).
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
    const double* A = (const double*)a.A + (size_t)b * a.sA;
    const double* Ht = (const double*)a.Ht + (size_t)b * a.sH;
    const double* Kb = (const double*)a.K + (size_t)b * a.sK;

    // ---- matrices: A, K_j -> registers ; H -> LDS (each element read from HBM once per solve)
    double ar[RL][CW];
#pragma unroll
    for (int r = 0; r < RL; ++r)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int row = RL * lane + r, col = CW * wave + c;
            ar[r][c] = (row < m && col < n) ? A[(size_t)row * ldn + col] : 0.0;
        }
    for (int i = tid; i < N * N; i += NT) {
        const int col = i / N, row = i % N;
        Hs[i] = (row < n && col < n) ? Ht[(size_t)col * ldn + row] : 0.0;      // Ht = sym(H): H[row][col] = Ht[col][row]
    }
    if (tid < 32) rhoL[tid] = (tid < a.nrho) ? a.rhos[tid] : 0.0;   // (nrho <= 32: rqp_res64_fits)
    double kr[KC];
    auto load_K = [&](int j) {
        const int krow = CW * wave + rr;
        int sl = j - wb;                                              // (iterate / compute_residuals: rqp_abi re-centres the windows first)
        sl = sl < 0 ? 0 : (sl >= a.kwin ? a.kwin - 1 : sl);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int col = KC * cc + c;
            kr[c] = (rr < CW && krow < n && col < n) ? Kb[((size_t)sl * n + krow) * ldn + col] : 0.0;
        }
    };
    load_K(ri);

    // ---- state.  Row i <-> thread i (i < M): z, lam, A x, bounds, rho in registers.  Column 13 w + rr <-> lane cc == 0.
    const bool rown = tid < M, rin = tid < m;
    double zt = 0.0, z = 0.0, lam = 0.0;
    if (rin) {
        z = a.z[(size_t)b * m + tid];
        lam = a.lam[(size_t)b * m + tid];
    }
    if (rown) {
        loL[tid] = rin ? ((const double*)a.l)[(size_t)b * m + tid] : 0.0;
        hiL[tid] = rin ? ((const double*)a.u)[(size_t)b * m + tid] : 0.0;
    }
    auto set_rho = [&](int j) {                    // (row i is written and read by thread i only: no barrier)
        if (!rown) return;
        const double rv = a.rhos[j] * (rin ? ((const double*)a.c)[(size_t)b * m + tid] : 1.0);
        rvL[tid] = rv;
        invL[tid] = 1.0 / rv;
    };
    set_rho(ri);
    const int xcol = CW * wave + rr;
    const bool xown = cc == 0 && rr < CW;
    const bool xin = xown && xcol < n;
    double x = xin ? a.x[(size_t)b * n + xcol] : 0.0;
    if (xown) {
        xL[xcol] = x;
        gL[xcol] = xin ? ((const double*)a.g)[(size_t)b * n + xcol] : 0.0;
    }
    if (rown) nuL[tid] = 0.0;
    __syncthreads();

    // ---- products ------------------------------------------------------------------------------------------------
    // wave partial of A v over the wave's 13 columns -> part[wave][row].  v[13 w + c] is held by lane 4 c of this very
    // wave (the column owners), so it is broadcast through v_readlane into a scalar register pair -- no LDS hop on the
    // critical path, no 13 broadcast reads (an LDS broadcast still returns 64 x 8 bytes), no 26 live VGPRs
    auto prod_A = [&](const double vown) {
        const unsigned long long vb = __builtin_bit_cast(unsigned long long, vown);
        const int vlo = (int)(unsigned)vb, vhi = (int)(unsigned)(vb >> 32);
        double s[RL];
#pragma unroll
        for (int r = 0; r < RL; ++r) s[r] = 0.0;
        double vc[CW];                                                // all 13 broadcasts first (scalar registers): a v_readlane result
#pragma unroll                                                        // used by the very next VALU instruction costs a hazard s_nop
        for (int c = 0; c < CW; ++c) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane(vlo, 4 * c), hi = (unsigned)__builtin_amdgcn_readlane(vhi, 4 * c);
            vc[c] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
            for (int r = 0; r < RL; ++r) s[r] = fma(ar[r][c], vc[c], s[r]);
#pragma unroll
        for (int r = 0; r < RL; ++r) part[(RL * lane + r) * R64_PS + wave] = s[r];
    };
    // [USE_A: A' w] + [USE_H: H xL] for the wave's own columns; lanes (c = lane >> 2 < 13, cc == 0) return the column sum
    auto prod_At = [&](bool use_a, bool use_h, const double* w) -> double {
        double* sl = slab + (size_t)wave * CW * RS;
        double wr[RL];
        if (use_a) {
#pragma unroll
            for (int r = 0; r < RL; ++r) wr[r] = w[RL * lane + r];
        }
        // H rows of this lane: 2 lane and 2 lane + 1 (lanes < 52), ONE ds_read_b128 per column.  (Rows lane and lane + 64 were
        // merged by the compiler into ds_read2_b64 across columns: 8 LDS cycles for what two ds_read_b64 or one ds_read_b128
        // do in 4 -- MI355X_MICROARCH.md, LDS table -- in the segment that runs at the LDS rate.)  Lanes >= 52 re-read the
        // last row pair with a zero multiplier: no divergent branch, so the reads of a chunk are requested together.
        const int hrow = (2 * lane + 1 < N) ? 2 * lane : N - 2;
        double x0 = 0.0, x1 = 0.0;
        if (use_h && 2 * lane + 1 < N) {
            const double2 xv = *(const double2*)(xL + hrow);
            x0 = xv.x;
            x1 = xv.y;
        }
        // columns in three chunks (4 + 4 + 5 accumulators live instead of 13: the register file is full of A and K)
        auto chunk = [&](auto c0c, auto c1c) __attribute__((always_inline)) {
            constexpr int C0 = decltype(c0c)::value, C1 = decltype(c1c)::value;
            double cs[C1 - C0];
#pragma unroll
            for (int c = C0; c < C1; ++c) cs[c - C0] = 0.0;
            if (use_a) {
#pragma unroll
                for (int r = 0; r < RL; ++r)
#pragma unroll
                    for (int c = C0; c < C1; ++c) cs[c - C0] = fma(ar[r][c], wr[r], cs[c - C0]);
            }
            if (use_h) {
#pragma unroll
                for (int c = C0; c < C1; ++c) {
                    const double2 hv = *(const double2*)(Hs + (size_t)(CW * wave + c) * N + hrow);
                    cs[c - C0] = fma(hv.x, x0, cs[c - C0]);
                    cs[c - C0] = fma(hv.y, x1, cs[c - C0]);
                }
            }
#pragma unroll
            for (int c = C0; c < C1; ++c) sl[c * RS + lane] = cs[c - C0];
        };
        chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        chunk(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
        chunk(std::integral_constant<int, 8>{}, std::integral_constant<int, CW>{});
        // transpose through the wave's slab: 64 partials per column -> lane (c, q) adds a quarter, two shuffle levels finish
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double s = 0.0;
        if (rr < CW) {
            // (Round 3 tried the two conflict-free forms of this sum -- lane (column, quarter q) adding the partials 4 k + q with
            //  slab stride 68, or the pairs (8 k + 2 q, + 1) by ds_read_b128 with stride 72: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
            //  fell from 0.20 to 0.10, but the kernel was 5 % / 3.5 % SLOWER on the same box (tools/ab_bench.sh: 6.29 -> 6.51 ms):
            //  the segment is bound by the latency of its dependent LDS round trips, not by replay cycles.  Contiguous form kept.)
            const double* src = sl + rr * RS + 16 * cc;
            double s0 = 0.0, s1 = 0.0;                                // two chains of 8
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                s0 += src[k];
                s1 += src[k + 1];
            }
            s = s0 + s1;
        }
        return quad_sum(s);
    };
    // lanes cc == 0 get sum_c K[13 w + rr][c] v[c]
    auto prod_K = [&](const double* v) -> double {
        double s0 = 0.0, s1 = 0.0;                                    // two chains: the 26 FMAs of a lane are one dependent chain otherwise
#pragma unroll
        for (int c = 0; c < KC; c += 2) {
            s0 = fma(kr[c], v[KC * cc + c], s0);
            s1 = fma(kr[c + 1], v[KC * cc + c + 1], s1);
        }
        return quad_sum(s0 + s1);
    };
    // row update.  do_a: A x += sum of the 8 wave partials; z = clamp(A x + lam / rho).  do_b: lam_hat, nu of the NEXT iteration.
    auto row_pass = [&](bool init, bool do_a, bool do_b) {
        if (!rown) return;
        if (init || do_a) {
            const double2* pp = (const double2*)(part + tid * R64_PS);   // 4 ds_read_b128 (ds_read2st64_b64 pairs cost twice that)
            const double2 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
            const double s = ((p0.x + p0.y) + (p1.x + p1.y)) + ((p2.x + p2.y) + (p3.x + p3.y));   // fixed pairwise order
            zt = (init ? 0.0 : zt) + s;
        }
        if (do_a) {
            const double lo = loL[tid], hi = hiL[tid];
            const double v = zt + lam * invL[tid];
            z = v;                                                    // torch.clamp: NaN stays NaN
            if (v < lo) z = lo;
            if (v > hi) z = hi;
        }
        if (do_b) {
            const double rv = rvL[tid];
            const double pr = zt - z;
            const double lh = lam + rv * pr;
            lam = lh;
            nuL[tid] = lh + rv * pr;
        }
    };

    if (tid == 0) {
        stat[0] = 0.0;                                                // pri
        stat[1] = 0.0;                                                // dua
        stat[2] = (a.mode == 2) ? a.rho_in : (exact ? a.cont_rho[b] : a.rhos[ri]);   // carried rho estimate, reluqpth.py:211
    }
    bool converged = false;
    int iters = 0;
    const int kmax = (a.mode == 2) ? 0 : a.max_iter;

    // A x of the incoming state (an exact continuation brought it along: only lam_hat / nu of the next iteration are due)
    if (!exact) {
        prod_A(x);
        __syncthreads();
        row_pass(true, false, kmax > 0);
    } else {
        zt = rin ? a.ax[(size_t)b * m + tid] : 0.0;
        row_pass(false, false, k0 < kmax);
    }

    // compute_residuals (reluqpth.py:307-318) on the current state; leaves (H x)[col] of the column owners in hxv
    // updates stat[0..4] = pri, dua, rho estimate (carried, Q4), scale of pri, scale of dua; ends with a barrier
    auto residuals = [&]() {
        double v[7] = {0, 0, 0, 0, 0, 0, 0};
        const size_t smat = (a.sA == 0) ? 0 : (size_t)b;              // (Ruiz scaling: caller-space norms, SolveArgs.scE)
        if (rown) {
            const double we = (a.scE && rin) ? 1.0 / a.scE[smat * m + tid] : 1.0;
            nuL[tid] = lam;
            v[0] = fabs(zt - z) * we;
            v[1] = fabs(zt) * we;
            v[2] = fabs(z) * we;
        }
        __syncthreads();
        const double t3 = prod_At(true, false, nuL);                  // A' lam
        __syncthreads();                                              // (slab reuse)
        const double t2 = prod_At(false, true, nuL);                  // H x
        if (xown) {
            const double wd = (a.scD && xin) ? 1.0 / (a.scC[smat] * a.scD[smat * n + xcol]) : 1.0;
            const double gx = gL[xcol];
            hxL[xcol] = t2;
            v[3] = fabs(t2 + t3 + gx) * wd;
            v[4] = fabs(t2) * wd;
            v[5] = fabs(t3) * wd;
            v[6] = fabs(gx) * wd;
        }
        // 7 maxima over the workgroup, NaN-propagating like torch.max / norm(inf): NaN flags travel as a bit mask, the values
        // through v_max_f64 (which skips NaN).  Wave level in registers; the 8 waves meet in LDS and wave 0 finishes:
        // lane (w, e) = (lane >> 3, lane & 7) takes wave w's value e (e == 7: its NaN mask), lanes 8 apart are combined.
        int nanm = 0;
#pragma unroll
        for (int e = 0; e < 7; ++e) nanm |= (v[e] != v[e]) ? (1 << e) : 0;
#pragma unroll
        for (int e = 0; e < 7; ++e) v[e] = wave_max64(v[e]);
        nanm = wave_or_i(nanm);
        if (lane == 0) {                                              // (red: the previous check's readers passed a barrier since)
#pragma unroll
            for (int e = 0; e < 7; ++e) red[wave * 8 + e] = v[e];
            ((int*)(red + wave * 8 + 7))[0] = nanm;
        }
        __syncthreads();
        if (wave == 0) {
            double r = red[lane];                                     // (lane & 7) == 7: NaN mask in the low dword
            int mk = ((const int*)(red + lane))[0];
            if ((lane & 7) == 7) r = 0.0;
            r = dpp_max64<0x128>(r);                                  // row_ror:8  -> lanes 8 apart inside a row
            mk |= dpp_i<0x128>(mk);
            r = rows_max64(r);
            mk = rows_or(mk);
            const int allnan = __builtin_amdgcn_readlane(mk, 7);
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                const unsigned long long u = __builtin_bit_cast(unsigned long long, r);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, e);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), e);
                v[e] = ((allnan >> e) & 1) ? __builtin_nan("") : __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
            }
        }
        if (tid == 0) {
            const double sp = tmx(v[1], v[2]), sd = tmx(tmx(v[4], v[5]), v[6]);
            const double num = v[0] / sp;                             // :315
            const double den = v[3] / sd;                             // :316
            double est = stat[2] * sqrt(num / den);                   // :317
            if (est < a.rho_min) est = a.rho_min;                     // torch.clamp: NaN stays NaN
            if (est > a.rho_max) est = a.rho_max;
            stat[0] = v[0];
            stat[1] = v[3];
            stat[2] = est;
            stat[3] = sp;
            stat[4] = sd;
        }
        __syncthreads();
    };

    unsigned long long t_last = 0, t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
    stamp(-1);
    int to_chk = a.check_interval;
    for (int k = k0 + 1; k <= kmax; ++k) {                            // (k0 > 0: an exact continuation, on the check grid)
        __syncthreads();                                              // B3: nu (and x) visible
        stamp(0);
        const double d = prod_At(true, true, nuL);                    // d = H x + g + A' nu   (own columns)
        if (xown) dL[xcol] = d + gL[xcol];
        stamp(1);
        __syncthreads();                                              // B1: d visible (the slabs are free: part may be written)
        stamp(2);
        {
            const double dx = -prod_K(dL);                            // (lanes cc == 0; rows >= 13 of the group are zero rows of K)
            if (xown) {
                x += dx;
                xL[xcol] = x;
            }
            stamp(3);
            prod_A(dx);
        }
        stamp(4);
        __syncthreads();                                              // B2: partials and x visible
        stamp(5);
        iters = k;
        const bool on_grid = (--to_chk == 0);
        if (on_grid) to_chk = a.check_interval;
        const bool check = (a.mode == 0) && on_grid;                  // reluqpth.py:218 (Q3 fixed)
        if (!check) {
            row_pass(false, true, k < kmax);
            stamp(6);
        } else {
            row_pass(false, true, false);
            const int ri_before = ri;
            residuals();                                              // :220 (Q4: carried estimate)
            const double pri = stat[0], dua = stat[1], rho_est = stat[2];
            if (rho_est > rhoL[ri] * a.tol && ri < a.nrho - 1)        // :223
                ri += 1;
            else if (rho_est < rhoL[ri] / a.tol && ri > 0)            // :226
                ri -= 1;
            if (a.info.trace && (k / a.check_interval) <= a.info.trace_cap && tid == 0) {
                double* tr = a.info.trace + ((size_t)b * a.info.trace_cap + (k / a.check_interval - 1)) * 4;
                tr[0] = pri; tr[1] = dua; tr[2] = rho_est; tr[3] = (double)ri_before;
            }
            const double tp = a.eps_rel > 0 ? a.thr_p + a.eps_rel * stat[3] : a.thr_p;    // :233 (+ relative term, 8(f)-3)
            const double td = a.eps_rel > 0 ? a.thr_d + a.eps_rel * stat[4] : a.thr_d;
            if (pri < tp && dua < td) {
                converged = true;
                break;
            }
            if (ri != ri_before) {                                    // adaptive-rho "re-factor": table lookup
                if (a.cstat && k < kmax && (ri < wb || ri >= wb + a.kwin)) {
                    // the new index has no K in this instance's window: leave with the exact state; rqp_solve re-factors a
                    // window around ri and continues the instance behind this check (SolveArgs.cont = 2)
                    if (xin) a.x[(size_t)b * n + xcol] = x;
                    if (rin) {
                        a.z[(size_t)b * m + tid] = z;
                        a.lam[(size_t)b * m + tid] = lam;
                        a.ax[(size_t)b * m + tid] = zt;
                    }
                    if (tid == 0) {
                        a.rho_ind[b] = ri;
                        a.cont_iter[b] = k;
                        a.cont_rho[b] = stat[2];
                        a.cstat[b] = 1;
                        atomicAdd(a.ncont, 1);
                    }
                    return;
                }
                load_K(ri);
                set_rho(ri);
            }
            if (k < kmax) row_pass(false, false, true);
            stamp(7);
        }
    }
    if constexpr (DIAG) {
        if (lane == 0)
            for (int e = 0; e < 9; ++e) dbg[((size_t)b * NW + wave) * 9 + e] = (e == 8) ? (unsigned long long)iters : t_acc[e];
    }

    __syncthreads();
    if (a.mode == 1) {                                                // iterate-only: keep the state
        if (xin) a.x[(size_t)b * n + xcol] = x;
        if (rin) {
            a.z[(size_t)b * m + tid] = z;
            a.lam[(size_t)b * m + tid] = lam;
        }
        return;
    }
    if (!converged) residuals();                                      // :243 (Q11 fixed: fresh state)
    const double pri = stat[0], dua = stat[1], rho_est = stat[2];

    // objective 1/2 x'Hx + g'x (compute_J :320-322): hxL = H x of the last check
    double jp = xown ? x * (0.5 * hxL[xcol] + gL[xcol]) : 0.0;
    for (int off = 32; off >= 1; off >>= 1) jp += __shfl_xor(jp, off, 64);
    if (lane == 0) red[wave] = jp;
    __syncthreads();
    double obj = red[0];
    for (int w = 1; w < NW; ++w) obj += red[w];

    if (a.mode == 2) {
        if (tid == 0) {
            if (a.r_pri) a.r_pri[b] = pri;
            if (a.r_dua) a.r_dua[b] = dua;
            if (a.r_rho) a.r_rho[b] = rho_est;
            if (a.r_obj) a.r_obj[b] = obj;
        }
        return;
    }

    // ---- update_results (reluqpth.py:278-305)
    if (a.out_x && xin) ((double*)a.out_x)[(size_t)b * n + xcol] = x;
    if (a.out_z && rin) ((double*)a.out_z)[(size_t)b * m + tid] = z;
    if (a.out_lam && rin) ((double*)a.out_lam)[(size_t)b * m + tid] = lam;
    if (tid == 0) {
        if (a.info.iter) a.info.iter[b] = converged ? iters : a.max_iter;
        if (a.last_iter) a.last_iter[b] = converged ? iters : a.max_iter;
        if (a.info.status) a.info.status[b] = converged ? RQP_STATUS_SOLVED : ((pri != pri || dua != dua) ? RQP_STATUS_NAN : RQP_STATUS_MAX_ITER);
        if (a.info.rho_ind) a.info.rho_ind[b] = ri;
        if (a.info.pri_res) a.info.pri_res[b] = pri;
        if (a.info.dua_res) a.info.dua_res[b] = dua;
        if (a.info.rho_estimate) a.info.rho_estimate[b] = rho_est;
        if (a.info.obj_val) a.info.obj_val[b] = obj;
    }
    const bool keep = a.warm_starting || a.keep_state;                // state + rho index persist (:304) or are cleared (:324-333)
    if (xin) a.x[(size_t)b * n + xcol] = keep ? x : 0.0;
    if (rin) {
        a.z[(size_t)b * m + tid] = keep ? z : 0.0;
        a.lam[(size_t)b * m + tid] = keep ? lam : 0.0;
    }
    if (tid == 0) {
        a.rho_ind[b] = keep ? ri : a.rho_ind0;
        if (a.cstat) a.cstat[b] = 0;
    }
}

bool rqp_res64_fits(const rqp_handle* h) { return h->esz == 8 && h->n <= R64_N && h->m <= R64_M && h->nrho <= 32; }

hipError_t rqp_prepare_res64(const rqp_handle* h) {
    const int lds = (int)(r64_lds_doubles() * sizeof(double));
    hipError_t e = rqp_raise_lds_limit((const void*)k_admm_res64<false>, (size_t)lds);
    if (e == hipSuccess && (h->debug & 2))
        e = rqp_raise_lds_limit((const void*)k_admm_res64<true>, (size_t)lds);
    return e;
}

hipError_t rqp_launch_solve_res64(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    const size_t lds = r64_lds_doubles() * sizeof(double);
    if (h->debug & 2) {          // diagnostic build: per-segment cycle shares of the iteration (synchronous, debug only)
        unsigned long long* dbg = nullptr;
        const size_t cnt = (size_t)h->B * R64_NW * 9;
        if (hipMalloc((void**)&dbg, cnt * 8) != hipSuccess) return hipErrorOutOfMemory;
        k_admm_res64<true><<<h->B, R64_NT, lds, s>>>(a, dbg);
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> hbuf(cnt);
        (void)hipMemcpy(hbuf.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        static const char* names[8] = {"B3 wait", "A'nu+Hx", "B1 wait", "Kd+x", "A dx", "B2 wait", "rows", "check"};
        for (int w = 0; w < R64_NW; ++w) {
            double tot[8] = {0}, its = 0;
            for (int b = 0; b < h->B; ++b) {
                for (int e = 0; e < 8; ++e) tot[e] += (double)hbuf[((size_t)b * R64_NW + w) * 9 + e];
                its += (double)hbuf[((size_t)b * R64_NW + w) * 9 + 8];
            }
            fprintf(stderr, "[rqp diag64] wave %d cycles/iteration:", w);
            double sum = 0;
            for (int e = 0; e < 8; ++e) { fprintf(stderr, " %s=%.0f", names[e], tot[e] / its); sum += tot[e] / its; }
            fprintf(stderr, " | total=%.0f\n", sum);
        }
        return hipGetLastError();
    }
    k_admm_res64<false><<<h->B, R64_NT, lds, s>>>(a, nullptr);
    return hipGetLastError();
}
