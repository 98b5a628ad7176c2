// rqp_setup.hip -- one-time factorisation path of ReLU_QP.setup (reference
// ReLU-QP-py/reluqp/reluqpth.py:102-157 -> ReLU_Layer.setup_matrices :40-78).
//
// The reference builds, for each of the ~18 rho values, a dense (n+2m)^2 matrix W.
// Here only K_j = (H + sigma I + rho_j A' diag(c) A)^-1 (n x n) is built per rho
// (SURVEY.md Appendix A): G = A' diag(c) A once, then M_j = sym(H) + sigma I + rho_j G
// is inverted in float64 by an in-place Gauss-Jordan sweep (M_j is SPD: no pivoting).
//
// Kernels: k_pack (casts/transposes, QP.__init__ classes.py:4-30), k_gram, k_factor,
// plus the small state/vector movers behind update()/warm_start()/get_state().
#include <algorithm>
#include <type_traits>
#include <utility>

#include "rqp_common.h"

template <int... Ks, class F>
__device__ __forceinline__ void rqp_static_for(std::integer_sequence<int, Ks...>, F&& f) {
    (f(std::integral_constant<int, Ks>{}), ...);
}

// ------------------------------------------------------------------------------ pack
// Ht = sym(H) = (H + H')/2 (the quadratic form only sees the symmetric part; every kernel may then read rows of Ht as
// rows of H, and K is built from the same matrix); A copied with padded leading dim; At[r][c] = A[c][r]; pads zeroed.
// H_in / A_in may be NULL (rqp_update_mats): that matrix keeps its packed copy.
template <typename T>
__global__ void k_pack_mats(SetupArgs a) {
    const int mat = blockIdx.y;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    if (a.H_in) {
        const T* H = (const T*)a.H_in + (size_t)mat * a.n * a.n;
        T* Ht = (T*)a.Ht + (size_t)mat * a.n * a.ldn;
        for (int i = tid; i < a.n * a.ldn; i += nth) {
            int r = i / a.ldn, c = i % a.ldn;
            Ht[i] = (c < a.n) ? T(0.5) * (H[(size_t)c * a.n + r] + H[(size_t)r * a.n + c]) : T(0);
        }
    }
    if (a.A_in && a.A != a.A_in) {                  // (a.A == a.A_in: the handle keeps no copy of A, rqp_handle.borrow_A)
        const T* A = (const T*)a.A_in + (size_t)mat * a.m * a.n;
        T* Ap = (T*)a.A + (size_t)mat * a.m * a.ldn;
        T* At = (T*)a.At + (size_t)mat * a.n * a.ldm;
        for (int i = tid; i < a.m * a.ldn; i += nth) {
            int r = i / a.ldn, c = i % a.ldn;
            Ap[i] = (c < a.n) ? A[(size_t)r * a.n + c] : T(0);
        }
        if (a.At) {                                // (NULL: no kernel of this handle reads the transposed copy, rqp_setup)
            for (int i = tid; i < a.n * a.ldm; i += nth) {
                int r = i / a.ldm, c = i % a.ldm;
                At[i] = (c < a.m) ? A[(size_t)c * a.n + r] : T(0);
            }
        }
    }
}

// g, l, u copies and the equality-row scale c (reluqpth.py:54: rho*1e3 where u-l <= eq_tol)
template <typename T>
__global__ void k_pack_vecs(SetupArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const T* g = (const T*)a.g_in;
    const T* l = (const T*)a.l_in;
    const T* u = (const T*)a.u_in;
    for (size_t i = tid; i < (size_t)a.B * a.n; i += nth) ((T*)a.g)[i] = g[i];
    for (size_t i = tid; i < (size_t)a.B * a.m; i += nth) {
        T li = l[i], ui = u[i];
        ((T*)a.l)[i] = li;
        ((T*)a.u)[i] = ui;
        ((T*)a.c)[i] = ((ui - li) <= (T)a.eq_tol) ? T(1e3) : T(1);   // NaN (inf-inf) compares false
    }
}

// sym(H) through LDS (n * n elements <= 100 KB): 16-byte row reads in (when n % 4 == 0 and the caller's matrix is 16-byte
// aligned), the transposed operand from LDS, padded rows out.  (k_pack_mats reads H[c][r] from global memory with 4-byte
// accesses a row apart: 0.21 ms for the 4096 matrices of the headline batch.)
template <typename T>
__global__ void k_sym_h(int n, int ldn, const T* __restrict__ H_in, T* __restrict__ Ht_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sym_raw[];
    T* st = (T*)sym_raw;
    const int mat = blockIdx.x, t = threadIdx.x;
    const T* H = H_in + (size_t)mat * n * n;
    T* Ht = Ht_out + (size_t)mat * n * ldn;
    constexpr int V = 16 / sizeof(T);
    if ((n % V) == 0 && (((size_t)H) & 15) == 0) {
        typedef T vec __attribute__((ext_vector_type(V)));
        for (int i = V * t; i < n * n; i += V * blockDim.x) *(vec*)(st + i) = *(const vec*)(H + i);
    } else {
        for (int i = t; i < n * n; i += blockDim.x) st[i] = H[i];
    }
    __syncthreads();
    for (int i = t; i < n * ldn; i += blockDim.x) {
        const int r = i / ldn, c = i - r * ldn;
        Ht[i] = (c < n) ? T(0.5) * (st[c * n + r] + st[r * n + c]) : T(0);
    }
}

hipError_t rqp_launch_pack_mats(const rqp_handle* h, const SetupArgs& a0, hipStream_t s) {
    SetupArgs a = a0;
    const size_t hb = (size_t)h->n * h->n * h->esz;
    if (a.H_in && hb <= 100 * 1024) {                                   // (float64 n = 100: 80 KB, one workgroup per CU)
        hipError_t e = rqp_raise_lds_limit(h->esz == 4 ? (const void*)k_sym_h<float> : (const void*)k_sym_h<double>, hb);
        if (e != hipSuccess) return e;
        if (h->esz == 4)
            k_sym_h<float><<<h->nmat, 256, hb, s>>>(h->n, h->ldn, (const float*)a.H_in, (float*)a.Ht);
        else
            k_sym_h<double><<<h->nmat, 256, hb, s>>>(h->n, h->ldn, (const double*)a.H_in, (double*)a.Ht);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        a.H_in = nullptr;
    }
    if (!a.H_in && !(a.A_in && a.A != a.A_in)) return hipSuccess;      // nothing left to copy
    // blocks per matrix: ~8 elements per thread of the largest copy (64 blocks of 256 threads on a 32 x 32 / 64 x 32 instance
    // were 16 384 threads for 2 048 elements: 0.40 ms for the 8 192 instances of the config-4 batch)
    const size_t big = (size_t)(h->m > h->n ? h->m : h->n) * (h->ldn > h->ldm ? h->ldn : h->ldm);
    const unsigned gx = (unsigned)std::min<size_t>(64, std::max<size_t>(1, (big + 2047) / 2048));
    dim3 grid(gx, h->nmat);
    if (h->esz == 4)
        k_pack_mats<float><<<grid, 256, 0, s>>>(a);
    else
        k_pack_mats<double><<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t rqp_launch_pack_vecs(const rqp_handle* h, const SetupArgs& a, hipStream_t s) {
    if (h->esz == 4)
        k_pack_vecs<float><<<256, 256, 0, s>>>(a);
    else
        k_pack_vecs<double><<<256, 256, 0, s>>>(a);
    return hipGetLastError();
}

// *flag = 1 when some instance's equality scale c differs from instance 0's (shared-matrix batches: K is built from
// instance 0's pattern, rqp_setup refuses a heterogeneous batch)
template <typename T>
__global__ void k_check_shared_c(int B, int m, const T* __restrict__ c, int32_t* flag) {
    const size_t total = (size_t)B * m;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        bad |= c[i] != c[i % m];
    if (bad) atomicOr(flag, 1);
}

hipError_t rqp_launch_check_shared_c(const rqp_handle* h, int32_t* flag, hipStream_t s) {
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    if (h->esz == 4)
        k_check_shared_c<float><<<256, 256, 0, s>>>(h->B, h->m, (const float*)h->c, flag);
    else
        k_check_shared_c<double><<<256, 256, 0, s>>>(h->B, h->m, (const double*)h->c, flag);
    return hipGetLastError();
}

// update(g,l,u): reluqpth.py:159-183.  c (and K) are NOT re-derived: the reference keeps the
// matrices built at setup when l/u change (W_ks untouched, :171-174).
template <typename T>
__global__ void k_vec_update(int B, int n, int m, const T* g, const T* l, const T* u, T* gd, T* ld, T* ud) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    if (g) for (size_t i = tid; i < (size_t)B * n; i += nth) gd[i] = g[i];
    if (l) for (size_t i = tid; i < (size_t)B * m; i += nth) ld[i] = l[i];
    if (u) for (size_t i = tid; i < (size_t)B * m; i += nth) ud[i] = u[i];
}

hipError_t rqp_launch_vec_update(const rqp_handle* h, const void* g, const void* l, const void* u, hipStream_t s) {
    if (h->esz == 4)
        k_vec_update<float><<<256, 256, 0, s>>>(h->B, h->n, h->m, (const float*)g, (const float*)l, (const float*)u,
                                                  (float*)h->g, (float*)h->l, (float*)h->u);
    else
        k_vec_update<double><<<256, 256, 0, s>>>(h->B, h->n, h->m, (const double*)g, (const double*)l,
                                                   (const double*)u, (double*)h->g, (double*)h->l, (double*)h->u);
    return hipGetLastError();
}

// g[b] = Gg p[b], l[b] = l0 + Glu p[b], u[b] = u0 + Glu p[b]  (rqp_update_affine: the MPC x0 update in one pass).
// One thread per output element; p[b] is read through the cache by the n + m threads of its instance, the maps
// (a few KB) stay in L1/L2.  Sums run over j in order (plain T arithmetic).
template <typename T>
__global__ void k_affine_update(int B, int n, int m, int np, const T* __restrict__ p, const T* __restrict__ Gg,
                                const T* __restrict__ Glu, const T* __restrict__ l0, const T* __restrict__ u0,
                                T* __restrict__ gd, T* __restrict__ ld, T* __restrict__ ud) {
    const size_t per = (size_t)n + m, total = (size_t)B * per;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t b = idx / per;
        const int i = (int)(idx - b * per);
        const T* pb = p + b * np;
        const T* row = (i < n) ? Gg + (size_t)i * np : Glu + (size_t)(i - n) * np;
        T acc = (T)0;
        for (int j = 0; j < np; ++j) acc += row[j] * pb[j];
        if (i < n) {
            gd[b * n + i] = acc;
        } else {
            const int r = i - n;
            ld[b * m + r] = l0[r] + acc;
            ud[b * m + r] = u0[r] + acc;
        }
    }
}

hipError_t rqp_launch_affine_update(const rqp_handle* h, const void* p, int np, const void* Gg, const void* Glu,
                                    const void* l0, const void* u0, hipStream_t s) {
    const size_t total = (size_t)h->B * (h->n + h->m);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (h->esz == 4)
        k_affine_update<float><<<blocks, 256, 0, s>>>(h->B, h->n, h->m, np, (const float*)p, (const float*)Gg, (const float*)Glu,
                                                      (const float*)l0, (const float*)u0, (float*)h->g, (float*)h->l, (float*)h->u);
    else
        k_affine_update<double><<<blocks, 256, 0, s>>>(h->B, h->n, h->m, np, (const double*)p, (const double*)Gg,
                                                       (const double*)Glu, (const double*)l0, (const double*)u0, (double*)h->g,
                                                       (double*)h->l, (double*)h->u);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ gram
// G[r][c] = sum_k c_k A[k][r] A[k][c] in float64.  One workgroup = one 64x64 output tile of
// one matrix; 16x16 threads, 4x4 register tile each; A staged through LDS 16 rows at a time.
#define GR_T 64
#define GR_K 16
template <typename T>
__global__ void __launch_bounds__(256) k_gram(SetupArgs a) {
    __shared__ double sR[GR_K][GR_T + 1];
    __shared__ double sC[GR_K][GR_T + 1];
    const int mat = blockIdx.z;
    const int r0 = blockIdx.y * GR_T, c0 = blockIdx.x * GR_T;
    const T* A = (const T*)a.A + (size_t)mat * a.m * a.ldn;
    const T* cv = (const T*)a.c + (size_t)mat * a.m;   // shared mats: instance 0's pattern (mat = 0)
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = 0; k0 < a.m; k0 += GR_K) {
        for (int i = threadIdx.x; i < GR_K * GR_T; i += 256) {
            int kk = i / GR_T, cc = i % GR_T;
            int k = k0 + kk;
            double vr = 0.0, vc = 0.0;
            if (k < a.m) {
                if (r0 + cc < a.n) vr = (double)A[(size_t)k * a.ldn + r0 + cc] * (double)cv[k];
                if (c0 + cc < a.n) vc = (double)A[(size_t)k * a.ldn + c0 + cc];
            }
            sR[kk][cc] = vr;
            sC[kk][cc] = vc;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GR_K; ++kk) {
            double rv[4], cvv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = sR[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) cvv[j] = sC[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(rv[i], cvv[j], acc[i][j]);
        }
        __syncthreads();
    }
    double* G = a.G + (size_t)mat * a.n * a.n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = r0 + ty * 4 + i, c = c0 + tx * 4 + j;
            if (r < a.n && c < a.n) G[(size_t)r * a.n + c] = acc[i][j];
        }
}

// n <= 112 (the sizes of k_factor_reg2, which reads the upper 16 x 16 blocks of G only): the product on the float64 matrix pipe.
// G tile (I, J) = sum_k (c_k A[k][16 I + .])' A[k][16 J + .] is one chain of
// v_mfma_f64_16x16x4_f64 over the rows of A, four rows per instruction.  Both operands of every tile are the SAME per-lane
// values -- lane (kq, i16) holds A[k0 + kq][16 T + i16] for the RT column tiles T (A operand: row i16 of the transposed tile,
// k = kq, times c; B operand: k = kq, column i16) -- so a k-step is RT loads straight from global memory (64-byte row
// segments, no LDS, no barrier) for up to 7 MFMAs per wave.  The RT (RT + 1) / 2 upper tiles are dealt to the 4 waves (tile g to
// wave g mod 4: compile-time (I, J) per wave behind a wave-uniform switch, or the operand registers would be indexed dynamically);
// loads run KU k-steps ahead in a second register set.  D layout (rqp_mfmad.hip): register r of lane (kq, i16) is row kq + 4 r,
// column i16 of the tile.  (Its predecessor k_gram2 -- 16 x 16 threads x RT x RT register tiles, rows of A staged in LDS, 2 RT LDS
// reads per RT^2 / 2 float64 FMAs -- took 0.79 ms for the 4096 matrices of the headline batch; this kernel 0.44 ms.)
typedef double gm_d4 __attribute__((ext_vector_type(4)));
constexpr int gm_tile_I(int g, int RT) { int I = 0; while (g >= RT - I) { g -= RT - I; ++I; } return I; }
constexpr int gm_tile_J(int g, int RT) { int I = 0; while (g >= RT - I) { g -= RT - I; ++I; } return I + g; }

template <typename T, int RT, int W>
__device__ __forceinline__ void gram_mfma_wave(const SetupArgs& a, int mat, int lane) {
    constexpr int NTILE = RT * (RT + 1) / 2, TPW = (NTILE - W + 3) / 4, KU = (sizeof(T) == 4) ? 2 : 4;
    const int n = a.n, m = a.m, ldn = a.ldn, i16 = lane & 15, kq = lane >> 4;
    const T* A = (const T*)a.A + (size_t)mat * m * ldn;
    const T* cv = (const T*)a.c + (size_t)mat * m;                       // shared mats: instance 0's pattern (mat = 0)
    gm_d4 acc[TPW > 0 ? TPW : 1];
#pragma unroll
    for (int e = 0; e < TPW; ++e) acc[e] = (gm_d4){0.0, 0.0, 0.0, 0.0};
    // Branch-free loads: row and column indices are clamped into the matrix.  A row k >= m is switched off through its scale
    // (c = 0: the A operand of the MFMA is zero); a column >= n only feeds tile rows / columns >= n, which are never stored.
    // (Guarded loads compiled to one exec-masked branch per load: 1.2 ms for the headline batch against 0.79 ms of the VALU kernel.)
    int coff[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) coff[t] = min(16 * t + i16, n - 1);
    auto load = [&](int k0, T (&v)[KU][RT], T (&c)[KU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = k0 + 4 * u + kq, kc = min(k, m - 1);
            const T cl = cv[kc];
            c[u] = (k < m) ? cl : T(0);
            const T* Ar = A + (size_t)kc * ldn;
#pragma unroll
            for (int t = 0; t < RT; ++t) v[u][t] = Ar[coff[t]];
        }
    };
    auto compute = [&](const T (&v)[KU][RT], const T (&c)[KU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            double vd[RT], vc[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                vd[t] = (double)v[u][t];
                vc[t] = vd[t] * (double)c[u];
            }
            rqp_static_for(std::make_integer_sequence<int, TPW>{}, [&](auto ec) __attribute__((always_inline)) {
                constexpr int e = decltype(ec)::value, I = gm_tile_I(W + 4 * e, RT), J = gm_tile_J(W + 4 * e, RT);   // (compile time:
                acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[I], vd[J], acc[e], 0, 0, 0);    //  left to the optimiser, the index loops ran on the SALU)
            });
        }
    };
    T v0[KU][RT], v1[KU][RT], c0[KU], c1[KU];
    load(0, v0, c0);
    for (int k0 = 0; k0 < m; k0 += 8 * KU) {                             // two groups of KU k-steps per trip: explicit ping-pong
        load(k0 + 4 * KU, v1, c1);                                       // (rows >= m load zeros)
        compute(v0, c0);
        load(k0 + 8 * KU, v0, c0);
        compute(v1, c1);
    }
    double* G = a.G + (size_t)mat * n * n;
    rqp_static_for(std::make_integer_sequence<int, TPW>{}, [&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value, I = gm_tile_I(W + 4 * e, RT), J = gm_tile_J(W + 4 * e, RT);
        const int c = 16 * J + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + kq + 4 * r;
            if (row < n && c < n) G[(size_t)row * n + c] = acc[e][r];
        }
    });
}

template <typename T, int RT>
__global__ void __launch_bounds__(256, (sizeof(T) == 4) ? 3 : 2) k_gram_mfma(SetupArgs a) {
    const int mat = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    switch (__builtin_amdgcn_readfirstlane(wave)) {
        case 0: gram_mfma_wave<T, RT, 0>(a, mat, lane); break;
        case 1: gram_mfma_wave<T, RT, 1>(a, mat, lane); break;
        case 2: gram_mfma_wave<T, RT, 2>(a, mat, lane); break;
        default: gram_mfma_wave<T, RT, 3>(a, mat, lane); break;
    }
}

hipError_t rqp_launch_gram(const rqp_handle* h, const SetupArgs& a, hipStream_t s) {
    // n <= 112: the float64 MFMA kernel (upper 16 x 16 tiles only, like k_factor_reg2 reads them)
    if (h->n <= 32) {
        if (h->esz == 4) k_gram_mfma<float, 2><<<h->nmat, 256, 0, s>>>(a); else k_gram_mfma<double, 2><<<h->nmat, 256, 0, s>>>(a);
        return hipGetLastError();
    }
    if (h->n <= 64) {
        if (h->esz == 4) k_gram_mfma<float, 4><<<h->nmat, 256, 0, s>>>(a); else k_gram_mfma<double, 4><<<h->nmat, 256, 0, s>>>(a);
        return hipGetLastError();
    }
    if (h->n <= 112 && h->ldn <= 112) {           // (the same predicate as rqp_launch_factor: k_factor_reg2 reads the upper blocks only)
        if (h->esz == 4) k_gram_mfma<float, 7><<<h->nmat, 256, 0, s>>>(a); else k_gram_mfma<double, 7><<<h->nmat, 256, 0, s>>>(a);
        return hipGetLastError();
    }

    int t = (h->n + GR_T - 1) / GR_T;
    dim3 grid(t, t, h->nmat);
    if (h->esz == 4)
        k_gram<float><<<grid, 256, 0, s>>>(a);
    else
        k_gram<double><<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- factor
// One workgroup per (matrix, rho index): M = sym(H) + sigma I + rho_j G, inverted in place by
// Gauss-Jordan (SPD, no pivoting), float64.  LDS_MODE: M lives in LDS (n*n*8 B <= ~150 KB);
// otherwise in a global scratch slab (L2-resident).  Output K_j in T, padded rows zeroed.
template <typename T, bool LDS_MODE>
__global__ void __launch_bounds__(256) k_factor(SetupArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = a.n;
    const int mat = blockIdx.x / a.kwin, j = blockIdx.x % a.kwin;        // j: K slot; ladder index = window base + slot
    if (a.only && !a.only[mat]) return;
    double* colb = (double*)smem_raw;            // [n]   column k before the sweep
    double* rowb = colb + n;                     // [n]   scaled pivot row
    double* M = LDS_MODE ? (rowb + n) : (a.fscratch + (size_t)blockIdx.x * n * n);
    const T* Ht = (const T*)a.Ht + (size_t)mat * n * a.ldn;
    const double* G = a.G + (size_t)mat * n * n;
    const double rho = a.rhos[(a.wbase ? a.wbase[mat] : 0) + j];
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += 256) {
        int r = i / n, c = i % n;
        double hs = 0.5 * ((double)Ht[(size_t)r * a.ldn + c] + (double)Ht[(size_t)c * a.ldn + r]);
        M[i] = hs + (r == c ? a.sigma : 0.0) + rho * G[i];
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const double p = 1.0 / M[(size_t)k * n + k];
        for (int i = tid; i < n; i += 256) {
            colb[i] = M[(size_t)i * n + k];
            rowb[i] = (i == k) ? p : M[(size_t)k * n + i] * p;
        }
        __syncthreads();
        for (int i = tid; i < n * n; i += 256) {
            int r = i / n, c = i % n;
            double v;
            if (r == k)
                v = rowb[c];
            else if (c == k)
                v = -colb[r] * p;
            else
                v = M[i] - colb[r] * rowb[c];
            M[i] = v;
        }
        __syncthreads();
    }
    T* K = (T*)a.K + ((size_t)mat * a.kwin + j) * n * a.ldn;
    for (int i = tid; i < n * a.ldn; i += 256) {
        int r = i / a.ldn, c = i % a.ldn;
        // symmetrise the rounded result so that column-oriented products see one matrix
        K[i] = (c < n) ? (T)(0.5 * (M[(size_t)r * n + c] + M[(size_t)c * n + r])) : T(0);
    }
}

// Large n (142 < n <= 320, e.g. the sparse linear-MPC form n = 320): BLOCKED Gauss-Jordan, 16 pivots per pass over the matrix.
// With M = [[A, B], [C, D]] and pivot block A (16 x 16): M <- [[A^-1, A^-1 B], [-C A^-1, D - C A^-1 B]] -- exactly 16 steps of the
// unblocked sweep of k_factor, but the matrix (float64, in the L2-resident scratch slab) is read and written ONCE per 16 pivots:
// column panel C, row panel B and A live in LDS, the trailing update is a rank-16 product from LDS.  1024 threads per matrix,
// thread (ty, tx) of a 32 x 32 grid owns the elements (ty + 32 a, tx + 32 b).  [k_factor<T, false>: 30 ms for the 18 matrices of the
// sparse MPC problem; one pivot per pass with 1024 threads: 7.8 ms; this kernel: DESIGN.md section 5]
constexpr int FB_N = 320, FB_P = 16;
constexpr size_t fb_lds_bytes() { return ((size_t)FB_N * (FB_P + 1) + (size_t)FB_P * FB_N + (size_t)FB_P * (FB_P + 1) + 8) * sizeof(double); }

template <typename T>
__global__ void __launch_bounds__(1024) k_factor_blk(SetupArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int P = FB_P, RT = FB_N / 32;
    double* Cb = (double*)smem_raw;              // [n][P + 1]  column panel M[:, K] (before the pass)
    double* Rb = Cb + FB_N * (P + 1);            // [P][n]      row panel M[K, :], then A^-1 M[K, :]
    double* Pb = Rb + P * FB_N;                  // [P][P + 1]  pivot block, then its inverse
    const int n = a.n;
    const int mat = blockIdx.x / a.kwin, jslot = blockIdx.x % a.kwin;
    if (a.only && !a.only[mat]) return;
    double* M = a.fscratch + (size_t)blockIdx.x * n * n;
    const T* Ht = (const T*)a.Ht + (size_t)mat * n * a.ldn;
    const double* G = a.G + (size_t)mat * n * n;
    const double rho = a.rhos[(a.wbase ? a.wbase[mat] : 0) + jslot];
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    for (int i = tid; i < n * n; i += 1024) {
        const int r = i / n, c = i % n;
        const double hs = 0.5 * ((double)Ht[(size_t)r * a.ldn + c] + (double)Ht[(size_t)c * a.ldn + r]);
        M[i] = hs + (r == c ? a.sigma : 0.0) + rho * G[i];
    }
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += P) {
        const int kw = (n - k0 < P) ? (n - k0) : P;
        // ---- panels -> LDS
        for (int e = tid; e < n * P; e += 1024) {
            const int i = e / P, q = e % P;                          // column panel: P consecutive doubles of row i
            Cb[i * (P + 1) + q] = (q < kw) ? M[(size_t)i * n + k0 + q] : 0.0;
        }
        for (int e = tid; e < P * n; e += 1024) {
            const int p = e / n, j = e % n;
            Rb[p * FB_N + j] = (p < kw) ? M[(size_t)(k0 + p) * n + j] : 0.0;
        }
        if (tid < P * P) {
            const int p = tid / P, q = tid % P;
            Pb[p * (P + 1) + q] = (p < kw && q < kw) ? M[(size_t)(k0 + p) * n + k0 + q] : (p == q ? 1.0 : 0.0);
        }
        __syncthreads();
        // ---- A^-1: unblocked sweep of the pivot block (threads (p, q) of a 16 x 16 grid; identity padding past kw)
        for (int s2 = 0; s2 < P; ++s2) {
            double nv = 0.0;
            if (tid < P * P) {
                const int p = tid / P, q = tid % P;
                const double piv = 1.0 / Pb[s2 * (P + 1) + s2];
                const double cs = Pb[p * (P + 1) + s2], rs = Pb[s2 * (P + 1) + q], cur = Pb[p * (P + 1) + q];
                if (p == s2) nv = (q == s2) ? piv : rs * piv;
                else nv = (q == s2) ? -cs * piv : cur - cs * (rs * piv);
            }
            __syncthreads();
            if (tid < P * P) Pb[(tid / P) * (P + 1) + tid % P] = nv;
            __syncthreads();
        }
        // ---- row panel: R = A^-1 M[K, :] (threads 0 .. n-1, one column each); column panel out: -C A^-1 (threads 512 ..)
        if (tid < n) {
            const int j = tid;
            double rv[P];
#pragma unroll
            for (int q = 0; q < P; ++q) rv[q] = Rb[q * FB_N + j];
            const bool inK = j >= k0 && j < k0 + kw;
#pragma unroll 1
            for (int p = 0; p < P; ++p) {                            // (column j is this thread's alone: written in place)
                double s3 = 0.0;
#pragma unroll
                for (int q = 0; q < P; ++q) s3 = fma(Pb[p * (P + 1) + q], rv[q], s3);
                Rb[p * FB_N + j] = inK ? Pb[p * (P + 1) + (j - k0)] : s3;   // (K columns of the K rows: A^-1)
            }
        } else if (tid >= 512 && tid < 512 + n) {
            const int i = tid - 512;
            if (i < k0 || i >= k0 + kw) {
                double cv[P];
#pragma unroll
                for (int q = 0; q < P; ++q) cv[q] = Cb[i * (P + 1) + q];
#pragma unroll 1
                for (int q2 = 0; q2 < kw; ++q2) {
                    double s3 = 0.0;
#pragma unroll
                    for (int q = 0; q < P; ++q) s3 = fma(cv[q], Pb[q * (P + 1) + q2], s3);
                    M[(size_t)i * n + k0 + q2] = -s3;
                }
            }
        }
        __syncthreads();
        // ---- trailing update D - C R (rows outside K), rows of K <- R
#pragma unroll 1
        for (int a0 = 0; a0 < RT; a0 += 2) {                         // two rows of the thread at a time: 2 x 10 accumulators
            const int i0 = ty + 32 * a0, i1 = i0 + 32;
            if (i0 >= n) break;
            double acc0[RT], acc1[RT];
#pragma unroll
            for (int b2 = 0; b2 < RT; ++b2) { acc0[b2] = 0.0; acc1[b2] = 0.0; }
            const bool k_0 = i0 >= k0 && i0 < k0 + kw, k_1 = i1 >= k0 && i1 < k0 + kw, ok1 = i1 < n;
            if (!(k_0 && (k_1 || !ok1))) {
#pragma unroll 1
                for (int p = 0; p < P; ++p) {
                    const double c0 = Cb[i0 * (P + 1) + p], c1 = ok1 ? Cb[i1 * (P + 1) + p] : 0.0;
#pragma unroll
                    for (int b2 = 0; b2 < RT; ++b2) {
                        const double r = Rb[p * FB_N + tx + 32 * b2];
                        acc0[b2] = fma(c0, r, acc0[b2]);
                        acc1[b2] = fma(c1, r, acc1[b2]);
                    }
                }
            }
#pragma unroll
            for (int b2 = 0; b2 < RT; ++b2) {
                const int j = tx + 32 * b2;
                if (j < n) {
                    const bool jK = j >= k0 && j < k0 + kw;
                    if (k_0) M[(size_t)i0 * n + j] = Rb[(i0 - k0) * FB_N + j];
                    else if (!jK) M[(size_t)i0 * n + j] -= acc0[b2];
                    if (ok1) {
                        if (k_1) M[(size_t)i1 * n + j] = Rb[(i1 - k0) * FB_N + j];
                        else if (!jK) M[(size_t)i1 * n + j] -= acc1[b2];
                    }
                }
            }
        }
        __syncthreads();
    }
    T* K = (T*)a.K + ((size_t)mat * a.kwin + jslot) * n * a.ldn;
    for (int i = tid; i < n * a.ldn; i += 1024) {
        const int r = i / a.ldn, c = i % a.ldn;
        // symmetrise the rounded result so that column-oriented products see one matrix
        K[i] = (c < n) ? (T)(0.5 * (M[(size_t)r * n + c] + M[(size_t)c * n + r])) : T(0);
    }
}

// Fast path for n <= 128: same in-place Gauss-Jordan, but thread t owns column c = t & (CN-1) of the row slice
// rs = t / CN (no integer division in the sweep, row k of the step in a register, column k broadcast from LDS).
// CN = 64 or 128 columns (power of two >= n).
template <typename T, int CN>
__global__ void __launch_bounds__(256) k_factor_fast(SetupArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int RS = 256 / CN;                 // row slices
    const int n = a.n;
    const int mat = blockIdx.x / a.kwin, j = blockIdx.x % a.kwin;
    if (a.only && !a.only[mat]) return;
    double* colb = (double*)smem_raw;            // [n] column k before the sweep
    double* rowb = colb + n;                     // [n] scaled pivot row
    double* M = rowb + n;                        // [n][n] row-major (consecutive c -> consecutive banks); 2 WGs/CU at n = 100
    const T* Ht = (const T*)a.Ht + (size_t)mat * n * a.ldn;
    const double* G = a.G + (size_t)mat * n * n;
    const double rho = a.rhos[(a.wbase ? a.wbase[mat] : 0) + j];
    const int tid = threadIdx.x, c = tid & (CN - 1), rs = tid / CN;
    const bool cin = c < n;
    if (cin)
        for (int r = rs; r < n; r += RS) {
            const double hs = 0.5 * ((double)Ht[(size_t)r * a.ldn + c] + (double)Ht[(size_t)c * a.ldn + r]);
            M[r * n + c] = hs + (r == c ? a.sigma : 0.0) + rho * G[(size_t)r * n + c];
        }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const double p = 1.0 / M[k * n + k];
        // column k (before the sweep) -> colb ; scaled row k -> rowb (written by the slice that owns row k)
        if (tid < n) colb[tid] = M[tid * n + k];
        if (cin && rs == (k & (RS - 1))) rowb[c] = (c == k) ? p : M[k * n + c] * p;
        __syncthreads();
        if (cin) {
            const double rb = rowb[c];
            const double f = (c == k) ? 0.0 : 1.0;            // column k: M[r][k] = -colb[r] * p  (rb = p there)
            // rows in batches of UB: all loads of a batch are issued before its FMAs (LDS latency overlaps)
            constexpr int UB = 8;
            for (int r0 = rs; r0 < n; r0 += RS * UB) {
                double mv[UB], cb[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int r = r0 + u * RS;
                    const bool ok = r < n;
                    mv[u] = ok ? M[r * n + c] : 0.0;
                    cb[u] = ok ? colb[r] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) mv[u] = fma(-cb[u], rb, f * mv[u]);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int r = r0 + u * RS;
                    if (r < n) M[r * n + c] = mv[u];
                }
            }
            if (rs == (k & (RS - 1))) M[k * n + c] = rb;      // row k: the scaled pivot row (same thread as the sweep)
        }
        __syncthreads();
    }
    T* K = (T*)a.K + ((size_t)mat * a.kwin + j) * n * a.ldn;
    for (int r = rs; r < n; r += RS)
        if (c < a.ldn) K[(size_t)r * a.ldn + c] = cin ? (T)(0.5 * (M[r * n + c] + M[c * n + r])) : T(0);
}


// Register-resident factorisation for n <= 112 (the sizes of every resident ADMM tile): M_j never leaves the VGPRs.
// In-place SYMMETRIC SWEEP (M -> -M^-1, SPD, no pivoting): at step k, with row = old row k (= old column k), d = row[k],
//     M[r][c] -= row[r] row[c] / d   (r, c != k);   M[r][k] = M[k][r] = row[r] / d;   M[k][k] = -1 / d
// so only ROW k has to be shared: its owners write it to LDS (double-buffered: ONE barrier per step).  The LDS Gauss-Jordan
// (k_factor_fast below, round 1) moves 24 bytes of LDS per FMA.  The step loop is fully instantiated (a fold over an
// integer sequence: the pivot row's register index must be a compile-time constant, and `#pragma unroll` gives up on 100+
// steps of this size -- a rolled loop would index registers dynamically, i.e. put M in scratch memory).
// Thread layout: a first version gave thread (column, row slice) a column strip of R = 52 rows; every lane then read the R
// row values of its slice per step (wave-uniform addresses, but a broadcast ds_read still returns 64 x 16 B): 26 KB of LDS
// return per wave and step against 52 FMAs per lane -- it ran at the LDS return bandwidth (12.4 ms for the 73 728 matrices
// of the headline batch).  Thread (ty, tx) of a 16 x 16 grid now owns the RT x RT elements (ty + 16 i, tx + 16 j): RT row
// values for its rows and RT for its columns -- 2 RT reads for RT^2 FMAs (n <= 16 RT; RT = 7: 14 reads for 49 FMAs) --
// and the column-k / row-k fix-ups touch RT elements instead of R: 7.6 ms.
// The sweep keeps M symmetric, so only the blocks (i, j) with i <= j of a thread are stored and updated (RT = 7: 28 of 49
// FMAs per step, 56 instead of 98 matrix registers -> 4 workgroups per CU): element (r, c) of a lower block is element
// (c, r) of thread (tx, ty)'s upper block.  Row k is posted by its two sets of owners: the threads with ty == ko hold
// (k, c) for the blocks j >= kb, the threads with tx == ko hold (c, k) = (k, c) for the blocks i < kb.  [7.6 -> ms below]
// TG: side of the thread grid.  16 (256 threads, four waves) for n <= 112; 8 (ONE wave per matrix, RT = 4: the barrier of a
// step is free) for n <= 32 on the table path -- 256 threads on a 32 x 32 matrix hold 3 elements each and spend a step on its
// barrier: 0.33 ms for the 40 960 inversions of the config-4 batch.
template <typename T, int RT, int TG = 16>
__global__ void __launch_bounds__(TG * TG, TG == 16 ? 4 : 8) k_factor_reg2(SetupArgs a) {
    constexpr int NMAX = TG * RT;
    __shared__ __attribute__((aligned(16))) double rowbuf[2][NMAX];
    __shared__ __attribute__((aligned(16))) double tb[RT][TG][TG + 1];       // one output pass: RT blocks, closed under transposition
    const int n = a.n;
    const int mat = blockIdx.x / a.kwin, jrho = blockIdx.x % a.kwin;     // jrho: K slot; ladder index = window base + slot
    if (a.only && !a.only[mat]) return;                                   // (uniform) re-factor of moved windows only
    const int tid = threadIdx.x, tx = tid & (TG - 1), ty = tid / TG;
    const T* Ht = (const T*)a.Ht + (size_t)mat * n * a.ldn;
    const double* G = a.G + (size_t)mat * n * n;
    const double rho = a.rhos[(a.wbase ? a.wbase[mat] : 0) + jrho];
    double mreg[RT][RT];                                                  // blocks i <= j only
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = i; j < RT; ++j) {
            const int r = ty + TG * i, c = tx + TG * j;
            mreg[i][j] = 0.0;
            if (r < n && c < n) mreg[i][j] = (double)Ht[(size_t)r * a.ldn + c] + (r == c ? a.sigma : 0.0) + rho * G[(size_t)r * n + c];
        }
    // (every thread divides by the pivot after the barrier: letting only the owners of row k divide -- they would post
    //  row / d as well -- puts a cross-lane fetch of d and the division in front of the barrier: 8.9 ms instead of 7.6)
    auto step = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k < n) {                                                      // uniform
            constexpr int kb = k / TG, ko = k % TG;                       // row / column block and offset of the pivot
            double* rb = rowbuf[k & 1];
            if (ty == ko) {                                               // (k, c) for c in the blocks j >= kb
#pragma unroll
                for (int j = kb; j < RT; ++j) rb[tx + TG * j] = mreg[kb][j];
            }
            if (tx == ko) {                                               // (k, c) = (c, k) for c in the blocks i < kb
#pragma unroll
                for (int i = 0; i < kb; ++i) rb[ty + TG * i] = mreg[i][kb];
            }
            __syncthreads();
            // 1 / d: hardware reciprocal + two Newton steps (5 instructions; the IEEE division sequence is ~15, and every
            // thread runs it every step).  d > 0 is a Schur-complement pivot of an SPD matrix: no special cases to fix up.
            const double dk = rb[k];
            double p = __builtin_amdgcn_rcp(dk);
            p = fma(p, fma(-dk, p, 1.0), p);
            p = fma(p, fma(-dk, p, 1.0), p);
            double rr[RT], tc[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) rr[i] = rb[ty + TG * i];
#pragma unroll
            for (int j = 0; j < RT; ++j) tc[j] = rb[tx + TG * j];
#pragma unroll
            for (int j = 0; j < RT; ++j) tc[j] = -tc[j] * p;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = i; j < RT; ++j) mreg[i][j] = fma(rr[i], tc[j], mreg[i][j]);
            if (tx == ko) {                                               // column k: row[r] / d
#pragma unroll
                for (int i = 0; i <= kb; ++i) mreg[i][kb] = rr[i] * p;
            }
            if (ty == ko) {                                               // row k: row[c] / d ; pivot: -1 / d
#pragma unroll
                for (int j = kb; j < RT; ++j) mreg[kb][j] = -tc[j];
                if (tx == ko) mreg[kb][kb] = -p;
            }
        }
    };
    rqp_static_for(std::make_integer_sequence<int, NMAX>{}, step);
    // K_j = -M.  Pass d moves the RT blocks with (i + j) mod RT == d (a set closed under transposition), slot i = block
    // (i, (d - i) mod RT): an upper block comes from the thread's own registers, a lower block (i, j), i > j, is the
    // transpose of block (j, i) of thread (tx, ty) -- slot j of the same pass -- and a diagonal block averages the two
    // roundings of (r, c) and (c, r), which both exist there.
    if constexpr (std::is_same<T, float>::value && TG == 16) {
        if (a.kp_img) {
            // Straight into the register image of k_admm_res2 (rqp_resident2.hip, k_pack_res2's layout):
            //   Kpack[mat][slot][pair = kp*KC + c][t = 64 w + lane][h] = K_j[CW w + KR rr + 2 kp + h][KC cc + c],  rr = lane >> 3, cc = lane & 7
            // Two halves of the rows (waves 0-1, then 2-3 of the solve kernel) pass through a float stage in LDS of 2 CW x ldn
            // elements -- with the whole matrix staged, four workgroups of this kernel would no longer share a CU --: the same
            // stage is filled (below), then all 256 threads gather the half's pairs (two thread groups take half of the pairs
            // each) and write them as float2, 512 contiguous bytes per wave.
            extern __shared__ __attribute__((aligned(16))) float kstage[];
            const int CW = a.kp_cw, KR = a.kp_kr, KC = a.kp_kc, KE2 = (KR / 2) * KC, ldn = a.ldn;
            float2* Kp = (float2*)a.kp_img + ((size_t)mat * a.kwin + jrho) * KE2 * 256;
            // The stage is filled WITHOUT the transposing passes of the table path: an upper block's element (r, c) is also element
            // (c, r) of the lower block (the sweep keeps M symmetric; the table path reads exactly that value through LDS), so its
            // owner writes both; only the diagonal blocks need the partner's rounding (their average), one exchange for all RT.
            float fv[RT][RT];                                          // K_j = -M in float32 (upper blocks; the doubles are dead from here)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < RT; ++i) tb[i][ty][tx] = mreg[i][i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = i; j < RT; ++j) fv[i][j] = (float)(-(i == j ? 0.5 * (mreg[i][i] + tb[i][tx][ty]) : mreg[i][j]));
            for (int half = 0; half < 2; ++half) {
                const int r0 = 2 * half * CW, r1 = r0 + 2 * CW;
                if (half) __syncthreads();                              // (the gather of the first half is done with the stage)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = i; j < RT; ++j) {
                        const int r = ty + TG * i, c = tx + TG * j;
                        const float v = fv[i][j];
                        if (r >= r0 && r < r1 && r < n && c < ldn) kstage[(r - r0) * ldn + c] = (c < n) ? v : 0.f;
                        if (i != j && c >= r0 && c < r1 && c < n && r < ldn) kstage[(c - r0) * ldn + r] = (r < n) ? v : 0.f;
                    }
                __syncthreads();
                const int tt = tid & 127, wl = tt >> 6, lane = tt & 63, rr = lane >> 3, cc = lane & 7;
                const int tk = 64 * (2 * half + wl) + lane;
                const int ph = (KE2 + 1) / 2, p0 = (tid >> 7) * ph, p1 = min(KE2, p0 + ph);
                int kp2 = 2 * (p0 / KC), c0 = p0 % KC;                  // (KC is a run-time value here: one division, then counters)
                for (int pr = p0; pr < p1; ++pr) {
                    const int lr = KR * rr + kp2, c = KC * cc + c0;
                    if (++c0 == KC) { c0 = 0; kp2 += 2; }
                    const int r = r0 + CW * wl + lr;
                    float2 v;
                    v.x = (lr < CW && r < n && c < n) ? kstage[(CW * wl + lr) * ldn + c] : 0.f;
                    v.y = (lr + 1 < CW && r + 1 < n && c < n) ? kstage[(CW * wl + lr + 1) * ldn + c] : 0.f;
                    Kp[(size_t)pr * 256 + tk] = v;
                }
            }
            return;
        }
    }
    T* K = (T*)a.K + ((size_t)mat * a.kwin + jrho) * n * a.ldn;
#pragma unroll
    for (int d = 0; d < RT; ++d) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int j = (d - i + RT) % RT;
            if (i <= j) tb[i][ty][tx] = mreg[i][j];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int j = (d - i + RT) % RT;
            const int r = ty + TG * i, c = tx + TG * j;
            double v;
            if (i < j) v = mreg[i][j];
            else if (i > j) v = tb[j][tx][ty];                            // M[c][r]: thread (tx, ty), block (j, i)
            else v = 0.5 * (mreg[i][i] + tb[i][tx][ty]);
            if (r < n && c < a.ldn) K[(size_t)r * a.ldn + c] = (c < n) ? (T)(-v) : T(0);
        }
    }
}

template <typename T, int RT, int TG = 16>
static hipError_t launch_factor_reg2(rqp_handle* h, const SetupArgs& a, hipStream_t s) {
    // (kpack_direct: + a float stage of half of the register image's rows; 17 KB static + <= 23 KB: four workgroups still share a CU)
    const size_t stage = a.kp_img ? (size_t)2 * a.kp_cw * a.ldn * sizeof(float) : 0;
    k_factor_reg2<T, RT, TG><<<a.nmat * a.kwin, TG * TG, stage, s>>>(a);
    return hipGetLastError();
}

template <typename T, int CN>
static hipError_t launch_factor_fast(rqp_handle* h, const SetupArgs& a, hipStream_t s) {
    const size_t lds = (2 * (size_t)h->n + (size_t)h->n * h->n) * sizeof(double);
    hipError_t e = rqp_raise_lds_limit((const void*)k_factor_fast<T, CN>, (size_t)lds);
    if (e != hipSuccess) return e;
    k_factor_fast<T, CN><<<a.nmat * a.kwin, 256, lds, s>>>(a);
    return hipGetLastError();
}

hipError_t rqp_launch_factor(rqp_handle* h, const SetupArgs& a, hipStream_t s) {
    const int n = h->n;
    if (n <= 32 && !a.kp_img) return h->esz == 4 ? launch_factor_reg2<float, 4, 8>(h, a, s) : launch_factor_reg2<double, 4, 8>(h, a, s);
    if (n <= 32) return h->esz == 4 ? launch_factor_reg2<float, 2>(h, a, s) : launch_factor_reg2<double, 2>(h, a, s);
    if (n <= 64) return h->esz == 4 ? launch_factor_reg2<float, 4>(h, a, s) : launch_factor_reg2<double, 4>(h, a, s);
    if (n <= 112 && h->ldn <= 112)
        return h->esz == 4 ? launch_factor_reg2<float, 7>(h, a, s) : launch_factor_reg2<double, 7>(h, a, s);
    if (n <= 64) return h->esz == 4 ? launch_factor_fast<float, 64>(h, a, s) : launch_factor_fast<double, 64>(h, a, s);
    if (n <= 128 && h->ldn <= 128)
        return h->esz == 4 ? launch_factor_fast<float, 128>(h, a, s) : launch_factor_fast<double, 128>(h, a, s);
    const size_t lds_need = ((size_t)n * n + 2 * (size_t)n) * sizeof(double);
    const bool lds_mode = lds_need <= 160 * 1024 - 512;
    const int grid = a.nmat * a.kwin;
    hipError_t e;
    if (lds_mode) {
        if (h->esz == 4) {
            e = rqp_raise_lds_limit((const void*)k_factor<float, true>, (size_t)lds_need);
            if (e != hipSuccess) return e;
            k_factor<float, true><<<grid, 256, lds_need, s>>>(a);
        } else {
            e = rqp_raise_lds_limit((const void*)k_factor<double, true>, (size_t)lds_need);
            if (e != hipSuccess) return e;
            k_factor<double, true><<<grid, 256, lds_need, s>>>(a);
        }
    } else if (n <= FB_N) {
        if (h->esz == 4) {
            e = rqp_raise_lds_limit((const void*)k_factor_blk<float>, fb_lds_bytes());
            if (e != hipSuccess) return e;
            k_factor_blk<float><<<grid, 1024, fb_lds_bytes(), s>>>(a);
        } else {
            e = rqp_raise_lds_limit((const void*)k_factor_blk<double>, fb_lds_bytes());
            if (e != hipSuccess) return e;
            k_factor_blk<double><<<grid, 1024, fb_lds_bytes(), s>>>(a);
        }
    } else {
        const size_t small = 2 * (size_t)n * sizeof(double);
        if (h->esz == 4)
            k_factor<float, false><<<grid, 256, small, s>>>(a);
        else
            k_factor<double, false><<<grid, 256, small, s>>>(a);
    }
    return hipGetLastError();
}

// ----------------------------------------------------------------------- state movers
template <typename T>
__global__ void k_state_set(int B, int n, int m, const T* x, const T* z, const T* lam, double* xs, double* zs,
                            double* ls, int set_rho, int rho_ind, int32_t* ri) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    if (x) for (size_t i = tid; i < (size_t)B * n; i += nth) xs[i] = (double)x[i];
    if (z) for (size_t i = tid; i < (size_t)B * m; i += nth) zs[i] = (double)z[i];
    if (lam) for (size_t i = tid; i < (size_t)B * m; i += nth) ls[i] = (double)lam[i];
    if (set_rho) for (int i = tid; i < B; i += nth) ri[i] = rho_ind;
}

hipError_t rqp_launch_state_set(const rqp_handle* h, const void* x, const void* z, const void* lam, int set_rho,
                                int rho_ind, hipStream_t s) {
    if (h->esz == 4)
        k_state_set<float><<<256, 256, 0, s>>>(h->B, h->n, h->m, (const float*)x, (const float*)z, (const float*)lam,
                                                 h->x, h->z, h->lam, set_rho, rho_ind, h->rho_ind);
    else
        k_state_set<double><<<256, 256, 0, s>>>(h->B, h->n, h->m, (const double*)x, (const double*)z,
                                                  (const double*)lam, h->x, h->z, h->lam, set_rho, rho_ind, h->rho_ind);
    return hipGetLastError();
}

template <typename T>
__global__ void k_state_get(int B, int n, int m, T* x, T* z, T* lam, const double* xs, const double* zs,
                            const double* ls, int32_t* ri_out, const int32_t* ri) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    if (x) for (size_t i = tid; i < (size_t)B * n; i += nth) x[i] = (T)xs[i];
    if (z) for (size_t i = tid; i < (size_t)B * m; i += nth) z[i] = (T)zs[i];
    if (lam) for (size_t i = tid; i < (size_t)B * m; i += nth) lam[i] = (T)ls[i];
    if (ri_out) for (int i = tid; i < B; i += nth) ri_out[i] = ri[i];
}

hipError_t rqp_launch_state_get(const rqp_handle* h, void* x, void* z, void* lam, int32_t* rho_ind, hipStream_t s) {
    if (h->esz == 4)
        k_state_get<float><<<256, 256, 0, s>>>(h->B, h->n, h->m, (float*)x, (float*)z, (float*)lam, h->x, h->z, h->lam,
                                                 rho_ind, h->rho_ind);
    else
        k_state_get<double><<<256, 256, 0, s>>>(h->B, h->n, h->m, (double*)x, (double*)z, (double*)lam, h->x, h->z,
                                                  h->lam, rho_ind, h->rho_ind);
    return hipGetLastError();
}

// order[] = instances sorted by descending last_iter (counting sort, one workgroup; ties in arrival order of the atomics --
// the order only schedules workgroups, results do not depend on it)
__global__ void __launch_bounds__(1024) k_order_lpt(int B, const int32_t* __restrict__ last_iter, int32_t* __restrict__ order) {
    constexpr int NBK = 4096;
    __shared__ int hist[NBK];                     // indexed by NBK-1 - min(iter, NBK-1): ascending index = descending iterations
    __shared__ int part[1024];
    const int t = threadIdx.x;
    for (int i = t; i < NBK; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = t; i < B; i += 1024) atomicAdd(&hist[NBK - 1 - min(max(last_iter[i], 0), NBK - 1)], 1);
    __syncthreads();
    int loc[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        loc[j] = sum;
        sum += hist[4 * t + j];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {    // inclusive Hillis-Steele scan of the 1024 partial sums
        const int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int excl = part[t] - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) hist[4 * t + j] = excl + loc[j];
    __syncthreads();
    for (int i = t; i < B; i += 1024) order[atomicAdd(&hist[NBK - 1 - min(max(last_iter[i], 0), NBK - 1)], 1)] = i;
}

// order[] = instances grouped by key (descending), STABLE: ties keep their index order, so the result is a pure function of the keys
// (the MFMA kernels group their slots by rho index; which instances share a tile decides when a tile hands its stragglers over, so an
// arrival-order sort would make those solves irreproducible).  Keys 0 .. 63; one workgroup; one block scan per DISTINCT key.
__global__ void __launch_bounds__(1024) k_order_stable(int B, const int32_t* __restrict__ key, int32_t* __restrict__ order) {
    __shared__ unsigned long long present;
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int C = (B + 1023) / 1024, lo = t * C, hi = min(B, lo + C);
    if (t == 0) present = 0ull;
    __syncthreads();
    unsigned long long mine = 0ull;
    for (int i = lo; i < hi; ++i) mine |= 1ull << min(max(key[i], 0), 63);
    for (int off = 32; off >= 1; off >>= 1) mine |= __shfl_xor(mine, off, 64);
    if (lane == 0 && mine) atomicOr(&present, mine);
    __syncthreads();
    unsigned long long pm = present;
    int b0 = 0;
    while (pm) {
        const int k = 63 - __clzll((long long)pm);
        pm &= ~(1ull << k);
        int c = 0;
        for (int i = lo; i < hi; ++i) c += (min(max(key[i], 0), 63) == k);
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
        }
        int pos = b0 + woff + incl - c;
        for (int i = lo; i < hi; ++i)
            if (min(max(key[i], 0), 63) == k) order[pos++] = i;
        b0 += tot;
        __syncthreads();
    }
}

hipError_t rqp_launch_order_by(const rqp_handle* h, const int32_t* key, hipStream_t s) {
    k_order_stable<<<1, 1024, 0, s>>>(h->B, key, h->order_d);
    return hipGetLastError();
}

hipError_t rqp_launch_order_lpt(const rqp_handle* h, hipStream_t s) {
    k_order_lpt<<<1, 1024, 0, s>>>(h->B, h->last_iter_d, h->order_d);
    return hipGetLastError();
}

template <typename T>
__global__ void k_get_K(int n, int ldn, const T* K, T* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += gridDim.x * blockDim.x)
        out[i] = K[(size_t)(i / n) * ldn + (i % n)];
}

// Kmat: one n x ldn matrix of the K table (or of a scratch copy a single-matrix factor call just filled)
hipError_t rqp_launch_get_K(const rqp_handle* h, const void* Kmat, void* out, hipStream_t s) {
    if (h->esz == 4)
        k_get_K<float><<<64, 256, 0, s>>>(h->n, h->ldn, (const float*)Kmat, (float*)out);
    else
        k_get_K<double><<<64, 256, 0, s>>>(h->n, h->ldn, (const double*)Kmat, (double*)out);
    return hipGetLastError();
}

// Window bookkeeping of a windowed handle.  all = 0: the instances that left their window mid-solve (cstat = 1) get a window
// centred on their current index.  all = 1 (before rqp_iterate / rqp_compute_residuals, which run no exit-and-continue
// protocol): every instance whose index lies outside its window is marked (cstat = 1) and re-centred, the others cleared.
__global__ void k_rewindow(int B, int nrho, int kwin, int all, const int32_t* __restrict__ rho_ind, int32_t* __restrict__ cstat,
                           int32_t* __restrict__ wbase) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const int ri = rho_ind[b];
        if (all) cstat[b] = (ri < wbase[b] || ri >= wbase[b] + kwin) ? 1 : 0;
        if (cstat[b]) wbase[b] = min(max(ri - kwin / 2, 0), nrho - kwin);
    }
}

hipError_t rqp_launch_rewindow(const rqp_handle* h, int all, hipStream_t s) {
    k_rewindow<<<(h->B + 255) / 256, 256, 0, s>>>(h->B, h->nrho, h->kwin, all, h->rho_ind, h->cstat_d, h->wbase_d);
    return hipGetLastError();
}
