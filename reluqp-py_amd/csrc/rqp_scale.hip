// rqp_scale.hip -- problem scaling (Ruiz equilibration), SURVEY.md 8(f)-3.
//
// The reference accepts a `scaling` argument and never uses it (`# TODO: implement scaling`, reluqpth.py:105,335;
// classes.py:35).  Here settings.scaling = k > 0 runs k passes of the modified Ruiz equilibration of OSQP on the KKT
// matrix [[H, A'], [A, 0]] at setup:
//     delta_j = 1 / sqrt(max(|Hbar[:, j]|_inf, |Abar[:, j]|_inf))   (variables),   eps_i = 1 / sqrt(|Abar[i, :]|_inf)   (rows)
//     D <- D delta,  E <- E eps,  Hbar = D H D,  Abar = E A D            (norms below 1e-4 count as 1; factors clamped to [1e-4, 1e4])
// followed by the cost scaling c = 1 / max(mean_j |Hbar[:, j]|_inf, 1e-4) (per matrix; q does not enter, so that a shared
// (H, A) has ONE scaling for the whole batch).  The ADMM kernels then solve, unchanged,
//     min 1/2 xb' (c Hbar) xb + (c D g)' xb   s.t.  E l <= Abar xb <= E u,        x = D xb,  z = E^-1 zb,  lam = E lamb / c
// and every boundary of the C ABI converts: inputs (g, l, u, warm starts) are scaled on the way in, outputs (x, z, lam,
// obj) un-scaled on the way out.  compute_residuals takes every term back to the caller's space before its inf-norm (row i of
// the primal side x 1 / E_i, column j of the dual side x 1 / (c D_j): SolveArgs.scE, every ADMM kernel), so the termination
// test certifies eps_abs / eps_rel in the caller's units and pri_res / dua_res are reported there (OSQP's default,
// scaled_termination = 0); the rho estimate is formed from the same caller-space ratios.  rqp_get_K / rqp_iterate expose
// the scaled space as it is.
#include "rqp_common.h"

namespace {
constexpr double MIN_SCALING = 1e-4, MAX_SCALING = 1e4;

__device__ __forceinline__ double limit_scaling(double nrm) {          // OSQP limit_scaling + 1/sqrt
    if (nrm < MIN_SCALING) nrm = 1.0;
    if (nrm > MAX_SCALING) nrm = MAX_SCALING;
    return 1.0 / sqrt(nrm);
}
}   // namespace

// One workgroup per matrix.  Ht (= sym(H), n x ldn) and A (m x ldn) are scaled IN PLACE, pass after pass; At is rebuilt at
// the end.  D [nmat][n], E [nmat][m], cs [nmat] (float64).
template <typename T>
__global__ void __launch_bounds__(256) k_ruiz(int n, int m, int ldn, int ldm, int passes, T* __restrict__ Ht, T* __restrict__ A,
                                              T* __restrict__ At, double* __restrict__ D, double* __restrict__ E,
                                              double* __restrict__ cs) {
    const int mat = blockIdx.x, tid = threadIdx.x;
    T* H = Ht + (size_t)mat * n * ldn;
    T* Am = A + (size_t)mat * m * ldn;
    T* Atm = At + (size_t)mat * n * ldm;
    double* Dm = D + (size_t)mat * n;
    double* Em = E + (size_t)mat * m;
    extern __shared__ double sh[];                 // [n] delta | [m] eps | [256] reduction
    double* dl = sh;
    double* ep = sh + n;
    double* red = ep + m;
    for (int j = tid; j < n; j += 256) Dm[j] = 1.0;
    for (int i = tid; i < m; i += 256) Em[i] = 1.0;
    __syncthreads();
    for (int it = 0; it < passes; ++it) {
        // column norms of [Hbar; Abar] (H symmetric: column j = row j) and row norms of Abar
        for (int j = tid; j < n; j += 256) {
            double nv = 0.0;
            for (int i = 0; i < n; ++i) nv = fmax(nv, fabs((double)H[(size_t)i * ldn + j]));
            for (int i = 0; i < m; ++i) nv = fmax(nv, fabs((double)Am[(size_t)i * ldn + j]));
            dl[j] = limit_scaling(nv);
        }
        for (int i = tid; i < m; i += 256) {
            double nv = 0.0;
            for (int j = 0; j < n; ++j) nv = fmax(nv, fabs((double)Am[(size_t)i * ldn + j]));
            ep[i] = limit_scaling(nv);
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += 256) {
            const int i = e / n, j = e % n;
            H[(size_t)i * ldn + j] = (T)(((double)H[(size_t)i * ldn + j] * dl[i]) * dl[j]);
        }
        for (int e = tid; e < m * n; e += 256) {
            const int i = e / n, j = e % n;
            Am[(size_t)i * ldn + j] = (T)(((double)Am[(size_t)i * ldn + j] * ep[i]) * dl[j]);
        }
        for (int j = tid; j < n; j += 256) Dm[j] *= dl[j];
        for (int i = tid; i < m; i += 256) Em[i] *= ep[i];
        __syncthreads();
    }
    // cost scaling: c = 1 / max(mean column norm of Hbar, MIN_SCALING), clamped like the other factors
    double part = 0.0;
    for (int j = tid; j < n; j += 256) {
        double nv = 0.0;
        for (int i = 0; i < n; ++i) nv = fmax(nv, fabs((double)H[(size_t)i * ldn + j]));
        part += nv;
    }
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int t = 0; t < 256; ++t) s += red[t];          // fixed order
        double cv = s / (double)n;
        if (cv < MIN_SCALING) cv = 1.0;
        if (cv > MAX_SCALING) cv = MAX_SCALING;
        red[0] = passes > 0 ? 1.0 / cv : 1.0;
        cs[mat] = red[0];
    }
    __syncthreads();
    const double c = red[0];
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e % n;
        H[(size_t)i * ldn + j] = (T)((double)H[(size_t)i * ldn + j] * c);
    }
    if (At) {                                      // (no transposed copy on handles whose kernels never read it: rqp_setup)
        for (int e = tid; e < n * ldm; e += 256) {
            const int r = e / ldm, cc = e % ldm;
            Atm[e] = (cc < m) ? Am[(size_t)cc * ldn + r] : T(0);
        }
    }
}

// In place on the handle's vectors: g *= c D (per matrix), l, u *= E.  rows of g / l / u are [B][n] / [B][m].
template <typename T>
__global__ void k_scale_vecs(int B, int n, int m, int shared, T* g, T* l, T* u, const double* __restrict__ D,
                             const double* __restrict__ E, const double* __restrict__ cs) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if (g) for (size_t i = tid; i < (size_t)B * n; i += nth) {
        const size_t b = i / n, j = i % n, mt = shared ? 0 : b;
        g[i] = (T)((double)g[i] * (cs[mt] * D[mt * n + j]));
    }
    for (size_t i = tid; i < (size_t)B * m; i += nth) {
        const size_t b = i / m, r = i % m, mt = shared ? 0 : b;
        const double e = E[mt * m + r];
        if (l) l[i] = (T)((double)l[i] * e);
        if (u) u[i] = (T)((double)u[i] * e);
    }
}

// state <-> caller space.  dir = +1: caller -> scaled (x / D, z * E, lam * c / E), in place on float64 state arrays;
// dir = -1: scaled -> caller on output arrays of type T (x * D, z / E, lam * E / c) and obj / c.
__global__ void k_scale_state(int B, int n, int m, int shared, double* x, double* z, double* lam, const double* __restrict__ D,
                              const double* __restrict__ E, const double* __restrict__ cs) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if (x) for (size_t i = tid; i < (size_t)B * n; i += nth) {
        const size_t b = i / n, mt = shared ? 0 : b;
        x[i] = x[i] / D[mt * n + i % n];
    }
    for (size_t i = tid; i < (size_t)B * m; i += nth) {
        const size_t b = i / m, mt = shared ? 0 : b;
        const double e = E[mt * m + i % m];
        if (z) z[i] = z[i] * e;
        if (lam) lam[i] = lam[i] * (cs[mt] / e);
    }
}

template <typename T>
__global__ void k_unscale_out(int B, int n, int m, int shared, T* x, T* z, T* lam, double* obj, const double* __restrict__ D,
                              const double* __restrict__ E, const double* __restrict__ cs) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if (x) for (size_t i = tid; i < (size_t)B * n; i += nth) {
        const size_t b = i / n, mt = shared ? 0 : b;
        x[i] = (T)((double)x[i] * D[mt * n + i % n]);
    }
    for (size_t i = tid; i < (size_t)B * m; i += nth) {
        const size_t b = i / m, mt = shared ? 0 : b;
        const double e = E[mt * m + i % m];
        if (z) z[i] = (T)((double)z[i] / e);
        if (lam) lam[i] = (T)((double)lam[i] * (e / cs[mt]));
    }
    if (obj) for (size_t b = tid; b < (size_t)B; b += nth) obj[b] = obj[b] / cs[shared ? 0 : b];
}

// New scaling factors after a matrix update: the handle's (already scaled) vectors and state move from the old space
// to the new one.  g *= (c' D') / (c D);  l, u *= E' / E;  xb *= D / D';  zb *= E' / E;  lamb *= (c' / c) (E / E').
template <typename T>
__global__ void k_rescale(int B, int n, int m, int shared, T* g, T* l, T* u, double* x, double* z, double* lam,
                          const double* __restrict__ D0, const double* __restrict__ E0, const double* __restrict__ c0,
                          const double* __restrict__ D1, const double* __restrict__ E1, const double* __restrict__ c1) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < (size_t)B * n; i += nth) {
        const size_t b = i / n, mt = shared ? 0 : b, k = mt * n + i % n;
        g[i] = (T)((double)g[i] * ((c1[mt] * D1[k]) / (c0[mt] * D0[k])));
        x[i] = x[i] * (D0[k] / D1[k]);
    }
    for (size_t i = tid; i < (size_t)B * m; i += nth) {
        const size_t b = i / m, mt = shared ? 0 : b, k = mt * m + i % m;
        const double e = E1[k] / E0[k];
        l[i] = (T)((double)l[i] * e);
        u[i] = (T)((double)u[i] * e);
        z[i] = z[i] * e;
        lam[i] = lam[i] * ((c1[mt] / c0[mt]) / e);
    }
}

// old = the factors in force before the matrix update (copies), handle fields = the new ones
hipError_t rqp_launch_rescale(const rqp_handle* h, const double* D0, const double* E0, const double* c0, hipStream_t s) {
    const int sh = h->dims.shared_mats != 0;
    if (h->esz == 4)
        k_rescale<float><<<256, 256, 0, s>>>(h->B, h->n, h->m, sh, (float*)h->g, (float*)h->l, (float*)h->u, h->x, h->z, h->lam, D0, E0,
                                             c0, h->Dsc, h->Esc, h->csc);
    else
        k_rescale<double><<<256, 256, 0, s>>>(h->B, h->n, h->m, sh, (double*)h->g, (double*)h->l, (double*)h->u, h->x, h->z, h->lam, D0,
                                              E0, c0, h->Dsc, h->Esc, h->csc);
    return hipGetLastError();
}

hipError_t rqp_launch_ruiz(const rqp_handle* h, hipStream_t s) {
    const size_t lds = ((size_t)h->n + h->m + 256) * sizeof(double);
    if (h->esz == 4)
        k_ruiz<float><<<h->nmat, 256, lds, s>>>(h->n, h->m, h->ldn, h->ldm, h->st.scaling, (float*)h->Ht, (float*)h->A, (float*)h->At,
                                                h->Dsc, h->Esc, h->csc);
    else
        k_ruiz<double><<<h->nmat, 256, lds, s>>>(h->n, h->m, h->ldn, h->ldm, h->st.scaling, (double*)h->Ht, (double*)h->A,
                                                 (double*)h->At, h->Dsc, h->Esc, h->csc);
    return hipGetLastError();
}

// g / l / u point into the handle's own (already copied) vectors; NULL = leave alone
hipError_t rqp_launch_scale_vecs(const rqp_handle* h, void* g, void* l, void* u, hipStream_t s) {
    const int sh = h->dims.shared_mats != 0;
    if (h->esz == 4)
        k_scale_vecs<float><<<256, 256, 0, s>>>(h->B, h->n, h->m, sh, (float*)g, (float*)l, (float*)u, h->Dsc, h->Esc, h->csc);
    else
        k_scale_vecs<double><<<256, 256, 0, s>>>(h->B, h->n, h->m, sh, (double*)g, (double*)l, (double*)u, h->Dsc, h->Esc, h->csc);
    return hipGetLastError();
}

hipError_t rqp_launch_scale_state(const rqp_handle* h, double* x, double* z, double* lam, hipStream_t s) {
    k_scale_state<<<256, 256, 0, s>>>(h->B, h->n, h->m, h->dims.shared_mats != 0, x, z, lam, h->Dsc, h->Esc, h->csc);
    return hipGetLastError();
}

hipError_t rqp_launch_unscale_out(const rqp_handle* h, void* x, void* z, void* lam, double* obj, hipStream_t s) {
    const int sh = h->dims.shared_mats != 0;
    if (h->esz == 4)
        k_unscale_out<float><<<256, 256, 0, s>>>(h->B, h->n, h->m, sh, (float*)x, (float*)z, (float*)lam, obj, h->Dsc, h->Esc, h->csc);
    else
        k_unscale_out<double><<<256, 256, 0, s>>>(h->B, h->n, h->m, sh, (double*)x, (double*)z, (double*)lam, obj, h->Dsc, h->Esc, h->csc);
    return hipGetLastError();
}
