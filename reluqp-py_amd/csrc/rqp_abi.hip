// rqp_abi.hip -- the extern "C" boundary declared in include/rqp_abi.h.
// Host orchestration only: argument validation, workspace ownership, kernel dispatch.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#include "rqp_common.h"

#define RQP_VERSION "rqp-hip 0.1 gfx950"

namespace {

int fail_hip(rqp_handle* h, hipError_t e, const char* what) {
    if (h) h->err = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorOutOfMemory) ? RQP_ERR_OOM : RQP_ERR_HIP;
}
int fail_arg(rqp_handle* h, const char* what) {
    if (h) h->err = what;
    return RQP_ERR_ARG;
}

#define HIP_TRY(h, call)                                      \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return fail_hip(h, e__, #call); \
    } while (0)

bool settings_valid(const rqp_settings& s) {
    return s.rho > 0 && s.rho_min > 0 && s.rho_max >= s.rho_min && s.sigma >= 0 && s.adaptive_rho_tolerance > 1 &&
           s.eps_abs >= 0 && s.max_iter >= 0 && s.check_interval >= 1;
}

// setup_rhos, reluqpth.py:20-38: repeated division / multiplication in doubles, sorted.
std::vector<double> build_rhos(const rqp_settings& s) {
    std::vector<double> r{s.rho};
    if (s.adaptive_rho) {
        double v = s.rho / s.adaptive_rho_tolerance;
        while (v >= s.rho_min) {
            r.push_back(v);
            v = v / s.adaptive_rho_tolerance;
        }
        v = s.rho * s.adaptive_rho_tolerance;
        while (v <= s.rho_max) {
            r.push_back(v);
            v = v * s.adaptive_rho_tolerance;
        }
        std::sort(r.begin(), r.end());
    }
    return r;
}

int argmin_abs(const std::vector<double>& r, double v) {   // np.argmin(np.abs(rhos - v)): first minimum
    int best = 0;
    double bd = std::fabs(r[0] - v);
    for (size_t i = 1; i < r.size(); ++i) {
        double d = std::fabs(r[i] - v);
        if (d < bd) {
            bd = d;
            best = (int)i;
        }
    }
    return best;
}

void free_ws(rqp_handle* h) {
    void** ptrs[] = {&h->Ht, &h->A, &h->At, &h->K, &h->g, &h->l, &h->u, &h->c, (void**)&h->G,
                     (void**)&h->x, (void**)&h->z, (void**)&h->lam, (void**)&h->rho_ind, (void**)&h->rhos_d,
                     (void**)&h->fscratch, (void**)&h->Apack, (void**)&h->Kpack, (void**)&h->Hpack, (void**)&h->W1img, (void**)&h->queue};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    h->is_setup = false;
    h->resident = false;
    h->res_kind = 0;
    h->use_mfma = false;
    h->use_wave = false;
    h->kernel_name = "generic";
}

hipError_t launch_solve(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    if (h->use_mfma && a.mode == 0) return rqp_launch_solve_mfma(h, a, s);
    if (h->use_wave && a.mode == 0) return rqp_launch_solve_wave(h, a, s);
    if (h->resident && h->res_kind == 2) return rqp_launch_solve_res2(h, a, s);
    if (h->resident) return rqp_launch_solve_resident(h, a, s);
    return rqp_launch_solve_generic(h, a, s);
}

SolveArgs make_solve_args(const rqp_handle* h) {
    SolveArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = h->n; a.m = h->m; a.ldn = h->ldn; a.ldm = h->ldm; a.nrho = h->nrho; a.B = h->B;
    a.max_iter = h->st.max_iter;
    a.check_interval = h->st.check_interval;
    a.warm_starting = h->st.warm_starting;
    a.rho_ind0 = h->rho_ind0;
    a.sigma = h->st.sigma;
    a.tol = h->st.adaptive_rho_tolerance;
    a.rho_min = h->st.rho_min;
    a.rho_max = h->st.rho_max;
    a.thr_p = h->st.eps_abs * std::sqrt((double)h->m);   // reluqpth.py:233
    a.thr_d = h->st.eps_abs * std::sqrt((double)h->n);
    a.Ht = h->Ht; a.A = h->A; a.At = h->At; a.K = h->K;
    const bool sh = h->dims.shared_mats != 0;
    a.sH = sh ? 0 : (size_t)h->n * h->ldn;
    a.sA = sh ? 0 : (size_t)h->m * h->ldn;
    a.sAt = sh ? 0 : (size_t)h->n * h->ldm;
    a.sK = sh ? 0 : (size_t)h->nrho * h->n * h->ldn;
    a.g = h->g; a.l = h->l; a.u = h->u; a.c = h->c;
    a.rhos = h->rhos_d;
    a.x = h->x; a.z = h->z; a.lam = h->lam; a.rho_ind = h->rho_ind;
    return a;
}

}  // namespace

extern "C" {

int rqp_default_settings(rqp_settings* s) {
    if (!s) return RQP_ERR_ARG;
    s->rho = 0.1;            // classes.py:36-46
    s->rho_min = 1e-6;
    s->rho_max = 1e6;
    s->sigma = 1e-6;
    s->adaptive_rho_tolerance = 5;
    s->eps_abs = 1e-3;
    s->eq_tol = 1e-6;
    s->adaptive_rho = 1;
    s->max_iter = 4000;
    s->check_interval = 25;
    s->warm_starting = 1;
    return RQP_OK;
}

int rqp_create(rqp_handle** out, const rqp_dims* dims, const rqp_settings* settings, int device) {
    if (!out || !dims || !settings) return RQP_ERR_ARG;
    *out = nullptr;
    if (dims->n < 1 || dims->m < 1 || dims->batch < 1) return RQP_ERR_ARG;
    if (dims->dtype != RQP_F32 && dims->dtype != RQP_F64) return RQP_ERR_ARG;
    if (!settings_valid(*settings)) return RQP_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RQP_ERR_HIP;
    rqp_handle* h = new rqp_handle();
    h->dims = *dims;
    h->st = *settings;
    h->device = device;
    h->n = dims->n; h->m = dims->m; h->B = dims->batch;
    h->nmat = dims->shared_mats ? 1 : dims->batch;
    h->esz = dims->dtype == RQP_F32 ? 4 : 8;
    h->ldn = rqp_round_up(h->n, 4);
    h->ldm = rqp_round_up(h->m, 4);
    h->rhos = build_rhos(*settings);
    h->nrho = (int)h->rhos.size();
    h->rho_ind0 = argmin_abs(h->rhos, settings->rho);     // reluqpth.py:153
    *out = h;
    return RQP_OK;
}

int rqp_destroy(rqp_handle* h) {
    if (!h) return RQP_ERR_ARG;
    (void)hipSetDevice(h->device);
    free_ws(h);
    delete h;
    return RQP_OK;
}

int rqp_setup(rqp_handle* h, const void* H, const void* g, const void* A, const void* l, const void* u,
              void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!H || !g || !A || !l || !u) return fail_arg(h, "rqp_setup: null input pointer");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->device));
    free_ws(h);
    const size_t n = h->n, m = h->m, B = h->B, nm = h->nmat, e = h->esz;
    HIP_TRY(h, hipMalloc(&h->Ht, nm * n * h->ldn * e));
    HIP_TRY(h, hipMalloc(&h->A, nm * m * h->ldn * e));
    HIP_TRY(h, hipMalloc(&h->At, nm * n * h->ldm * e));
    HIP_TRY(h, hipMalloc(&h->K, nm * h->nrho * n * h->ldn * e));
    HIP_TRY(h, hipMalloc(&h->g, B * n * e));
    HIP_TRY(h, hipMalloc(&h->l, B * m * e));
    HIP_TRY(h, hipMalloc(&h->u, B * m * e));
    HIP_TRY(h, hipMalloc(&h->c, B * m * e));
    HIP_TRY(h, hipMalloc((void**)&h->G, nm * n * n * sizeof(double)));
    HIP_TRY(h, hipMalloc((void**)&h->x, B * n * sizeof(double)));
    HIP_TRY(h, hipMalloc((void**)&h->z, B * m * sizeof(double)));
    HIP_TRY(h, hipMalloc((void**)&h->lam, B * m * sizeof(double)));
    HIP_TRY(h, hipMalloc((void**)&h->rho_ind, B * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc((void**)&h->rhos_d, h->nrho * sizeof(double)));
    HIP_TRY(h, hipMemcpyAsync(h->rhos_d, h->rhos.data(), h->nrho * sizeof(double), hipMemcpyHostToDevice, s));
    const size_t lds_need = (n * n + 2 * n) * sizeof(double);
    if (lds_need > 160 * 1024 - 512) {   // factor scratch in global memory
        h->fscratch_elems = nm * h->nrho * n * n;
        HIP_TRY(h, hipMalloc((void**)&h->fscratch, h->fscratch_elems * sizeof(double)));
    }
    SetupArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = h->n; a.m = h->m; a.ldn = h->ldn; a.ldm = h->ldm; a.nrho = h->nrho; a.B = h->B; a.nmat = h->nmat;
    a.sigma = h->st.sigma;
    a.eq_tol = h->st.eq_tol;
    a.H_in = H; a.A_in = A; a.g_in = g; a.l_in = l; a.u_in = u;
    a.Ht = h->Ht; a.A = h->A; a.At = h->At; a.K = h->K; a.g = h->g; a.l = h->l; a.u = h->u; a.c = h->c;
    a.G = h->G;
    a.rhos = h->rhos_d;
    a.fscratch = h->fscratch;
    HIP_TRY(h, rqp_launch_pack(h, a, s));
    HIP_TRY(h, rqp_launch_gram(h, a, s));
    HIP_TRY(h, rqp_launch_factor(h, a, s));
    const char* force = getenv("RQP_FORCE_GENERIC");
    const char* kind = getenv("RQP_RESIDENT");            // "1": first resident layout (A/B runs); default: layout 2
    const bool want_v1 = kind && kind[0] == '1';
    // small problems: one wavefront per instance (solve() only; iterate / residuals then run on the streaming kernel, so
    // the resident images are not built at all)
    const char* wv = getenv("RQP_WAVE");                  // "0": never
    const bool want_wave = rqp_wave_fits(h) && !(force && force[0] == '1') && !(wv && wv[0] == '0') && !want_v1;
    if (!want_wave && !(force && force[0] == '1') && (want_v1 ? rqp_resident_fits(h) : rqp_res2_fits(h))) {
        size_t ae, ke, he;
        if (want_v1)
            rqp_resident_pack_elems(h, &ae, &ke, &he);
        else
            rqp_res2_pack_elems(h, &ae, &ke, &he);
        HIP_TRY(h, hipMalloc((void**)&h->Apack, ae * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->Kpack, ke * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->Hpack, he * sizeof(float)));
        HIP_TRY(h, want_v1 ? rqp_launch_pack_resident(h, s) : rqp_launch_pack_res2(h, s));
        h->resident = true;
        h->res_kind = want_v1 ? 1 : 2;
        h->kernel_name = want_v1 ? "resident" : "resident2";
    }
    // shared-(H,A) batches large enough to fill the chip with 16-instance tiles go to the MFMA kernel
    // (crossover measured on the condensed-MPC shape: 1024 -> resident 1.7x faster, 2048 -> even cold / MFMA 1.3x closed loop,
    //  3072 -> MFMA 1.35x / 1.8x)
    const char* mf = getenv("RQP_MFMA");                  // "0": never, "1": whenever it fits
    // default: from ~2k instances (below, the per-instance kernels still win) and only for problems beyond the small /
    // mid per-instance tiles -- the MFMA tile costs the same whatever the problem size (minus skipped zero groups), and
    // on n=30, m=60 the one-wavefront kernel is 3-4x faster, on n=20, m=80 the mid resident tile is on par (measured)
    const bool mfma_pays = h->B >= 2048 && (h->n > 56 || h->m > 128);
    h->use_mfma = rqp_mfma_fits(h) && !(force && force[0] == '1') && !(mf && mf[0] == '0') &&
                  (mfma_pays || (mf && mf[0] == '1'));
    if (h->use_mfma) {
        HIP_TRY(h, hipMalloc((void**)&h->W1img, rqp_mfma_img_elems(h) * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->queue, sizeof(int)));
        HIP_TRY(h, rqp_launch_pack_mfma(h, s));
        h->kernel_name = "mfma";
    }
    h->use_wave = want_wave && !h->use_mfma;
    if (h->use_wave) h->kernel_name = "wave";
    h->is_setup = true;
    return rqp_clear_primal_dual(h, stream);    // zero state, rho_ind0 (reluqpth.py:148-153)
}

int rqp_update(rqp_handle* h, const void* g, const void* l, const void* u, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    if (!g && !l && !u) return RQP_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, rqp_launch_vec_update(h, g, l, u, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_update_affine(rqp_handle* h, const void* p, int32_t np, const void* Gg, const void* Glu, const void* l0,
                      const void* u0, void* stream) {
    if (!h || !p || !Gg || !Glu || !l0 || !u0 || np <= 0 || np > 64) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, rqp_launch_affine_update(h, p, np, Gg, Glu, l0, u0, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_update_settings(rqp_handle* h, const rqp_settings* s) {
    if (!h || !s) return RQP_ERR_ARG;
    const rqp_settings& o = h->st;
    if (s->rho != o.rho || s->rho_min != o.rho_min || s->rho_max != o.rho_max || s->sigma != o.sigma ||
        s->adaptive_rho != o.adaptive_rho || s->adaptive_rho_tolerance != o.adaptive_rho_tolerance ||
        s->eq_tol != o.eq_tol)
        return fail_arg(h, "rqp_update_settings: only max_iter, eps_abs, check_interval, warm_starting may change");
    if (s->max_iter < 0 || s->check_interval < 1 || s->eps_abs < 0) return fail_arg(h, "rqp_update_settings: bad value");
    h->st.max_iter = s->max_iter;
    h->st.eps_abs = s->eps_abs;
    h->st.check_interval = s->check_interval;
    h->st.warm_starting = s->warm_starting;
    return RQP_OK;
}

int rqp_warm_start(rqp_handle* h, const void* x, const void* z, const void* lam, int has_rho, double rho,
                   void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    const int ri = has_rho ? argmin_abs(h->rhos, rho) : 0;          // reluqpth.py:273-274
    HIP_TRY(h, rqp_launch_state_set(h, x, z, lam, has_rho, ri, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_clear_primal_dual(rqp_handle* h, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemsetAsync(h->x, 0, (size_t)h->B * h->n * sizeof(double), s));
    HIP_TRY(h, hipMemsetAsync(h->z, 0, (size_t)h->B * h->m * sizeof(double), s));
    HIP_TRY(h, hipMemsetAsync(h->lam, 0, (size_t)h->B * h->m * sizeof(double), s));
    HIP_TRY(h, rqp_launch_state_set(h, nullptr, nullptr, nullptr, 1, h->rho_ind0, s));
    return RQP_OK;
}

int rqp_solve(rqp_handle* h, void* x, void* z, void* lam, const rqp_info* info, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    SolveArgs a = make_solve_args(h);
    a.mode = 0;
    a.out_x = x; a.out_z = z; a.out_lam = lam;
    if (info) a.info = *info;
    if (a.info.trace && a.info.trace_cap < 1) return fail_arg(h, "rqp_solve: trace without capacity");
    HIP_TRY(h, launch_solve(h, a, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_iterate(rqp_handle* h, int32_t k, void* stream) {
    if (!h || k < 0) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    SolveArgs a = make_solve_args(h);
    a.mode = 1;
    a.max_iter = k;
    HIP_TRY(h, launch_solve(h, a, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_compute_residuals(rqp_handle* h, double rho_in, double* pri, double* dua, double* rho_out, double* obj,
                          void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    SolveArgs a = make_solve_args(h);
    a.mode = 2;
    a.rho_in = rho_in;
    a.r_pri = pri; a.r_dua = dua; a.r_rho = rho_out; a.r_obj = obj;
    HIP_TRY(h, launch_solve(h, a, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_get_state(rqp_handle* h, void* x, void* z, void* lam, int32_t* rho_ind, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, rqp_launch_state_get(h, x, z, lam, rho_ind, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_get_rhos(const rqp_handle* h, double* rhos, int32_t cap, int32_t* count) {
    if (!h) return RQP_ERR_ARG;
    if (count) *count = h->nrho;
    if (rhos) {
        if (cap < h->nrho) return RQP_ERR_ARG;
        for (int i = 0; i < h->nrho; ++i) rhos[i] = h->rhos[i];
    }
    return RQP_OK;
}

int rqp_get_K(rqp_handle* h, int32_t b, int32_t j, void* out, void* stream) {
    if (!h || !out) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    if (b < 0 || b >= h->B || j < 0 || j >= h->nrho) return fail_arg(h, "rqp_get_K: index out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, rqp_launch_get_K(h, b, j, out, (hipStream_t)stream));
    return RQP_OK;
}

const char* rqp_kernel_name(const rqp_handle* h) { return h ? h->kernel_name : ""; }

const char* rqp_strerror(int err) {
    switch (err) {
        case RQP_OK: return "ok";
        case RQP_ERR_ARG: return "invalid argument";
        case RQP_ERR_STATE: return "invalid call order (setup first)";
        case RQP_ERR_HIP: return "HIP runtime error";
        case RQP_ERR_OOM: return "out of device memory";
        case RQP_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown error";
    }
}

const char* rqp_last_error(const rqp_handle* h) { return h ? h->err.c_str() : ""; }

const char* rqp_version(void) { return RQP_VERSION; }

}  // extern "C"
