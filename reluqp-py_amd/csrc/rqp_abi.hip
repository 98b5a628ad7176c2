// rqp_abi.hip -- the extern "C" boundary declared in include/rqp_abi.h.
// Host orchestration only: argument validation, workspace ownership, kernel dispatch.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#include <map>
#include <mutex>
#include <utility>

#include "rqp_common.h"

hipError_t rqp_raise_lds_limit(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> limit;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = limit[std::make_pair(fn, dev)];
    if (bytes <= cur) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}

#define RQP_VERSION "rqp-hip 0.3 gfx950"

namespace {

int fail_hip(rqp_handle* h, hipError_t e, const char* what) {
    if (h) h->err = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorOutOfMemory) ? RQP_ERR_OOM : RQP_ERR_HIP;
}
int fail_arg(rqp_handle* h, const char* what) {
    if (h) h->err = what;
    return RQP_ERR_ARG;
}
int fail_unsupported(rqp_handle* h, const char* what) {
    if (h) h->err = what;
    return RQP_ERR_UNSUPPORTED;
}

#define HIP_TRY(h, call)                                      \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return fail_hip(h, e__, #call); \
    } while (0)

bool settings_valid(const rqp_settings& s) {
    return s.rho > 0 && s.rho_min > 0 && s.rho_max >= s.rho_min && s.sigma >= 0 && s.adaptive_rho_tolerance > 1 &&
           s.eps_abs >= 0 && s.max_iter >= 0 && s.check_interval >= 1 && s.eps_rel >= 0 && s.eps_prim_inf >= 0 &&
           s.eps_dual_inf >= 0 && s.scaling >= 0;
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
                     (void**)&h->fscratch, (void**)&h->Apack, (void**)&h->Kpack, (void**)&h->Hpack, (void**)&h->Kscale, (void**)&h->W1img, (void**)&h->queue,
                     (void**)&h->flag_d, (void**)&h->order_d, (void**)&h->last_iter_d, (void**)&h->cont_iter_d, (void**)&h->cont_rho_d,
                     (void**)&h->Dsc, (void**)&h->Esc, (void**)&h->csc, (void**)&h->wbase_d, (void**)&h->ax_d, (void**)&h->cstat_d, (void**)&h->key_d,
                     (void**)&h->ncont_d};
    // hipFree is one of the calls that invalidate a stream capture in progress (global / thread-local capture modes).  A handle
    // may be destroyed while this thread captures something else (a Python finaliser, an explicit `del`): free under the
    // relaxed mode, which exists for exactly this.
    hipStreamCaptureMode cmode = hipStreamCaptureModeRelaxed;
    const bool swapped = hipThreadExchangeStreamCaptureMode(&cmode) == hipSuccess;
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (h->ncont_h) (void)hipHostFree(h->ncont_h);
    if (swapped) (void)hipThreadExchangeStreamCaptureMode(&cmode);
    h->ncont_h = nullptr;
    h->windowed = false;
    h->borrow_A = false;
    h->kpack_direct = false;
    h->is_setup = false;
    h->order_valid = false;
    h->resident = false;
    h->resident64 = false;
    h->use_mfma = false;
    h->use_wave = false;
    h->kernel_name = "generic";
}

hipError_t launch_solve(const rqp_handle* h, const SolveArgs& a, hipStream_t s) {
    if (h->use_mfma && a.mode == 0)
        return h->mfmad ? rqp_launch_solve_mfmad(h, a, s)
                        : (h->mfmal ? rqp_launch_solve_mfmal(h, a, s) : (h->mfma16 ? rqp_launch_solve_mfma16(h, a, s) : rqp_launch_solve_mfma(h, a, s)));
    if (h->use_wave && a.mode == 0) return rqp_launch_solve_wave(h, a, s);
    if (h->resident) return rqp_launch_solve_res2(h, a, s);
    if (h->resident64) return rqp_launch_solve_res64(h, a, s);
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
    a.eps_rel = h->st.eps_rel;
    a.check_infeas = h->st.check_infeasibility;
    a.eps_pinf = h->st.eps_prim_inf;
    a.eps_dinf = h->st.eps_dual_inf;
    a.Ht = h->Ht; a.A = h->A; a.At = h->At; a.K = h->K;
    const bool sh = h->dims.shared_mats != 0;
    a.sH = sh ? 0 : (size_t)h->n * h->ldn;
    a.sA = sh ? 0 : (size_t)h->m * h->ldn;
    a.sAt = sh ? 0 : (size_t)h->n * h->ldm;
    a.sK = sh ? 0 : (size_t)h->kwin * h->n * h->ldn;
    a.kwin = h->kwin;
    if (h->windowed) {
        a.wbase = h->wbase_d;
        a.ax = h->ax_d;
        a.cstat = h->cstat_d;
        a.ncont = h->ncont_d;
        a.cont_iter = h->cont_iter_d;
        a.cont_rho = h->cont_rho_d;
    }
    a.g = h->g; a.l = h->l; a.u = h->u; a.c = h->c;
    a.rhos = h->rhos_d;
    if (h->st.scaling > 0) { a.scD = h->Dsc; a.scE = h->Esc; a.scC = h->csc; }
    a.x = h->x; a.z = h->z; a.lam = h->lam; a.rho_ind = h->rho_ind;
    return a;
}


SetupArgs make_setup_args(const rqp_handle* h, const void* H, const void* g, const void* A, const void* l, const void* u) {
    SetupArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = h->n; a.m = h->m; a.ldn = h->ldn; a.ldm = h->ldm; a.nrho = h->nrho; a.B = h->B; a.nmat = h->nmat;
    a.sigma = h->st.sigma;
    a.eq_tol = h->st.eq_tol;
    a.H_in = H; a.A_in = A; a.g_in = g; a.l_in = l; a.u_in = u;
    a.Ht = h->Ht; a.A = h->borrow_A ? const_cast<void*>(A) : h->A; a.At = h->At; a.K = h->K; a.g = h->g; a.l = h->l; a.u = h->u; a.c = h->c;
    a.G = h->G;
    a.rhos = h->rhos_d;
    a.fscratch = h->fscratch;
    a.kwin = h->kwin;
    a.wbase = h->windowed ? h->wbase_d : nullptr;
    return a;
}

// Kernel selection (rqp_dims.kernel; AUTO = measured crossovers).  Pure function of the handle: no environment.
int select_kernels(rqp_handle* h) {
    int req = h->dims.kernel;
    if (h->dims.tile_dtype == RQP_TILE_F16) {     // the fp16 K tile lives in the register-resident kernel; the MFMA kernel takes
                                                  // the same rounded K into its float32 operand image (k_pack_mfma)
        const bool mfma_ok = rqp_mfma_fits(h) && rqp_res2_fits(h);
        if (req == RQP_KERNEL_AUTO && rqp_res2_fits(h) && !(mfma_ok && h->B >= 2048 && (h->n > 56 || h->m > 128))) req = RQP_KERNEL_RESIDENT;
        if (req != RQP_KERNEL_RESIDENT && !((req == RQP_KERNEL_MFMA || req == RQP_KERNEL_AUTO) && mfma_ok))
            return fail_unsupported(h, "tile_dtype = f16 needs the resident kernel (float32, n <= 104, m <= 320) or the MFMA kernel");
    }
    h->resident = h->resident64 = h->use_wave = h->use_mfma = h->mfma16 = h->mfmal = h->mfmad = false;
    h->kernel_name = "generic";
    if (h->dims.tile_dtype == RQP_TILE_BF16) {     // the bf16-plane tile exists in the MFMA kernel only: an explicit request for it
        if (!rqp_mfma_fits(h) || (req != RQP_KERNEL_AUTO && req != RQP_KERNEL_MFMA))
            return fail_unsupported(h, "tile_dtype = bf16 needs the MFMA kernel: float32, shared (H, A), n <= 80, m <= 320");
        req = RQP_KERNEL_MFMA;
        h->mfma16 = true;
    }
    switch (req) {
        case RQP_KERNEL_GENERIC:
            break;
        case RQP_KERNEL_RESIDENT:
            if (rqp_res2_fits(h)) h->resident = true;
            else if (rqp_res64_fits(h)) h->resident64 = true;
            else return fail_unsupported(h, "kernel=resident: needs n <= 104, m <= 320");
            break;
        case RQP_KERNEL_WAVE:
            if (!rqp_wave_fits(h)) return fail_unsupported(h, "kernel=wave: needs n <= 32, m <= 64 (float32 also n <= 32, m <= 128 and 56 < n <= 64, m <= 128)");
            h->use_wave = true;
            break;
        case RQP_KERNEL_MFMA:
            if (rqp_mfma_fits(h)) h->use_mfma = true;                         // operands resident in registers
            else if (rqp_mfmal_fits(h) && !h->mfma16) h->use_mfma = h->mfmal = true;   // operands streamed from L2 (rqp_mfmal.hip)
            else if (rqp_mfmad_fits(h) && !h->mfma16) h->use_mfma = h->mfmad = true;   // float64 MFMA, streamed operands (rqp_mfmad.hip)
            else return fail_unsupported(h, "kernel=mfma: needs shared (H, A) and float32 with n <= 320, m <= 640 (bf16 tile: n <= 80, m <= 320) "
                                            "or float64 with n <= 160, m <= 320");
            break;
        default: {
            // shared-(H,A) batches large enough to fill the chip with 16-instance tiles go to the MFMA kernel: from ~2k
            // instances (below, the per-instance kernels still win; crossover measured on the condensed-MPC shape: 1024 ->
            // resident 1.7x faster, 2048 -> even cold / MFMA 1.3x closed loop, 3072 -> MFMA 1.35x / 1.8x) and only for
            // problems beyond the small / mid per-instance tiles -- the MFMA tile costs the same whatever the problem size
            // (minus skipped zero groups); on n=30, m=60 the one-wavefront kernel is 3-4x faster, on n=20, m=80 the mid
            // resident tile is on par (measured)
            const bool mfma_pays = h->B >= 2048 && (h->n > 56 || h->m > 128);
            // check_infeasibility: only the streaming kernel tests the certificates at every check (an infeasible instance
            // leaves at the first check where one holds, like the oracle; the other kernels would run it to max_iter first)
            if (h->st.check_infeasibility)
                break;
            if (rqp_mfma_fits(h) && mfma_pays)
                h->use_mfma = true;
            else if (rqp_mfmal_fits(h) && !rqp_res2_fits(h) && !rqp_wave_fits(h))
                h->use_mfma = h->mfmal = true;   // beyond every resident tile (the sparse linear-MPC form): 16-instance MFMA tiles with
                                                 // streamed operands instead of the streaming kernel's 2 MB of matrices per instance-iteration
                                                 // (any batch: ONE instance solves in 1.0 ms against 2.8 ms, tools/mfmal_check.py 1)
            else if (rqp_mfmad_fits(h) && ((mfma_pays && h->B >= 3072) || (!rqp_res64_fits(h) && !rqp_wave_fits(h))))
                h->use_mfma = h->mfmad = true;   // float64 shared batches: 16-instance tiles on v_mfma_f64_16x16x4_f64 from ~3k instances
                                                 // (measured on the condensed-MPC shape: 4096 -> 1.5x the float64 resident kernel; at
                                                 // 2048 one resident instance per CU still wins), or whenever no resident kernel fits
            else if (rqp_wave_fits(h))       // small problems: one wavefront per instance
                h->use_wave = true;
            else if (rqp_res2_fits(h))
                h->resident = true;
            else if (rqp_res64_fits(h))      // float64 (the reference's default precision) at the headline sizes
                h->resident64 = true;
        }
    }
    // iterate / residuals modes of an MFMA or wave handle run on the resident tile when one fits, else on the streaming kernel
    if (h->use_mfma && rqp_res2_fits(h)) h->resident = true;
    if (h->dims.tile_dtype == RQP_TILE_F16 && !h->resident) return fail_unsupported(h, "tile_dtype = f16 needs the resident kernel");
    // RQP_FLAG_LOW_MEMORY: the float32 resident kernel reads K from the row-major table (no effect on the other kernels, which
    // have no packed copy of it; the fp16 tile IS the smaller copy)
    h->k_direct = (h->dims.flags & RQP_FLAG_LOW_MEMORY) && h->resident && h->dims.tile_dtype != RQP_TILE_F16;
    // rho-ladder window (rqp_common.h): batches of per-instance matrices whose solve kernel runs the exit-and-continue protocol
    // (resident float32 / float64 tiles, one-wavefront kernel, streaming kernel).  Not with check_infeasibility (its certificate pass reads K at the final index)
    // and not on request (RQP_FLAG_FULL_LADDER: rqp_solve then never synchronises the host, e.g. for graph capture).
    h->kwin = h->nrho;
    h->windowed = !h->dims.shared_mats && h->nmat >= 32 && h->nrho > RQP_WINDOW && !(h->dims.flags & RQP_FLAG_FULL_LADDER) &&
                  !h->st.check_infeasibility && !h->use_mfma;
    if (h->windowed) h->kwin = RQP_WINDOW;
    h->borrow_A = h->windowed && h->resident && h->st.scaling <= 0 && h->ldn == h->n;     // (rqp_common.h)
    h->kpack_direct = h->windowed && h->resident && !h->k_direct && h->dims.tile_dtype != RQP_TILE_F16;
    if (h->use_mfma) h->kernel_name = h->mfmad ? "mfmad" : (h->mfmal ? "mfmal" : (h->mfma16 ? "mfma16" : "mfma"));
    else if (h->use_wave) h->kernel_name = "wave";
    else if (h->resident) h->kernel_name = "resident2";
    else if (h->resident64) h->kernel_name = "resident64";
    return RQP_OK;
}

// kpack_direct handles: the factor kernel's output is the register image of k_admm_res2
void set_kp_image(const rqp_handle* h, SetupArgs& f) {
    if (!h->kpack_direct) return;
    f.kp_img = h->Kpack;
    rqp_res2_kp_layout(h, &f.kp_cw, &f.kp_kr, &f.kp_kc);
}

// pack (QP.__init__ casts) -> G = A'cA -> K_j ladder -> kernel images, for new H and/or A (NULL: keep the packed copy).
// Shared by rqp_setup and rqp_update_mats.
int build_matrices(rqp_handle* h, const SetupArgs& a, hipStream_t s) {
    HIP_TRY(h, rqp_launch_pack_mats(h, a, s));
    if (h->st.scaling > 0) HIP_TRY(h, rqp_launch_ruiz(h, s));       // Ht, A, At scaled in place; D, E, c kept for the boundary
    if (a.A) HIP_TRY(h, rqp_launch_gram(h, a, s));                  // (NULL: a handle without a copy of A that keeps its A -- G = A'cA stands)
    if (h->resident && !h->Apack) {                                 // (before the factor launch: kpack_direct writes Kpack there)
        size_t ae, ke, he;
        rqp_res2_pack_elems(h, &ae, &ke, &he);
        HIP_TRY(h, hipMalloc((void**)&h->Apack, ae * sizeof(float)));
        if (ke) HIP_TRY(h, hipMalloc((void**)&h->Kpack, ke * sizeof(float)));   // (none with RQP_FLAG_LOW_MEMORY)
        HIP_TRY(h, hipMalloc((void**)&h->Hpack, he * sizeof(float)));
        if (h->dims.tile_dtype == RQP_TILE_F16) HIP_TRY(h, hipMalloc((void**)&h->Kscale, (size_t)h->nmat * h->nrho * sizeof(float)));
        HIP_TRY(h, rqp_prepare_res2(h));
    }
    {
        SetupArgs f = a;
        set_kp_image(h, f);
        HIP_TRY(h, rqp_launch_factor(h, f, s));
    }
    if (h->resident) HIP_TRY(h, rqp_launch_pack_res2(h, a.A, nullptr, s));
    if (h->resident64) HIP_TRY(h, rqp_prepare_res64(h));
    if (h->use_mfma) {
        if (!h->W1img) {
            const size_t elems = h->mfmad ? rqp_mfmad_img_elems(h) : (h->mfmal ? rqp_mfmal_img_elems(h) : (h->mfma16 ? rqp_mfma16_img_elems(h) : rqp_mfma_img_elems(h)));
            HIP_TRY(h, hipMalloc((void**)&h->W1img, elems * sizeof(float)));
            HIP_TRY(h, hipMalloc((void**)&h->queue, sizeof(int)));
            HIP_TRY(h, h->mfmad ? rqp_prepare_mfmad(h) : (h->mfmal ? rqp_prepare_mfmal(h) : (h->mfma16 ? rqp_prepare_mfma16(h) : rqp_prepare_mfma(h))));
        }
        HIP_TRY(h, h->mfmad ? rqp_launch_pack_mfmad(h, s) : (h->mfmal ? rqp_launch_pack_mfmal(h, s) : (h->mfma16 ? rqp_launch_pack_mfma16(h, s) : rqp_launch_pack_mfma(h, s))));
    }
    return RQP_OK;
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
    s->eps_rel = 0.0;        // extensions: off = the reference's behaviour
    s->eps_prim_inf = 1e-4;
    s->eps_dual_inf = 1e-4;
    s->scaling = 0;
    s->check_infeasibility = 0;
    return RQP_OK;
}

int rqp_create(rqp_handle** out, const rqp_dims* dims, const rqp_settings* settings, int device) {
    if (!out || !dims || !settings) return RQP_ERR_ARG;
    *out = nullptr;
    if (dims->n < 1 || dims->m < 1 || dims->batch < 1) return RQP_ERR_ARG;
    if (dims->dtype != RQP_F32 && dims->dtype != RQP_F64) return RQP_ERR_ARG;
    if (dims->kernel < RQP_KERNEL_AUTO || dims->kernel > RQP_KERNEL_MFMA) return RQP_ERR_ARG;
    if (dims->tile_dtype != RQP_TILE_SAME &&
        !((dims->tile_dtype == RQP_TILE_F16 || dims->tile_dtype == RQP_TILE_BF16) && dims->dtype == RQP_F32)) return RQP_ERR_ARG;
    if (!settings_valid(*settings)) return RQP_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RQP_ERR_HIP;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return RQP_ERR_HIP;
    rqp_handle* h = new rqp_handle();
    h->ncu = pr.multiProcessorCount;
    {   // diagnostics only (occupancy print / s_memtime build); read once, never consulted for kernel selection
        const char* d1 = getenv("RQP_DEBUG");
        const char* d2 = getenv("RQP_DIAG");
        h->debug = ((d1 && d1[0] == '1') ? 1 : 0) | ((d2 && d2[0] == '1') ? 2 : 0);
    }
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
    const auto t_begin = std::chrono::steady_clock::now();
    free_ws(h);
    {
        const int rc = select_kernels(h);
        if (rc != RQP_OK) return rc;
    }
    {
        // the streaming kernel backs every handle (iterate / residuals modes, sizes beyond the tiles): its vectors must fit LDS
        const size_t lds = rqp_generic_lds_bytes(h);
        if (lds > 160 * 1024) {
            char buf[256];
            snprintf(buf, sizeof(buf), "rqp_setup: n=%d, m=%d needs %zu B of LDS for the vector state of the streaming kernel "
                     "(limit 163840 B per workgroup)", h->n, h->m, lds);
            h->err = buf;
            return RQP_ERR_UNSUPPORTED;
        }
        HIP_TRY(h, rqp_prepare_generic(h));
    }
    const size_t n = h->n, m = h->m, B = h->B, nm = h->nmat, e = h->esz;
    HIP_TRY(h, hipMalloc(&h->Ht, nm * n * h->ldn * e));
    if (!h->borrow_A) HIP_TRY(h, hipMalloc(&h->A, nm * m * h->ldn * e));     // (rqp_common.h: borrow_A)
    // A' (the streaming kernel's A dx operand and the wavefront kernel's column role): not on a windowed resident handle, whose
    // solve / iterate / residuals all run on k_admm_res2 / k_admm_res64 (neither reads A') and which refuses the certificate pass
    // (float32: 0.5 GB and 0.4 ms at B = 4096; float64: 1 GB and 1 ms)
    if (!(h->windowed && (h->resident || h->resident64))) HIP_TRY(h, hipMalloc(&h->At, nm * n * h->ldm * e));
    if (!h->kpack_direct) {   // (+ a zeroed tail: the low-memory K load of the resident kernel reads up to one vector past a row's end)
        const size_t kb = nm * h->kwin * n * h->ldn * e;
        HIP_TRY(h, hipMalloc(&h->K, kb + 256));
        HIP_TRY(h, hipMemsetAsync((char*)h->K + kb, 0, 256, s));
    }
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
        h->fscratch_elems = nm * h->kwin * n * n;
        HIP_TRY(h, hipMalloc((void**)&h->fscratch, h->fscratch_elems * sizeof(double)));
    }
    HIP_TRY(h, hipMalloc((void**)&h->flag_d, sizeof(int32_t)));
    if ((h->mfmal || h->mfmad) && h->B > 16)       // slot order of the streamed-operand MFMA kernels (grouped by starting rho index)
        HIP_TRY(h, hipMalloc((void**)&h->order_d, (size_t)h->B * sizeof(int32_t)));
    if (h->mfmal && h->B > 16) {                   // regrouped cold solve (rqp_mfmal.hip): exact state of the instances between its two launches
        HIP_TRY(h, hipMalloc((void**)&h->ax_d, B * m * sizeof(double)));
        HIP_TRY(h, hipMalloc((void**)&h->cont_rho_d, B * sizeof(double)));
        HIP_TRY(h, hipMalloc((void**)&h->key_d, B * sizeof(int32_t)));
    }
    if (!h->use_mfma && h->B >= (h->resident64 ? 2 : 4) * h->ncu) {      // dispatch order (see rqp_common.h): batches that outlast one wave of workgroups
        HIP_TRY(h, hipMalloc((void**)&h->order_d, (size_t)h->B * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc((void**)&h->last_iter_d, (size_t)h->B * sizeof(int32_t)));
    }
    if (h->windowed) {
        const int w0 = std::min(std::max(h->rho_ind0 - 1, 0), h->nrho - h->kwin);
        HIP_TRY(h, hipMalloc((void**)&h->wbase_d, nm * sizeof(int32_t)));
        HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)h->wbase_d, w0, nm, s));
        HIP_TRY(h, hipMalloc((void**)&h->ax_d, B * m * sizeof(double)));
        HIP_TRY(h, hipMalloc((void**)&h->cstat_d, B * sizeof(int32_t)));
        HIP_TRY(h, hipMemsetAsync(h->cstat_d, 0, B * sizeof(int32_t), s));
        HIP_TRY(h, hipMalloc((void**)&h->ncont_d, sizeof(int32_t)));
        HIP_TRY(h, hipHostMalloc((void**)&h->ncont_h, sizeof(int32_t), hipHostMallocDefault));
        HIP_TRY(h, hipMalloc((void**)&h->cont_iter_d, B * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc((void**)&h->cont_rho_d, B * sizeof(double)));
    }
    h->handoff_cols = 0;
    if (h->use_mfma && h->resident && (h->B + 15) / 16 <= h->ncu) {   // straggler hand-off (SolveArgs): one tile per CU at most
        HIP_TRY(h, hipMalloc((void**)&h->cont_iter_d, (size_t)h->B * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc((void**)&h->cont_rho_d, (size_t)h->B * sizeof(double)));
        h->handoff_cols = 6;    // measured on the config-3 batch: 4.7 M QP/s without, 5.0 M at 2-4, 6.2 M at 6-10, 5.5 M at 12 (tools/)
        if (const char* ho = getenv("RQP_TUNE_HANDOFF")) h->handoff_cols = atoi(ho);   // (tuning aid of tools/mfma16_check.py, not a dispatch input)
    }
    SetupArgs a = make_setup_args(h, H, g, A, l, u);
    HIP_TRY(h, rqp_launch_pack_vecs(h, a, s));
    if (h->dims.shared_mats && h->B > 1) {
        // K is built from ONE equality pattern c (rho x 1e3 on rows with u - l <= eq_tol, reluqpth.py:54); the kernels
        // scale rho by every instance's own c.  A shared-matrix batch must therefore share the pattern.
        int32_t bad = 0;
        HIP_TRY(h, rqp_launch_check_shared_c(h, h->flag_d, s));
        HIP_TRY(h, hipMemcpyAsync(&bad, h->flag_d, sizeof(bad), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        if (bad) {
            free_ws(h);
            h->err = "rqp_setup: shared (H, A) batch whose instances differ in which rows are equalities (u - l <= eq_tol); "
                     "K(rho) is built per matrix, so pass H and A with a batch dimension for such a batch";
            return RQP_ERR_UNSUPPORTED;
        }
    }
    if (h->st.scaling > 0) {
        HIP_TRY(h, hipMalloc((void**)&h->Dsc, nm * n * sizeof(double)));
        HIP_TRY(h, hipMalloc((void**)&h->Esc, nm * m * sizeof(double)));
        HIP_TRY(h, hipMalloc((void**)&h->csc, nm * sizeof(double)));
    }
    const auto t_alloc = std::chrono::steady_clock::now();
    int rc = build_matrices(h, a, s);
    if (rc != RQP_OK) {
        free_ws(h);
        return rc;
    }
    if (h->st.scaling > 0) HIP_TRY(h, rqp_launch_scale_vecs(h, h->g, h->l, h->u, s));    // g <- c D g, l/u <- E l/u
    h->is_setup = true;
    h->cold_state = true;
    rc = rqp_clear_primal_dual(h, stream);      // zero state, rho_ind0 (reluqpth.py:148-153)
    if (h->debug & 1) {                         // host-side split of a setup call (synchronous, debug only)
        const auto t_enq = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(s);
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](auto a0, auto a1) { return std::chrono::duration<double, std::milli>(a1 - a0).count(); };
        fprintf(stderr, "[rqp] setup host split: allocate %.2f ms, enqueue %.2f ms, device drain %.2f ms\n", ms(t_begin, t_alloc),
                ms(t_alloc, t_enq), ms(t_enq, t_end));
    }
    return rc;
}

int rqp_update_mats(rqp_handle* h, const void* H, const void* A, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    if (!H && !A) return RQP_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->device));
    SetupArgs a = make_setup_args(h, H, nullptr, A, nullptr, nullptr);
    if (h->st.scaling <= 0) return build_matrices(h, a, s);             // state, rho indices, g, l, u, c untouched
    // With Ruiz scaling the packed copies are D H D / E A D: a new matrix changes D, E, c, so BOTH raw matrices are needed, and
    // the handle's vectors and state move from the old scaled space to the new one.
    if (!H || !A) return fail_unsupported(h, "rqp_update_mats with scaling needs both H and A (the packed copies are scaled)");
    const size_t nm = h->nmat, nD = nm * h->n, nE = nm * h->m;
    double* old = nullptr;
    HIP_TRY(h, hipMalloc((void**)&old, (nD + nE + nm) * sizeof(double)));
    hipError_t e1 = hipMemcpyAsync(old, h->Dsc, nD * sizeof(double), hipMemcpyDeviceToDevice, s);
    hipError_t e2 = hipMemcpyAsync(old + nD, h->Esc, nE * sizeof(double), hipMemcpyDeviceToDevice, s);
    hipError_t e3 = hipMemcpyAsync(old + nD + nE, h->csc, nm * sizeof(double), hipMemcpyDeviceToDevice, s);
    int rc = (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) ? fail_hip(h, hipErrorUnknown, "copy of the scaling factors") : RQP_OK;
    if (rc == RQP_OK) rc = build_matrices(h, a, s);                      // pack -> Ruiz (new D, E, c) -> gram -> factor -> images
    if (rc == RQP_OK && rqp_launch_rescale(h, old, old + nD, old + nD + nE, s) != hipSuccess) rc = fail_hip(h, hipGetLastError(), "k_rescale");
    (void)hipStreamSynchronize(s);                                       // `old` is freed below
    (void)hipFree(old);
    return rc;
}

int rqp_update(rqp_handle* h, const void* g, const void* l, const void* u, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    if (!g && !l && !u) return RQP_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, rqp_launch_vec_update(h, g, l, u, (hipStream_t)stream));
    if (h->st.scaling > 0)
        HIP_TRY(h, rqp_launch_scale_vecs(h, g ? h->g : nullptr, l ? h->l : nullptr, u ? h->u : nullptr, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_update_affine(rqp_handle* h, const void* p, int32_t np, const void* Gg, const void* Glu, const void* l0,
                      const void* u0, void* stream) {
    if (!h || !p || !Gg || !Glu || !l0 || !u0 || np <= 0 || np > 64) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, rqp_launch_affine_update(h, p, np, Gg, Glu, l0, u0, (hipStream_t)stream));
    if (h->st.scaling > 0) HIP_TRY(h, rqp_launch_scale_vecs(h, h->g, h->l, h->u, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_update_settings(rqp_handle* h, const rqp_settings* s) {
    if (!h || !s) return RQP_ERR_ARG;
    const rqp_settings& o = h->st;
    if (s->rho != o.rho || s->rho_min != o.rho_min || s->rho_max != o.rho_max || s->sigma != o.sigma ||
        s->adaptive_rho != o.adaptive_rho || s->adaptive_rho_tolerance != o.adaptive_rho_tolerance ||
        s->eq_tol != o.eq_tol || s->scaling != o.scaling)
        return fail_arg(h, "rqp_update_settings: only max_iter, eps_abs, eps_rel, check_interval, warm_starting, "
                           "check_infeasibility, eps_prim_inf, eps_dual_inf may change");
    if (s->max_iter < 0 || s->check_interval < 1 || s->eps_abs < 0 || s->eps_rel < 0 || s->eps_prim_inf < 0 || s->eps_dual_inf < 0)
        return fail_arg(h, "rqp_update_settings: bad value");
    if (h->windowed && s->check_infeasibility)
        return fail_unsupported(h, "rqp_update_settings: check_infeasibility on a handle set up with a rho-ladder window; "
                                   "pass check_infeasibility (or RQP_FLAG_FULL_LADDER) at setup");
    h->st.eps_rel = s->eps_rel;
    h->st.check_infeasibility = s->check_infeasibility;
    h->st.eps_prim_inf = s->eps_prim_inf;
    h->st.eps_dual_inf = s->eps_dual_inf;
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
    if (x || z || lam) h->cold_state = false;   // (a rho alone moves every instance to one index: still the common state)
    if (h->st.scaling > 0 && (x || z || lam))      // caller space -> scaled space
        HIP_TRY(h, rqp_launch_scale_state(h, x ? h->x : nullptr, z ? h->z : nullptr, lam ? h->lam : nullptr, (hipStream_t)stream));
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
    h->cold_state = true;
    return RQP_OK;
}

// Windowed handles: re-centre and re-factor the windows of the marked instances (cstat = 1): k_rewindow -> K_j of the new
// windows -> their kernel images.  Every kernel filters on cstat, so nothing here needs the host to know which instances.
static int refactor_windows(rqp_handle* h, int all, hipStream_t s) {
    HIP_TRY(h, rqp_launch_rewindow(h, all, s));
    SetupArgs f = make_setup_args(h, nullptr, nullptr, nullptr, nullptr, nullptr);
    f.only = h->cstat_d;
    set_kp_image(h, f);
    HIP_TRY(h, rqp_launch_factor(h, f, s));
    if (h->resident) HIP_TRY(h, rqp_launch_pack_res2(h, nullptr, h->cstat_d, s));
    return RQP_OK;
}

int rqp_solve(rqp_handle* h, void* x, void* z, void* lam, const rqp_info* info, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    SolveArgs a = make_solve_args(h);
    a.mode = 0;
    a.cold = h->cold_state ? 1 : 0;
    h->cold_state = !h->st.warm_starting;          // (a cold-start handle clears its state in the kernel)
    a.out_x = x; a.out_z = z; a.out_lam = lam;
    if (info) a.info = *info;
    if (a.info.trace && a.info.trace_cap < 1) return fail_arg(h, "rqp_solve: trace without capacity");
    if (h->order_d && h->last_iter_d && h->use_history) {
        a.order = h->order_valid ? h->order_d : nullptr;
        a.last_iter = h->last_iter_d;
    }
    if (h->windowed) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail_unsupported(h, "rqp_solve: a windowed handle synchronises the stream (rho-ladder window); set up with "
                                       "RQP_FLAG_FULL_LADDER to capture solves into a HIP graph");
        HIP_TRY(h, hipMemsetAsync(h->ncont_d, 0, sizeof(int32_t), s));
    }
    // Infeasibility certificates: the streaming kernel tests them at every check; the register-resident / MFMA kernels
    // keep their loops untouched and a mode-3 pass of the streaming kernel examines the instances that ran out of iterations.
    const bool post_cert = h->st.check_infeasibility && a.info.status && (h->use_mfma || h->use_wave || h->resident || h->resident64);
    if (post_cert) a.keep_state = 1;
    const bool handoff = h->use_mfma && h->handoff_cols > 0 && a.info.status != nullptr;
    if (handoff) {
        a.handoff_cols = h->handoff_cols;
        a.cont_iter = h->cont_iter_d;
        a.cont_rho = h->cont_rho_d;
        a.keep_state = 1;                           // the continue pass clears what warm_starting = 0 asks to clear
    }
    HIP_TRY(h, launch_solve(h, a, s));
    if (handoff) {                                  // stragglers finish on the per-instance resident kernel
        SolveArgs c = a;
        c.cont = 1;
        c.handoff_cols = 0;
        c.keep_state = post_cert ? 1 : 0;
        c.order = nullptr;
        c.last_iter = nullptr;
        HIP_TRY(h, rqp_launch_solve_res2(h, c, s));
    }
    const bool ranks = h->order_d && h->last_iter_d && h->use_history;   // rank the instances by what they just needed: next launch goes longest-first
    if (h->windowed) {
        // instances whose rho index left their window stopped with their exact state: new windows, then they continue
        // (at most one window move per `RQP_WINDOW / 2` index moves, i.e. per >= 2 checks of an instance).  The ranking is
        // enqueued BEFORE the host reads the count (nothing left the window in the common case: the host's wake-up latency
        // then hides behind it) and again after a continuation pass.
        if (ranks) HIP_TRY(h, rqp_launch_order_lpt(h, s));
        for (;;) {
            HIP_TRY(h, hipMemcpyAsync(h->ncont_h, h->ncont_d, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
            if (*h->ncont_h <= 0) break;
            const int rc = refactor_windows(h, 0, s);
            if (rc != RQP_OK) return rc;
            HIP_TRY(h, hipMemsetAsync(h->ncont_d, 0, sizeof(int32_t), s));
            SolveArgs c = a;
            c.cont = 2;
            c.order = nullptr;
            HIP_TRY(h, launch_solve(h, c, s));
            if (ranks) HIP_TRY(h, rqp_launch_order_lpt(h, s));
        }
        if (ranks) h->order_valid = true;
    }
    if (post_cert) {
        SolveArgs c = a;
        c.mode = 3;
        c.order = nullptr;
        c.last_iter = nullptr;
        c.keep_state = 0;
        HIP_TRY(h, rqp_launch_solve_generic(h, c, s));
    }
    if (ranks && !h->windowed) {
        HIP_TRY(h, rqp_launch_order_lpt(h, s));
        h->order_valid = true;
    }
    if (h->st.scaling > 0)                          // x = D xb, z = zb / E, lam = E lamb / c, obj / c
        HIP_TRY(h, rqp_launch_unscale_out(h, x, z, lam, a.info.obj_val, s));
    return RQP_OK;
}

int rqp_iterate(rqp_handle* h, int32_t k, void* stream) {
    if (!h || k < 0) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    SolveArgs a = make_solve_args(h);
    a.mode = 1;
    a.max_iter = k;
    h->cold_state = false;
    if (h->windowed) {                              // (test hook: no exit-and-continue here -- every window is centred first)
        const int rc = refactor_windows(h, 1, (hipStream_t)stream);
        if (rc != RQP_OK) return rc;
    }
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
    if (h->st.scaling > 0 && obj) HIP_TRY(h, rqp_launch_unscale_out(h, nullptr, nullptr, nullptr, obj, (hipStream_t)stream));
    return RQP_OK;
}

int rqp_get_state(rqp_handle* h, void* x, void* z, void* lam, int32_t* rho_ind, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, rqp_launch_state_get(h, x, z, lam, rho_ind, (hipStream_t)stream));
    if (h->st.scaling > 0) HIP_TRY(h, rqp_launch_unscale_out(h, x, z, lam, nullptr, (hipStream_t)stream));
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
    hipStream_t s = (hipStream_t)stream;
    const size_t mat = h->dims.shared_mats ? 0 : (size_t)b, one = (size_t)h->n * h->ldn;
    if (!h->windowed) {
        HIP_TRY(h, rqp_launch_get_K(h, (const char*)h->K + (mat * h->nrho + j) * one * h->esz, out, s));
        return RQP_OK;
    }
    // windowed handle: the entry may not exist in the table -- factor this one (matrix, rho) pair into a scratch matrix
    void* tmp = nullptr;
    double* fs = nullptr;
    HIP_TRY(h, hipMalloc(&tmp, one * h->esz));
    SetupArgs f = make_setup_args(h, nullptr, nullptr, nullptr, nullptr, nullptr);
    f.nmat = 1;
    f.kwin = 1;
    f.wbase = nullptr;
    f.only = nullptr;
    f.Ht = (char*)h->Ht + mat * one * h->esz;
    f.G = h->G + mat * (size_t)h->n * h->n;
    f.K = tmp;
    f.rhos = h->rhos_d + j;
    int rc = RQP_OK;
    if (h->fscratch) {                              // (large n: the factor kernel works in a global scratch slab)
        if (hipMalloc((void**)&fs, (size_t)h->n * h->n * sizeof(double)) != hipSuccess) rc = RQP_ERR_OOM;
        f.fscratch = fs;
    }
    if (rc == RQP_OK && rqp_launch_factor(h, f, s) != hipSuccess) rc = fail_hip(h, hipGetLastError(), "factor (rqp_get_K)");
    if (rc == RQP_OK && rqp_launch_get_K(h, tmp, out, s) != hipSuccess) rc = fail_hip(h, hipGetLastError(), "k_get_K");
    (void)hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (fs) (void)hipFree(fs);
    return rc;
}

int rqp_get_window(rqp_handle* h, int32_t* slots, int32_t* wbase, void* stream) {
    if (!h) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    if (slots) *slots = h->kwin;
    if (wbase && h->windowed) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(wbase, h->wbase_d, (size_t)h->nmat * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return RQP_OK;
}

int rqp_dispatch_history(rqp_handle* h, int32_t mode) {
    if (!h || mode < 0 || mode > 2) return RQP_ERR_ARG;
    h->order_valid = false;
    if (mode != 2) h->use_history = mode == 1;
    return RQP_OK;
}

int rqp_get_dispatch(rqp_handle* h, int32_t* order, int32_t* last_iter, int32_t* valid, void* stream) {
    if (!h || !valid) return RQP_ERR_ARG;
    if (!h->is_setup) return RQP_ERR_STATE;
    *valid = (h->order_d && h->order_valid) ? 1 : 0;
    if (!*valid) return RQP_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nb = (size_t)h->B * sizeof(int32_t);
    if (order) HIP_TRY(h, hipMemcpyAsync(order, h->order_d, nb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (last_iter) HIP_TRY(h, hipMemcpyAsync(last_iter, h->last_iter_d, nb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
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
