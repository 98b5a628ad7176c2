// Internal declarations shared by the translation units of librqp_hip.so.
// gfx950 (MI355X, CDNA4) only.  Not installed; the public boundary is include/rqp_abi.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "rqp_abi.h"

#define RQP_STATUS_CONTINUE (-2)   // internal: handed from the MFMA kernel to the per-instance kernel mid-solve (never returned)
#define RQP_WINDOW 5        // K(rho) slots per matrix of a windowed handle (rho_ind0 - 1 .. rho_ind0 + 3 at setup)
#define RQP_NT 256          // threads per workgroup of the generic kernels (4 wavefronts)
#define RQP_WAVE 64         // CDNA wavefront

// ---------------------------------------------------------------------------------------
// Workspace layout (all device memory, owned by the handle).  T = float | double.
//   Ht  [nmat][n][ldn]   T   H transposed (so that H x is a column-oriented product for any H)
//   A   [nmat][m][ldn]   T   row-major, leading dim padded to a multiple of 16 B
//   At  [nmat][n][ldm]   T   A transposed
//   K   [nmat][kwin][n][ldn] T  K_j = (H + sigma I + A' diag(rho_j c) A)^-1, symmetric; slot s of matrix `mat` holds
//                               ladder index wbase[mat] + s (kwin = nrho, wbase = 0: the whole ladder, as the reference builds it)
//   G   [nmat][n][n]     double  A' diag(c) A   (setup only)
//   g   [B][n], l,u,c [B][m]   T   (c_i = 1e3 on equality rows else 1)
//   x   [B][n], z,lam [B][m]   double   ADMM state (float64 accumulators)
//   rho_ind [B] int32
// nmat = 1 when dims.shared_mats else B.
// ---------------------------------------------------------------------------------------
struct rqp_handle {
    rqp_dims dims;
    rqp_settings st;
    int device = 0;
    int n = 0, m = 0, B = 0, ldn = 0, ldm = 0, nrho = 0, rho_ind0 = 0, nmat = 0;
    size_t esz = 4;
    bool is_setup = false;
    bool cold_state = false;      // every instance sits at the zero state and ONE rho index (after setup / clear_primal_dual, or a solve
                                  // with warm_starting = 0): k_admm_mfmal then regroups the solve at its first check
    std::vector<double> rhos;

    void *Ht = nullptr, *A = nullptr, *At = nullptr, *K = nullptr;
    void *g = nullptr, *l = nullptr, *u = nullptr, *c = nullptr;
    double* G = nullptr;
    double *Dsc = nullptr, *Esc = nullptr, *csc = nullptr;   // Ruiz scaling (settings.scaling > 0): D [nmat][n], E [nmat][m], c [nmat]
    double *x = nullptr, *z = nullptr, *lam = nullptr;
    int32_t* rho_ind = nullptr;
    double* rhos_d = nullptr;
    double* fscratch = nullptr;   // factor workspace when n*n doubles exceed LDS
    size_t fscratch_elems = 0;
    // resident kernel images (rqp_resident2.hip): lane-linear register / LDS layouts
    float *Apack = nullptr, *Kpack = nullptr, *Hpack = nullptr;
    bool k_direct = false;        // RQP_FLAG_LOW_MEMORY on the float32 resident kernel: no Kpack, K read from the row-major table
    float* Kscale = nullptr;      // [nmat][nrho] power-of-two scale of the fp16 K tile (tile_dtype = RQP_TILE_F16)
    bool resident = false;        // rqp_resident2.hip: A, K in VGPRs (solve, iterate and residuals modes)
    bool resident64 = false;      // rqp_res64.hip: the float64 resident kernel (n <= 104, m <= 320), all modes
    bool use_wave = false;        // rqp_wave.hip: one wavefront per instance (n <= 32, m <= 64), solve() only
    bool use_mfma = false;        // rqp_mfma.hip: shared-(H,A) batches, solve() only
    bool mfma16 = false;          // ... on the bf16 matrix pipe (rqp_mfma16.hip, tile_dtype = RQP_TILE_BF16)
    bool mfmal = false;           // ... large / sparse problems, operands streamed from L2 (rqp_mfmal.hip: n <= 320, m <= 640)
    bool mfmad = false;           // ... the same in float64 on v_mfma_f64_16x16x4_f64 (rqp_mfmad.hip: n <= 160, m <= 320)
    float* W1img = nullptr;       // lane-linear MFMA operand images ([A; H'], A, K_j)
    int* queue = nullptr;         // next-instance counter of the persistent MFMA grid
    int32_t* flag_d = nullptr;    // device scratch flag (setup-time validation)
    // rho-ladder WINDOW (batches of per-instance matrices on the resident float32 / streaming kernels): only kwin = RQP_WINDOW
    // of the nrho entries of K(rho) exist per matrix -- the reference builds all of them (reluqpth.py:52-78) although a solve
    // visits 2-4.  An instance whose index leaves its window at a check exits with its exact state (x, z, lam, A x, carried
    // rho estimate, iteration count); rqp_solve re-centres its window, re-factors it and continues the instance: bit-identical
    // to a solve on the whole ladder.
    int kwin = 0;                 // slots per matrix (nrho when not windowed)
    bool windowed = false;
    // Windowed float32 resident handles without Ruiz scaling keep NO row-major copy of A: nothing reads it after setup (solve,
    // iterate and residuals run on the register image Apack; a window move needs only sym(H) and G), so k_gram_mfma and k_pack_res2
    // read the caller's A during rqp_setup / rqp_update_mats (same stream order as every other input) -- 0.5 GB and 0.3 ms of
    // the headline batch's setup.  Needs the caller's row pitch to be the packed one (n % 4 == 0).
    bool borrow_A = false;
    // ... and no row-major K(rho) table either: k_factor_reg2 writes every K_j in the lane-linear layout of Kpack (two halves of
    // the rows through an LDS stage), k_pack_res2's K launch is gone (0.8 GB and 0.37 ms of the headline batch's setup).  Not with
    // the fp16 tile (its scale needs the block maximum first) nor with RQP_FLAG_LOW_MEMORY (which keeps the TABLE instead).
    bool kpack_direct = false;
    int32_t* wbase_d = nullptr;   // [nmat] ladder index of slot 0
    double* ax_d = nullptr;       // [B][m] A x of an instance that left its window (exact continuation)
    int32_t* cstat_d = nullptr;   // [B] 1: left its window, continue after the re-factor
    int32_t* ncont_d = nullptr;   // instances that left their window in the last pass
    int32_t* ncont_h = nullptr;   // (pinned host copy)
    // Dispatch order of the per-instance kernels.  Workgroups are issued in grid order and an instance runs as long as its
    // iteration count, so the launch ends with a tail of late, long solves (14 % of the headline launch, measured).  After
    // every solve the instances are ranked by the iteration count they just needed (counting sort on the device) and the next
    // launch issues them longest-first.  Pure scheduling: results do not depend on it.  Batches of >= 4 workgroups per CU only.
    int32_t* cont_iter_d = nullptr;   // straggler hand-off MFMA -> resident (SolveArgs)
    double* cont_rho_d = nullptr;
    int handoff_cols = 0;
    int32_t* key_d = nullptr;     // [B] sort keys of the regrouped cold solve (rqp_mfmal.hip)
    int32_t* order_d = nullptr;   // [B] instance of workgroup i
    int32_t* last_iter_d = nullptr;
    bool order_valid = false;
    bool use_history = true;      // rqp_dispatch_history
    int ncu = 0;                  // compute units of `device` (cached at rqp_create)
    int debug = 0;                // bit 0: RQP_DEBUG (occupancy print at setup), bit 1: RQP_DIAG (s_memtime build); read ONCE
                                  // at rqp_create -- diagnostics only, kernel selection never depends on the environment

    const char* kernel_name = "generic";
    std::string err;
};

// kernel argument blocks (POD, passed by value) --------------------------------------------
struct SolveArgs {
    int n, m, ldn, ldm, nrho, B;
    int max_iter, check_interval, warm_starting, rho_ind0;
    int mode;                 // 0 solve, 1 iterate-only (k = max_iter), 2 residuals-only, 3 certificates of the persisted state
    double sigma, tol, rho_min, rho_max, thr_p, thr_d, rho_in;
    double eps_pinf, eps_dinf;   // infeasibility certificates (check_infeas)
    int check_infeas;         // 1: test the OSQP certificates (generic kernel: at every check; others: rqp_solve runs the
                              //    generic kernel in mode 3 afterwards on the instances that spent their iterations)
    int keep_state;           // 1: persist the state even when warm_starting = 0 (a mode-3 pass follows and clears it)
    double eps_rel;           // 0: absolute test only (reluqpth.py:233); > 0: thresholds grow by eps_rel * the residual's scale
    // Ruiz scaling (settings.scaling > 0; NULL otherwise): D [nmat][n], E [nmat][m], c [nmat].  The kernels iterate in the scaled
    // space; every quantity of compute_residuals is taken back to the CALLER's space before its inf-norm -- row i of the primal
    // side times 1 / E_i, column j of the dual side times 1 / (c D_j) -- so that "solved" certifies eps_abs / eps_rel in the
    // caller's units (OSQP's default, scaled_termination = 0) and pri_res / dua_res are reported there.
    const double *scD, *scE, *scC;
    const void *Ht, *A, *At, *K;
    size_t sH, sA, sAt, sK;   // per-instance strides in elements (0 when shared)
    const void *g, *l, *u, *c;
    const double* rhos;
    double *x, *z, *lam;
    int32_t* rho_ind;
    void *out_x, *out_z, *out_lam;
    rqp_info info;
    // Straggler hand-off (shared-matrix batches with at most one tile per CU): a 16-instance MFMA tile iterates as long as
    // its slowest member; once at most `handoff_cols` of its columns are still unsolved at a check, the tile stops and those
    // instances finish on the per-instance resident kernel (`cont` = 1), whose iteration is ~3x shorter than a tile's.
    int handoff_cols;         // MFMA kernel: 0 = off
    int cold;                 // the batch starts from the common cold state (rqp_handle.cold_state)
    int leave_at, k0;         // k_admm_mfmal, regrouped cold solve: leave behind the check of iteration leave_at (0: off) / cont = 3: resume at k0
    int cont;                 // per-instance kernels: 1 = only instances with status RQP_STATUS_CONTINUE (MFMA hand-off), resumed at
                              // cont_iter with A x recomputed; 2 = only instances with cstat = 1 (they left their rho window),
                              // resumed exactly: cont_iter >= 0: behind the check of that iteration with A x = ax; < 0: from the
                              // start of their solve (the incoming rho index was outside the window)
    const int32_t* wbase;     // [nmat] ladder index of K slot 0 (NULL: 0)
    int kwin;                 // K slots per matrix
    double* ax;               // [B][m]
    int32_t* cstat;           // [B] (NULL: not a windowed handle)
    int32_t* ncont;
    int32_t* cont_iter;       // [B] iterations done at the hand-off
    double* cont_rho;         // [B] carried rho estimate at the hand-off (Q4)
    const int32_t* order;     // workgroup -> instance (NULL: identity)
    int32_t* last_iter;       // iteration count of this solve, for the next launch's order (NULL: not recorded)
    double *r_pri, *r_dua, *r_rho, *r_obj;   // mode 2 outputs
};

struct SetupArgs {
    int n, m, ldn, ldm, nrho, B, nmat;
    double sigma, eq_tol;
    const void *H_in, *A_in, *g_in, *l_in, *u_in;
    void *Ht, *A, *At, *K, *g, *l, *u, *c;
    double* G;
    const double* rhos;
    double* fscratch;
    int kwin;                 // K slots per matrix; slot s <-> ladder index (wbase ? wbase[mat] : 0) + s
    const int32_t* wbase;
    const int32_t* only;      // [nmat] (NULL: all) build only the matrices with only[mat] != 0 (re-factor of a moved window)
    // k_factor_reg2 writes K_j straight into the register image of k_admm_res2 (rqp_handle.kpack_direct; NULL: row-major table K)
    float* kp_img;            // Kpack[mat][slot][pair][256][2]
    int kp_cw, kp_kr, kp_kc;  // tile constants of the handle's Res2Cfg: columns per wave, K rows per lane, K columns per lane
};

// launchers (defined in the .hip files); return hipError_t of the launch
hipError_t rqp_launch_pack_mats(const rqp_handle* h, const SetupArgs& a, hipStream_t s);
hipError_t rqp_launch_pack_vecs(const rqp_handle* h, const SetupArgs& a, hipStream_t s);
hipError_t rqp_launch_check_shared_c(const rqp_handle* h, int32_t* flag, hipStream_t s);
hipError_t rqp_launch_affine_update(const rqp_handle* h, const void* p, int np, const void* Gg, const void* Glu,
                                    const void* l0, const void* u0, hipStream_t s);
hipError_t rqp_launch_vec_update(const rqp_handle* h, const void* g, const void* l, const void* u, hipStream_t s);
hipError_t rqp_launch_gram(const rqp_handle* h, const SetupArgs& a, hipStream_t s);
hipError_t rqp_launch_factor(rqp_handle* h, const SetupArgs& a, hipStream_t s);
hipError_t rqp_launch_solve_generic(const rqp_handle* h, const SolveArgs& a, hipStream_t s);
hipError_t rqp_launch_state_set(const rqp_handle* h, const void* x, const void* z, const void* lam, int set_rho,
                                int rho_ind, hipStream_t s);
hipError_t rqp_launch_state_get(const rqp_handle* h, void* x, void* z, void* lam, int32_t* rho_ind, hipStream_t s);
hipError_t rqp_launch_ruiz(const rqp_handle* h, hipStream_t s);
hipError_t rqp_launch_rescale(const rqp_handle* h, const double* D0, const double* E0, const double* c0, hipStream_t s);
hipError_t rqp_launch_scale_vecs(const rqp_handle* h, void* g, void* l, void* u, hipStream_t s);
hipError_t rqp_launch_scale_state(const rqp_handle* h, double* x, double* z, double* lam, hipStream_t s);
hipError_t rqp_launch_unscale_out(const rqp_handle* h, void* x, void* z, void* lam, double* obj, hipStream_t s);
hipError_t rqp_launch_order_lpt(const rqp_handle* h, hipStream_t s);
hipError_t rqp_launch_order_by(const rqp_handle* h, const int32_t* key, hipStream_t s);
hipError_t rqp_launch_get_K(const rqp_handle* h, const void* Kmat, void* out, hipStream_t s);
hipError_t rqp_launch_rewindow(const rqp_handle* h, int all, hipStream_t s);

// one-time launch preparation (dynamic-LDS function attributes), called from rqp_setup for the selected kernels
hipError_t rqp_prepare_generic(const rqp_handle* h);
hipError_t rqp_prepare_res2(const rqp_handle* h);
hipError_t rqp_prepare_mfma(const rqp_handle* h);
size_t rqp_generic_lds_bytes(const rqp_handle* h);

// resident (register/LDS) variant, float32 only
bool rqp_res2_fits(const rqp_handle* h);
void rqp_res2_pack_elems(const rqp_handle* h, size_t* a_elems, size_t* k_elems, size_t* h_elems);
void rqp_res2_kp_layout(const rqp_handle* h, int* cw, int* kr, int* kc);
hipError_t rqp_launch_pack_res2(const rqp_handle* h, const void* A_src, const int32_t* only, hipStream_t s);   // A_src NULL: Apack is kept
hipError_t rqp_launch_solve_res2(const rqp_handle* h, const SolveArgs& a, hipStream_t s);

bool rqp_res64_fits(const rqp_handle* h);
hipError_t rqp_prepare_res64(const rqp_handle* h);
hipError_t rqp_launch_solve_res64(const rqp_handle* h, const SolveArgs& a, hipStream_t s);

bool rqp_wave_fits(const rqp_handle* h);
hipError_t rqp_launch_solve_wave(const rqp_handle* h, const SolveArgs& a, hipStream_t s);

bool rqp_mfma_fits(const rqp_handle* h);
size_t rqp_mfma_img_elems(const rqp_handle* h);
hipError_t rqp_launch_pack_mfma(const rqp_handle* h, hipStream_t s);
hipError_t rqp_launch_solve_mfma(const rqp_handle* h, const SolveArgs& a, hipStream_t s);
// large / sparse shared-matrix problems (rqp_mfmal.hip): operands streamed from L2, non-zero k-step lists
bool rqp_mfmal_fits(const rqp_handle* h);
size_t rqp_mfmal_img_elems(const rqp_handle* h);
hipError_t rqp_launch_pack_mfmal(const rqp_handle* h, hipStream_t s);
hipError_t rqp_prepare_mfmal(const rqp_handle* h);
hipError_t rqp_launch_solve_mfmal(const rqp_handle* h, const SolveArgs& a, hipStream_t s);
bool rqp_mfmad_fits(const rqp_handle* h);
size_t rqp_mfmad_img_elems(const rqp_handle* h);
hipError_t rqp_launch_pack_mfmad(const rqp_handle* h, hipStream_t s);
hipError_t rqp_prepare_mfmad(const rqp_handle* h);
hipError_t rqp_launch_solve_mfmad(const rqp_handle* h, const SolveArgs& a, hipStream_t s);
// the same kernel on the bf16 matrix pipe (rqp_mfma16.hip; rqp_dims.tile_dtype = RQP_TILE_BF16): same shapes as rqp_mfma_fits
size_t rqp_mfma16_img_elems(const rqp_handle* h);
hipError_t rqp_launch_pack_mfma16(const rqp_handle* h, hipStream_t s);
hipError_t rqp_prepare_mfma16(const rqp_handle* h);
hipError_t rqp_launch_solve_mfma16(const rqp_handle* h, const SolveArgs& a, hipStream_t s);

// Raise -- never lower -- the dynamic-LDS limit of kernel `fn` on the current device.  The attribute belongs to the FUNCTION, not
// to a handle: a later, smaller handle must not shrink the limit an earlier, larger handle's launches rely on (process-wide
// maximum per function and device, rqp_abi.hip).
hipError_t rqp_raise_lds_limit(const void* fn, size_t bytes);

static inline int rqp_round_up(int v, int q) { return (v + q - 1) / q * q; }
