/*
 * rqp_abi.h -- C ABI of librqp_hip.so, the MI355X (gfx950) ReLU-QP hot path.
 *
 * The reference (gstoica27/ReLUQP-py) has NO foreign-function interface: its hot
 * path is the Python class reluqp.reluqpth.ReLU_QP calling torch.  This header is
 * therefore the boundary a maintainer would bind *under* that class (ctypes stub in
 * INTEGRATION.md); each entry point names the reference method it replaces
 * (paths relative to /root/reference/ReLU-QP-py/reluqp/).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / C++ types cross the ABI.
 *  - Every `const void*` / `void*` data pointer is a DEVICE pointer (e.g.
 *    torch.Tensor.data_ptr()) of element type `dims.dtype` unless stated; the caller
 *    owns all of them.  The library owns only its workspace (packed H/A copies, the
 *    K(rho) table, per-instance ADMM state), allocated in rqp_setup, freed in
 *    rqp_destroy.
 *  - `stream` is a hipStream_t passed as void* (0 = default stream).  All work is
 *    enqueued on it; nothing synchronises the host except where stated.
 *  - Every function returns 0 on success or a negative rqp_error; nothing throws.
 *  - A handle is not thread-safe: one handle per host thread / GPU.
 *  - Batched: every instance b in [0,batch) has its own g,l,u, state, rho index,
 *    iteration count and exit; H and A are per instance, or shared by all instances
 *    (dims.shared_mats = 1, then H is [n,n] and A is [m,n]).
 */
#ifndef RQP_ABI_H
#define RQP_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rqp_handle rqp_handle;

enum rqp_dtype { RQP_F32 = 0, RQP_F64 = 1 };

/* Storage type of the preconditioner tile K_j in the register-resident kernels (BASELINE
 * config 5, SURVEY.md 7.3 "matrix tile fp16, x/z/lam and residual fp32").  K only
 * preconditions the residual correction dx = -K d: its rounding
 * changes the convergence rate, never the fixed point; H, A and every residual stay in
 * dims.dtype.  RQP_TILE_F16 needs dims.dtype == RQP_F32.  The MFMA kernel (shared H, A)
 * takes the same fp16-rounded K into its float32 operand image.                        */
enum rqp_tile_dtype {
    RQP_TILE_SAME = 0,
    RQP_TILE_F16 = 1,
    /* Shared-(H, A) batches on the MFMA kernel: every matrix tile ([A; H'], A, K_j) and every vector operand is kept as two
     * bf16 planes (hi + mid = 16 significant bits) and every product runs as three v_mfma_f32_16x16x32_bf16 with float32
     * accumulation -- the 16-bit matrix pipe (16x the float32 MFMA rate); state, residuals and the checks stay float32.
     * Same recurrence as the float32 kernel; operand error 2^-16 relative (stated tolerance: eps_abs >= 1e-5).  Needs
     * dims.dtype == RQP_F32, shared_mats, n <= 80, m <= 320 (the MFMA kernel's shapes).                                     */
    RQP_TILE_BF16 = 2
};

/* Solve-kernel request (rqp_dims.kernel).  AUTO = measured dispatch by size / batch /
 * sharing; an explicit kernel that cannot hold the problem makes rqp_setup return
 * RQP_ERR_UNSUPPORTED (nothing falls back silently).                                    */
enum rqp_kernel {
    RQP_KERNEL_AUTO = 0,
    RQP_KERNEL_GENERIC = 1,   /* streaming, any n, m, f32/f64                              */
    RQP_KERNEL_RESIDENT = 2,  /* A, K in VGPRs, one workgroup per instance (f32; f64 tile) */
    RQP_KERNEL_WAVE = 3,      /* one wavefront per instance, small problems                */
    RQP_KERNEL_MFMA = 4       /* shared (H, A), f32: batch on the MFMA N axis -- operands in registers (n <= 80,
                               * m <= 320), else streamed from L2 as non-zero 16 x 16 blocks (n <= 320, m <= 640);
                               * f64: streamed operands on v_mfma_f64_16x16x4_f64 (n <= 160, m <= 320)              */
};

enum rqp_error {
    RQP_OK = 0,
    RQP_ERR_ARG = -1,        /* null pointer / bad dimension / bad setting       */
    RQP_ERR_STATE = -2,      /* call order (e.g. solve before setup)             */
    RQP_ERR_HIP = -3,        /* a HIP runtime call failed: see rqp_last_error    */
    RQP_ERR_OOM = -4,        /* workspace allocation failed                      */
    RQP_ERR_UNSUPPORTED = -5 /* size / mode no kernel of this build covers        */
};

/* per-instance exit status written to rqp_info.status (reluqpth.py:236,245) */
enum rqp_status {
    RQP_STATUS_SOLVED = 0,          /* "solved"             */
    RQP_STATUS_MAX_ITER = 1,        /* "max_iters_reached"  */
    RQP_STATUS_NAN = 2,             /* "nan_detected": a residual of the last check is NaN (Q17 made visible;
                                       only raised when the iteration budget is spent -- the loop control is
                                       the reference's)                                                        */
    RQP_STATUS_PRIMAL_INFEASIBLE = 3, /* "primal_infeasible": certificate found (check_infeasibility)          */
    RQP_STATUS_DUAL_INFEASIBLE = 4,   /* "dual_infeasible"                                                     */
    RQP_STATUS_UNSOLVED = -1        /* never solved          */
};

typedef struct rqp_dims {
    int32_t n;            /* decision variables   (QP.nx, classes.py:29)              */
    int32_t m;            /* constraints          (QP.nc, classes.py:30)              */
    int32_t batch;        /* independent instances (>= 1)                             */
    int32_t shared_mats;  /* 1: one (H, A) shared by all instances (linear MPC)       */
    int32_t dtype;        /* rqp_dtype of every data pointer                          */
    int32_t kernel;       /* rqp_kernel request (0 = auto)                            */
    int32_t tile_dtype;   /* rqp_tile_dtype of the resident K tile (0 = same as dtype) */
    int32_t flags;        /* RQP_FLAG_* bits (0 = defaults)                           */
} rqp_dims;

/* rqp_dims.flags */
enum {
    RQP_FLAG_LOW_MEMORY = 1,  /* resident float32 kernel: keep K(rho) as the row-major table (n x ldn per entry) and load it
                                 from there, instead of the larger lane-linear register image (padded to the kernel's tile):
                                 less workspace for ~1 % more solve time; results are bit-identical.  (A windowed handle
                                 holds ONE of the two: by default the factor kernel writes the register image directly.)  */
    RQP_FLAG_FULL_LADDER = 2  /* build K(rho) for EVERY entry of the rho ladder of every matrix, as the reference does
                                 (reluqpth.py:52-78).  Default for batches of >= 32 per-instance matrices on the resident
                                 float32 / streaming kernels: a WINDOW of 5 entries around each instance's index
                                 (rho_ind0 - 1 .. rho_ind0 + 3 at setup; a solve visits 2-4 entries).  An instance whose
                                 index leaves its window exits with its exact state, rqp_solve re-factors a window around
                                 the new index and continues it -- results are bit-identical to the full ladder, setup and
                                 workspace shrink by ~3x, and rqp_solve synchronises `stream` once per pass (it cannot be
                                 captured into a HIP graph: use this flag there).                                    */
};

/* Settings of classes.py:32-65 that reach the device (same names, same defaults). */
typedef struct rqp_settings {
    double rho;                    /* 0.1   */
    double rho_min;                /* 1e-6  */
    double rho_max;                /* 1e6   */
    double sigma;                  /* 1e-6  */
    double adaptive_rho_tolerance; /* 5     */
    double eps_abs;                /* 1e-3  */
    double eq_tol;                 /* 1e-6  */
    int32_t adaptive_rho;          /* 1     */
    int32_t max_iter;              /* 4000  */
    int32_t check_interval;        /* 25    */
    int32_t warm_starting;         /* 1     */
    /* --- extensions (all default to the reference's behaviour) --- */
    double eps_rel;                /* 0: absolute test only (reluqpth.py:233).  > 0: OSQP-style
                                      pri < eps_abs*sqrt(m) + eps_rel*max(|Ax|,|z|),
                                      dua < eps_abs*sqrt(n) + eps_rel*max(|Hx|,|A'lam|,|g|)  (inf-norms)   */
    double eps_prim_inf;           /* 1e-4  certificate tolerances (check_infeasibility)                    */
    double eps_dual_inf;           /* 1e-4  */
    int32_t scaling;               /* 0: none (the reference's `scaling` is an unused TODO, reluqpth.py:105);
                                      k > 0: k Ruiz equilibration passes at setup; results, residuals and the
                                      termination test in the caller's (un-scaled) units                       */
    int32_t check_infeasibility;   /* 0.  1: OSQP's primal / dual infeasibility certificates.  The streaming kernel
                                      (RQP_KERNEL_GENERIC) tests them at every check and exits at the first one that
                                      holds; the register-resident / wavefront / MFMA kernels keep their loops
                                      untouched: an instance that spends max_iter (or ends nan_detected) is examined
                                      once, after the launch, and labelled -- it has then run its whole budget.
                                      AUTO therefore dispatches a handle with this option to the streaming kernel;
                                      the other kernels take it on explicit request only.                          */
} rqp_settings;

/* Per-instance results of one solve (classes.py:67-88), struct of DEVICE arrays,
 * each of length `batch`; any pointer may be NULL to skip that field.           */
typedef struct rqp_info {
    int32_t* iter;         /* Info.iter                                             */
    int32_t* status;       /* rqp_status                                            */
    int32_t* rho_ind;      /* rho index after the solve (ReLU_QP.rho_ind)           */
    double* pri_res;       /* Info.pri_res                                          */
    double* dua_res;       /* Info.dua_res                                          */
    double* rho_estimate;  /* Info.rho_estimate                                     */
    double* obj_val;       /* Info.obj_val = 1/2 x'Hx + g'x   (reluqpth.py:320-322)  */
    /* optional per-check trace, [batch][trace_cap][4] doubles:
     * (pri, dua, rho_estimate, rho index before the move), check c = iteration
     * (c+1)*check_interval; rows past the last check are left untouched.         */
    double* trace;
    int32_t trace_cap;
    int32_t reserved;
} rqp_info;

/* Fill *s with the defaults of classes.py:32-65. */
int rqp_default_settings(rqp_settings* s);

/* ReLU_QP.__init__ + Settings (reluqpth.py:93-100,128-142): validates dims and
 * settings, builds the rho ladder (setup_rhos, reluqpth.py:20-38) and binds `device`. */
int rqp_create(rqp_handle** out, const rqp_dims* dims, const rqp_settings* settings, int device);

/* ReLU_QP.setup (reluqpth.py:102-157): QP.__init__ casts (classes.py:4-30), the
 * per-rho KKT inverses of ReLU_Layer.setup_matrices (reluqpth.py:40-78; here
 * K_j = (H + sigma I + A' diag(rho_j c) A)^-1 for every ladder entry, c_i = 1e3 on
 * rows with u_i - l_i <= eq_tol), zero state and rho_ind = argmin|rhos - rho|.
 * H [batch|1][n][n], g [batch][n], A [batch|1][m][n], l,u [batch][m], row-major.
 * The inputs are read by kernels enqueued on `stream` (every one of them more than once:
 * some handles keep no copy of A and build A'cA and the register image of A straight from
 * the caller's buffer): they must stay valid and unchanged until that work has completed,
 * like the operands of any stream-ordered call.  Nothing is read after that.           */
int rqp_setup(rqp_handle* h, const void* H, const void* g, const void* A, const void* l,
              const void* u, void* stream);

/* ReLU_QP.update (reluqpth.py:159-183): new g and/or l and/or u ([batch][n] /
 * [batch][m]); NULL = unchanged.  State is untouched.                           */
int rqp_update(rqp_handle* h, const void* g, const void* l, const void* u, void* stream);

/* ReLU_QP.update(Hx=, Ax=) -- rejected upstream (`assert`, reluqpth.py:176-177), SURVEY.md
 * 8(f)-4: new H and/or A (same shapes as in rqp_setup; NULL = unchanged).  Re-runs the
 * device setup chain (pack -> A'cA -> K(rho) ladder -> kernel images) on the existing
 * workspace; g, l, u, the equality pattern c, the ADMM state and the rho indices are kept,
 * i.e. the next solve is warm-started exactly like after rqp_update.  With
 * settings.scaling > 0 BOTH raw matrices must be given (the packed copies are scaled):
 * the problem is re-equilibrated and the stored vectors and state move to the new scaled
 * space (the call then synchronises `stream`).                                           */
int rqp_update_mats(rqp_handle* h, const void* H, const void* A, void* stream);

/* Parametric form of ReLU_QP.update for linear MPC (the x0 update of the reference's driver,
 * loose_code/RandomLinMPC.py / SURVEY.md Appendix C, evaluated on the device in one pass):
 *   g[b] = Gg p[b],   l[b] = l0 + Glu p[b],   u[b] = u0 + Glu p[b]
 * p [batch][np] (np <= 64), Gg [n][np], Glu [m][np], l0/u0 [m], all device pointers in the
 * handle's dtype.  Same effect as rqp_update with those vectors; state is untouched.       */
int rqp_update_affine(rqp_handle* h, const void* p, int32_t np, const void* Gg, const void* Glu,
                      const void* l0, const void* u0, void* stream);

/* ReLU_QP.update_settings (reluqpth.py:185-199, whose whitelist is max_iter, eps_abs, verbose,
 * check_interval).  Changeable after setup: max_iter, eps_abs, check_interval, warm_starting and
 * the extension fields eps_rel, check_infeasibility, eps_prim_inf, eps_dual_inf.  A difference in
 * any other field (rho, rho_min, rho_max, sigma, adaptive_rho, adaptive_rho_tolerance, eq_tol,
 * scaling -- they shape the K(rho) ladder built at setup) returns RQP_ERR_ARG (the reference
 * raises ValueError).                                                                          */
int rqp_update_settings(rqp_handle* h, const rqp_settings* settings);

/* ReLU_QP.warm_start (reluqpth.py:251-276) with Q6 fixed (values are written into
 * the iterate).  x [batch][n], z, lam [batch][m]; NULL = unchanged.  `rho` selects
 * rho_ind = argmin|rhos - rho| for every instance when has_rho != 0.              */
int rqp_warm_start(rqp_handle* h, const void* x, const void* z, const void* lam, int has_rho,
                   double rho, void* stream);

/* ReLU_QP.clear_primal_dual (reluqpth.py:324-333): zero state, reset rho index.  */
int rqp_clear_primal_dual(rqp_handle* h, void* stream);

/* ReLU_QP.solve + update_results (reluqpth.py:201-249,278-305): the whole ADMM
 * loop of every instance -- iterate (jit_forward :84-89), every check_interval
 * iterations compute_residuals (:307-318), rho-index move (:223-227), termination
 * (:233) -- in ONE kernel launch, one workgroup (or wavefront, or MFMA column) per
 * instance, no host round trip.  Small follow-up launches on the same stream where
 * they apply: the dispatch-order ranking for the next solve, the straggler pass of a
 * shared-matrix batch, the certificate pass (check_infeasibility), the un-scaling.
 * Writes x [batch][n], z [batch][m], lam [batch][m] (the state's dual, Q7/Q16) and
 * *info; any of them may be NULL.  Asynchronous on `stream`.                      */
int rqp_solve(rqp_handle* h, void* x, void* z, void* lam, const rqp_info* info, void* stream);

/* k plain iterations at each instance's current rho index, no checks: exactly k
 * applications of ReLU_Layer.forward (reluqpth.py:80-89).  For parity tests.     */
int rqp_iterate(rqp_handle* h, int32_t k, void* stream);

/* ReLU_QP.compute_residuals + compute_J (reluqpth.py:307-322) on the current
 * state with carried estimate rho_in (host scalar): pri, dua, rho_out, obj are
 * device arrays [batch] of doubles (NULL to skip).                               */
int rqp_compute_residuals(rqp_handle* h, double rho_in, double* pri, double* dua,
                          double* rho_out, double* obj, void* stream);

/* Copy the current state out (x [batch][n], z, lam [batch][m] in dims.dtype,
 * rho_ind [batch] int32); NULL to skip.                                           */
int rqp_get_state(rqp_handle* h, void* x, void* z, void* lam, int32_t* rho_ind, void* stream);

/* The rho ladder (host doubles).  *count receives its length; rhos may be NULL.  */
int rqp_get_rhos(const rqp_handle* h, double* rhos, int32_t cap, int32_t* count);

/* K_j of instance b (ReLU_Layer.kkt_rhs_invs[j], reluqpth.py:56) -> out [n][n]
 * in dims.dtype (device pointer).  For parity tests.                              */
int rqp_get_K(rqp_handle* h, int32_t b, int32_t j, void* out, void* stream);

/* Dispatch order of the per-instance kernels.  After every solve the library ranks the instances by
 * the iteration count they just needed and issues the next launch longest-first (pure scheduling:
 * results never depend on it; it pays when consecutive solves resemble each other -- closed loops,
 * parameter sweeps).  mode 1 = on (default), 0 = off: every launch in grid order, nothing recorded
 * (a handle then behaves at every solve like a fresh handle on a fresh batch), 2 = forget the
 * recorded order now and stay on.  No reference counterpart (one QP per object there).            */
int rqp_dispatch_history(rqp_handle* h, int32_t mode);

/* Inspection hook for the tests: the workgroup -> instance permutation the NEXT launch would use and the
 * iteration counts it was ranked by (device arrays [batch] of int32, NULL to skip).  *valid (host)
 * receives 1 when a recorded order exists, else 0 (then nothing is copied).                       */
int rqp_get_dispatch(rqp_handle* h, int32_t* order, int32_t* last_iter, int32_t* valid, void* stream);

/* K(rho) slots per matrix of this handle: *slots = the ladder length (whole ladder) or the window size
 * (see RQP_FLAG_FULL_LADDER); wbase (device int32 [batch], NULL to skip; windowed handles only) receives the
 * ladder index of slot 0 of every instance's window.  For tests.                                        */
int rqp_get_window(rqp_handle* h, int32_t* slots, int32_t* wbase, void* stream);

/* Which solve kernel the handle dispatches to ("generic", "resident", ...).       */
const char* rqp_kernel_name(const rqp_handle* h);

int rqp_destroy(rqp_handle* h);

const char* rqp_strerror(int err);
/* Text of the last failure on this handle (HIP error string etc.).                */
const char* rqp_last_error(const rqp_handle* h);
/* Library / ABI version, e.g. "rqp-hip 0.1 gfx950".                               */
const char* rqp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RQP_ABI_H */
