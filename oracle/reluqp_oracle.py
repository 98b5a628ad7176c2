"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (numpy) restatement of the reference's ReLU-QP algorithm, function by
function, for the hot path of SURVEY.md section 8(a).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / reported baseline.  The product
(`reluqp-py_amd/`) never imports it and fails loudly without its HIP library.

Parity is PINNED: `tests/test_oracle_golden.py` checks this restatement against
`tests/golden/*.npz`, which were produced by running the reference's own code in
the build container (`tests/golden/make_golden.py`; harness of SURVEY.md 8(c)).

Every function cites the reference lines it follows (paths relative to
/root/reference/ReLU-QP-py/reluqp/).  Two statement forms are provided:

* W form      -- the reference's literal formulation: dense (n+2m)^2 matrix per
                 rho (reluqpth.py:71-77), s <- clamp_z(W s + b).
* factored    -- the algebraically identical recurrence that only needs
                 K = (H + sigma I + A' diag(rho) A)^-1 and A.
* refine      -- the same recurrence again, re-associated as a residual correction
                 x+ = x - K (H x + g + A' nu) with the vector state (x, lam, A x)
                 accumulated in float64 and the matrix-vector products applied in
                 the working dtype.  In exact arithmetic it is identical to the two
                 forms above; in float32 it is the only one of the three that keeps
                 the reference's float64 trajectory (DESIGN.md "fp32 numerics").
                 THIS is the statement the HIP kernels implement.

`quirks=True` replicates the reference including the behaviours the build fixes
(Q3, Q6, Q11); `quirks=False` is the build's disposition
(fixes applied, everything marked "replicate" still replicated).
"""
from __future__ import annotations

import time

import numpy as np

STATUS_SOLVED = "solved"
STATUS_MAX_ITER = "max_iters_reached"
# extensions of the build (not in the reference): SURVEY.md section 5 / 8(f)-3
STATUS_NAN = "nan_detected"
STATUS_PRIMAL_INFEASIBLE = "primal_infeasible"
STATUS_DUAL_INFEASIBLE = "dual_infeasible"


# --------------------------------------------------------------------------- a2
class Settings:
    """Defaults of classes.py:32-65 (field names are the contract)."""

    def __init__(self, verbose=False, warm_starting=True, scaling=False, rho=0.1,
                 rho_min=1e-6, rho_max=1e6, sigma=1e-6, adaptive_rho=True,
                 adaptive_rho_interval=1, adaptive_rho_tolerance=5, max_iter=4000,
                 eps_abs=1e-3, eq_tol=1e-6, check_interval=25, dtype=np.float64,
                 eps_rel=0.0, check_infeasibility=False, eps_prim_inf=1e-4, eps_dual_inf=1e-4):
        self.verbose = verbose
        self.warm_starting = warm_starting
        self.scaling = scaling
        self.rho = rho
        self.rho_min = rho_min
        self.rho_max = rho_max
        self.sigma = sigma
        self.adaptive_rho = adaptive_rho
        self.adaptive_rho_interval = adaptive_rho_interval
        self.adaptive_rho_tolerance = adaptive_rho_tolerance
        self.max_iter = max_iter
        self.eps_abs = eps_abs
        self.eq_tol = eq_tol
        self.check_interval = check_interval
        self.dtype = dtype
        # extensions of the build, absent from the reference; the defaults are the reference's behaviour
        self.eps_rel = eps_rel
        self.check_infeasibility = check_infeasibility
        self.eps_prim_inf = eps_prim_inf
        self.eps_dual_inf = eps_dual_inf


# --------------------------------------------------------------------------- a3
def setup_rhos(rho, rho_min, rho_max, tol, adaptive_rho=True):
    """rho ladder, reluqpth.py:20-38: repeated /tol and *tol in Python floats,
    then sorted ascending; a single element when adaptive_rho is False."""
    rhos = [rho]
    if adaptive_rho:
        r = rho / tol
        while r >= rho_min:
            rhos.append(r)
            r = r / tol
        r = rho * tol
        while r <= rho_max:
            rhos.append(r)
            r = r * tol
        rhos.sort()
    return np.array(rhos, dtype=np.float64)


def rho_vector(rho_scalar, l, u, eq_tol):
    """Per-row penalty, reluqpth.py:53-54 / :64-65: x1e3 on rows with u-l <= eq_tol."""
    rho = rho_scalar * np.ones(l.shape[0], dtype=l.dtype)
    with np.errstate(invalid="ignore"):
        rho[(u - l) <= eq_tol] = rho_scalar * 1e3
    return rho


# ------------------------------------------------------- build extension: scaling
def ruiz_scale(H, A, passes):
    """Modified Ruiz equilibration of the KKT matrix [[H, A'], [A, 0]] (OSQP's scaling; the reference only carries a
    `scaling` TODO, reluqpth.py:105,335).  Statement of csrc/rqp_scale.hip:k_ruiz, same order of operations:
    returns (D[n], E[m], c, Hbar = c D H D, Abar = E A D)."""
    n, m = H.shape[0], A.shape[0]
    Hs, As = H.astype(np.float64).copy(), A.astype(np.float64).copy()
    D, E = np.ones(n), np.ones(m)

    def lim(nrm):
        nrm = np.where(nrm < 1e-4, 1.0, nrm)
        return 1.0 / np.sqrt(np.minimum(nrm, 1e4))
    for _ in range(passes):
        dl = lim(np.maximum(np.abs(Hs).max(axis=0), np.abs(As).max(axis=0) if m else 0.0))
        ep = lim(np.abs(As).max(axis=1))
        Hs = (Hs * dl[:, None]) * dl[None, :]
        As = (As * ep[:, None]) * dl[None, :]
        D, E = D * dl, E * ep
    cv = np.abs(Hs).max(axis=0).sum() / n if passes > 0 else 1.0
    cv = 1.0 if cv < 1e-4 else min(cv, 1e4)
    c = 1.0 / cv
    return D, E, c, Hs * c, As


# --------------------------------------------------------------------------- a4
def kkt_inverse(H, A, rho_vec, sigma):
    """K = inverse(H + sigma I + A' diag(rho) A), reluqpth.py:56."""
    n = H.shape[0]
    return np.linalg.inv(H + sigma * np.eye(n, dtype=H.dtype) + A.T @ (rho_vec[:, None] * A))


def setup_matrices_W(H, g, A, l, u, rhos, sigma, eq_tol):
    """W_ks, B_ks, b_ks for every rho, reluqpth.py:40-78 (block rows :72-74,
    B at :76, b = B g at :77)."""
    n, m = H.shape[0], A.shape[0]
    Ix, Ic = np.eye(n, dtype=H.dtype), np.eye(m, dtype=H.dtype)
    W_ks, B_ks, b_ks = [], [], []
    for rho_scalar in rhos:
        rv = rho_vector(H.dtype.type(rho_scalar), l, u, eq_tol)
        K = kkt_inverse(H, A, rv, sigma)
        Rho = np.diag(rv)
        Rho_inv = np.diag(1.0 / rv)
        AtRA = A.T @ (Rho @ A)
        W = np.block([
            [K @ (sigma * Ix - AtRA), 2 * K @ A.T @ Rho, -K @ A.T],
            [A @ K @ (sigma * Ix - AtRA) + A, 2 * A @ K @ A.T @ Rho - Ic, -A @ K @ A.T + Rho_inv],
            [Rho @ A, -Rho, Ic]])
        B = np.vstack([-K, -A @ K, np.zeros((m, n), dtype=H.dtype)])
        W_ks.append(np.ascontiguousarray(W))
        B_ks.append(B)
        b_ks.append(B @ g)
    return W_ks, B_ks, b_ks


# --------------------------------------------------------------------------- a5
def _clamp(v, lo, hi):
    """torch.clamp semantics (NaN stays NaN), reluqpth.py:88."""
    out = v.copy()
    with np.errstate(invalid="ignore"):
        out = np.where(v < lo, lo, out)
        out = np.where(v > hi, hi, out)
    return out


def forward_W(s, W, b, l, u, n, m):
    """One ADMM iteration in W form: s <- W s + b; clamp z-slice.  Non-aliased
    statement of reluqpth.py:84-89 (Q1: the shipped out=input aliasing is a bug)."""
    out = W @ s + b
    out[n:n + m] = _clamp(out[n:n + m], l, u)
    return out


def forward_factored(x, z, lam, zt, K, A, g, l, u, rho_vec, sigma):
    """One ADMM iteration in factored form, carrying
    zt = A x so that A is applied once forward and once transposed per iteration.

        lam_hat = lam + rho*(A x - z)
        x+      = K (sigma x - g + A'(rho*z - lam_hat))
        z+      = clamp(A x+ + lam_hat/rho, l, u)
        lam+    = lam_hat
    """
    lam_hat = lam + rho_vec * (zt - z)
    r = sigma * x - g + A.T @ (rho_vec * z - lam_hat)
    xn = K @ r
    ztn = A @ xn
    zn = _clamp(ztn + lam_hat / rho_vec, l, u)
    return xn, zn, lam_hat, ztn


def forward_refine(x64, z64, lam64, zt64, K, H, A, At, g, l, u, rho_vec):
    """One ADMM iteration, residual-correction statement of Appendix A.2.

    x64, z64, lam64, zt64 (= A x) are float64 accumulators; K, H, A, At(=A'), g, l,
    u, rho_vec are in the working dtype T and every matrix-vector product is
    evaluated in T.  With M = H + sigma I + A' rho A and K = M^-1:

        p       = A x - z                        (tracked: zt - z)
        lam_hat = lam + rho*p
        nu      = lam_hat + rho*p
        d       = H x + g + A' nu                (= M x - (sigma x - g + A'(rho z - lam_hat)))
        dx      = -K d                           (so x + dx = K (sigma x - g + A'(rho z - lam_hat)))
        x+      = x + dx ;  (A x)+ = A x + A dx
        z+      = clamp(A x+ + lam_hat/rho, l, u) ;  lam+ = lam_hat
    """
    T = K.dtype.type
    rv = rho_vec.astype(np.float64)
    p = zt64 - z64
    lam_hat = lam64 + rv * p
    nu = (lam_hat + rv * p).astype(T)
    d = H @ x64.astype(T) + g + At @ nu
    dx = -(K @ d)
    xn = x64 + dx.astype(np.float64)
    ztn = zt64 + (A @ dx).astype(np.float64)
    zn = _clamp(ztn + lam_hat / rv, l.astype(np.float64), u.astype(np.float64))
    return xn, zn, lam_hat, ztn


# --------------------------------------------------------------------------- a6
def _inf_norm(v):
    """torch.linalg.vector_norm(ord=inf): NaN-propagating max|v| (0 for empty)."""
    if v.size == 0:
        return v.dtype.type(0)
    a = np.abs(v)
    return a.max() if not np.isnan(a).any() else v.dtype.type(np.nan)


def _tmax(a, b):
    """torch.max(a, b) on 0-dim tensors: NaN-propagating."""
    if np.isnan(a) or np.isnan(b):
        return type(a)(np.nan)
    return a if a >= b else b


def residual_scales(H, A, g, x, z, lam, wE=None, wD=None):
    """(max(|Ax|,|z|), max(|Hx|,|A'lam|,|g|)) in inf-norms: the denominators of reluqpth.py:315-316, reused by the
    build's eps_rel extension as the scales of the two residuals.  wE / wD: see compute_residuals."""
    T = x.dtype.type
    wE = T(1) if wE is None else wE.astype(x.dtype)
    wD = T(1) if wD is None else wD.astype(x.dtype)
    sp = _tmax(T(_inf_norm((A @ x) * wE)), T(_inf_norm(z * wE)))
    sd = _tmax(_tmax(T(_inf_norm((H @ x) * wD)), T(_inf_norm((A.T @ lam) * wD))), T(_inf_norm(g * wD)))
    return sp, sd


def compute_residuals(H, A, g, x, z, lam, rho, rho_min, rho_max, wE=None, wD=None):
    """reluqpth.py:307-318.  Returns (primal_res, dual_res, rho_estimate).
    Build extension (Ruiz scaling): wE = 1 / E (rows) and wD = 1 / (c D) (columns) take every term back to the caller's
    space before its inf-norm (H, A, g, x, z, lam are the scaled quantities): "solved" then certifies the tolerances in
    the caller's units (OSQP's scaled_termination = 0).  Statement of the weighted maxima of the HIP kernels."""
    T = x.dtype.type
    wE = T(1) if wE is None else wE.astype(x.dtype)
    wD = T(1) if wD is None else wD.astype(x.dtype)
    t1 = A @ x                                             # :309
    t2 = H @ x                                             # :310
    t3 = A.T @ lam                                         # :311
    pri = T(_inf_norm(np.abs(t1 - z) * wE))                # :313
    dua = T(_inf_norm(np.abs(t2 + t3 + g) * wD))           # :314
    with np.errstate(invalid="ignore", divide="ignore"):
        num = pri / _tmax(T(_inf_norm(t1 * wE)), T(_inf_norm(z * wE)))                          # :315
        den = dua / _tmax(_tmax(T(_inf_norm(t2 * wD)), T(_inf_norm(t3 * wD))), T(_inf_norm(g * wD)))  # :316
        est = T(rho) * np.sqrt(num / den)                  # :317
    # torch.clamp keeps NaN (Q17)
    if est < rho_min:
        est = T(rho_min)
    elif est > rho_max:
        est = T(rho_max)
    return pri, dua, T(est)


def infeasibility_certificate(H, A, g, l, u, x, z, dx, rho_vec, eps_prim_inf, eps_dual_inf):
    """Build extension (the reference has no infeasibility test): OSQP's certificates on
    dy = rho (A x - z) -- the dual increment the next iteration applies, projected on the polar of the recession cone of
    [l, u] -- and on dx, the last x step.  Returns STATUS_PRIMAL_INFEASIBLE / STATUS_DUAL_INFEASIBLE / None.
    Statement of csrc/rqp_admm.hip:certificates."""
    dy = rho_vec * (A @ x - z)
    uinf, linf = np.isposinf(u), np.isneginf(l)
    dy = np.where(uinf & linf, 0.0, np.where(uinf, np.minimum(dy, 0.0), np.where(linf, np.maximum(dy, 0.0), dy)))
    ndy = np.abs(dy).max() if dy.size else 0.0
    with np.errstate(invalid="ignore"):
        lhs = np.sum(np.where(dy > 0, u * dy, 0.0)) + np.sum(np.where(dy < 0, l * dy, 0.0))
    if ndy > 0 and lhs < -eps_prim_inf * ndy and np.abs(A.T @ dy).max() < eps_prim_inf * ndy:
        return STATUS_PRIMAL_INFEASIBLE
    ndx = np.abs(dx).max()
    Adx = A @ dx
    tol = eps_dual_inf * ndx
    bad = np.any((~uinf & (Adx > tol)) | (~linf & (Adx < -tol)))
    if ndx > 0 and g @ dx < -eps_dual_inf * ndx and np.abs(H @ dx).max() < eps_dual_inf * ndx and not bad:
        return STATUS_DUAL_INFEASIBLE
    return None


def compute_J(H, g, x):
    """Objective, reluqpth.py:320-322."""
    return 0.5 * np.dot(x, H @ x) + np.dot(g, x)


# ----------------------------------------------------------------- a1, a7..a12
class Info:
    """classes.py:67-88."""

    def __init__(self):
        self.iter = None
        self.status = None
        self.obj_val = None
        self.pri_res = None
        self.dua_res = None
        self.setup_time = 0
        self.solve_time = 0
        self.update_time = 0
        self.run_time = 0
        self.rho_estimate = None


class Results:
    """classes.py:91-95 (+ y, the build's Q7 extension)."""

    def __init__(self, info):
        self.x = None
        self.z = None
        self.y = None
        self.info = info


class OracleQP:
    """Restatement of class ReLU_QP (reluqpth.py:92-333), single instance."""

    def __init__(self, form="W", quirks=False):
        assert form in ("W", "factored", "refine")
        self.form = form
        self.quirks = quirks
        self.info = Info()
        self.results = Results(self.info)
        self.trace = []

    # a12 -------------------------------------------------------------- setup
    def setup(self, H, g, A, l, u, **kw):
        """reluqpth.py:102-157."""
        t0 = time.perf_counter()
        self.settings = st = Settings(**kw)
        dt = st.dtype
        # a1: classes.py:4-30
        self.H = np.ascontiguousarray(H, dtype=dt)
        self.g = np.ascontiguousarray(g, dtype=dt)
        self.A = np.ascontiguousarray(A, dtype=dt)
        self.l = np.ascontiguousarray(l, dtype=dt)
        self.u = np.ascontiguousarray(u, dtype=dt)
        self.nx, self.nc = self.H.shape[0], self.A.shape[0]
        self._lu_setup = (self.l.copy(), self.u.copy())    # the equality pattern (rho x 1e3 rows) is fixed at setup (raw l, u)
        self._sc = None
        passes = 10 if st.scaling is True else int(st.scaling or 0)
        self._raw = dict(H=self.H.copy(), A=self.A.copy(), g=self.g.copy(), l=self.l.copy(), u=self.u.copy())
        if passes > 0:                                     # build extension (8(f)-3): solve the equilibrated problem
            assert not self.quirks, "the reference has no scaling"
            D, E, c, Hs, As = ruiz_scale(self.H, self.A, passes)
            self._sc = (D, E, c)
            self.H, self.A = Hs.astype(dt), As.astype(dt)
            self.g = (self.g.astype(np.float64) * (c * D)).astype(dt)
            self.l = (self.l.astype(np.float64) * E).astype(dt)
            self.u = (self.u.astype(np.float64) * E).astype(dt)
        self.rhos = setup_rhos(st.rho, st.rho_min, st.rho_max, st.adaptive_rho_tolerance,
                               st.adaptive_rho).astype(dt)
        self._build_matrices()
        self.clear_primal_dual()
        self.info.setup_time = time.perf_counter() - t0

    def _build_matrices(self):
        st = self.settings
        l0, u0 = self._lu_setup
        if self.form == "W":
            self.W_ks, self.B_ks, self.b_ks = setup_matrices_W(
                self.H, self.g, self.A, l0, u0, self.rhos, st.sigma, st.eq_tol)
        else:
            self.At = np.ascontiguousarray(self.A.T)
            self.rho_vecs = [rho_vector(r, l0, u0, st.eq_tol) for r in self.rhos]
            # K is computed in float64 and rounded to the working dtype: this is
            # what the HIP setup kernel does (fp64 Cholesky)
            H64, A64 = self.H.astype(np.float64), self.A.astype(np.float64)
            self.K_ks = [kkt_inverse(H64, A64, rv.astype(np.float64), st.sigma).astype(st.dtype)
                         for rv in self.rho_vecs]

    def clear_primal_dual(self):
        """reluqpth.py:324-333."""
        dt = self.settings.dtype
        self.x = np.zeros(self.nx, dt)
        self.z = np.zeros(self.nc, dt)
        self.lam = np.zeros(self.nc, dt)
        self.output = np.concatenate([self.x, self.z, self.lam])
        self._acc = None        # refine form: float64 accumulators (x, z, lam, A x)
        self.rho_ind = int(np.argmin(np.abs(self.rhos - self.settings.rho)))

    # a10 ------------------------------------------------------------- update
    def update(self, g=None, l=None, u=None, Hx=None, Ax=None):
        """reluqpth.py:159-183.  Note (replicated): l/u changes do NOT re-derive
        the equality-row rho vectors nor K -- the reference keeps the matrices
        built at setup (W_ks untouched, :171-174)."""
        t0 = time.perf_counter()
        dt = self.settings.dtype
        sc = getattr(self, "_sc", None)
        for k_, v_ in (("g", g), ("l", l), ("u", u)):
            if v_ is not None:
                self._raw[k_] = np.ascontiguousarray(v_, dtype=dt)
        if g is not None:
            self.g = np.ascontiguousarray(g, dtype=dt)
            if sc is not None:
                self.g = (self.g.astype(np.float64) * (sc[2] * sc[0])).astype(dt)
            if self.form == "W":
                self.b_ks = [B @ self.g for B in self.B_ks]
        if l is not None:
            self.l = np.ascontiguousarray(l, dtype=dt)
            if sc is not None:
                self.l = (self.l.astype(np.float64) * sc[1]).astype(dt)
        if u is not None:
            self.u = np.ascontiguousarray(u, dtype=dt)
            if sc is not None:
                self.u = (self.u.astype(np.float64) * sc[1]).astype(dt)
        if Hx is not None or Ax is not None:
            # The reference asserts here ("updating Hx and Ax is not supported yet", reluqpth.py:176-177).  The build
            # defines it as: new dense H / A, the matrices of setup_matrices (:40-78) rebuilt with the
            # equality pattern of setup, state and rho index carried -- this restatement is that definition.
            assert not self.quirks, "updating Hx and Ax is not supported yet"
            if Hx is not None:
                self._raw["H"] = np.ascontiguousarray(Hx, dtype=dt)
            if Ax is not None:
                self._raw["A"] = np.ascontiguousarray(Ax, dtype=dt)
            self.H, self.A = self._raw["H"], self._raw["A"]
            if sc is not None:                             # new D, E, c: vectors and state move to the new scaled space
                D0, E0, c0 = sc
                passes = 10 if self.settings.scaling is True else int(self.settings.scaling)
                D, E, c, Hs, As = ruiz_scale(self.H, self.A, passes)
                self._sc = (D, E, c)
                self.H, self.A = Hs.astype(dt), As.astype(dt)
                self.g = (self._raw["g"].astype(np.float64) * (c * D)).astype(dt)
                self.l = (self._raw["l"].astype(np.float64) * E).astype(dt)
                self.u = (self._raw["u"].astype(np.float64) * E).astype(dt)
                n_, m_ = self.nx, self.nc
                o = self.output.astype(np.float64)
                o[:n_] *= D0 / D
                o[n_:n_ + m_] *= E / E0
                o[n_ + m_:] *= (c / c0) * (E0 / E)
                self.output = o.astype(dt)
            self._build_matrices()
            self._acc = None
        self.info.update_time = time.perf_counter() - t0

    def update_settings(self, **kwargs):
        """reluqpth.py:185-199 with Q8 fixed (accept eps_abs, tolerate eps_ab)."""
        for key, value in kwargs.items():
            if key in ("max_iter", "eps_abs", "verbose", "check_interval"):
                setattr(self.settings, key, value)
            elif key == "eps_ab":
                self.settings.eps_abs = value
            elif key in ("rho", "rho_min", "rho_max", "sigma", "adaptive_rho",
                         "adaptive_rho_interval", "adaptive_rho_tolerance"):
                raise ValueError("Cannot change {} after setup".format(key))
            else:
                raise ValueError("Invalid setting: {}".format(key))

    def warm_start(self, x=None, z=None, lam=None, rho=None):
        """reluqpth.py:251-276.  quirks=True: attributes only (Q6: no effect on the
        iterate); quirks=False: written into the state."""
        n, m = self.nx, self.nc
        dt = self.settings.dtype
        sc = getattr(self, "_sc", None)
        if sc is not None:                                 # caller space -> scaled space
            x = None if x is None else np.asarray(x, np.float64) / sc[0]
            z = None if z is None else np.asarray(z, np.float64) * sc[1]
            lam = None if lam is None else np.asarray(lam, np.float64) * (sc[2] / sc[1])
        if x is not None:
            self.x = np.asarray(x, dt).copy()
            if not self.quirks:
                self.output[:n] = self.x
        if z is not None:
            self.z = np.asarray(z, dt).copy()
            if not self.quirks:
                self.output[n:n + m] = self.z
        if lam is not None:
            self.lam = np.asarray(lam, dt).copy()
            if not self.quirks:
                self.output[n + m:] = self.lam
        if rho is not None:
            self.rho_ind = int(np.argmin(np.abs(self.rhos - rho)))
        if not self.quirks:
            self._acc = None

    # a5 dispatch ----------------------------------------------------- iterate
    def _iterate(self):
        n, m = self.nx, self.nc
        if self.form == "W":
            self.output = forward_W(self.output, self.W_ks[self.rho_ind], self.b_ks[self.rho_ind],
                                    self.l, self.u, n, m)
        elif self.form == "refine":
            if self._acc is None:
                s64 = self.output.astype(np.float64)
                x0 = s64[:n]
                self._acc = [x0, s64[n:n + m], s64[n + m:], self.A.astype(np.float64) @ x0]
            x, z, lam, zt = self._acc
            self._acc = list(forward_refine(x, z, lam, zt, self.K_ks[self.rho_ind], self.H, self.A, self.At,
                                            self.g, self.l, self.u, self.rho_vecs[self.rho_ind]))
            self.output = np.concatenate(self._acc[:3]).astype(self.settings.dtype)
        else:
            s = self.output
            x, z, lam = s[:n], s[n:n + m], s[n + m:]
            zt = self.A @ x
            xn, zn, lamn, _ = forward_factored(x, z, lam, zt, self.K_ks[self.rho_ind], self.A, self.g,
                                               self.l, self.u, self.rho_vecs[self.rho_ind],
                                               self.settings.dtype(self.settings.sigma))
            self.output = np.concatenate([xn, zn, lamn])

    def iterate(self, k):
        """k plain iterations at the current rho index (no checks)."""
        for _ in range(k):
            self._iterate()
        return self.output

    # a8 --------------------------------------------------------------- solve
    def solve(self):
        """reluqpth.py:201-249 (loop control a8, check a6, index move + termination a7)."""
        t0 = time.perf_counter()
        st = self.settings
        n, m = self.nx, self.nc
        rho = self.rhos[self.rho_ind]                      # :211 (Q4: carried below)
        nrho = len(self.rhos)
        thr_p = st.eps_abs * np.sqrt(m)
        thr_d = st.eps_abs * np.sqrt(n)
        tol = st.adaptive_rho_tolerance
        self.trace = []
        for k in range(1, st.max_iter + 1):
            x_prev = self.output[:n].copy()
            self._iterate()                                # :215
            do_check = (k % st.check_interval == 0)
            if self.quirks:
                do_check = do_check and st.adaptive_rho    # :218 (Q3)
            if do_check:
                s = self.output
                self.x, self.z, self.lam = s[:n], s[n:n + m], s[n + m:]      # :219
                pri, dua, rho = compute_residuals(self.H, self.A, self.g, self.x, self.z, self.lam,
                                                  rho, st.rho_min, st.rho_max, *self._unscale_weights())   # :220
                self.trace.append((float(pri), float(dua), float(rho), self.rho_ind))
                if rho > self.rhos[self.rho_ind] * tol and self.rho_ind < nrho - 1:   # :223
                    self.rho_ind += 1
                elif rho < self.rhos[self.rho_ind] / tol and self.rho_ind > 0:        # :226
                    self.rho_ind -= 1
                if st.verbose:
                    print('Iter: {}, rho: {:.2e}, res_p: {:.2e}, res_d: {:.2e}'.format(k, rho, pri, dua))
                tp, td = thr_p, thr_d
                if st.eps_rel > 0:                         # build extension (8(f)-3): OSQP-style relative term
                    sp, sd = residual_scales(self.H, self.A, self.g, self.x, self.z, self.lam, *self._unscale_weights())
                    tp, td = thr_p + st.eps_rel * sp, thr_d + st.eps_rel * sd
                if pri < tp and dua < td:                  # :233
                    self._update_results(k, STATUS_SOLVED, pri, dua, rho, t0)
                    return self.results
                if st.check_infeasibility and not self.quirks:
                    rv = rho_vector(self.rhos[self.rho_ind], *self._lu_setup, st.eq_tol)
                    code = infeasibility_certificate(self.H, self.A, self.g, self.l, self.u, self.x, self.z,
                                                     self.x - x_prev, rv, st.eps_prim_inf, st.eps_dual_inf)
                    if code is not None:
                        self._update_results(k, code, pri, dua, rho, t0)
                        return self.results
        if not self.quirks:                                # Q11 fix: re-slice
            s = self.output
            self.x, self.z, self.lam = s[:n], s[n:n + m], s[n + m:]
        pri, dua, rho = compute_residuals(self.H, self.A, self.g, self.x, self.z, self.lam,
                                          rho, st.rho_min, st.rho_max, *self._unscale_weights())           # :243
        status = STATUS_MAX_ITER
        if not self.quirks and (np.isnan(pri) or np.isnan(dua)):
            status = STATUS_NAN                            # build extension: the reference reports max_iters_reached (Q17)
        self._update_results(st.max_iter, status, pri, dua, rho, t0)
        return self.results

    def _unscale_weights(self):
        """(1 / E, 1 / (c D)) with Ruiz scaling, else (None, None): compute_residuals."""
        sc = getattr(self, "_sc", None)
        if sc is None:
            return None, None
        D, E, c = sc
        return 1.0 / E, 1.0 / (c * D)

    # a9 ------------------------------------------------------ update_results
    def _update_results(self, it, status, pri, dua, rho, t0):
        """reluqpth.py:278-305."""
        n, m = self.nx, self.nc
        self.results.x = self.x.copy()
        self.results.z = self.z.copy()
        self.results.lam = self.lam.copy()
        self.info.obj_val = compute_J(self.H, self.g, self.x)
        sc = getattr(self, "_sc", None)
        if sc is not None:                                 # scaled space -> caller space (the residuals already are: compute_residuals)
            D, E, c = sc
            self.results.x = self.results.x * D
            self.results.z = self.results.z / E
            self.results.lam = self.results.lam * (E / c)
            self.info.obj_val = self.info.obj_val / c
        self.results.y = self.results.lam.copy()           # Q7 extension: dual of the state
        self.info.iter = it
        self.info.status = status
        self.info.pri_res = pri
        self.info.dua_res = dua
        self.info.rho_estimate = rho
        self.info.run_time = time.perf_counter() - t0
        self.info.solve_time = self.info.update_time + self.info.run_time
        self.lam = np.zeros(m, self.settings.dtype)        # :303 (attribute only, Q7)
        if not self.settings.warm_starting:
            self.clear_primal_dual()                       # :304-305


def solve_batch(H, g, A, l, u, form="W", quirks=False, **kw):
    """Loop the single-instance oracle over a batch (leading dim on every array
    that has one); returns dict of stacked outputs.  Used by the parity tests
    and by bench.py's cpu_baseline leg."""
    B = g.shape[0]
    n, m = g.shape[-1], l.shape[-1]
    out = dict(x=np.zeros((B, n)), z=np.zeros((B, m)), lam=np.zeros((B, m)),
               iter=np.zeros(B, np.int64), status=[], pri_res=np.zeros(B), dua_res=np.zeros(B),
               rho_estimate=np.zeros(B), rho_ind=np.zeros(B, np.int64), obj_val=np.zeros(B),
               setup_time=0.0, solve_time=0.0)
    for i in range(B):
        Hi = H[i] if H.ndim == 3 else H
        Ai = A[i] if A.ndim == 3 else A
        qp = OracleQP(form=form, quirks=quirks)
        qp.setup(Hi, g[i], Ai, l[i], u[i], **kw)
        r = qp.solve()
        out["x"][i], out["z"][i], out["lam"][i] = r.x, r.z, r.lam
        out["iter"][i] = r.info.iter
        out["status"].append(r.info.status)
        out["pri_res"][i], out["dua_res"][i] = r.info.pri_res, r.info.dua_res
        out["rho_estimate"][i] = r.info.rho_estimate
        out["rho_ind"][i] = qp.rho_ind
        out["obj_val"][i] = r.info.obj_val
        out["setup_time"] += qp.info.setup_time
        out["solve_time"] += r.info.run_time
    return out
