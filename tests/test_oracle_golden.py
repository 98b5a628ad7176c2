"""Pin the oracle (oracle/reluqp_oracle.py) against outputs of the REFERENCE itself.

The fixtures in tests/golden/ were produced by tests/golden/make_golden.py, which
imports and runs /root/reference/ReLU-QP-py/reluqp (fp64, CPU, SURVEY.md 8(c)
harness).  Tolerances: the oracle uses numpy/OpenBLAS, the reference torch -- the
same IEEE operations in possibly different summation order, so floating-point
results are compared at 1e-9..1e-12 relative; integers/strings bit-exact.
"""
import hashlib

import numpy as np
import pytest

from oracle import reluqp_oracle as O
from reluqp import utils

RTOL_W = 1e-10     # W-form oracle vs reference (same formulation)
RTOL_F = 1e-7      # factored-form oracle vs reference (different association, K vs W)


def _qp(gold, p=""):
    return [gold[p + k] for k in ("H", "g", "A", "l", "u")]


def _check_result(gold, p, qp, res, rtol, atol=1e-12, check_rho=True):
    assert res.info.iter == int(gold[p + "iter"])
    assert res.info.status == str(gold[p + "status"])
    np.testing.assert_allclose(res.x, gold[p + "x"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(res.z, gold[p + "z"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(qp.output, gold[p + "state"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(res.info.obj_val, gold[p + "obj_val"], rtol=rtol, atol=atol)
    if not check_rho:
        return
    assert qp.rho_ind == int(gold[p + "rho_ind_final"])
    tr = np.array(qp.trace).reshape(-1, 4)
    gt = gold[p + "trace"]
    assert tr.shape == gt.shape
    assert np.array_equal(tr[:, 3], gt[:, 3])                      # rho_ind trajectory: exact
    # residuals are differences of O(1..100) terms built from an inverse of a matrix with
    # cond ~1e9 (rho*1e3 on equality rows vs sigma=1e-6): LAPACK-order noise is ~1e-6 relative
    np.testing.assert_allclose(tr[:, :2], gt[:, :2], rtol=max(100 * rtol, 1e-5), atol=1e-6)
    # the rho estimate is a ratio of the two residuals: when either is pure
    # rounding noise (G1 converges to ~1e-14) the estimate is noise too -- skip those rows
    # It also compounds from check to check (Q4), so noise accumulates along a long run:
    # 1e-2 relative is ample for what it drives (index moves in factor-of-5 bands, compared exactly above)
    ok = ~np.isnan(gt[:, 2]) & (gt[:, 0] > 1e-8) & (gt[:, 1] > 1e-8)
    np.testing.assert_allclose(tr[ok, 2], gt[ok, 2], rtol=1e-2)
    assert np.array_equal(np.isnan(tr[:, 2]), np.isnan(gt[:, 2]))


# ------------------------------------------------------------------ G1: builtin
def test_g1_rho_ladder(golden):
    g = golden("g1_builtin.npz")
    rhos = O.setup_rhos(0.1, 1e-6, 1e6, 5, True)
    assert rhos.shape == (18,)
    assert np.array_equal(rhos, g["rhos"])                         # bit-exact
    assert int(np.argmin(np.abs(rhos - 0.1))) == int(g["rho_ind0"]) == 7
    assert np.array_equal(O.setup_rhos(0.1, 1e-6, 1e6, 5, False), g["rhos_noadapt"])
    alt = O.setup_rhos(0.4, 1e-3, 1e3, 3, True)
    assert np.array_equal(alt, g["rhos_alt"])
    assert int(np.argmin(np.abs(alt - 0.4))) == int(g["rho_ind0_alt"])


def test_g1_W_and_iterates(golden):
    g = golden("g1_builtin.npz")
    H, gg, A, l, u = _qp(g)
    qp = O.OracleQP(form="W", quirks=True)
    qp.setup(H, gg, A, l, u)
    np.testing.assert_allclose(qp.W_ks[7], g["W7"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(qp.b_ks[7], g["b7"], rtol=1e-11, atol=1e-13)
    for k, key in ((1, "state_k1"), (2, "state_k2"), (25, "state_k25")):
        q = O.OracleQP(form="W", quirks=True)
        q.setup(H, gg, A, l, u)
        np.testing.assert_allclose(q.iterate(k), g[key], rtol=1e-10, atol=1e-12)
        f = O.OracleQP(form="factored")
        f.setup(H, gg, A, l, u)
        np.testing.assert_allclose(f.iterate(k), g[key], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("form,quirks,rtol", [("W", True, RTOL_W), ("W", False, RTOL_W),
                                              ("factored", False, RTOL_F)])
def test_g1_solve_and_warm_resolve(golden, form, quirks, rtol):
    g = golden("g1_builtin.npz")
    H, gg, A, l, u = _qp(g)
    qp = O.OracleQP(form=form, quirks=quirks)
    qp.setup(H, gg, A, l, u)
    res = qp.solve()
    _check_result(g, "", qp, res, rtol, atol=1e-9)
    np.testing.assert_allclose(res.x, [2.0, -1.0, 1.0], atol=1e-8)   # reluqpth.py:360
    res2 = qp.solve()                                              # warm: state + rho_ind persist
    # the first solve converged to rounding noise (pri = 0, dua ~ 1e-13), so the
    # warm re-solve's rho estimate -- a ratio of two noise residuals -- and the
    # index move it drives are not reproducible across BLAS backends: values only
    _check_result(g, "warm_", qp, res2, rtol, atol=1e-9, check_rho=False)


def test_g1_cold_and_maxiter(golden):
    g = golden("g1_builtin.npz")
    H, gg, A, l, u = _qp(g)
    qp = O.OracleQP(form="W", quirks=True)
    qp.setup(H, gg, A, l, u, warm_starting=False)
    res = qp.solve()
    assert res.info.iter == int(g["cold_iter"])
    assert np.array_equal(qp.output, g["cold_state_after"])        # zeros
    assert qp.rho_ind == int(g["cold_rho_ind_after"])
    for quirks in (True, False):
        q = O.OracleQP(form="W", quirks=quirks)
        q.setup(H, gg, A, l, u, max_iter=10)
        r = q.solve()
        assert r.info.status == str(g["mi10_status"]) == "max_iters_reached"
        assert r.info.iter == int(g["mi10_iter"]) == 10
        np.testing.assert_allclose(q.output, g["mi10_state"], rtol=1e-10, atol=1e-12)
        if quirks:      # Q11: the reference returns the never-updated zero slices
            assert np.all(r.x == 0)
        else:           # build disposition: fresh slices
            np.testing.assert_allclose(r.x, g["mi10_state"][:3], rtol=1e-10)


def test_g1_tight(golden):
    g = golden("g1_builtin.npz")
    H, gg, A, l, u = _qp(g)
    qp = O.OracleQP(form="W", quirks=True)
    qp.setup(H, gg, A, l, u, eps_abs=1e-8)
    res = qp.solve()
    _check_result(g, "tight_", qp, res, 1e-9, atol=1e-10)


# ------------------------------------------------------------ G2: compat rand_qp
@pytest.mark.parametrize("seed", range(5))
def test_g2_generator_and_solve(golden, seed):
    g = golden("g2_randqp_compat.npz")
    p = "s%d_" % seed
    H, gg, A, l, u, _ = utils.rand_qp(nx=10, n_eq=5, n_ineq=5, seed=seed, compute_sol=False)
    for a, k in ((H, "H"), (gg, "g"), (A, "A"), (l, "l"), (u, "u")):
        assert np.array_equal(a, g[p + k])                         # generator: bit-exact
    for form, rtol in (("W", 1e-8), ("factored", 1e-6)):
        qp = O.OracleQP(form=form, quirks=(form == "W"))
        qp.setup(H, gg, A, l, u)
        res = qp.solve()
        _check_result(g, p, qp, res, rtol, atol=1e-9)


# ------------------------------------------------------- G3: C1 feasible n=10 m=20
@pytest.mark.parametrize("seed", range(3))
def test_g3_c1(golden, seed):
    g = golden("g3_c1_feasible.npz")
    p = "s%d_" % seed
    H, gg, A, l, u, xs = utils.rand_qp(nx=10, n_eq=5, n_ineq=15, seed=seed, feasible=True)
    for a, k in ((H, "H"), (gg, "g"), (A, "A"), (l, "l"), (u, "u"), (xs, "x_planted")):
        assert np.array_equal(a, g[p + k])
    for form, rtol in (("W", 1e-8), ("factored", 1e-6)):
        qp = O.OracleQP(form=form)
        qp.setup(H, gg, A, l, u)
        res = qp.solve()
        _check_result(g, p, qp, res, rtol, atol=1e-9)
    # tight run converges to the planted optimum (independent truth)
    qp = O.OracleQP(form="factored")
    qp.setup(H, gg, A, l, u, eps_abs=1e-9, max_iter=20000)
    res = qp.solve()
    assert res.info.status == "solved"
    # eps_abs*sqrt(n) = 3e-9 sits at the rounding floor of the dual residual, so the
    # terminating check is marginal: allow one check (25 iterations) either way
    assert abs(res.info.iter - int(g[p + "tight_iter"])) <= 25
    np.testing.assert_allclose(res.x, xs, atol=1e-8)


# ----------------------------------------------- G4: C2 (n=100,m=300) and C4 shapes
def _sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("seed", range(3))
def test_g4_c2(golden, seed):
    g = golden("g4_c2_feasible.npz")
    p = "s%d_" % seed
    H, gg, A, l, u, xs = utils.rand_qp(nx=100, n_eq=25, n_ineq=275, seed=seed, feasible=True)
    assert _sha(H, gg, A, l, u) == str(g[p + "input_sha256"])     # inputs regenerated, pinned by hash
    qp = O.OracleQP(form="factored")
    qp.setup(H, gg, A, l, u)
    res = qp.solve()
    _check_result(g, p, qp, res, 1e-6, atol=1e-8)
    if seed == 0:   # the W form at D=700 is slow (18 inverses + GEMMs): one seed only
        qw = O.OracleQP(form="W", quirks=True)
        qw.setup(H, gg, A, l, u)
        rw = qw.solve()
        _check_result(g, p, qw, rw, 1e-7, atol=1e-9)
    if seed == 1:   # tight tolerance: 350 iterations in the reference
        qt = O.OracleQP(form="factored")
        qt.setup(H, gg, A, l, u, eps_abs=1e-6)
        rt = qt.solve()
        _check_result(g, p + "e6_", qt, rt, 1e-6, atol=1e-8)


@pytest.mark.parametrize("seed", range(3))
def test_g4_c4_shape(golden, seed):
    g = golden("g4_c2_feasible.npz")
    p = "c4s%d_" % seed
    H, gg, A, l, u, xs = utils.rand_qp(nx=32, n_eq=8, n_ineq=56, seed=seed, feasible=True)
    assert _sha(H, gg, A, l, u) == str(g[p + "input_sha256"])
    qp = O.OracleQP(form="factored")
    qp.setup(H, gg, A, l, u)
    res = qp.solve()
    _check_result(g, p, qp, res, 1e-6, atol=1e-8)


# --------------------------------------------------------------- G5: update()
@pytest.mark.parametrize("form", ["W", "factored"])
def test_g5_update(golden, form):
    g1 = golden("g1_builtin.npz")
    g = golden("g5_update.npz")
    H, gg, A, l, u = _qp(g1)
    qp = O.OracleQP(form=form)
    qp.setup(H, gg, A, l, u)
    qp.solve()
    qp.update(g=g["g_new"])
    res = qp.solve()
    _check_result(g, "upd_g_", qp, res, 1e-7, atol=1e-9)
    qp.update(l=g["l_new"], u=g["u_new"])
    res = qp.solve()
    _check_result(g, "upd_lu_", qp, res, 1e-7, atol=1e-9)
    qq = O.OracleQP(form=form, quirks=True)                       # the reference's behaviour (reluqpth.py:176-177)
    qq.setup(H, gg, A, l, u)
    with pytest.raises(AssertionError):
        qq.update(Hx=np.eye(3))
    # the build's definition: same as a fresh setup with the new matrices and the carried state
    H2 = H + 0.5 * np.eye(3)
    qp.update(Hx=H2)
    res = qp.solve()
    fresh = O.OracleQP(form=form)
    fresh.setup(H2, g["g_new"], A, g["l_new"], g["u_new"])
    rf = fresh.solve()
    assert res.info.status == rf.info.status == "solved"
    np.testing.assert_allclose(res.x, rf.x, atol=5e-3)             # both within eps_abs of the same optimum


# ------------------------------------------------ G6: compute_residuals / compute_J
def test_g6_residuals(golden):
    g = golden("g6_residuals.npz")
    for i in range(int(g["n_cases"])):
        p = "c%d_" % i
        pri, dua, rho = O.compute_residuals(g[p + "H"], g[p + "A"], g[p + "g"], g[p + "x"], g[p + "z"],
                                            g[p + "lam"], float(g[p + "rho_in"]), 1e-6, 1e6)
        np.testing.assert_allclose(pri, g[p + "pri"], rtol=1e-12, atol=1e-13)   # atol: cancellation floor of sums of O(1..10) terms
        np.testing.assert_allclose(dua, g[p + "dua"], rtol=1e-12, atol=1e-13)   # atol: cancellation floor of sums of O(1..10) terms
        if np.isnan(g[p + "rho_out"]):
            assert np.isnan(rho)                                   # Q17: 0/0 stays NaN through clamp
        else:
            np.testing.assert_allclose(rho, g[p + "rho_out"], rtol=1e-5)
        np.testing.assert_allclose(O.compute_J(g[p + "H"], g[p + "g"], g[p + "x"]), g[p + "J"], rtol=1e-12)
    assert np.isnan(g["c4_rho_out"])
    assert float(g["c5_rho_out"]) == 1e6 and float(g["c6_rho_out"]) == 1e-6   # both clamps exercised


# --------------------------------------- G7: K, W with equality rows; W == factored
def test_g7_matrices(golden):
    g = golden("g7_matrices.npz")
    H, gg, A, l, u = _qp(g)
    rhos = g["rhos"]
    n, m = 10, 10
    W_ks, B_ks, b_ks = O.setup_matrices_W(H, gg, A, l, u, rhos, 1e-6, 1e-6)
    rs = np.random.RandomState(7)
    for j in (3, 7, 12):
        rv = O.rho_vector(rhos[j], l, u, 1e-6)
        assert np.array_equal(rv[:5], np.full(5, rhos[j] * 1e3))  # x1e3 rule on equality rows (Q15)
        assert np.array_equal(rv[5:], np.full(5, rhos[j]))
        K = O.kkt_inverse(H, A, rv, 1e-6)
        np.testing.assert_allclose(K, g["K%d" % j], rtol=1e-9, atol=1e-14)
        Wg = g["W%d" % j]   # blocks like 2AKA'rho - I cancel: absolute floor relative to max|W|
        np.testing.assert_allclose(W_ks[j], Wg, rtol=1e-8, atol=1e-10 * np.abs(Wg).max())
        np.testing.assert_allclose(b_ks[j], g["b%d" % j], rtol=1e-8, atol=1e-10)
        # W-form step == factored step on a random state (Appendix A.2 identity)
        s = rs.randn(n + 2 * m)
        sw = O.forward_W(s, g["W%d" % j], g["b%d" % j], l, u, n, m)
        x, z, lam = s[:n], s[n:n + m], s[n + m:]
        xn, zn, lamn, _ = O.forward_factored(x, z, lam, A @ x, K, A, gg, l, u, rv, 1e-6)
        np.testing.assert_allclose(np.concatenate([xn, zn, lamn]), sw, rtol=1e-8, atol=1e-9)


# ------------------------------------------------------------ API-level behaviours
def test_update_settings_q8():
    qp = O.OracleQP()
    qp.setup(np.eye(2), np.ones(2), np.eye(2), -np.ones(2), np.ones(2))
    qp.update_settings(eps_abs=1e-5, max_iter=10)
    assert qp.settings.eps_abs == 1e-5 and qp.settings.max_iter == 10
    qp.update_settings(eps_ab=1e-4)
    assert qp.settings.eps_abs == 1e-4
    with pytest.raises(ValueError):
        qp.update_settings(rho=1.0)
    with pytest.raises(ValueError):
        qp.update_settings(bogus=1)


@pytest.mark.parametrize("seed", range(3))
def test_fp32_refine_tracks_fp64_and_naive_fp32_does_not(golden, seed):
    """Sets the stated fp32 tolerance and records WHY the HIP kernels use the
    residual-correction ("refine") statement (DESIGN.md, fp32 numerics):

    * refine form with float32 matrices/matvecs + float64 vector accumulators: the SAME
      iteration count and rho-index trajectory as the float64 reference, x within 2e-5*max|x|;
    * the plain factored recurrence x+ = K(sigma x - g + A'(rho z - lam)) in float32
      cancels catastrophically once rho*1e3 on equality rows reaches ~500 and needs
      several times more iterations (observed 325..900 instead of 125..200).
    """
    g = golden("g4_c2_feasible.npz")
    p = "s%d_" % seed
    H, gg, A, l, u, xs = utils.rand_qp(nx=100, n_eq=25, n_ineq=275, seed=seed, feasible=True)
    qp = O.OracleQP(form="refine")
    qp.setup(H, gg, A, l, u, dtype=np.float32)
    res = qp.solve()
    assert res.info.status == "solved"
    assert res.info.iter == int(g[p + "iter"])
    assert [t[3] for t in qp.trace] == g[p + "trace"][:, 3].astype(int).tolist()
    assert qp.rho_ind == int(g[p + "rho_ind_final"])
    assert np.abs(res.x - g[p + "x"]).max() <= 2e-5 * np.abs(g[p + "x"]).max()
    # float64 refine == reference to rounding
    q64 = O.OracleQP(form="refine")
    q64.setup(H, gg, A, l, u)
    r64 = q64.solve()
    _check_result(g, p, q64, r64, 1e-6, atol=1e-8)
    naive = O.OracleQP(form="factored")
    naive.setup(H, gg, A, l, u, dtype=np.float32)
    rn = naive.solve()
    assert rn.info.iter >= int(g[p + "iter"]) + 100


@pytest.mark.parametrize("fixture,prefixes", [("g2_randqp_compat.npz", ["s%d_" % s for s in range(5)]),
                                              ("g3_c1_feasible.npz", ["s%d_" % s for s in range(3)])])
def test_refine_form_on_small_fixtures(golden, fixture, prefixes):
    g = golden(fixture)
    for p in prefixes:
        for dt, rtol, atol in ((np.float64, 1e-6, 1e-9), (np.float32, 1e-4, 2e-5)):
            qp = O.OracleQP(form="refine")
            qp.setup(*_qp(g, p), dtype=dt)
            res = qp.solve()
            assert res.info.iter == int(g[p + "iter"]) and res.info.status == "solved"
            assert qp.rho_ind == int(g[p + "rho_ind_final"])
            np.testing.assert_allclose(res.x, g[p + "x"], rtol=rtol, atol=atol * max(1, np.abs(g[p + "x"]).max()))
